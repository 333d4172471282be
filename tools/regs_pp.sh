#!/bin/bash
# resource table of the gemm_pp instantiations (tools/regs.sh for a file outside its default list)
exec "$(dirname "$0")/regs.sh" gemm_pp "$@"
