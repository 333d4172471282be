#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gemm_ablate.py > gpurun_out/gemm_ablate.log 2>&1; echo "== exit $?"; tail -30 gpurun_out/gemm_ablate.log
