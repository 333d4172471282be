#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_rccl.py tests/test_checkpoint.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_rccl.log 2>&1; echo "== exit $?"; tail -n 25 gpurun_out/test_rccl.log | cut -c1-400
