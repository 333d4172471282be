#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/gemm_bench.py "$@" > gpurun_out/gemm_bench.log 2>&1; echo "== gemm_bench exit $?"; cat gpurun_out/gemm_bench.log | tail -40
timeout 300 python -m pytest tests/test_gpu_gemm.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_gemm.log 2>&1; echo "== test_gpu_gemm exit $?"; tail -n 5 gpurun_out/test_gemm.log
