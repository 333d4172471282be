#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 400 python -m pytest tests/test_gpu_conv.py tests/test_gpu_unet.py -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/test_conv.log 2>&1; echo "== conv/unet tests exit $?"; tail -n 25 gpurun_out/test_conv.log | cut -c1-200
timeout 150 python tools/unet_bench.py --img 64 --batch 8 --steps 5 --warmup 2 > gpurun_out/unet_64.log 2>&1; echo "== unet 64 exit $?"; tail -n 1 gpurun_out/unet_64.log | cut -c1-400
