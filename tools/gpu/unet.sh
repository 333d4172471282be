#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 150 python tools/unet_bench.py --img 64 --batch 8 --steps 5 --warmup 2 > gpurun_out/unet_64.log 2>&1; echo "== unet 64 exit $?"; tail -n 3 gpurun_out/unet_64.log | cut -c1-600
timeout 240 python tools/unet_bench.py --img 256 --batch 1 --steps 2 --warmup 1 > gpurun_out/unet_256.log 2>&1; echo "== unet 256 exit $?"; tail -n 3 gpurun_out/unet_256.log | cut -c1-600
