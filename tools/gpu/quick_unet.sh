#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 12 gpurun_out/pytest_gpu.log | cut -c1-220
timeout 300 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('vit b128', d['value'], d['ms_per_step'])"
timeout 150 python tools/unet_bench.py --img 64 --batch 8 --steps 5 --warmup 2 > gpurun_out/unet_64.log 2>&1; echo "== unet 64 exit $?"; tail -n 1 gpurun_out/unet_64.log | cut -c1-400
timeout 240 python tools/unet_bench.py --img 256 --batch 1 --steps 2 --warmup 1 > gpurun_out/unet_256.log 2>&1; echo "== unet 256 exit $?"; tail -n 1 gpurun_out/unet_256.log | cut -c1-400
