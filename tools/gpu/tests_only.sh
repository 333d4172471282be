#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 30 gpurun_out/pytest_gpu.log | cut -c1-250
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 1 gpurun_out/smoke.log
timeout 200 python tools/clip_bench.py --batch 256 --steps 8 --warmup 3 > gpurun_out/clip_bench.log 2>&1; echo "== clip bench exit $?"; tail -n 2 gpurun_out/clip_bench.log | cut -c1-500
