#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_train.py -m gpu -q --tb=short -p no:cacheprovider -x 2>&1 | tail -n 4 | cut -c1-300
run() { timeout 150 python tools/unet_bench.py --img 64 --batch 8 --steps 7 --warmup 1 "$@" 2>&1 | tail -n 1 | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(d['ms_per_step'], d['losses'], d['grad_checksum'])"; }
( echo "side on : $(run)"; echo "side off: $(run --no-side-stream)"; echo "side on, poison: $(run --poison)" ) | tee gpurun_out/unet_side_stream_check.log
