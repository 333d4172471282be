#!/bin/bash
# quick correctness + bandwidth + bench cycle
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for f in gemm norm_elem attn modules train; do
  timeout 600 python -m pytest tests/test_gpu_$f.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_$f.log 2>&1
  echo "== test_gpu_$f exit $?"; tail -n 2 gpurun_out/test_$f.log
done
timeout 300 python tools/bw_bench.py 2>&1 | grep -E "LN|colsum" 
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.log 2>&1
echo "== bench graph exit $?"; grep -E "timed region" gpurun_out/bench_graph.log; tail -c 300 gpurun_out/bench_graph.log
