#!/bin/bash
# One gpurun call: per-file GPU tests (separate processes so a fault in one file does not hide the
# others), smoke, bench (graph + eager), rocprofv3 kernel stats.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/env.log
for f in gemm norm_elem attn modules train; do
  timeout 900 python -m pytest tests/test_gpu_$f.py -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/test_$f.log 2>&1
  echo "== test_gpu_$f exit $?"; tail -n 3 gpurun_out/test_$f.log
done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 2 gpurun_out/smoke.log
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_graph.log 2>&1; echo "== bench graph exit $?"; tail -c 600 gpurun_out/bench_graph.log
timeout 600 python bench.py --steps 10 --warmup 3 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/bench_eager.log 2>&1; echo "== bench eager exit $?"; tail -c 400 gpurun_out/bench_eager.log
