#!/bin/bash
# Second kind of gpurun call: graph bench with watchdog, rocprofv3 kernel stats of the eager step,
# GEMM roofline table.  Logs under gpurun_out/.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-roofline --watchdog 60 > gpurun_out/bench_graph.log 2>&1
echo "== bench graph exit $?"; tail -c 1500 gpurun_out/bench_graph.log
timeout 400 python bench.py --steps 5 --warmup 2 --no-graph --no-cpu-baseline > gpurun_out/bench_roofline.log 2>&1
echo "== bench eager+roofline exit $?"; tail -c 300 gpurun_out/bench_roofline.log
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats -d "$OLDPWD/gpurun_out/prof_eager" -o eager -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline ) > gpurun_out/prof_eager.log 2>&1
echo "== rocprof exit $?"; ls gpurun_out/prof_eager* | head; 
f=$(ls gpurun_out/prof_eager/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && head -25 "$f"
timeout 300 python bench.py --steps 3 --warmup 1 --no-graph --no-roofline --watchdog 120 > gpurun_out/bench_cpu.log 2>&1
echo "== bench cpu baseline exit $?"; tail -c 700 gpurun_out/bench_cpu.log
