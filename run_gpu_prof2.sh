#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o step -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline ) > gpurun_out/prof.log 2>&1
echo "== rocprof exit $?"; ls gpurun_out/prof | head
f=$(ls gpurun_out/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" 7
