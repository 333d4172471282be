#!/bin/bash
# rocprofv3 kernel-trace stats of the default bench command (7 steps) -> gpurun_out/$1/prof_summary.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/${1:-prof}
mkdir -p $OUT
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$OUT/prof" -o step -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline ) > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" 7 > $OUT/prof_summary.txt && cp "$f" $OUT/kernel_stats.csv; head -30 $OUT/prof_summary.txt; rm -rf $OUT/prof
