#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/trace
mkdir -p $OUT
( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$OLDPWD/$OUT/prof" -o step -- python "$OLDPWD/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline ${BENCH_ARGS} ) > $OUT/prof.log 2>&1
f=$(ls $OUT/prof/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/trace_reduce.py "$f" $OUT/timeline.csv && gzip -f $OUT/timeline.csv; rm -rf $OUT/prof; tail -3 $OUT/prof.log; ls -la $OUT
