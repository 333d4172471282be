#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
( cd /tmp && timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_clip -o clip -- python $R/tools/clip_bench.py --batch 256 --steps 3 --warmup 1 ) > gpurun_out/prof_clip.log 2>&1
echo "== exit $?"
f=$(ls gpurun_out/prof_clip/*kernel_stats.csv | head -1); python tools/prof_summary.py "$f" 4 > gpurun_out/prof_clip_summary.txt; head -24 gpurun_out/prof_clip_summary.txt | cut -c1-130; tail -1 gpurun_out/prof_clip_summary.txt
rm -f gpurun_out/prof_clip/*kernel_trace.csv
