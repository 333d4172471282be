#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
( cd /tmp && timeout 150 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_unet256 -o unet -- python $R/tools/unet_bench.py --img 256 --batch 1 --steps 2 --warmup 1 ) > gpurun_out/prof_unet256.log 2>&1
echo "== exit $?"; tail -n 1 gpurun_out/prof_unet256.log | cut -c1-300
f=$(ls gpurun_out/prof_unet256/*kernel_stats.csv | head -1); python tools/prof_summary.py "$f" 3 > gpurun_out/prof_unet256_summary.txt; head -16 gpurun_out/prof_unet256_summary.txt | cut -c1-130; tail -1 gpurun_out/prof_unet256_summary.txt
rm -f gpurun_out/prof_unet256/*kernel_trace.csv
