#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_unet -o unet -- python $R/tools/unet_bench.py --img 64 --batch 8 --steps 3 --warmup 1 ) > gpurun_out/prof_unet.log 2>&1
echo "== exit $?"; tail -n 1 gpurun_out/prof_unet.log | cut -c1-300
f=$(ls gpurun_out/prof_unet/*kernel_stats.csv | head -1); python tools/prof_summary.py "$f" 4 > gpurun_out/prof_unet_summary.txt; cat gpurun_out/prof_unet_summary.txt | cut -c1-130
rm -f gpurun_out/prof_unet/*kernel_trace.csv
