#!/bin/bash
# HBM traffic of one training step (default bench workload) from two separate PMC passes
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/pmc_step
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc_step/$ctr -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline ) > gpurun_out/pmc_step/$ctr.log 2>&1
  echo "== $ctr exit $?"; tail -n 2 gpurun_out/pmc_step/$ctr.log | cut -c1-300
done
f=$(ls gpurun_out/pmc_step/FETCH_SIZE/*counter_collection.csv | head -1); w=$(ls gpurun_out/pmc_step/WRITE_SIZE/*counter_collection.csv | head -1)
python tools/pmc_step_summary.py "$f" "$w" 3 > gpurun_out/pmc_step/pmc_step_b128.json; cat gpurun_out/pmc_step/pmc_step_b128.json
rm -rf gpurun_out/pmc_step/FETCH_SIZE gpurun_out/pmc_step/WRITE_SIZE
