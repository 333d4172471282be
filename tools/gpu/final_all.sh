#!/bin/bash
# PMC passes first (bench.py's roofline.traffic / hbm objects read profiles/r01/pmc_step_b128.json), then the evidence run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
bash tools/gpu/pmc_step.sh > gpurun_out/pmc_step.log 2>&1; tail -n 3 gpurun_out/pmc_step.log | cut -c1-400
[ -s gpurun_out/pmc_step/pmc_step.json ] && cp gpurun_out/pmc_step/pmc_step.json profiles/r01/pmc_step_b128.json
bash tools/gpu/final.sh
