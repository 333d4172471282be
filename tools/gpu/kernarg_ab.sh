#!/bin/bash
# HIP_FORCE_DEV_KERNARG (kernel arguments in device memory: shorter dispatch-to-start latency per kernel) 0 / 1 on the three workloads,
# alternating, each run its own process:  bash tools/gpu/kernarg_ab.sh > gpurun_out/kernarg_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "default in this process: HIP_FORCE_DEV_KERNARG=${HIP_FORCE_DEV_KERNARG:-unset}"
for rnd in 1 2; do
  for v in 0 1; do
    for wl in "--workload unet --img 64 --steps 6 --warmup 3" "--workload clip --steps 8 --warmup 3" "--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-workloads"; do
      line=$(HIP_FORCE_DEV_KERNARG=$v python bench.py $wl 2>/dev/null | tail -1)
      echo "HIP_FORCE_DEV_KERNARG=$v | $wl | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, host", d["host_issue_ms_per_step"])')"
    done
  done
done
