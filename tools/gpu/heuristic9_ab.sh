#!/bin/bash
# gemm_heuristic 8 (r3c: four waves up to N = 1024, eight above) against 9 (four waves for every M >= 1024 GEMM) on the three workloads,
# alternating, each its own process:  bash tools/gpu/heuristic9_ab.sh > gpurun_out/heuristic9_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rnd in 1 2; do
  for h in 8 9; do
    for wl in "--workload unet --img 64 --steps 6 --warmup 3" "--workload clip --steps 8 --warmup 3" "--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-workloads"; do
      line=$(python bench.py $wl --set-option gemm_heuristic=$h 2>/dev/null | tail -1)
      echo "heuristic $h | $wl | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, host", d["host_issue_ms_per_step"])')"
    done
  done
done
