#!/bin/bash
# CLIP: the two towers side by side on two streams (one batch pipeline each) against one tower after the other in two batch slices,
# alternating processes:  bash tools/gpu/clip_towers_ab.sh > gpurun_out/clip_towers_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rnd in 1 2 3; do
  for v in 0 1 2; do
    line=$(CFHIP_CLIP_TOWERS=$v python bench.py --workload clip --steps 10 --warmup 4 2>/dev/null | tail -1)
    echo "CFHIP_CLIP_TOWERS=$v | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, host", d["host_issue_ms_per_step"], "loss", d["config"]["loss_last_step"], d["optimizer_in_backward"])')"
  done
done
