#!/bin/bash
# functional.NHWC (the UNet's activations as NHWC rows between the convolutions) on / off, alternating processes:
#   bash tools/gpu/unet_nhwc_ab.sh > gpurun_out/unet_nhwc_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rnd in 1 2; do
  for v in 0 1; do
    for wl in "--workload unet --img 64 --steps 6 --warmup 3" "--workload unet --img 256 --steps 3 --warmup 2"; do
      line=$(CFHIP_UNET_NHWC=$v python bench.py $wl 2>/dev/null | tail -1)
      echo "CFHIP_UNET_NHWC=$v | $wl | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, host", d["host_issue_ms_per_step"], "loss", d["config"]["loss_last_step"])')"
    done
  done
done
