#!/bin/bash
# One environment switch on / off over alternating processes of the UNet / CLIP workloads:
#   bash tools/gpu/env_ab.sh CFHIP_GN_GRADS_ASIDE "unet64 unet256" > gpurun_out/env_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
var=$1; which=${2:-"unet64"}
for rnd in 1 2 3; do
  for v in 0 1; do
    for w in $which; do
      case $w in
        unet64) wl="--workload unet --img 64 --steps 6 --warmup 3";;
        unet256) wl="--workload unet --img 256 --steps 3 --warmup 2";;
        clip) wl="--workload clip --steps 8 --warmup 3";;
        vit) wl="--steps 20 --warmup 5 --no-other-workloads";;
      esac
      line=$(env $var=$v python bench.py $wl 2>/dev/null | tail -1)
      echo "$var=$v | $w | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step, host", d.get("host_issue_ms_per_step"), "loss", d["config"].get("loss_last_step"))')"
    done
  done
done
