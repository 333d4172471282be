#!/bin/bash
# One cfhip_set_option value against the default over alternating processes of the UNet / CLIP / ViT workloads:
#   bash tools/gpu/option_ab.sh conv_form=-2 "unet64 unet256" [rounds] > gpurun_out/option_ab.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
opt=$1; which=${2:-"unet64"}; rounds=${3:-3}
for rnd in $(seq 1 $rounds); do
  for v in A B; do
    for w in $which; do
      case $w in
        unet64) wl="--workload unet --img 64 --steps 6 --warmup 3";;
        unet256) wl="--workload unet --img 256 --steps 3 --warmup 2";;
        clip) wl="--workload clip --steps 8 --warmup 3";;
        vit) wl="--steps 20 --warmup 5 --no-other-workloads --no-cpu-baseline";;
      esac
      if [ $v = A ]; then extra="--set-option $opt"; tag="$opt"; else extra=""; tag="default"; fi
      line=$(python bench.py $wl $extra 2>/dev/null | tail -1)
      echo "$tag | $w | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); t=d.get("telemetry") or {}; print(d["ms_per_step"], "ms/step, host", d.get("host_issue_ms_per_step"), "loss", d["config"].get("loss_last_step"), "sclk", t.get("sclk_mhz_avg"), "W", t.get("power_w_avg"))')"
    done
  done
done
