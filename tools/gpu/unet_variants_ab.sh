#!/bin/bash
# Round 6: which of the already-built launch-tail options moves the 64^2 x 8 UNet step now that it is device-bound
# (alternating processes; ms/step, host issue, sclk / W of the timed region)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
wl="--workload unet --img 64 --steps 8 --warmup 3"
run() {
  line=$(env "$@" python bench.py $wl 2>/dev/null | tail -1)
  echo "$* | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); t=d.get("telemetry") or {}; print(d["ms_per_step"], "ms/step, host", d.get("host_issue_ms_per_step"), "sclk", t.get("sclk_mhz_avg"), "W", t.get("power_w_avg"), "loss", d["config"].get("loss_last_step"))')"
}
for rnd in 1 2; do
  run CFHIP_NOP=1
  run CFHIP_LINEAR_DW_TILES=256 CFHIP_LINEAR_DW_KERNEL=2
  run CFHIP_TAPED_NODES=1
  run CFHIP_TAPED_NODES=1 CFHIP_LINEAR_DW_TILES=256 CFHIP_LINEAR_DW_KERNEL=2
done
