"""A module attribute of the package against its default over alternating bench.py processes (what tools/gpu/option_ab.sh does for a
library option):   python tools/attr_ab.py ops.SPLIT_K_DEEP=False[,fused.LINEAR_DW_TILES=0] "unet64 unet256" [rounds]
The attribute is set after import and before bench.main() in a child process per run; prints one line per run."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
WL = {"unet64": ["--workload", "unet", "--img", "64", "--steps", "6", "--warmup", "3"],
      "unet256": ["--workload", "unet", "--img", "256", "--steps", "3", "--warmup", "2"],
      "clip": ["--workload", "clip", "--steps", "8", "--warmup", "3"],
      "vit": ["--steps", "20", "--warmup", "5", "--no-other-workloads", "--no-cpu-baseline"]}


def run(setting, wl):
    pre = ""
    for one in (setting.split(",") if setting else ()):  # several attributes: comma-separated
        path, value = one.split("=", 1)
        mod, attr = path.rsplit(".", 1)
        pre += f"import cflearn_amd.{mod} as _m; _m.{attr} = {value}; "
    code = (f"import sys; sys.path.insert(0, {ROOT!r}); {pre}import bench; sys.argv = ['bench.py'] + {WL[wl]!r}; bench.main()")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT).stdout.strip().splitlines()
    d = json.loads(out[-1])
    t = d.get("telemetry") or {}
    return (f"{d['ms_per_step']} ms/step, host {d.get('host_issue_ms_per_step')} loss {d['config'].get('loss_last_step')} "
            f"sclk {t.get('sclk_mhz_avg')} W {t.get('power_w_avg')}")


def main():
    setting, which = sys.argv[1], sys.argv[2].split()
    rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    for _ in range(rounds):
        for s in (setting, ""):
            for wl in which:
                print(f"{s or 'default':<52} | {wl:<8} | {run(s, wl)}", flush=True)


if __name__ == "__main__":
    main()
