#!/bin/bash
# Is the step power-limited?  Samples rocm-smi (average socket power, sclk, mclk, temperature) every ~0.25 s while the ViT step runs for
# ~12 s, and once idle before / after:  bash tools/gpu/power_probe.sh > gpurun_out/power_probe.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
smi() { rocm-smi --showpower --showclocks --showtemp --showperflevel 2>/dev/null | grep -E "Power|sclk|mclk|Temperature \(Sensor (edge|junction|hotspot)|Performance Level" | tr -s ' ' | tr '\n' '|'; echo; }
echo "idle:   $(smi)"
rocm-smi --showmaxpower 2>/dev/null | grep -i "max" | head -3
python bench.py --steps 600 --warmup 20 --no-cpu-baseline --no-roofline --no-other-workloads > /tmp/pp_bench.json 2>/tmp/pp_bench.err &
BP=$!
sleep 8   # import + model build
for i in $(seq 1 40); do
  kill -0 $BP 2>/dev/null || break
  echo "t=$i:  $(smi)"
  sleep 0.25
done
wait $BP
tail -1 /tmp/pp_bench.json | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print("bench:", d["ms_per_step"], "ms/step over", d["steps"], "steps;", d["step_ms"])'
echo "after:  $(smi)"
