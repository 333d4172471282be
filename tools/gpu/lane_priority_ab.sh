# HIP stream priority of the weight-gradient lane (0) and the second batch slice's lane (1): separate processes, alternating
run() { tag=$1; shift; env "$@" timeout 200 python bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-roofline --no-other-workloads 2>/dev/null | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], d['step_ms']['median'])"; }
run warm A=1
for i in 1 2; do
run default_$i A=1
run dw_low_$i CFHIP_LANE_PRIORITY=0:1
run dw_high_$i CFHIP_LANE_PRIORITY=0:-1
run slice_high_$i CFHIP_LANE_PRIORITY=1:-1
done
