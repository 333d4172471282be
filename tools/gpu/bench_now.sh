mkdir -p gpurun_out/bench_now
timeout 420 python bench.py --steps 20 --warmup 5 > gpurun_out/bench_now/bench.json 2> gpurun_out/bench_now/bench.err; echo "rc=$?"; tail -6 gpurun_out/bench_now/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench_now/bench.json').read().strip().split('\n')[-1])
print({k:v for k,v in d.items() if k not in ('config','roofline','cpu_baseline')})
r=d.get('roofline',{}); print({k:v for k,v in r.items() if k not in ('shapes','shapes_isolated')}); print(d.get('cpu_baseline'))
for row in r.get('shapes', [])[:14]: print(row)
PY
