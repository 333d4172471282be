mkdir -p gpurun_out/bench_now
timeout 400 python bench.py > gpurun_out/bench_now/bench.json 2> gpurun_out/bench_now/bench.err; tail -1 gpurun_out/bench_now/bench.json | cut -c1-300; python -c "
import json; d=json.loads(open('gpurun_out/bench_now/bench.json').read().strip().split('\n')[-1]); r=d['roofline']; print(d['ms_per_step'], r['achieved'], r['frac'], r['traffic'], r.get('traffic_stale'), d.get('hbm'))"
