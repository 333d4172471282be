#!/bin/bash
# round-end evidence: full GPU test suite, smoke, bench JSON (graph), eager bench, rocprofv3 kernel stats
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 1 gpurun_out/smoke.log
timeout 900 python bench.py --steps 30 --warmup 5 > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "== bench exit $?"; cat gpurun_out/bench.err | grep bench; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.json').read().strip().split('\n')[-1])
print({k:v for k,v in d.items() if k not in ('config','roofline','cpu_baseline')})
r=d.get('roofline',{}); print({k:v for k,v in r.items() if k!='shapes'}); print(d.get('cpu_baseline'))
PY
timeout 600 python bench.py --steps 20 --warmup 3 --no-graph --no-cpu-baseline --no-roofline > gpurun_out/bench_eager.json 2>/dev/null; echo "== eager exit $?"; python -c "
import json; d=json.loads(open('gpurun_out/bench_eager.json').read().strip().split('\n')[-1]); print('eager', d['value'], d['ms_per_step'])"
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o step -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-graph --no-cpu-baseline --no-roofline ) > gpurun_out/prof.log 2>&1
f=$(ls gpurun_out/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" 7 > gpurun_out/prof_summary.txt; head -12 gpurun_out/prof_summary.txt
