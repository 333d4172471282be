#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/final
mkdir -p $O
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; grep -E "passed|failed" $O/pytest_gpu.log | tail -1
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; echo "== bench exit $?"; grep "bench +" $O/bench.err | tail -3
timeout 200 python bench.py --batch 64 --no-cpu-baseline > $O/bench_b64.json 2>/dev/null; echo "== b64 exit $?"
python - <<'PY'
import json
for n in ('bench', 'bench_b64'):
    d=json.loads(open(f'gpurun_out/final/{n}.json').read().strip().split('\n')[-1]); r=d.get('roofline', {})
    print(n, d['value'], d['ms_per_step'], r.get('achieved'), r.get('frac'), r.get('wall', {}).get('achieved'), r.get('isolated', {}).get('achieved'), r.get('traffic'), d.get('cpu_baseline', {}).get('value'))
PY
