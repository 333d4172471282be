#!/bin/bash
# round-end evidence: full GPU test suite, smoke, bench JSON (default: eager, batch 128), batch-64 and
# hipGraph variants, the N > 1 dry run (2 gloo ranks on one GPU), rocprofv3 kernel stats of the bench command
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
mkdir -p gpurun_out
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > gpurun_out/env.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> gpurun_out/env.log
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 3 gpurun_out/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 1 gpurun_out/smoke.log
timeout 900 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "== bench exit $?"; grep bench gpurun_out/bench.err; python - <<'PY'
import json
d=json.loads(open('gpurun_out/bench.json').read().strip().split('\n')[-1])
print({k:v for k,v in d.items() if k not in ('config','roofline','cpu_baseline')})
r=d.get('roofline',{}); print({k:v for k,v in r.items() if k!='shapes'}); print(d.get('cpu_baseline'))
PY
timeout 600 python bench.py --batch 64 --no-cpu-baseline > gpurun_out/bench_b64.json 2>/dev/null; echo "== b64 exit $?"
timeout 600 python bench.py --batch 256 --no-cpu-baseline --no-roofline > gpurun_out/bench_b256.json 2>/dev/null; echo "== b256 exit $?"
timeout 600 python bench.py --graph --no-cpu-baseline --no-roofline > gpurun_out/bench_graph.json 2>/dev/null; echo "== graph exit $?"
timeout 600 python bench.py --input host --no-cpu-baseline --no-roofline > gpurun_out/bench_hostinput.json 2>/dev/null; echo "== host-input exit $?"
python - <<'PY'
import json
for n in ('bench_b64', 'bench_b256', 'bench_graph', 'bench_hostinput'):
    try:
        d=json.loads(open(f'gpurun_out/{n}.json').read().strip().split('\n')[-1]); print(n, d['value'], d['ms_per_step'], d.get('roofline', {}).get('achieved'))
    except Exception as e: print(n, 'failed', e)
PY
timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --batch 16 --backend gloo --all-on-gpu0 --no-roofline > gpurun_out/ddp_dry.log 2>&1; echo "== ddp dry run exit $?"; tail -n 1 gpurun_out/ddp_dry.log | cut -c1-200
( cd /tmp && timeout 500 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/gpurun_out/prof" -o step -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline ) > gpurun_out/prof.log 2>&1
f=$(ls gpurun_out/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" 7 > gpurun_out/prof_summary.txt; head -14 gpurun_out/prof_summary.txt
