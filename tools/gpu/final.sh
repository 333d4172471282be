#!/bin/bash
# round-end evidence -> gpurun_out/final/: full GPU test suite, smoke, bench JSON (default: eager, batch 128), batch-64 /
# batch-256 / host-input variants, the N > 1 dry run (2 gloo ranks on one GPU), rocprofv3 kernel stats of the bench
# command, PMC passes (FETCH_SIZE / WRITE_SIZE) for the HBM-side traffic.  Every step is time-limited.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp MASTER_ADDR=127.0.0.1
O=gpurun_out/final
mkdir -p $O
python -c "import torch; print(torch.__version__, torch.cuda.get_device_name(0))" > $O/env.log 2>&1
rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8 >> $O/env.log
nproc >> $O/env.log
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > $O/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 3 $O/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 1 $O/smoke.log
timeout 420 python bench.py > $O/bench.json 2> $O/bench.err; echo "== bench exit $?"; grep bench $O/bench.err | tail -4
timeout 200 python bench.py --batch 64 --no-cpu-baseline > $O/bench_b64.json 2>/dev/null; echo "== b64 exit $?"
timeout 200 python bench.py --batch 256 --no-cpu-baseline --no-roofline > $O/bench_b256.json 2>/dev/null; echo "== b256 exit $?"
timeout 200 python bench.py --input host --no-cpu-baseline --no-roofline > $O/bench_hostinput.json 2>/dev/null; echo "== host-input exit $?"
python - <<'PY'
import json
for n in ('bench', 'bench_b64', 'bench_b256', 'bench_hostinput'):
    try:
        d=json.loads(open(f'gpurun_out/final/{n}.json').read().strip().split('\n')[-1]); r=d.get('roofline', {})
        print(n, d['value'], d['ms_per_step'], r.get('achieved'), r.get('wall', {}).get('achieved'), r.get('isolated', {}).get('achieved'), d.get('cpu_baseline', {}).get('value'))
    except Exception as e: print(n, 'failed', e)
PY
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 4 --warmup 2 --batch 16 --backend gloo --all-on-gpu0 --no-roofline --no-cpu-baseline > $O/ddp_dryrun_2ranks_gloo.log 2>&1; echo "== ddp dry run exit $?"; tail -n 1 $O/ddp_dryrun_2ranks_gloo.log | cut -c1-300
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$OLDPWD/$O/prof" -o step -- python "$OLDPWD/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline ) > $O/prof.log 2>&1
f=$(ls $O/prof/*kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/prof_summary.py "$f" 7 > $O/prof_summary.txt && cp "$f" $O/rocprofv3_kernel_stats_b128_7steps.csv; head -16 $O/prof_summary.txt; rm -rf $O/prof
R=$PWD
for ctr in FETCH_SIZE WRITE_SIZE; do
  ( cd /tmp && timeout 400 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$O/pmc/$ctr -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline ) > $O/pmc_$ctr.log 2>&1
  echo "== $ctr exit $?"
done
f=$(ls $O/pmc/FETCH_SIZE/*counter_collection.csv 2>/dev/null | head -1); w=$(ls $O/pmc/WRITE_SIZE/*counter_collection.csv 2>/dev/null | head -1)
[ -n "$f" ] && [ -n "$w" ] && python tools/pmc_step_summary.py "$f" "$w" 3 > $O/pmc_step_b128.json && head -c 900 $O/pmc_step_b128.json
rm -rf $O/pmc
