#!/bin/bash
# quick check: all GPU tests + default bench (no cpu baseline) + batch 64
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider -x > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 3 gpurun_out/pytest_gpu.log
timeout 600 python bench.py --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('b128', d['value'], d['ms_per_step'])"
timeout 600 python bench.py --no-cpu-baseline --no-roofline --batch 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('b64', d['value'], d['ms_per_step'])"
