#!/bin/bash
# reduced round-end evidence (when little GPU budget is left): full GPU suite, smoke, the default bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests -m gpu -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu exit $?"; tail -n 3 gpurun_out/pytest_gpu.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "== smoke exit $?"; tail -n 1 gpurun_out/smoke.log
timeout 400 python bench.py > gpurun_out/bench.json 2> gpurun_out/bench.err; echo "== bench exit $?"; grep "timed region\|GEMM roofline" gpurun_out/bench.err; tail -c 600 gpurun_out/bench.json
