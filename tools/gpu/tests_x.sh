#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python -m pytest tests -m gpu -x -q --tb=short -p no:cacheprovider > gpurun_out/pytest_gpu.log 2>&1; echo "== pytest -m gpu -x exit $?"; tail -n 6 gpurun_out/pytest_gpu.log | cut -c1-250
