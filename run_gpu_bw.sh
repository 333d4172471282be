#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 300 python tools/bw_bench.py > gpurun_out/bw_bench.log 2>&1; echo "== exit $?"; cat gpurun_out/bw_bench.log
