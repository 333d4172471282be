#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out
for b in "$@"; do
timeout 300 python tools/step_ab.py $b > gpurun_out/step_ab$b.log 2>&1; echo "== ab$b exit $?"; tail -6 gpurun_out/step_ab$b.log
done
