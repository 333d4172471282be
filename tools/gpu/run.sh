#!/bin/bash
# ONE parametrised script for everything that runs on the GPU box (replaces the 40 one-off scripts of rounds 1-2):
#   gpurun --timeout 900 -- 'bash tools/gpu/run.sh <recipe> [args...] ; bash tools/gpu/run.sh <recipe> ...'
# Every recipe writes under gpurun_out/<tag>/ (merged back by gpurun); copy what should be judged into profiles/rNN/.
#
#   tests [pytest args]            pytest -m gpu (default: the whole suite)            -> gpurun_out/tests/pytest.log
#   bench <tag> [bench.py args]    one bench.py run                                    -> gpurun_out/<tag>/bench.json
#   prof <tag> [bench.py args]     rocprofv3 --kernel-trace --stats of bench.py        -> gpurun_out/<tag>/prof_summary.txt
#   profw <tag> <n> [bench.py args]  the same for --workload unet | clip (n = steps + warm-up + 1 FLOP-count step in the trace)
#   pmc <tag> [bench.py args]      HBM bytes: FETCH_SIZE and WRITE_SIZE in separate passes -> gpurun_out/<tag>/pmc_step.json
#   trace <tag> [bench.py args]    kernel timeline (start / end / stream per launch)   -> gpurun_out/<tag>/timeline.csv.gz
#   py <tag> <script> [args]       python <script> args                                -> gpurun_out/<tag>/<script>.log
#   ddp <tag> <n> [bench.py args]  bench.py --gpus n self-launched (gloo ranks on one GPU when n > visible GPUs)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
recipe=$1; shift
case "$recipe" in
  tests)
    mkdir -p gpurun_out/tests
    if [ $# -eq 0 ]; then set -- tests; fi
    timeout ${TEST_TIMEOUT:-1500} python -m pytest "$@" -x -q -m gpu > gpurun_out/tests/pytest.log 2>&1
    echo "== tests exit $?"; tail -n 6 gpurun_out/tests/pytest.log | cut -c1-300 ;;
  bench)
    tag=$1; shift; mkdir -p gpurun_out/$tag
    timeout ${BENCH_TIMEOUT:-600} python bench.py "$@" > gpurun_out/$tag/bench.json 2> gpurun_out/$tag/bench.err
    echo "== bench $tag exit $?"; tail -n 1 gpurun_out/$tag/bench.json | cut -c1-600; tail -n 3 gpurun_out/$tag/bench.err | cut -c1-300 ;;
  prof)
    tag=$1; shift; OUT=gpurun_out/$tag; mkdir -p $OUT
    ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o step -- python "$R/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads --no-calibration "$@" ) > $OUT/prof.log 2>&1
    f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python tools/prof_summary.py "$f" 7 > $OUT/prof_summary.txt && cp "$f" $OUT/kernel_stats.csv
    head -40 $OUT/prof_summary.txt | cut -c1-150; rm -rf $OUT/prof ;;
  profw)  # rocprofv3 kernel stats of another workload: profw <tag> <steps-in-trace> [bench.py args, e.g. --workload unet --img 64 --steps 3 --warmup 2]
    tag=$1; nsteps=$2; shift 2; OUT=gpurun_out/$tag; mkdir -p $OUT
    ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/$OUT/prof" -o step -- python "$R/bench.py" "$@" ) > $OUT/prof.log 2>&1
    f=$(ls $OUT/prof/*kernel_stats.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python tools/prof_summary.py "$f" $nsteps > $OUT/prof_summary.txt && cp "$f" $OUT/kernel_stats.csv
    head -32 $OUT/prof_summary.txt | cut -c1-150; tail -2 $OUT/prof.log | cut -c1-300; rm -rf $OUT/prof ;;
  pmc)
    tag=$1; shift; OUT=gpurun_out/$tag; mkdir -p $OUT
    for ctr in FETCH_SIZE WRITE_SIZE; do
      ( cd /tmp && timeout 900 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/$OUT/$ctr -o p -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-other-workloads --no-calibration "$@" ) > $OUT/$ctr.log 2>&1
      echo "== $ctr exit $?"; tail -n 2 $OUT/$ctr.log | cut -c1-300
    done
    f=$(ls $OUT/FETCH_SIZE/*counter_collection.csv | head -1); w=$(ls $OUT/WRITE_SIZE/*counter_collection.csv | head -1)
    python tools/pmc_step_summary.py "$f" "$w" 3 > $OUT/pmc_step.json; cat $OUT/pmc_step.json | cut -c1-1500
    rm -rf $OUT/FETCH_SIZE $OUT/WRITE_SIZE ;;
  trace)
    tag=$1; shift; OUT=gpurun_out/$tag; mkdir -p $OUT
    ( cd /tmp && timeout 400 rocprofv3 --kernel-trace --output-format csv -d "$R/$OUT/prof" -o step -- python "$R/bench.py" --steps 3 --warmup 2 --no-cpu-baseline --no-roofline --no-other-workloads --no-calibration "$@" ) > $OUT/prof.log 2>&1
    f=$(ls $OUT/prof/*kernel_trace.csv 2>/dev/null | head -1); [ -n "$f" ] && python tools/trace_reduce.py "$f" $OUT/timeline.csv && gzip -f $OUT/timeline.csv
    rm -rf $OUT/prof; tail -3 $OUT/prof.log ;;
  py)
    tag=$1; script=$2; shift 2; mkdir -p gpurun_out/$tag
    log=gpurun_out/$tag/$(basename $script .py)${LOG_SUFFIX}.log
    timeout ${PY_TIMEOUT:-600} python $script "$@" > $log 2>&1
    echo "== $script exit $?"; tail -n ${PY_TAIL:-40} $log | cut -c1-400 ;;
  ddp)
    tag=$1; n=$2; shift 2; mkdir -p gpurun_out/$tag
    timeout 600 python bench.py --gpus $n --no-cpu-baseline --no-roofline --no-other-workloads --no-calibration "$@" > gpurun_out/$tag/bench_ddp$n.json 2> gpurun_out/$tag/bench_ddp$n.err
    echo "== ddp $n exit $?"; tail -n 1 gpurun_out/$tag/bench_ddp$n.json | cut -c1-600; tail -n 4 gpurun_out/$tag/bench_ddp$n.err | cut -c1-300 ;;
  *) echo "unknown recipe $recipe"; exit 2 ;;
esac
