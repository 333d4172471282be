mkdir -p gpurun_out/r02a
python -m pytest tests -m gpu -x -q > gpurun_out/r02a/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02a/pytest.log
python tools/gemm_shapes_time.py > gpurun_out/r02a/gemm_default.log 2>&1; tail -25 gpurun_out/r02a/gemm_default.log
CFHIP_LIB=tools/libcfhip_pf8.so python tools/gemm_shapes_time.py > gpurun_out/r02a/gemm_pf8.log 2>&1; tail -1 gpurun_out/r02a/gemm_pf8.log
python tools/gemm_bench.py --batch 128 --configs 7,8 > gpurun_out/r02a/gemm_bench.log 2>&1
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err; tail -3 gpurun_out/r02a/bench.err; python -c "import json; d=json.load(open('gpurun_out/r02a/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
