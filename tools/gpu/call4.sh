mkdir -p gpurun_out/r02d
python -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/r02d/pytest.log
python tools/gemm_shapes_time.py > gpurun_out/r02d/gemm_default.log 2>&1; grep -v amdgpu.ids gpurun_out/r02d/gemm_default.log
python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r02d/bench.json 2> gpurun_out/r02d/bench.err; python -c "import json; d=json.load(open('gpurun_out/r02d/bench.json')); print(d['value'], d['ms_per_step'], d['roofline']['achieved'])"
