mkdir -p gpurun_out/r02b
python -m pytest tests/test_gpu_norm_elem.py tests/test_gpu_train.py -x -q > gpurun_out/r02b/pytest.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r02b/pytest.log
python tools/ln_bench.py > gpurun_out/r02b/ln_bench.log 2>&1; cat gpurun_out/r02b/ln_bench.log | grep -v amdgpu.ids
python tools/step_ab.py 128 > gpurun_out/r02b/step_ab.log 2>&1; tail -4 gpurun_out/r02b/step_ab.log
