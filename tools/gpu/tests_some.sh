mkdir -p gpurun_out/tests
timeout 500 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_modules.py tests/test_gpu_train.py -x -q -s > gpurun_out/tests/pytest_some.log 2>&1; echo "pytest rc=$?"; grep -v "amdgpu.ids" gpurun_out/tests/pytest_some.log | tail -30
