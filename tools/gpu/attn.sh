mkdir -p gpurun_out/attn
timeout 900 python -m pytest tests/test_gpu_attn.py tests/test_gpu_stochastic_tabular.py tests/test_gpu_modules.py -x -q -m gpu > gpurun_out/attn/pytest.log 2>&1; tail -25 gpurun_out/attn/pytest.log
