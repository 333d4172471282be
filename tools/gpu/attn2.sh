mkdir -p gpurun_out/attn
timeout 600 python -m pytest tests/test_gpu_attn.py tests/test_gpu_unet.py -x -q -m gpu > gpurun_out/attn/pytest.log 2>&1; tail -3 gpurun_out/attn/pytest.log
timeout 200 python bench.py --workload unet --img 64 --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
timeout 300 python bench.py --workload unet --img 256 --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | cut -c1-400
