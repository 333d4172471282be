mkdir -p gpurun_out/tail
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu > gpurun_out/tail/pytest_gemm.log 2>&1; tail -5 gpurun_out/tail/pytest_gemm.log
timeout 500 python tools/option_ab.py gemm_tail 0,1 128 > gpurun_out/tail/tail_ab.log 2>&1; grep -v amdgpu gpurun_out/tail/tail_ab.log | cut -c1-420 | tail -5
