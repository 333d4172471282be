mkdir -p gpurun_out/cfgtest
timeout 600 python -m pytest tests/test_gpu_gemm.py -x -q -m gpu > gpurun_out/cfgtest/pytest_gemm.log 2>&1; tail -5 gpurun_out/cfgtest/pytest_gemm.log
timeout 300 python tools/gemm_bench.py --batch 128 --configs 1,14 > gpurun_out/cfgtest/gemm_bench.log 2>&1; grep -v amdgpu gpurun_out/cfgtest/gemm_bench.log | cut -c1-200 | grep "^tn\|config"
