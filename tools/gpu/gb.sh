mkdir -p gpurun_out/gb
timeout 400 python tools/gemm_bench.py --batch 128 --configs ${1:-1,2,4,8,11,12} > gpurun_out/gb/gemm_bench.log 2>&1; grep -v amdgpu gpurun_out/gb/gemm_bench.log | cut -c1-330
