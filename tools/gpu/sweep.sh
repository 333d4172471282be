mkdir -p gpurun_out/sweep
timeout 500 python tools/gemm_group_sweep.py $1 $2 > gpurun_out/sweep/group_sweep.log 2>&1; grep -v amdgpu gpurun_out/sweep/group_sweep.log | tail -34
