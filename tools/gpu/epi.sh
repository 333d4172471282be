mkdir -p gpurun_out/epi
CFHIP_LIB=tools/libcfhip_ablate.so timeout 300 python tools/gemm_probe_epi.py > gpurun_out/epi/epi.log 2>&1; grep -v amdgpu gpurun_out/epi/epi.log | tail -12
