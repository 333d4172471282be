mkdir -p gpurun_out/r02c
CFHIP_LIB=tools/libcfhip_ablate.so python tools/gemm_probe_ring.py > gpurun_out/r02c/probe_ring.log 2>&1; grep -v amdgpu.ids gpurun_out/r02c/probe_ring.log
python -m pytest tests/test_gpu_gemm.py -x -q 2>&1 | tail -2
