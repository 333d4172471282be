mkdir -p gpurun_out/quant
timeout 300 python tools/gemm_quant_probe.py ${1:-8} > gpurun_out/quant/quant.log 2>&1; grep -v amdgpu gpurun_out/quant/quant.log | cut -c1-420 | tail -30
