mkdir -p gpurun_out/stag
timeout 400 python tools/stagger_ab.py 128 > gpurun_out/stag/stagger_ab.log 2>&1; grep -v amdgpu gpurun_out/stag/stagger_ab.log | cut -c1-400 | tail -14
