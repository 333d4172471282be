mkdir -p gpurun_out/ab
timeout 300 python tools/step_ab.py 128 > gpurun_out/ab/step_ab.log 2>&1; tail -4 gpurun_out/ab/step_ab.log
