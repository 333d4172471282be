mkdir -p gpurun_out/opt
timeout 500 python tools/option_ab.py $1 $2 ${3:-128} > gpurun_out/opt/$1.log 2>&1; grep -v amdgpu gpurun_out/opt/$1.log | cut -c1-420 | tail -14
