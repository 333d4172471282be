mkdir -p gpurun_out/aux
python tools/gemm_shapes_time.py --reps 5 > /dev/null 2>&1
V="default $*"
for r in 1 2; do
for v in $V; do
  if [ $v = default ]; then L=""; else L="tools/libcfhip_$v.so"; fi
  CFHIP_LIB=$L timeout 120 python tools/gemm_shapes_time.py --reps 20 2>&1 | grep -v amdgpu > gpurun_out/aux/$v.$r.log
  echo "$v $r: $(tail -1 gpurun_out/aux/$v.$r.log)"
done; done
for v in $V; do echo $v; cut -c1-60 gpurun_out/aux/$v.2.log | head -16; done
