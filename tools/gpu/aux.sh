mkdir -p gpurun_out/aux
python tools/gemm_shapes_time.py --reps 5 > /dev/null 2>&1
for r in 1 2; do
for v in default stsc1 stnt stsc1ldnt stntldnt; do
  if [ $v = default ]; then L=""; else L="tools/libcfhip_$v.so"; fi
  CFHIP_LIB=$L timeout 120 python tools/gemm_shapes_time.py --reps 20 2>&1 | grep -v amdgpu > gpurun_out/aux/$v.$r.log
  echo "$v $r: $(tail -1 gpurun_out/aux/$v.$r.log)"
done; done
paste -d'|' <(cut -c1-52 gpurun_out/aux/default.2.log) <(cut -c40-52 gpurun_out/aux/stsc1.2.log) <(cut -c40-52 gpurun_out/aux/stnt.2.log) <(cut -c40-52 gpurun_out/aux/stsc1ldnt.2.log) <(cut -c40-52 gpurun_out/aux/stntldnt.2.log)
