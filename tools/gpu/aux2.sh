mkdir -p gpurun_out/aux2
python bench.py --steps 5 --warmup 2 --no-cpu-baseline > /dev/null 2>&1
for r in 1 2; do
for v in default stnt st3 st18 stsc1; do
  if [ $v = default ]; then L=""; else L="tools/libcfhip_$v.so"; fi
  CFHIP_LIB=$L timeout 200 python bench.py --steps 40 --warmup 10 --no-cpu-baseline --no-roofline 2>/dev/null | tail -1 > gpurun_out/aux2/$v.$r.json
  echo "$v $r: $(python -c "import json,sys; d=json.load(open('gpurun_out/aux2/$v.$r.json')); print(d['ms_per_step'], d['value'])")"
done; done
