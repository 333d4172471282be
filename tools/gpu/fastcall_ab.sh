# host path with / without the vectorcall entry module (CFHIP_FASTCALL=0: plain ctypes), same box, alternating runs
mkdir -p gpurun_out/fast
run() { tag=$1; wl=$2; shift 2; env "$@" timeout 300 python bench.py $wl --no-cpu-baseline --no-roofline --no-other-workloads 2> gpurun_out/fast/$tag.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], 'ms/step, host issue', d['host_issue_ms_per_step'])"; }
for rep in 1 2; do
run unet_fast$rep "--workload unet --img 64 --steps 12 --warmup 3" CFHIP_FASTCALL=1
run unet_ctypes$rep "--workload unet --img 64 --steps 12 --warmup 3" CFHIP_FASTCALL=0
done
run clip_fast "--workload clip --steps 20 --warmup 5" CFHIP_FASTCALL=1
run clip_ctypes "--workload clip --steps 20 --warmup 5" CFHIP_FASTCALL=0
run vit_fast "--steps 20 --warmup 5" CFHIP_FASTCALL=1
run vit_ctypes "--steps 20 --warmup 5" CFHIP_FASTCALL=0
run vit_fast_noplan "--steps 20 --warmup 5" CFHIP_FASTCALL=1 CFHIP_STACK_PLANS=0
run vit_ctypes_noplan "--steps 20 --warmup 5" CFHIP_FASTCALL=0 CFHIP_STACK_PLANS=0
