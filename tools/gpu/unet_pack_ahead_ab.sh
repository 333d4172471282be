# rotated 3x3 filters for dX packed in the forward (CFHIP_PACK_AHEAD=2: caller's stream, 1: side lane) or in the backward (0)
mkdir -p gpurun_out/pack
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload unet --img 64 --steps 15 --warmup 4 --no-cpu-baseline 2> gpurun_out/pack/$tag.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['loss_last_step'])"; }
run warm CFHIP_PACK_AHEAD=0
for i in 1 2 3; do
run ahead2_$i CFHIP_PACK_AHEAD=2
run ahead0_$i CFHIP_PACK_AHEAD=0
done
