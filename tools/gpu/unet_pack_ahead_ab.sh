mkdir -p gpurun_out/pack
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload unet --img 64 --steps 12 --warmup 3 --no-cpu-baseline 2> gpurun_out/pack/$tag.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['loss_last_step'])"; }
run ahead1 CFHIP_PACK_AHEAD=1
run ahead0 CFHIP_PACK_AHEAD=0
run ahead1b CFHIP_PACK_AHEAD=1
run ahead0b CFHIP_PACK_AHEAD=0
