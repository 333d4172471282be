mkdir -p gpurun_out/lin
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload unet --img 64 --steps 10 --warmup 3 --no-cpu-baseline 2> gpurun_out/lin/$tag.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['loss_last_step'])"; }
run base A=1
run t512k2 CFHIP_LINEAR_DW_TILES=512 CFHIP_LINEAR_DW_KERNEL=2
run t256k2 CFHIP_LINEAR_DW_TILES=256 CFHIP_LINEAR_DW_KERNEL=2
run t1024k2 CFHIP_LINEAR_DW_TILES=1024 CFHIP_LINEAR_DW_KERNEL=2
run t512k1 CFHIP_LINEAR_DW_TILES=512 CFHIP_LINEAR_DW_KERNEL=1
run t192k0 CFHIP_LINEAR_DW_TILES=192
run base2 A=1
