# GroupNorm split rule on the UNet 64^2 x 8 step: CFHIP_GN_TARGET (workgroups wanted per launch) x CFHIP_GN_MIN_SLICE
mkdir -p gpurun_out/gn
run() { tag=$1; shift; env "$@" timeout 300 python bench.py --workload unet --img 64 --steps 12 --warmup 3 --no-cpu-baseline 2> gpurun_out/gn/$tag.err | python -c "import sys,json; d=json.loads([l for l in sys.stdin if l.startswith('{')][0]); print('$tag', d['ms_per_step'], d['host_issue_ms_per_step'], d['config']['loss_last_step'])"; }
run warm A=1
run t1024 CFHIP_GN_TARGET=1024
run t2048 CFHIP_GN_TARGET=2048
run t4096 CFHIP_GN_TARGET=4096
run t512 CFHIP_GN_TARGET=512
run t2048_s1024 CFHIP_GN_TARGET=2048 CFHIP_GN_MIN_SLICE=1024
run t1024b CFHIP_GN_TARGET=1024
