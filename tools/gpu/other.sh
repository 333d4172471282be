mkdir -p gpurun_out/other
timeout 200 python bench.py --workload clip --steps 10 --warmup 3 > gpurun_out/other/bench_clip.json 2> gpurun_out/other/clip.err; tail -1 gpurun_out/other/bench_clip.json | cut -c1-900
timeout 200 python bench.py --workload unet --img 64 --steps 5 --warmup 2 > gpurun_out/other/bench_unet64.json 2> gpurun_out/other/unet64.err; tail -1 gpurun_out/other/bench_unet64.json | cut -c1-900
timeout 300 python bench.py --workload unet --img 256 --steps 3 --warmup 1 > gpurun_out/other/bench_unet256.json 2> gpurun_out/other/unet256.err; tail -1 gpurun_out/other/bench_unet256.json | cut -c1-900
