mkdir -p gpurun_out/rccl1
export MASTER_ADDR=127.0.0.1
for comm in cfhip; do
  timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --force-ddp --comm $comm --no-cpu-baseline --no-roofline > gpurun_out/rccl1/bench_$comm.json 2> gpurun_out/rccl1/bench_$comm.err
  python -c "import json; d=json.loads(open('gpurun_out/rccl1/bench_$comm.json').read().strip().split('\n')[-1]); print('$comm', d['ms_per_step'], d.get('allreduce_exposed_ms',{}).get('mean'), d['config']['grad_exchange'])"
  tail -3 gpurun_out/rccl1/bench_$comm.err | cut -c1-200
done
timeout 300 python -m pytest tests/test_gpu_rccl.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -2
