mkdir -p gpurun_out/rccl1
export MASTER_ADDR=127.0.0.1
for comm in torch cfhip torch cfhip; do
  timeout 200 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 1 --steps 20 --warmup 5 --force-ddp --comm $comm --no-cpu-baseline --no-roofline > gpurun_out/rccl1/bench_$comm.json 2> gpurun_out/rccl1/bench_$comm.err
  python -c "import json; d=json.loads(open('gpurun_out/rccl1/bench_$comm.json').read().strip().split('\n')[-1]); print('$comm', d['ms_per_step'], d.get('allreduce_exposed_ms',{}).get('mean'), d['config']['grad_exchange'])"
done
timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print('no ddp', d['ms_per_step'])"
