#!/bin/bash
# the distributed code path over RCCL with one rank (the only RCCL configuration a 1-GPU box allows),
# alternating the number of ROCclr hardware queues
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p gpurun_out
run() {
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 \
  bench.py --gpus 1 --steps 20 --warmup 5 --force-ddp --no-cpu-baseline --no-roofline "$@" 2>&1 | grep -o "timed region.*\|UserWarning.*"
}
for rep in 1 2 3; do
for q in 8 4 2; do echo "ddp, GPU_MAX_HW_QUEUES=$q: $(GPU_MAX_HW_QUEUES=$q run)"; done
done 2>&1 | tee gpurun_out/rccl_1rank_queues.log
echo "no ddp (8): $(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-roofline 2>&1 | grep -o 'timed region.*')" | tee -a gpurun_out/rccl_1rank_queues.log
