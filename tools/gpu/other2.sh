bash tools/gpu/other.sh
bash tools/gpu/rccl1.sh
