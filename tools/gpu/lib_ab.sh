#!/bin/bash
# The product library against a variant build (tools/build_variant.sh <name> <flags>) on the ViT and CLIP steps, alternating processes:
#   bash tools/gpu/lib_ab.sh noslp > gpurun_out/lib_ab_noslp.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
NAME=$1
for rnd in 1 2 3; do
  for lib in "" "tools/libcfhip_$NAME.so"; do
    for wl in "--steps 20 --warmup 5 --no-cpu-baseline --no-roofline --no-other-workloads" "--workload clip --steps 8 --warmup 3"; do
      line=$(CFHIP_LIB=$lib python bench.py $wl 2>/dev/null | tail -1)
      echo "${lib:-product} | $wl | $(echo "$line" | python -c 'import json,sys; d=json.loads(sys.stdin.read()); print(d["ms_per_step"], "ms/step")')"
    done
  done
done
