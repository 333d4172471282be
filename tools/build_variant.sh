#!/bin/bash
# Benchmark-only builds of the C-ABI with extra -D flags, written next to the tools (never the product library):
#   tools/build_variant.sh ablate -DCFHIP_ABLATE        -> tools/libcfhip_ablate.so  (phase-timing ablation masks)
#   tools/build_variant.sh pf8 -DCFHIP_PF_BF16=8 ...    -> tools/libcfhip_pf8.so
# Select one with CFHIP_LIB=tools/libcfhip_<name>.so in front of any tool.
set -e
NAME=$1; shift
ROOT="$(dirname "$(readlink -f "$0")")/.."
cd "$ROOT/carefree-learn_amd/csrc"
mkdir -p /tmp/cfhip_build_$NAME
pids=()
for f in errors gemm gemm_grouped attn attn_probs norm elementwise conv conv_grouped embed random tabular comm; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics "$@" -c $f.hip -o /tmp/cfhip_build_$NAME/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/libcfhip_$NAME.so" /tmp/cfhip_build_$NAME/*.o -ldl
echo "built $ROOT/tools/libcfhip_$NAME.so"
