#!/bin/bash
# Build libcfhip.so (gfx950 only) in-tree: carefree-learn_amd/libcfhip.so
set -e
cd "$(dirname "$0")/carefree-learn_amd/csrc"
OUT=../libcfhip.so
mkdir -p ../_build
pids=()
for f in errors gemm gemm_grouped attn attn_probs norm elementwise conv conv_grouped embed random tabular comm; do
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -c $f.hip -o ../_build/$f.o &
  pids+=($!)
done
for p in "${pids[@]}"; do wait $p; done
hipcc --offload-arch=gfx950 -shared -fPIC -o $OUT ../_build/*.o -ldl
echo "built $(realpath $OUT)"
# the vectorcall entry module of the host path (generated from _lib.SIGNATURES)
python3 gen_fastcall.py ../_build/fastcall_gen.c
gcc -O2 -shared -fPIC $(python3-config --includes) ../_build/fastcall_gen.c -o ../_cfhip_fast$(python3 -c "import sysconfig; print(sysconfig.get_config_var('EXT_SUFFIX'))")
echo "built _cfhip_fast"
