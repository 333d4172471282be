CFHIP_LIB=tools/libcfhip_ablate.so timeout 200 python tools/attn_vit_time.py 128 2>&1 | grep ablate
timeout 100 python tools/attn_vit_time.py 128 2>&1 | grep ablate
