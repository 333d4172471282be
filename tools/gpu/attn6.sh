timeout 600 python -m pytest tests/test_gpu_attn.py tests/test_gpu_unet.py tests/test_gpu_clip.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python tools/attn_vit_time.py 128 2>&1 | grep ablate
for a in "40 16384 8" "64 16384 8"; do timeout 120 python tools/attn_bwd_time.py $a 2>&1 | grep lib=; done
