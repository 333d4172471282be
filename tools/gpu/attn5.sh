timeout 600 python -m pytest tests/test_gpu_attn.py -x -q -m gpu 2>&1 | tail -2
timeout 100 python tools/attn_vit_time.py 128 2>&1 | grep ablate
