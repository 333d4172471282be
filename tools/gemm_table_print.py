"""Print the `roofline.gemm_table` of a `bench.py --workload unet|clip --gemm-table` line.   python tools/gemm_table_print.py <bench.json>"""
import json, sys

j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = j["roofline"]["gemm_table"]
print(f"{j['config']['workload'][:60]}: step {j['ms_per_step']} ms; GEMM launches {t['kernel_ms_per_step']} ms/step, "
      f"{t['flops_per_step'] / 1e12:.2f} TFLOP/step -> {t['flops_per_step'] / t['kernel_ms_per_step'] / 1e9:.0f} TFLOP/s in step")
print(f"{'layout':<7}{'M':>7}{'N':>6}{'K':>7} epi {'n/step':>7}{'us':>8}{'TF/s':>7}{'ms/step':>9}")
for r in t["rows"]:
    if r["layout"] == "tn-grouped":
        print(f"grouped {r['problems']} problems {r['tiles']} tiles K {r['K']}: {r['launches_per_step']}/step {r['us']} us {r['tflops']} TF/s {r['ms_per_step']} ms")
    else:
        print(f"{r['layout']:<7}{r['M']:>7}{r['N']:>6}{r['K']:>7} {r['epilogue']:>3} {r['launches_per_step']:>7}{r['us']:>8}{r['tflops']:>7}{r['ms_per_step']:>9}")
