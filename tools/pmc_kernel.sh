#!/bin/bash
# SQ counters of one kernel: tools/pmc_kernel.sh <tag> <kernel-name-substring> -- <command...>
#   -> gpurun_out/<tag>/pmc_<substring>.txt  (per-counter mean over the matching dispatches, and the derived shares)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
tag=$1; pat=$2; shift 3
OUT=gpurun_out/$tag; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_BUSY_CYCLES --kernel-trace --output-format csv -d $R/$OUT/pmc_a -o p -- "$@" ) > $OUT/pmc_a.log 2>&1
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$OUT/pmc_b -o p -- "$@" ) > $OUT/pmc_b.log 2>&1
python - "$OUT" "$pat" <<'PY'
import csv, glob, sys, collections
out, pat = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]:
            acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
res = {k: sum(v) / len(v) for k, v in acc.items()}
lines = [f"{k:28s} {v:16.0f}  (n={len(acc[k])})" for k, v in sorted(res.items())]
wc = res.get("SQ_WAVE_CYCLES")
if wc:
    for k in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_WAIT_INST_LDS"):
        if k in res:
            lines.append(f"{k} / SQ_WAVE_CYCLES = {res[k] / wc:.3f}")
    if "SQ_VALU_MFMA_BUSY_CYCLES" in res and "SQ_BUSY_CYCLES" in res:
        lines.append(f"SQ_VALU_MFMA_BUSY_CYCLES / SQ_BUSY_CYCLES = {res['SQ_VALU_MFMA_BUSY_CYCLES'] / res['SQ_BUSY_CYCLES']:.3f} (per-SE busy cycles: indicative)")
open(f"{out}/pmc_{pat}.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/pmc_a $OUT/pmc_b
