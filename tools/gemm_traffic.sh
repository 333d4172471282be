#!/bin/bash
# tools/gemm_traffic.sh <tag> [--gn list]  -> gpurun_out/<tag>/gemm_traffic.txt
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
tag=$1; shift
OUT=gpurun_out/$tag; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_f -o p -- python $R/tools/gemm_traffic.py "$@" ) > $OUT/pmc_f.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys
out = sys.argv[1]
rows = []
for f in glob.glob(out + "/pmc_f/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            rows.append((int(r["Dispatch_Id"]), r["Kernel_Name"], r["Counter_Value"]))
rows.sort()
csv.writer(open(out + "/dispatches.csv", "w")).writerows([(k, v) for _, k, v in rows])
PY
python tools/gemm_traffic.py "$@" --join $OUT/dispatches.csv | tee $OUT/gemm_traffic.txt
rm -rf $OUT/pmc_f
