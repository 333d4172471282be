#!/bin/bash
# L2-fill bytes (FETCH_SIZE x 2, MI355X_MICROARCH.md HBM section) per launch of every kernel a command runs:
#   tools/pmc_fetch.sh <tag> -- <command...>   -> gpurun_out/<tag>/fetch.txt   (kernel, grid, launches, MB read per launch)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
tag=$1; shift 2
OUT=gpurun_out/$tag; mkdir -p $OUT
( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $R/$OUT/pmc_f -o p -- "$@" ) > $OUT/pmc_f.log 2>&1
python - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(list)
for f in glob.glob(out + "/pmc_f/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE":
            name = r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0][:100]
            acc[(name, r.get("Grid_Size", ""))].append(float(r["Counter_Value"]) * 2048.0)
lines = [f"{k[0]:<102}{k[1]:>9}{len(v):>5} {sum(v) / len(v) / 1e6:10.1f} MB" for k, v in sorted(acc.items(), key=lambda kv: -sum(kv[1]))[:30]]
open(out + "/fetch.txt", "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
PY
rm -rf $OUT/pmc_f
