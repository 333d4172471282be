#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
./tools/micro/store_pattern
( cd /tmp && rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/sp -o p -- $OLDPWD/tools/micro/store_pattern ) > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
f = glob.glob("/tmp/sp/**/*counter_collection.csv", recursive=True)[0]
d = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    d[r["Kernel_Name"][:48]].append(float(r["Counter_Value"]))
for k, v in d.items():
    print(k, "FETCH_SIZE KB mean", round(sum(v) / len(v), 1))
PY
