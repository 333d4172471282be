#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc
R=$PWD
for shape in "nt 12608 768 3072 residual" "nt 12608 2304 768 bias" "nt 12608 3072 768 gelu" "tn 3072 768 12608 none"; do
  tag=$(echo $shape | tr ' ' '_')
  for ctr in FETCH_SIZE WRITE_SIZE; do
    ( cd /tmp && timeout 300 rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $R/gpurun_out/pmc/${tag}_$ctr -o p -- python $R/tools/gemm_one.py $shape ) > gpurun_out/pmc/${tag}_$ctr.log 2>&1
    f=$(ls gpurun_out/pmc/${tag}_$ctr/*counter_collection.csv 2>/dev/null | head -1)
    [ -n "$f" ] && python - "$f" "$ctr" "$tag" <<'PY'
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if 'gemm_bf16_kernel' in r.get('Kernel_Name', '')]
vals = [float(r['Counter_Value']) for r in rows if r.get('Counter_Name') == sys.argv[2]]
if vals: print(sys.argv[3], sys.argv[2], 'launches', len(vals), 'mean', sum(vals) / len(vals))
else: print(sys.argv[3], sys.argv[2], 'no rows; columns:', list(rows[0].keys()) if rows else 'none')
PY
  done
done
