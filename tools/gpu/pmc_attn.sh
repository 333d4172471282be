#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_attn
R=$PWD
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES"; do
  i=$((i+1))
  ( cd /tmp && timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc_attn/p$i -o p -- python $R/tools/attn_vit_time.py 128 ) > gpurun_out/pmc_attn/p$i.log 2>&1
  f=$(ls gpurun_out/pmc_attn/p$i/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
acc = collections.defaultdict(lambda: [0, 0.0])
for r in csv.DictReader(open(sys.argv[1])):
    k = r.get('Kernel_Name', '')
    if 'attn' not in k: continue
    k = k.split('(')[0].replace('(anonymous namespace)::', '').replace('void ', '')[:40]
    a = acc[(k, r['Counter_Name'])]
    a[0] += 1; a[1] += float(r['Counter_Value'])
for (k, c), (n, v) in sorted(acc.items()):
    print(f"{k:42s} {c:28s} launches {n:4d} mean {v / n:14.0f}")
PY
  tail -2 gpurun_out/pmc_attn/p$i.log | cut -c1-200
  rm -rf gpurun_out/pmc_attn/p$i
done
