#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
R=$PWD
mkdir -p gpurun_out/pmc2
rocprofv3 -L 2>/dev/null | grep -oE "SQ_[A-Z_0-9]+" | sort -u | tr '\n' ' ' | head -c 3000 > gpurun_out/pmc2/sq_counters.txt
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES" "GRBM_GUI_ACTIVE GRBM_COUNT"; do
  tag=$(echo $set | cut -d' ' -f1)
  ( cd /tmp && timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $R/gpurun_out/pmc2/$tag -o p -- python $R/tools/gemm_one.py nt 12608 2304 768 bias ) > gpurun_out/pmc2/$tag.log 2>&1
  f=$(ls gpurun_out/pmc2/$tag/*counter_collection.csv 2>/dev/null | head -1)
  [ -n "$f" ] && python - "$f" <<'PY'
import csv, sys, collections
d = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if 'gemm_bf16_kernel' in r.get('Kernel_Name', ''):
        d[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in d.items():
    print(f"{k:34s} {sum(v)/len(v):16.0f}")
PY
  [ -z "$f" ] && tail -3 gpurun_out/pmc2/$tag.log
done
