#!/bin/bash
# Compact per-kernel resource report (VGPRs / SGPRs / scratch / occupancy) for one .hip source:
#   tools/regs.sh gemm [filter-regex]
cd "$(dirname "$(readlink -f "$0")")/../carefree-learn_amd/csrc"
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics ${CFHIP_EXTRA_FLAGS} -c $1.hip -o /tmp/regs_$1.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import re, sys, subprocess
cur = None; rows = []
for line in sys.stdin:
    m = re.search(r'remark: \s*(.*?) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        cur = {'name': t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':',1); cur[k.strip()] = v.strip()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
flt = re.compile(sys.argv[1]) if len(sys.argv) > 1 else None
for r, n in zip(rows, names):
    n = n.replace('(anonymous namespace)::', '')
    if flt and not flt.search(n): continue
    print(f\"{n[:110]:110s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} sgpr {r.get('TotalSGPRs','?'):>4} scratch {r.get('ScratchSize [bytes/lane]','?'):>4} occ {r.get('Occupancy [waves/SIMD]','?'):>2} lds {r.get('LDS Size [bytes/block]','?')}\")
" "${2:-}"
