#!/bin/bash
# tools/kernel_resources.sh [extra hipcc flags]: registers / scratch / LDS / occupancy of the production kernels (hipcc remarks; no GPU needed)
cd "$(dirname "$0")/.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-function "$@" -Rpass-analysis=kernel-resource-usage -c sumcheck_amd/csrc/kernels.hip -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
name=None; d={}
for line in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        if 'error' in line: print(line, end='')
        continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'): name=t.split(':',1)[1].strip(); d[name]={}
    elif name and ':' in t:
        k,v=t.split(':',1); d[name][k.strip()]=v.strip()
for n,v in d.items():
    if any(x in n for x in ('round1_tree_split','round_tree_split','tail_rounds','finalize_mb','k_resident')):
        print(re.sub(r'^_ZN3scd\d+','',n)[:28].ljust(28), 'VGPR', v.get('VGPRs'), 'scratch', v.get('ScratchSize [bytes/lane]'), 'occ', v.get('Occupancy [waves/SIMD]'), 'LDS', v.get('LDS Size [bytes/block]'))
"
