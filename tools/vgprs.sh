#!/bin/bash
# usage: tools/vgprs.sh <file.hip> [pattern]  -> registers, spills, LDS and occupancy the compiler reports for each kernel of a source (same flags as the build)
f=$1; pat=${2:-.}
/opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -fvisibility=hidden -fno-gpu-rdc -Wno-unused-function -mllvm -amdgpu-mfma-vgpr-form=1 \
  -Rpass-analysis=kernel-resource-usage -c $f -o /dev/null 2>&1 | python3 -c "
import re, sys, subprocess
cur = None
for line in sys.stdin:
    m = re.search(r'remark: (.*) \[-Rpass', line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        name = re.sub(r'\(.*', '', name)
        cur = [name]; rows = cur
        print()
        print(name[:90].ljust(92), end='')
    elif any(t.startswith(k) for k in ('VGPRs:', 'AGPRs', 'VGPR Spill', 'SGPR Spill', 'Occupancy', 'LDS Size', 'ScratchSize')):
        print(t.replace(' [bytes/lane]', '').replace(' [bytes/block]', '').replace(' [waves/SIMD]', ''), end='  ')
print()
" | grep -E "$pat"
