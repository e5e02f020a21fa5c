#!/bin/bash
# usage: scratch/isa.sh [extra hipcc flags]: device-only listing of the library in /tmp/sv.s + register / spill figures of the cost kernels
cd /root/repo/stereo-vision_amd/csrc && /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -ffp-contract=off --cuda-device-only -S -o /tmp/sv.s "$@" sv_hip.hip 2>/dev/null
python3 - <<'PY'
import re
t=open('/tmp/sv.s').read()
for m in re.finditer(r'\.name:\s+(\S*k_cost2\S*)\n(?:.*\n)*?\s+\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)', t):
    print(m.group(1)[:40], 'vgpr', m.group(2), 'spill', m.group(3))
PY
