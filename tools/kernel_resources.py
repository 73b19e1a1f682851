#!/usr/bin/env python
"""Print VGPR/AGPR/spill/occupancy of every kernel of one .hip file (hipcc -Rpass-analysis=kernel-resource-usage).
    python tools/kernel_resources.py ttscube_amd/csrc/resblock.hip [filter]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
cmd = ['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '--offload-arch=gfx950'] + [a for a in sys.argv if a.startswith('-D')] + [
       '-Rpass-analysis=kernel-resource-usage', '-c', src, '-o', '/tmp/_kr.o']
out = subprocess.run(cmd, capture_output=True, text=True).stderr
cur = None
rows = []
for line in out.splitlines():
    m = re.search(r'remark:\s+(.*?) \[-Rpass', line)
    if not m:
        continue
    t = m.group(1).strip()
    if t.startswith('Function Name:'):
        name = t.split(':', 1)[1].strip()
        name = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
        cur = {'name': name}
        rows.append(cur)
    elif cur is not None and ':' in t:
        k, v = t.split(':', 1)
        cur[k.strip()] = v.strip()
for r in rows:
    if flt and flt not in r['name']:
        continue
    print('%-90s vgpr %3s agpr %3s spill %3s scratch %4s occ %s sgpr %s' % (
        r['name'][:90], r.get('VGPRs'), r.get('AGPRs'), r.get('VGPRs Spill'), r.get('ScratchSize [bytes/lane]'),
        r.get('Occupancy [waves/SIMD]'), r.get('SGPRs')))
