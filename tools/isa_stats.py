#!/usr/bin/env python
"""Instruction mix per kernel of a .hip file (device ISA via hipcc -S).   python tools/isa_stats.py file.hip [filter] [--dump name.s]"""
import re
import subprocess
import sys

src = sys.argv[1]
flt = sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith('-') else ''
subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '-ffp-contract=off', '--offload-arch=gfx950'] + [a for a in sys.argv if a.startswith('-D')] + [ '-S',
                '--cuda-device-only', '-o', '/tmp/_isa.s', src], capture_output=True)
s = open('/tmp/_isa.s').read()
parts = re.split(r'\n(?=_Z[^\n]*:\s*; @)', s)
for f in parts[1:]:
    mangled = f.split(':')[0]
    name = subprocess.run(['c++filt', mangled], capture_output=True, text=True).stdout.strip()
    if flt and flt not in name:
        continue
    body = f.split('s_endpgm')[0]
    n = lambda pat: len(re.findall(pat, body))
    print('%-70s lines %5d mfma %4d ds_read %4d ds_write %4d gload %4d gstore %4d waitcnt %4d vmcnt0 %3d barrier %2d scratch %3d valu %5d salu %5d' % (
        name[:70], body.count('\n'), n(r'v_mfma'), n(r'ds_read'), n(r'ds_write'), n(r'global_load'), n(r'global_store'),
        n(r's_waitcnt'), n(r'vmcnt\(0\)'), n(r's_barrier'), n(r'scratch_'), n(r'\n\s+v_(?!mfma)'), n(r'\n\s+s_(?!waitcnt|barrier|nop)')))
    if '--dump' in sys.argv:
        open(sys.argv[sys.argv.index('--dump') + 1], 'w').write(f)
