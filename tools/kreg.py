#!/usr/bin/env python3
"""Register / scratch / occupancy table of every kernel in a .hip file (cross-compiles to gfx950 ISA, no GPU needed).
usage: tools/kreg.py vmp-for-svae_amd/csrc/vmp_mix.hip [substring filter]"""
import re, subprocess, sys, os, tempfile
src = sys.argv[1]; flt = sys.argv[2] if len(sys.argv) > 2 else ''
out = os.path.join(tempfile.gettempdir(), 'kreg_' + os.path.basename(src) + '.s')
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
subprocess.run(['/opt/rocm/bin/hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-S', '--cuda-device-only', '-I' + root + '/include',
                '-o', out, src], check=True, stderr=subprocess.DEVNULL)
name = None
for line in open(out):
    m = re.match(r'^(_Z\w+):\s', line)
    if m: name = m.group(1)
    for key in ('NumVgprs', 'ScratchSize', 'Occupancy', 'LDSByteSize'):
        m = re.match(r'^; %s: (\d+)' % key, line)
        if m and name:
            globals().setdefault('row', {})[key] = int(m.group(1))
            if key == 'Occupancy':
                dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
                dem = re.sub(r'\(anonymous namespace\)::', '', dem).split('(')[0]
                if flt in dem:
                    print('%-60s vgpr %3d scratch %4d occ %d' % (dem[:60], row['NumVgprs'], row['ScratchSize'], row['Occupancy']))
                name = None
