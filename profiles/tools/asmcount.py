#!/usr/bin/env python3
"""static instruction mix of the kernels of a device assembly file (hipcc --cuda-device-only -S): asmcount.py file.s [name filter]"""
import sys, re, subprocess
txt = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ''
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\s*s_endpgm', txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    dem = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip().replace('elba::(anonymous namespace)::', '')
    if flt and not re.search(flt, dem): continue
    ins = [l.strip() for l in body.split('\n') if l.strip() and not l.strip().startswith(('.', ';', '//')) and not l.strip().endswith(':')]
    c = lambda p: sum(1 for l in ins if l.startswith(p))
    print('%-60s valu %5d salu %5d ds %4d vmem %4d' % (dem[:60], c('v_'), c('s_'), c('ds_'), c(('global_', 'flat_', 'buffer_', 'scratch_'))))
