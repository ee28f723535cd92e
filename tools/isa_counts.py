#!/usr/bin/env python3
"""Static instruction mix per kernel of a gfx950 assembly file (hipcc --save-temps): VALU / MFMA / LDS / VMEM / SALU counts.
Straight-line (unrolled) kernels only: loops are counted once.   tools/isa_counts.py file.s [name-filter ...]"""
import re
import subprocess
import sys

src = open(sys.argv[1]).read()
filters = sys.argv[2:]
rows = []
for m in re.finditer(r'^(_Z\w+):.*?\n(.*?)^\.Lfunc_end', src, re.S | re.M):
    name = subprocess.run(['c++filt', m.group(1)], capture_output=True, text=True).stdout.strip()
    if filters and not all(f in name for f in filters):
        continue
    ins = [l.split()[0] for l in m.group(2).split('\n') if l.startswith('\t') and l.strip() and l.strip()[0] not in '.;']
    valu = sum(1 for i in ins if i.startswith('v_') and 'mfma' not in i)
    rows.append((re.sub(r'^void scs::|scs::|\(.*', '', name)[:100], valu, sum('mfma' in i for i in ins), sum(i.startswith('ds_') for i in ins),
                 sum(i.startswith(('global_', 'buffer_', 'flat_')) for i in ins), sum(i.startswith('s_') for i in ins)))
for r in rows:
    print('%-100s valu %5d mfma %4d lds %4d vmem %4d salu %5d' % r)
