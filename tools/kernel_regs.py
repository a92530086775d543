#!/usr/bin/env python
"""Register / occupancy table of the kernels in a hipcc -S listing: python tools/kernel_regs.py file.s"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
for m in re.finditer(r'\n(_ZN3dva\S+):', txt):
    name = m.group(1)
    tail = txt[m.end():]
    meta = tail[tail.index('.Lfunc_end'):][:6000]
    get = lambda k: (re.search(r'; %s: (\d+)' % k, meta) or re.search(r'(\?)', '?')).group(1)
    try:
        dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.split('(')[0]
    except FileNotFoundError:
        dn = name
    print(f"{dn[:78]:78s} vgpr {get('NumVgprs'):>4s} agpr {get('NumAgprs'):>3s} scratch {get('ScratchSize'):>4s} "
          f"occ {get('Occupancy')}")
