"""Instruction histogram of the hot loop(s) of one kernel in a hipcc -S listing: the basic blocks between a label and the
backward branch to it.  python tools/isa_loop.py file.s <substring of mangled name> [min instructions]"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
minlen = int(sys.argv[3]) if len(sys.argv) > 3 else 60
m = re.search(r'\n(_Z[^\n]*' + re.escape(pat) + r'[^\n]*):\s*;[^\n]*\n', s)
body = s[m.end():s.index('s_endpgm', m.end())]
lines = [l.strip() for l in body.split('\n')]
labels = {}
for i, l in enumerate(lines):
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        labels[mm.group(1)] = i
def cls(i):
    if 'mfma' in i: return 'mfma'
    if i.startswith('ds_'): return 'lds'
    if i.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if i.startswith('s_'): return 'salu'
    return 'valu'
for i, l in enumerate(lines):
    mm = re.match(r'^s_cbranch\w*\s+(\.LBB\d+_\d+)|^s_branch\s+(\.LBB\d+_\d+)', l)
    if mm:
        tgt = mm.group(1) or mm.group(2)
        if tgt in labels and labels[tgt] < i:
            seg = [x.split()[0] for x in lines[labels[tgt]:i + 1] if x and not x.startswith(('.', ';')) and not x.endswith(':')]
            if len(seg) >= minlen:
                c = collections.Counter(cls(x) for x in seg)
                print(f"loop {tgt} .. line {i}: {len(seg)} instructions {dict(c)}")
                print("  ", collections.Counter(seg).most_common(28))
