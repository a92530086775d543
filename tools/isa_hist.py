"""Instruction histogram of one kernel in a hipcc -S listing: python tools/isa_hist.py file.s <substring of mangled name>"""
import collections, re, sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'\n(_Z[^\n]*' + re.escape(pat) + r'[^\n]*):\s*;[^\n]*\n', s)
start = m.end()
end = s.index('s_endpgm', start)
body = s[start:end]
lines = [l.strip() for l in body.split('\n')]
ins = [l.split()[0] for l in lines if l and not l.startswith(('.', ';')) and not l.endswith(':')]
def cls(i):
    if 'mfma' in i: return 'mfma'
    if i.startswith('ds_'): return 'lds'
    if i.startswith(('buffer_', 'global_', 'flat_', 'scratch_')): return 'vmem'
    if i.startswith('s_'): return 'salu'
    return 'valu'
c = collections.Counter(cls(i) for i in ins)
print(m.group(1)[:70], 'total', len(ins), dict(c))
print(collections.Counter(ins).most_common(45))
# loop structure: labels and backward branches
for i, l in enumerate(lines):
    if re.match(r's_cbranch|s_branch', l):
        pass
