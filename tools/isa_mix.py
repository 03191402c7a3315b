"""instruction mix of the kernels of a hipcc -save-temps .s file whose mangled name contains a pattern
usage: python tools/isa_mix.py file.s pattern"""
import re, sys
from collections import Counter
s = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r'^(_Z\S*):\s*; @\S*\n(.*?)s_endpgm', s, re.S | re.M):
    name, body = m.group(1), m.group(2)
    if pat not in name:
        continue
    ins = re.findall(r'^\s+([a-z][a-z_0-9]+)', body, re.M)
    c = Counter(ins)
    grp = Counter()
    for k, v in c.items():
        g = ('pk' if k.startswith('v_pk_') else 'valu' if k.startswith('v_') else 'salu' if k.startswith('s_') else
             'lds' if k.startswith('ds_') else 'vmem' if k.startswith(('global_', 'buffer_', 'scratch_', 'flat_')) else 'other')
        grp[g] += v
    print(name[:100])
    print('  total', len(ins), dict(grp))
    print('  ', ', '.join(f'{k}:{v}' for k, v in c.most_common(28)))
