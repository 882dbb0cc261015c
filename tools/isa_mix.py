#!/usr/bin/env python3
"""isa_mix.py <file.s> <substring> -- instruction mix of the kernels whose mangled name contains <substring>
(hipcc --cuda-device-only -S output). Used to size VALU/FP64/LDS/VMEM work per thread."""
import sys, re, collections
lines = open(sys.argv[1]).read().splitlines()
pat = sys.argv[2]
i = 0
while i < len(lines):
    m = re.match(r'^(_Z\w+):', lines[i])
    if m and pat in m.group(1):
        name = m.group(1); c = collections.Counter(); i += 1
        while i < len(lines) and not lines[i].strip().startswith('s_endpgm'):
            t = lines[i].strip(); i += 1
            if not t or t[0] in '.;' or t.endswith(':') or t.startswith('//'): continue
            c[t.split()[0]] += 1
        g = lambda f: sum(v for k, v in c.items() if f(k))
        print(name)
        print('  total', sum(c.values()), 'VALU', g(lambda k: k.startswith('v_')), 'f64', g(lambda k: k.endswith('_f64')),
              'ds', g(lambda k: k.startswith('ds_')), 'vmem', g(lambda k: k.startswith(('global_', 'buffer_', 'flat_', 'scratch_'))),
              's_load', g(lambda k: k.startswith('s_load')), 'waitcnt', c['s_waitcnt'], 'salu', g(lambda k: k.startswith('s_')))
        print('  ', ', '.join(f'{k}:{v}' for k, v in c.most_common(28)))
    i += 1
