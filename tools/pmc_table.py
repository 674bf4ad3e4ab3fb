#!/usr/bin/env python3
"""Per-kernel averages of every counter found in the rocprofv3 pmc DBs under a directory."""
import glob
import os
import re
import sqlite3
import sys

d = sys.argv[1]
tab = {}
order = []
for db in sorted(glob.glob(os.path.join(d, 'p*', '*.db'))):
    c = sqlite3.connect(db)
    q = 'select kernel_name, counter_name, count(*), avg(value) from counters_collection group by kernel_name, counter_name'
    for k, cn, n, v in c.execute(q):
        k = re.sub(r'\(.*\)$', '', re.sub(r'^void ', '', k))
        if not k.startswith('f8::'):
            continue
        tab.setdefault(k, {})[cn] = v
        tab[k]['#'] = n
        if cn not in order:
            order.append(cn)
filt = sys.argv[2] if len(sys.argv) > 2 else ''
print('kernel | launches | ' + ' | '.join(order))
for k in sorted(tab, key=lambda k: -tab[k].get('SQ_WAVE_CYCLES', tab[k].get(order[0], 0))):
    if filt and filt not in k:
        continue
    print(k.replace('f8::conv_igemm_kernel', 'conv') + ' | ' + str(tab[k]['#']) + ' | ' + ' | '.join(f'{tab[k].get(c, float("nan")):.4g}' for c in order))
