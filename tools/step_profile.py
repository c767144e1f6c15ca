#!/usr/bin/env python3
"""Per-kernel time of ONE steady-state step from a rocprofv3 kernel trace of bench.py (segments split at stem_im2col).
usage: step_profile.py <dir with *_kernel_trace.csv> [top]"""
import collections
import csv
import glob
import re
import sys

d = sys.argv[1]
top = int(sys.argv[2]) if len(sys.argv) > 2 else 30
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, e in enumerate(ev) if 'stem_im2col' in e[2]]
a, b = idx[-4], idx[-3]
seg = ev[a:b]
span = seg[-1][1] - seg[0][0]
cs, ce, uni = seg[0][0], seg[0][1], 0
for s, e, _ in seg[1:]:
    if s > ce:
        uni += ce - cs
        cs, ce = s, e
    else:
        ce = max(ce, e)
uni += ce - cs
print('kernels %d span %.3f ms  sum %.3f  union %.3f' % (len(seg), span / 1e6, sum(e - s for s, e, _ in seg) / 1e6, uni / 1e6))


def short(n):
    n = re.sub(r'void |sedt::|at::native::|\(anonymous namespace\)::', '', n)
    n = re.sub(r'^_ZN4sedt\d+', '', n)
    return n[:62]


agg = collections.defaultdict(lambda: [0, 0])
for s, e, n in seg:
    agg[short(n)][0] += 1
    agg[short(n)][1] += e - s
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])[:top]:
    print(f'{k:62s} {v[0]:4d} {v[1] / 1e3:9.1f}')
