#!/usr/bin/env python3
"""HBM-side traffic of ONE steady-state train step per kernel, from two rocprofv3 --pmc passes over bench.py
(FETCH_SIZE and WRITE_SIZE in separate runs, as MI355X_MICROARCH.md prescribes).  Steps are split at stem_im2col.
usage: pmc_step_traffic.py <dir of the FETCH_SIZE pass> <dir of the WRITE_SIZE pass> [out.csv]
FETCH_SIZE is printed uncorrected (KB) and doubled (the guide: gfx950 tallies 128-byte requests of wide coalesced reads at 64 B)."""
import collections
import csv
import glob
import re
import sys


def one_step(d, counter):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == counter]
    rows.sort(key=lambda r: int(r['Dispatch_Id']))
    marks = [i for i, r in enumerate(rows) if 'stem_im2col' in r['Kernel_Name']]
    seg = rows[marks[-2]:marks[-1]]
    agg = collections.OrderedDict()
    for r in seg:
        n = re.sub(r'void |sedt::|at::native::|\(anonymous namespace\)::', '', r['Kernel_Name'])
        n = re.sub(r'\(.*', '', n)[:60]
        a = agg.setdefault(n, [0, 0.0])
        a[0] += 1
        a[1] += float(r['Counter_Value'])
    return agg


fetch = one_step(sys.argv[1], 'FETCH_SIZE')
write = one_step(sys.argv[2], 'WRITE_SIZE')
names = sorted(set(fetch) | set(write), key=lambda n: -(fetch.get(n, [0, 0])[1] + write.get(n, [0, 0])[1]))
out = open(sys.argv[3], 'w') if len(sys.argv) > 3 else sys.stdout
w = csv.writer(out)
w.writerow(['kernel', 'launches_per_step', 'FETCH_SIZE_KB_uncorrected', 'FETCH_KB_x2', 'WRITE_SIZE_KB'])
tf = tw = 0.0
for n in names:
    fl, fv = fetch.get(n, [0, 0.0])
    wl, wv = write.get(n, [0, 0.0])
    w.writerow([n, max(fl, wl), round(fv), round(2 * fv), round(wv)])
    tf += fv
    tw += wv
w.writerow(['TOTAL', '', round(tf), round(2 * tf), round(tw)])
