#!/usr/bin/env python3
"""Per-kernel sums of arbitrary rocprofv3 --pmc counters over ONE steady-state step of bench.py (steps split at the stem forward launch).
usage: pmc_cache_summary.py <out.csv> <pmc_dir> [<pmc_dir> ...]
Every directory is one `rocprofv3 --pmc A B ... --kernel-trace --output-format csv -d <pmc_dir>` pass (the guide: TCC counters in their
own passes).  Output: one row per kernel with its launches per step and every counter found, plus the L2 hit rate
TCC_HIT_sum / (TCC_HIT_sum + TCC_MISS_sum) when both are present."""
import collections
import csv
import glob
import re
import sys


def short(n):
    n = re.sub(r'void |sedt::|at::native::|\(anonymous namespace\)::', '', n)
    n = re.sub(r'^_ZN4sedt\d+', '', n)
    return re.sub(r'\(.*', '', n)[:64]


def last_step(rows):
    rows = sorted(rows, key=lambda r: int(r['Dispatch_Id']))
    marks = [i for i, r in enumerate(rows) if 'stem_pool_fwd' in r['Kernel_Name']]
    return rows[marks[-2]:marks[-1]]


table = collections.OrderedDict()
cols = []
for d in sys.argv[2:]:
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        print('no counter_collection.csv under', d, file=sys.stderr)
        continue
    by = collections.defaultdict(list)
    for r in csv.DictReader(open(f[0])):
        by[r['Counter_Name']].append(r)
    for c, rs in by.items():
        if c not in cols:
            cols.append(c)
        for r in last_step(rs):
            row = table.setdefault(short(r['Kernel_Name']), collections.defaultdict(float))
            row[c] += float(r['Counter_Value'])
            row['_n_' + c] += 1
w = csv.writer(open(sys.argv[1], 'w'))
w.writerow(['kernel', 'launches_per_step'] + cols + ['l2_hit_rate'])
tot = collections.defaultdict(float)
for k, row in sorted(table.items(), key=lambda kv: -kv[1].get(cols[0], 0.0) if cols else 0):
    n = int(max(row['_n_' + c] for c in cols))
    h, m = row.get('TCC_HIT_sum', 0.0), row.get('TCC_MISS_sum', 0.0)
    w.writerow([k, n] + [int(row.get(c, 0.0)) for c in cols] + [round(h / (h + m), 4) if h + m else ''])
    for c in cols:
        tot[c] += row.get(c, 0.0)
h, m = tot.get('TCC_HIT_sum', 0.0), tot.get('TCC_MISS_sum', 0.0)
w.writerow(['TOTAL', ''] + [int(tot[c]) for c in cols] + [round(h / (h + m), 4) if h + m else ''])
