#!/usr/bin/env python3
"""One steady-state training step per kernel from the outputs of tools/profile_round.sh:
  time (kernel trace), HBM-side bytes (FETCH_SIZE x 2 as the guide prescribes for wide coalesced reads on gfx950, WRITE_SIZE),
  matrix-core busy share (SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs): rocprofv3 sums GRBM_GUI_ACTIVE over the 8
  XCDs; calibrated on the layer4 3x3 conv: 1.18 M MFMAs x 32 cycles over 1024 SIMDs), VALU instructions per MFMA.
usage: pmc_step_summary.py <gpurun_out/prof_<tag>> <profiles/rNN_step_c2.csv> <profiles/rNN_pmc_c2.json> [config, default c2] [stem launches per
step, default 1 (2 for c4: clips + patches, and c5: teacher + student)]
Steps are split at the stem forward launch (stem_pool_fwd, or stem_im2col in builds before the one-launch stem: one per forward of the supervised step)."""
import collections
import csv
import glob
import json
import re
import sys

root, out_csv, out_json = sys.argv[1], sys.argv[2], sys.argv[3]
cfg = sys.argv[4] if len(sys.argv) > 4 else 'c2'
PER = int(sys.argv[5]) if len(sys.argv) > 5 else 1


def short(n):
    n = re.sub(r'void |sedt::|at::native::|\(anonymous namespace\)::', '', n)
    n = re.sub(r'^_ZN4sedt\d+', '', n)
    return re.sub(r'\(.*', '', n)[:64]


def last_step(rows, key):
    rows = sorted(rows, key=key)
    marks = [i for i, r in enumerate(rows) if 'stem_im2col' in r['Kernel_Name'] or 'stem_pool_fwd' in r['Kernel_Name']]
    return rows[marks[-1 - PER]:marks[-1]]


def counters(d):
    f = glob.glob(d + '/**/*counter_collection.csv', recursive=True)
    if not f:
        return {}
    rows = list(csv.DictReader(open(f[0])))
    by = collections.defaultdict(list)
    for r in rows:
        by[r['Counter_Name']].append(r)
    res = {}
    for c, rs in by.items():
        agg = collections.OrderedDict()
        for r in last_step(rs, lambda r: int(r['Dispatch_Id'])):
            a = agg.setdefault(short(r['Kernel_Name']), [0, 0.0])
            a[0] += 1
            a[1] += float(r['Counter_Value'])
        res[c] = agg
    return res


tr = glob.glob(root + f'/trace_{cfg}/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(tr)))
seg = last_step(rows, lambda r: int(r['Start_Timestamp']))
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e6
tim = collections.OrderedDict()
for r in seg:
    a = tim.setdefault(short(r['Kernel_Name']), [0, 0.0])
    a[0] += 1
    a[1] += (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
cf, cw, cs = counters(root + '/pmc_fetch'), counters(root + '/pmc_write'), counters(root + '/pmc_sq')
fetch, write = cf.get('FETCH_SIZE', {}), cw.get('WRITE_SIZE', {})
names = sorted(tim, key=lambda n: -tim[n][1])
w = csv.writer(open(out_csv, 'w'))
w.writerow(['kernel', 'launches_per_step', 'time_us', 'FETCH_KB_x2', 'WRITE_KB', 'mfma_busy_share', 'valu_per_mfma', 'wave_cycles_waiting_share'])
tot_f = tot_w = 0.0
for n in names:
    f2 = 2 * fetch.get(n, [0, 0.0])[1]
    wr = write.get(n, [0, 0.0])[1]
    tot_f += f2
    tot_w += wr
    busy = cs.get('SQ_VALU_MFMA_BUSY_CYCLES', {}).get(n, [0, 0.0])[1]
    gui = cs.get('GRBM_GUI_ACTIVE', {}).get(n, [0, 0.0])[1]
    valu = cs.get('SQ_INSTS_VALU', {}).get(n, [0, 0.0])[1]
    mfma = cs.get('SQ_INSTS_MFMA', {}).get(n, [0, 0.0])[1]
    wc = cs.get('SQ_WAVE_CYCLES', {}).get(n, [0, 0.0])[1]
    wa = cs.get('SQ_WAIT_ANY', {}).get(n, [0, 0.0])[1]
    w.writerow([n, tim[n][0], round(tim[n][1], 1), round(f2), round(wr), round(busy / (gui / 8.0 * 1024), 4) if gui else '',
                round(valu / mfma, 1) if mfma else '', round(wa / wc, 3) if wc else ''])
w.writerow(['TOTAL', sum(v[0] for v in tim.values()), round(sum(v[1] for v in tim.values()), 1), round(tot_f), round(tot_w), '', '', ''])
import os
import subprocess
stamp = head = None
try:      # the build the passes ran on: tools/profile_round.sh leaves its source stamp beside the traces
    stamp = open(os.path.join(root, 'build_stamp.txt')).read().strip()
except OSError:
    pass
try:
    head = subprocess.run(['git', 'rev-parse', '--short=12', 'HEAD'], capture_output=True, text=True).stdout.strip() or None
except OSError:
    pass
json.dump({'build_stamp': stamp, 'git_head_at_summary': head, 'hbm_bytes_per_step': int((tot_f + tot_w) * 1024), 'fetch_bytes_x2': int(tot_f * 1024), 'write_bytes': int(tot_w * 1024),
           'kernels_per_step': sum(v[0] for v in tim.values()), 'step_span_ms_under_profiler': round(span, 3),
           'source': 'rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) over bench.py, one steady-state step; FETCH_SIZE doubled '
                     '(gfx950 tallies 128-B requests of wide coalesced reads at 64 B, MI355X_MICROARCH.md); ' + out_csv},
          open(out_json, 'w'), indent=1)
print('step span %.3f ms, %d kernels, HBM-side %.2f GB fetched (x2) + %.2f GB written' % (span, sum(v[0] for v in tim.values()), tot_f / 1e6 * 1.024 / 1.0, tot_w / 1e6 * 1.024))
