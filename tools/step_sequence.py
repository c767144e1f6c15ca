#!/usr/bin/env python3
"""The ordered kernel sequence of ONE steady-state step from a rocprofv3 kernel trace of bench.py: index, start offset, duration,
gap to the previous kernel's end, name.  Steps are split at the stem forward kernel.
usage: step_sequence.py <dir with *_kernel_trace.csv> [out.txt] [stem launches per step, default 1: C4 (clips + patches) and C5 (teacher +
student) run the stem twice per step]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
f = glob.glob(d + '/**/*_kernel_trace.csv', recursive=True)[0]
rows = list(csv.DictReader(open(f)))
ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in rows)
idx = [i for i, e in enumerate(ev) if 'stem_pool_fwd' in e[2] or 'stem_im2col' in e[2]]
per = int(sys.argv[3]) if len(sys.argv) > 3 else 1
# a step INSIDE bench.py's timed loop: the last three replays of a run are the host-issue burst that follows the closing barrier (round 6)
a, b = idx[-6 * per], idx[-5 * per]
seg = ev[a:b]


def short(n):
    n = re.sub(r'void |sedt::|at::native::|\(anonymous namespace\)::', '', n)
    return n[:110]


out = open(sys.argv[2], 'w') if len(sys.argv) > 2 else sys.stdout
t0 = seg[0][0]
prev = seg[0][0]
gaps = 0
for i, (s, e, n) in enumerate(seg):
    gap = s - prev
    gaps += max(gap, 0)
    print(f'{i:4d} {(s - t0) / 1e3:9.1f} {(e - s) / 1e3:7.1f} {gap / 1e3:6.1f}  {short(n)}', file=out)
    prev = e
print(f'# kernels {len(seg)} span {(seg[-1][1] - t0) / 1e6:.3f} ms, sum of kernel durations {sum(e - s for s, e, _ in seg) / 1e6:.3f} ms, '
      f'sum of gaps {gaps / 1e6:.3f} ms', file=out)
