"""same-box A/B of one captured step: `python tools/dev/ab_step.py --config c2 --dtype bf16x3 --set ops.X3_WGROUP=0 [--set ...] [--replays 200]`
builds the bench workload with the given module attributes overridden (before anything is captured), settles, times `replays` graph replays
with HIP events and prints one line.  Boxes of the pool differ by +-2.5 %: run the variants interleaved inside ONE gpurun call."""
import argparse
import importlib
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
ap = argparse.ArgumentParser()
ap.add_argument('--config', default='c2')
ap.add_argument('--dtype', default='bf16')
ap.add_argument('--set', action='append', default=[], help='module.attr=value inside the package, e.g. ops.X3_WGROUP=0')
ap.add_argument('--replays', type=int, default=200)
ap.add_argument('--tag', default='')
a = ap.parse_args()
import bench                                                                                    # noqa: E402
from sound_event_detection_transformer_amd import runtime                                       # noqa: E402
for s in a.set:
    name, val = s.split('=')
    mod, attr = name.rsplit('.', 1)
    m = importlib.import_module('sound_event_detection_transformer_amd.' + mod)
    old = getattr(m, attr)
    setattr(m, attr, type(old)(eval(val)) if not isinstance(old, bool) else bool(eval(val)))
sys.argv = ['bench.py', '--config', a.config, '--dtype', a.dtype]
args = bench.parse()
runtime.set_compute_dtype(a.dtype)
dev = torch.device('cuda:0')
step, clips, flop, what, graphed, ex = bench.build_workload(args, dev, 0, 1)
for _ in range(60):
    step()
torch.cuda.synchronize()
best = []
for rep in range(3):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(a.replays):
        step()
    e1.record()
    torch.cuda.synchronize()
    best.append(e0.elapsed_time(e1) / a.replays)
print(f'{a.tag or " ".join(a.set) or "default":40s} {a.config} {a.dtype}: ms/step ' + ' '.join(f'{b:.3f}' for b in best), flush=True)
