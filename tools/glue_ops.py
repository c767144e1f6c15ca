"""List the torch (non-sedt) device kernels of ONE eager pass through the code a captured step runs, with op name, shapes and python call
site - and the launch log (entry points, GEMM kernel instances incl. anything on the generic register-staged GEMM).
usage: python tools/glue_ops.py [c2|c3|c4|c5] [aten]      (aten: list only the non-sedt kernels)"""
import collections
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench                                                                                    # noqa: E402
from sound_event_detection_transformer_amd import lib, optim, runtime                          # noqa: E402

cfg = sys.argv[1] if len(sys.argv) > 1 else 'c2'
only_aten = len(sys.argv) > 2 and sys.argv[2] == 'aten'
sys.argv = ['bench.py', '--config', cfg]
args = bench.parse()
runtime.set_compute_dtype('bf16')
dev = torch.device('cuda:0')
step, clips, flop, what, graphed, ex = bench.build_workload(args, dev, 0, 1)
g = ex['stepper']
from sound_event_detection_transformer_amd.engine import train_stream                          # noqa: E402
side = train_stream(dev)


def one():
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side), g.optimizer.table_set(g._tabname):
        if hasattr(g, '_eager_device_step'):
            g._eager_device_step(g._example_targets if hasattr(g, '_example_targets') else ex.get('targets'), 32)
        else:
            g._body()
    torch.cuda.current_stream().wait_stream(side)


for _ in range(2):
    one()
torch.cuda.synchronize()
with lib.launch_log() as log:
    one()
torch.cuda.synchronize()
print('# launch log (entry points):', {k: v for k, v in sorted(log.items()) if not k.startswith('igemm')})
print('# GEMM kernel instances:', {k: v for k, v in sorted(log.items()) if k.startswith('igemm')})
from torch.profiler import profile, ProfilerActivity                                            # noqa: E402
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], record_shapes=True, with_stack=True) as prof:
    one()
    torch.cuda.synchronize()
agg = collections.Counter()
tim = collections.Counter()
for ev in prof.events():
    if ev.kernels and not any(c.kernels for c in ev.cpu_children):
        st = [s.split('sound_event_detection_transformer_amd/')[-1] for s in ev.stack if 'sound_event' in s or 'engine' in s]
        if only_aten and 'sedt' in ev.kernels[0].name:
            continue
        key = (ev.name + ' | ' + ev.kernels[0].name[:40], str(ev.input_shapes)[:60], ' <- '.join(x[-48:] for x in st[:3]) if st else ('<autograd>' if not ev.stack else ev.stack[0][-60:]))
        agg[key] += 1
        tim[key] += sum(k.duration for k in ev.kernels)
for k, n in sorted(agg.items(), key=lambda kv: -tim[kv[0]])[:90]:
    print(f'{n:4d} {tim[k]:8.1f}us  {k[0]:62s} {k[1]:60s} {k[2]}')
