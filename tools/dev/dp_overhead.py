"""What the data-parallel SCHEDULE itself costs on one GPU: the C2 step as one graph against the same step with the segmented
backward, one RCCL all-reduce per segment (process group of size 1: the collectives are real calls that move nothing) and the
optimizer graph - for the coarse / fine cuts, f32 / bf16 buckets and the single all-reduce.  usage: python tools/dev/dp_overhead.py"""
import os
import sys
import time
import torch
import torch.distributed as dist
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
from sound_event_detection_transformer_amd import runtime                                        # noqa: E402
from sound_event_detection_transformer_amd.sedt import build_model, default_args                 # noqa: E402
from sound_event_detection_transformer_amd.engine import build_optimizer, GraphedTrainStep        # noqa: E402
from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict, synthetic_batch   # noqa: E402

dev = torch.device('cuda', 0)
torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1)
runtime.set_compute_dtype('bf16')
x, targets = synthetic_batch(64, 500, 2020, torch.device('cpu'))
x = x.to(dev)


def run(name, **kw):
    model, crit, _ = build_model(default_args(dropout=0.1))
    model.load_state_dict(seeded_state_dict(model.state_dict(), 2020))
    model.to(dev).train()
    crit.to(dev)
    opt = build_optimizer(model)
    g = GraphedTrainStep(model, crit, opt, x, targets, None, slice(64), **kw)
    for _ in range(5):
        g(x, targets)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50):
        g(x, targets)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 50 * 1e3
    print(f'{name:44s} {ms:7.3f} ms/step   graphs {1 + len(g.g_seg) + (g.g_opt is not None)}  segments MB '
          f'{[round(v.numel() * v.element_size() / 1e6, 1) for v in g.flat_parts]}', flush=True)
    del g, model, opt


only = sys.argv[1] if len(sys.argv) > 1 else ''
variants = [('one', 'one graph (no data-parallel schedule)', dict(data_parallel=False)),
            ('flat', 'flat buffer, ONE all-reduce, optimizer graph', dict(data_parallel=True, overlap_allreduce=False)),
            ('coarse', 'coarse cuts: 4 segments / all-reduces', dict(data_parallel=True, dp_cuts='coarse')),
            ('fine', 'fine cuts: 5 segments / all-reduces', dict(data_parallel=True, dp_cuts='fine')),
            ('bf16', 'coarse cuts, bf16 buckets', dict(data_parallel=True, dp_cuts='coarse', grad_dtype=torch.bfloat16))]
for key, name, kw in variants:
    if not only or key == only:
        run(name, **kw)
dist.destroy_process_group()
