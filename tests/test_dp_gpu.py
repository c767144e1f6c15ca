"""GPU: the graphed data-parallel step with world_size 2 (two processes sharing the one test GPU, gloo transport, the
flat gradient buffer staged through the host by engine.allreduce_mean).  Checked: replicas stay bit-identical; the cut /
overlapped schedule equals the single all-reduce schedule; and ONE data-parallel step equals a single-process optimizer
step fed the explicit MEAN of the two ranks' gradients (each computed eagerly on that rank's batch).  Equality with a
single-process step on the concatenated batch is NOT expected: num_boxes is normalised per rank (sedt.py:322-324)."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, overlap=True, nsteps=2):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from sound_event_detection_transformer_amd import runtime
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from sound_event_detection_transformer_amd.engine import build_optimizer, GraphedTrainStep
    from oracle import sedt_oracle as O
    from sound_event_detection_transformer_amd.utilities.synthetic import synthetic_batch
    dev = torch.device('cuda', 0)
    runtime.set_compute_dtype('bf16')
    model, crit, _ = build_model(default_args(dropout=0.0))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 3))
    model.to(dev).train()
    crit.to(dev)
    opt = build_optimizer(model)
    if not overlap:          # same flat parameter order as the overlapped schedule: identical summation order of the grad norm
        body = model.backbone[0].body
        opt.set_tail_params([p for n, p in body.named_parameters()
                             if p.requires_grad and (n.startswith('conv0.') or n.startswith('layer2.'))])
    B = 2
    x, t = synthetic_batch(B, 500, 100 + rank, dev)
    stepper = GraphedTrainStep(model, crit, opt, x, t, None, slice(B), warmup=1, overlap_allreduce=overlap)
    assert (stepper.g_low is not None) == overlap
    for i in range(nsteps):
        x, t = synthetic_batch(B, 500, 200 + 10 * i + rank, dev)
        stepper(x, t)
    torch.cuda.synchronize()
    vec = torch.cat([p.detach().flatten().float().cpu() for p in model.parameters() if p.requires_grad])
    both = [torch.zeros_like(vec) for _ in range(world)]
    dist.all_gather(both, vec)
    if rank == 0:
        torch.save({'same': bool(torch.equal(both[0], both[1])), 'finite': bool(torch.isfinite(vec).all()),
                    'norm': float(vec.norm()), 'vec': vec}, out)
    dist.destroy_process_group()


def test_graphed_dp_world2_replicas_stay_identical(tmp_path):
    out = str(tmp_path / 'r.pt')
    mp.spawn(_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r = torch.load(out)
    assert r['same'] and r['finite'], {k: v for k, v in r.items() if k != 'vec'}
    # the same two steps without the backward cut / split all-reduce (same flat layout order): same parameters
    out2 = str(tmp_path / 'r2.pt')
    mp.spawn(_worker, args=(2, _free_port(), out2, False), nprocs=2, join=True)
    r2 = torch.load(out2)
    assert r2['same'] and r2['finite']
    d = (r['vec'] - r2['vec']).abs().max().item()
    assert d <= 1e-6 * max(1.0, r2['vec'].abs().max().item()), d


def _mean_grad_worker(rank, out):
    """single process: eager gradients of rank 0's and rank 1's first batch from the same initial weights, their explicit
    mean, one fused clip + AdamW step"""
    sys.path.insert(0, ROOT)
    from sound_event_detection_transformer_amd import runtime
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from sound_event_detection_transformer_amd.engine import build_optimizer
    from sound_event_detection_transformer_amd.utilities.synthetic import synthetic_batch
    from oracle import sedt_oracle as O
    dev = torch.device('cuda', 0)
    runtime.set_compute_dtype('bf16')
    model, crit, _ = build_model(default_args(dropout=0.0))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 3))
    model.to(dev).train()
    crit.to(dev)
    opt = build_optimizer(model)
    body = model.backbone[0].body
    opt.set_tail_params([p for n, p in body.named_parameters() if p.requires_grad and (n.startswith('conv0.') or n.startswith('layer2.'))])
    B = 2
    grads = []
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for r in range(2):
            x, t = synthetic_batch(B, 500, 200 + r, dev)
            crit(model(x), t, None, slice(B))
            crit.last_total.backward()
            crit.last_total = None
            grads.append([p.grad.clone() for p in model.parameters() if p.requires_grad])
            opt.zero_grad(set_to_none=True)
        for p, g0, g1 in zip([p for p in model.parameters() if p.requires_grad], *grads):
            p.grad = (g0 + g1) / 2
        opt.step(max_norm=0.1)
    torch.cuda.synchronize()
    vec = torch.cat([p.detach().flatten().float().cpu() for p in model.parameters() if p.requires_grad])
    torch.save({'vec': vec}, out)


def test_one_dp_step_equals_step_on_mean_of_rank_gradients(tmp_path):
    out = str(tmp_path / 'dp1.pt')
    mp.spawn(_worker, args=(2, _free_port(), out, True, 1), nprocs=2, join=True)
    ref = str(tmp_path / 'mean.pt')
    mp.spawn(_mean_grad_worker, args=(ref,), nprocs=1, join=True)
    a, b = torch.load(out), torch.load(ref)
    assert a['same'] and a['finite']
    d = (a['vec'] - b['vec']).abs().max().item()
    moved = (a['vec'] - _initial_vec()).abs().max().item()
    assert moved > 0 and d <= 2e-2 * moved, (d, moved)            # graph vs eager bf16 kernels: a few % of one AdamW step


def _initial_vec():
    sys.path.insert(0, ROOT)
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from oracle import sedt_oracle as O
    model, _, _ = build_model(default_args(dropout=0.0))
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 3))
    return torch.cat([p.detach().flatten().float() for p in model.parameters() if p.requires_grad])


def _nccl_worker(rank, port, out, dp):
    """one process, one GPU; dp=True: RCCL process group of size 1 with the multi-GPU schedule forced"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    from sound_event_detection_transformer_amd import runtime
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from sound_event_detection_transformer_amd.engine import build_optimizer, GraphedTrainStep
    from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict, synthetic_batch
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    if dp:
        dist.init_process_group('nccl', rank=0, world_size=1)
    runtime.set_compute_dtype('bf16')
    model, crit, _ = build_model(default_args(dropout=0.0))
    model.load_state_dict(seeded_state_dict(model.state_dict(), 3))
    model.to(dev).train()
    crit.to(dev)
    opt = build_optimizer(model)
    if not dp:
        # same flat parameter order as the data-parallel schedule picks, so that the global-norm summation order - and with
        # it every bit of the clipped AdamW update - is the same in both runs (bf16 training amplifies 1-ulp differences)
        body = model.backbone[0].body
        opt.set_tail_params([p for n, p in body.named_parameters()
                             if p.requires_grad and (n.startswith('conv0.') or n.startswith('layer2.'))])
    B = 2
    x, t = synthetic_batch(B, 500, 100, dev)
    stepper = GraphedTrainStep(model, crit, opt, x, t, None, slice(B), warmup=1, data_parallel=True if dp else None)
    assert (stepper.g_low is not None) == dp
    for i in range(2):
        x, t = synthetic_batch(B, 500, 200 + 10 * i, dev)
        stepper(x, t)
    torch.cuda.synchronize()
    vec = torch.cat([p.detach().flatten().float().cpu() for p in model.parameters() if p.requires_grad])
    torch.save({'vec': vec, 'finite': bool(torch.isfinite(vec).all())}, out)
    if dp:
        dist.destroy_process_group()


def test_rccl_schedule_on_one_gpu_matches_single_process_step(tmp_path):
    """the multi-GPU schedule (cut backward, two asynchronous RCCL AVG all-reduces of the flat buffer, optimizer graph) on a
    process group of size 1 - exercises the real RCCL calls between the graphs - equals the plain one-graph step"""
    a, b = str(tmp_path / 'dp.pt'), str(tmp_path / 'single.pt')
    mp.spawn(_nccl_worker, args=(_free_port(), a, True), nprocs=1, join=True)
    mp.spawn(_nccl_worker, args=(_free_port(), b, False), nprocs=1, join=True)
    ra, rb = torch.load(a), torch.load(b)
    assert ra['finite'] and rb['finite']
    d = (ra['vec'] - rb['vec']).abs().max().item()
    assert d <= 1e-6 * max(1.0, rb['vec'].abs().max().item()), d


def _ddp_worker(rank, port, out, ddp):
    """eager train_step with the model wrapped in torch DistributedDataParallel over RCCL (world size 1) - the path bench.py
    falls back to when the graphed data-parallel schedule cannot be built - against the same eager steps without the wrapper"""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY='0')
    from sound_event_detection_transformer_amd import runtime
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from sound_event_detection_transformer_amd.engine import build_optimizer, train_step, train_stream
    from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict, synthetic_batch
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    if ddp:
        dist.init_process_group('nccl', rank=0, world_size=1)
    runtime.set_compute_dtype('bf16')
    runtime.manual_seed(4)
    model, crit, _ = build_model(default_args(dropout=0.0))
    model.load_state_dict(seeded_state_dict(model.state_dict(), 3))
    model.to(dev).train()
    crit.to(dev)
    opt = build_optimizer(model)
    net = model
    if ddp:
        with torch.cuda.stream(train_stream(dev)):          # DDP's AccumulateGrad hooks belong to the stream the steps run on
            net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[0], broadcast_buffers=False, gradient_as_bucket_view=True)
    B = 2
    losses = []
    for i in range(3):
        x, t = synthetic_batch(B, 500, 300 + 10 * i, dev)
        total, ld = train_step(net, crit, opt, x, t, None, slice(B), max_norm=0.1)
        losses.append(float(total))
    torch.cuda.synchronize()
    vec = torch.cat([p.detach().flatten().float().cpu() for p in model.parameters() if p.requires_grad])
    torch.save({'vec': vec, 'losses': losses}, out)
    if ddp:
        dist.destroy_process_group()


def test_eager_step_under_torch_ddp_matches_plain_eager_step(tmp_path):
    a, b = str(tmp_path / 'ddp.pt'), str(tmp_path / 'plain.pt')
    mp.spawn(_ddp_worker, args=(_free_port(), a, True), nprocs=1, join=True)
    mp.spawn(_ddp_worker, args=(_free_port(), b, False), nprocs=1, join=True)
    ra, rb = torch.load(a), torch.load(b)
    # the first step sees identical parameters: identical losses; later steps differ by what bf16 training makes of a different
    # f32 summation order in the global gradient norm (bucket views vs per-parameter gradients): bounded, not bit-equal
    assert abs(ra['losses'][0] - rb['losses'][0]) <= 1e-6 * abs(rb['losses'][0]), (ra['losses'], rb['losses'])
    assert all(abs(x - y) <= 1e-2 * abs(y) for x, y in zip(ra['losses'], rb['losses'])), (ra['losses'], rb['losses'])
    assert torch.isfinite(ra['vec']).all()
    v0 = _initial_vec()
    ua, ub = ra['vec'] - v0, rb['vec'] - v0
    assert ub.norm().item() > 0
    assert (ua - ub).norm().item() <= 0.1 * ub.norm().item(), ((ua - ub).norm().item(), ub.norm().item())
