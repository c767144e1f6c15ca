#!/usr/bin/env python3
"""Headline benchmark: SEDT training throughput (audio clips/s) on synthetic URBAN-SED-shaped batches.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run, one rank per GPU)

A step = one full training step of BASELINE.json configs[1] (URBAN-SED SEDT, enc_layers=3, dec_at, num_queries=10,
B=64 per GPU, bf16): forward on the HIP path, Hungarian matching + SetCriterion (on the device inside the step's HIP
graph by default; --host-matching keeps the reference's host-side matching), backward on the HIP path,
clip_grad_norm_(0.1), AdamW - with dropout 0.1 active.  Inputs are resident in HBM before the timed region.
Prints ONE JSON line (rank 0) with the `roofline` (dominant kernel: the MFMA implicit GEMM, timed live with HIP events
on the launch stream) and `cpu_baseline` (the CPU oracle timed on this box's host cores, bounded sample) objects.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

FLOP_PER_CLIP_FWD_BWD = 27.37e9      # SURVEY.md 8(d), C2 geometry, 2*MAC, wgrad skipped for frozen conv1+layer1
MFMA_PEAK_BF16 = 2.5e15              # dense bf16 MFMA peak, MI355X_MICROARCH.md
MFMA_PEAK_F32 = 157.3e12


def synthetic_batch(B, T, seed, device):
    from sound_event_detection_transformer_amd.utilities.synthetic import synthetic_batch as sb
    return sb(B, T, seed, device)


def cpu_baseline(batch=8, steps=2):
    """the CPU oracle (pure PyTorch restatement of the reference path) on this box's host cores"""
    from oracle import sedt_oracle as O
    from oracle.criterion_oracle import build_oracle_criterion
    cores = torch.get_num_threads()
    model = O.build_oracle_model(10, 10, 3, 3, True, True, True, dropout=0.1)
    model.load_state_dict(O.seeded_state_dict(model.state_dict(), 2020))
    model.train()
    crit = build_oracle_criterion(10, 3, True, True)
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=1e-4, weight_decay=1e-4)
    x, targets = synthetic_batch(batch, 500, 2020, None)

    def step():
        ld, _ = crit(model(x), targets, None, slice(batch))
        loss = sum(ld[k] * crit.weight_dict[k] for k in ld if k in crit.weight_dict)
        loss.backward()
        torch.nn.utils.clip_grad_norm_(model.parameters(), 0.1)
        opt.step()
        opt.zero_grad()
    step()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    dt = time.perf_counter() - t0
    return {"value": round(batch * steps / dt, 3), "unit": "clips/s", "cores": cores, "kind": "port",
            "sample": f"{steps} full train steps of the CPU oracle at B={batch} (same model/config, f32, dropout on), 1 warm-up"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--batch', type=int, default=64)
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--dump-igemm', default=None, help='write per-launch igemm timings (json) to this path')
    ap.add_argument('--host-matching', action='store_true',
                    help='solve the Hungarian matching on the host between two HIP graphs (the reference split) instead of on the device')
    ap.add_argument('--coschedule', action='store_true', help='let weight-gradient GEMMs ride in the spare workgroup slots of the\n'
                    'dgrad launches instead of one grouped launch per layer (measured slightly slower)')
    ap.add_argument('--async-wgrad', action='store_true',
                    help='issue the weight-gradient GEMMs as a parallel branch of the step graph (second stream); measured slower than the\n'
                         'single-stream graph on ROCm 7.2: cross-queue dependencies cost 50-100 us each')
    ap.add_argument('--no-graph', action='store_true', help='issue every kernel from Python instead of replaying HIP graphs')
    ap.add_argument('--model-only', action='store_true', help='time fwd+bwd of the model with a fixed differentiable loss')
    args = ap.parse_args()

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        dist.init_process_group('nccl')
    dev = torch.device('cuda', local_rank)
    torch.cuda.set_device(dev)

    from sound_event_detection_transformer_amd import runtime, ops
    from sound_event_detection_transformer_amd.sedt import build_model, default_args
    from sound_event_detection_transformer_amd.engine import train_step, build_optimizer
    from sound_event_detection_transformer_amd.utilities.synthetic import seeded_state_dict

    runtime.set_compute_dtype(args.dtype)
    torch.manual_seed(2020)
    model, criterion, _ = build_model(default_args(enc_layers=3, num_queries=10, dec_at=True, dropout=0.1))
    model.load_state_dict(seeded_state_dict(model.state_dict(), 2020))
    model.to(dev).train()
    criterion.to(dev)
    net = model
    if world > 1 and args.no_graph:
        # eager path: torch DDP (bucketed RCCL all-reduce overlapped with backward)
        net = torch.nn.parallel.DistributedDataParallel(model, device_ids=[local_rank], broadcast_buffers=False,
                                                        gradient_as_bucket_view=True)
    opt = build_optimizer(model)
    B = args.batch
    x, targets = synthetic_batch(B, 500, 2020 + rank, dev)

    from sound_event_detection_transformer_amd.engine import GraphedTrainStep
    graphed = None
    if not args.no_graph and not args.model_only:
        graphed = GraphedTrainStep(net, criterion, opt, x, targets, None, slice(B), max_norm=0.1,
                                   device_matching=not args.host_matching, async_wgrad=args.async_wgrad, coschedule=args.coschedule)

    def step():
        if graphed is not None and ops.PROFILE is None:
            graphed(x, targets)
        elif args.model_only:
            o = net(x)
            loss = o['pred_logits'].square().mean() + o['pred_boxes'].mean() + o['at'].mean() + \
                sum(a['pred_logits'].square().mean() + a['pred_boxes'].mean() for a in o['aux_outputs'])
            loss.backward()
            opt.zero_grad(set_to_none=True)
        else:
            train_step(net, criterion, opt, x, targets, None, slice(B), max_norm=0.1)

    def barrier():
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        elapsed = t.item()

    # ---- roofline of the dominant kernel family: ONE extra (eager) step records the argument block of every GEMM launch and
    #      keeps its operands alive; the launches are then replayed back to back on the launch stream between two HIP events
    #      (queueing ~300 launches takes less host time than they run, so the stream never starves): elapsed / launches = the
    #      average GEMM launch duration, comparable with rocprofv3's per-kernel averages of the graphed step.
    roof = None
    if rank != 0 and world > 1 and args.no_graph:
        ops.PROFILE = []                      # eager DDP: the extra step contains collectives, every rank must take part
        step()
        torch.cuda.synchronize()
        ops.PROFILE = None
    if rank == 0:
        import ctypes
        from sound_event_detection_transformer_amd import lib as L_
        ops.PROFILE = []
        step()
        torch.cuda.synchronize()
        rec = ops.PROFILE
        ops.PROFILE = None
        n = len(rec)
        lib = L_.load()

        def replay():
            for a, dt_, _, _ in rec:
                L_.check(lib.sedt_igemm(ctypes.byref(a), dt_, L_.stream_ptr()), 'sedt_igemm')
        replay()
        torch.cuda.synchronize()
        reps = 3
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            replay()
        e1.record()
        torch.cuda.synchronize()
        tot_ms = e0.elapsed_time(e1) / reps
        if args.dump_igemm:
            per = []
            for a, dt_, sh, _ in rec:
                s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s0.record()
                for _ in range(5):
                    L_.check(lib.sedt_igemm(ctypes.byref(a), dt_, L_.stream_ptr()), 'sedt_igemm')
                s1.record()
                torch.cuda.synchronize()
                per.append({'ms': s0.elapsed_time(s1) / 5, 'shape': sh})
            with open(args.dump_igemm, 'w') as f:
                json.dump(per, f)
        flops_launch = FLOP_PER_CLIP_FWD_BWD * B / max(n, 1)
        avg_s = tot_ms / 1e3 / max(n, 1)
        peak = MFMA_PEAK_BF16 if args.dtype == 'bf16' else MFMA_PEAK_F32
        ach = flops_launch / avg_s / 1e12
        roof = {"bound": "mfma", "kernel": "sedt::igemm3_kernel / igemm3_w8_kernel + sedt::wgrad3_kernel / wgrad4_kernel (MFMA implicit-GEMM family: every conv/linear fwd, dgrad, wgrad launch)", "achieved": round(ach, 2), "peak": peak / 1e12, "unit": "TFLOP/s",
                "frac": round(ach / (peak / 1e12), 4), "traffic": None, "launches_per_step": n,
                "avg_launch_us": round(avg_s * 1e6, 2), "igemm_ms_per_step": round(tot_ms, 3),
                "algorithmic_flop_per_launch": flops_launch}
        del rec

    if rank == 0:
        cpu = None if (args.no_cpu_baseline or world > 1) else cpu_baseline()
        value = world * B * args.steps / elapsed
        out = {"metric": "audio clips/sec training throughput (B=64, 10s@64-mel)", "value": round(value, 2),
               "unit": "clips/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(elapsed / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
               "config": {"workload": "URBAN-SED SEDT enc_layers=3 dec_at num_queries=10 B=64/GPU, 10 s @ 64-mel "
                                      "(B,1,500,64), full train step: fwd + Hungarian matching (" + ("host" if (args.host_matching or args.no_graph) else "device") +
                                      ") + SetCriterion + bwd + clip 0.1 + AdamW, dropout 0.1"
                                      + (" [model-only timing]" if args.model_only else ""),
                          "global_batch": world * B, "parallelism": f"dp{world}"},
               "roofline": roof, "cpu_baseline": cpu, "hip_graph": graphed is not None}
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
