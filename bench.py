#!/usr/bin/env python3
"""Headline benchmark: sequences/second of the GCRNN time-step recurrence on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--batch B] [--dtype bf16|f32|f64] [--mode fwd|train]

Workload = BASELINE.json configs[1]: synthetic k-step prediction, sparse SBM graph N=1000 (p_in 0.04,
p_out 0.0025, nnz ~ 1e4), K=5 taps, T=32 steps, G=F=64 features, un-gated cell, h0 = 0. One "step" is one
pass of the hot path (GGCRNNCell.forward: pack -> T-step recurrence -> unpack) over a batch of B sequences
per GPU that is already resident in HBM. For N>1 the batch is sharded (weak scaling, no data-path
collective in inference). Rank 0 prints ONE JSON line (contract in the task statement) with two extra
objects: "roofline" (algorithmic HBM bytes / measured time vs the 8 TB/s peak) and "cpu_baseline" (the
numpy oracle -- a port of the reference's dense CPU algorithm -- timed on a bounded sample of the same
workload on this host).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import numpy as np
import torch

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s spec (6.3 TB/s achievable copy rate)
CFG = dict(N=1000, K=5, T=32, G=64, F=64)


def sbm_graph(N=1000, C=5, p_in=0.04, p_out=0.0025, seed=0):
    """Undirected SBM adjacency / lambda_max (mean degree ~10, nnz ~1e4), redrawn until connected."""
    rng = np.random.default_rng(seed)
    labels = np.arange(N) * C // N
    while True:
        P = np.where(labels[:, None] == labels[None, :], p_in, p_out)
        U = np.triu(rng.random((N, N)) < P, 1)
        W = (U + U.T).astype(np.float64)
        seen = np.zeros(N, bool); seen[0] = True; frontier = np.array([0])
        while frontier.size:
            nxt = np.nonzero((W[frontier].sum(0) > 0) & ~seen)[0]
            seen[nxt] = True; frontier = nxt
        if seen.all():
            break
    lam = np.max(np.linalg.eigvalsh(W))
    return (W / lam).reshape(1, N, N)


def algorithmic_bytes_per_seq(T, N, G, F, elt):
    """Compulsory HBM traffic of the fused recurrence: read x_t, read h_{t-1}, write h_t."""
    return T * elt * N * (G + 2 * F)


def flops_per_seq(T, N, nnz, K, G, F):
    return T * (2 * nnz * (K - 1) * (G + F) + 2 * N * K * F * (G + F))


def cpu_baseline(S, params, T, G, F, seconds_budget=20.0):
    """Time the numpy oracle (dense x@S per hop, the reference's algorithm) on a bounded sample."""
    from oracle import gcrnn_oracle as orc
    N = S.shape[1]
    rng = np.random.default_rng(1)
    Bc = 32                                                    # ~10-20 s of host work in all (5 passes)
    X = rng.standard_normal((Bc, T, G, N)).astype(np.float32)
    h0 = np.zeros((Bc, F, N), np.float32)
    p32 = {k: v.astype(np.float32) for k, v in params.items()}
    S32 = S.astype(np.float32)
    try:
        avail = len(os.sched_getaffinity(0))
    except AttributeError:
        avail = os.cpu_count()
    cores = min(16, avail)                                     # the GPU box's CPU share for one GPU; the BLAS pool is pinned to it
    try:
        from threadpoolctl import threadpool_limits
        limiter = threadpool_limits(limits=cores)
    except ImportError:
        limiter, cores = None, avail
    t0 = time.perf_counter()
    orc.ggcrnn_cell(p32, S32, X[:, :2], h0)                    # warm-up on 2 steps
    warm = time.perf_counter() - t0
    reps, times = 0, []
    while reps < 5 and (reps == 0 or sum(times) + times[-1] < seconds_budget):      # at least one full pass
        t0 = time.perf_counter()
        orc.ggcrnn_cell(p32, S32, X, h0)
        times.append(time.perf_counter() - t0)
        reps += 1
    best = min(times)
    if limiter is not None:
        limiter.restore_original_limits()
    return {'value': Bc / best, 'unit': 'sequences/s', 'cores': cores, 'kind': 'port',
            'sample': 'oracle/gcrnn_oracle.py ggcrnn_cell (dense x@S hops, numpy/BLAS fp32), B=%d full T=%d N=%d '
                      'K=5 G=%d F=%d sequences, best of %d passes (%.2f s each)' % (Bc, T, N, G, F, reps, best)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--batch', type=int, default=256, help='sequences per GPU per step')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'f64'])
    ap.add_argument('--mode', default='fwd', choices=['fwd', 'train'])
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--time-gating', action='store_true', help='secondary point: the time-gated cell (fwd or train); '
                    'the headline workload is the un-gated cell')
    ap.add_argument('--in-features', type=int, default=CFG['G'], help='secondary point: input features per node (the reference '
                    'drivers feed G = 1; the headline workload is G = F = 64)')
    ap.add_argument('--hipgraph', type=int, default=0, help='replay the fused forward as one captured hipGraph (bf16 fwd)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    assert world == args.gpus, 'launch with torch.distributed.run --nproc-per-node %d' % args.gpus
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist_on = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ          # launched by torch.distributed.run
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)                      # "nccl" is RCCL on ROCm

    import gated_gcrnns_amd.Utils.graphML as gml

    N, K, T, G, F = CFG['N'], CFG['K'], CFG['T'], args.in_features, CFG['F']
    B = args.batch
    dt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f64': torch.float64}[args.dtype]
    elt = {'bf16': 2, 'f32': 4, 'f64': 8}[args.dtype]
    S = sbm_graph(N)
    nnz = int(np.count_nonzero(S))
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, args.time_gating, None, 1, True)      # reference init U(+-1/sqrt(G*K))
    cell.addGSO(torch.tensor(S))
    params = {k: v.detach().numpy().copy() for k, v in cell.state_dict().items()}
    cell = cell.to(dev).to(dt)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(dt)
    h0 = torch.zeros(B, F, N, device=dev, dtype=dt)

    sync_grads = None
    if args.mode == 'train':
        # one optimiser step of the k-step-prediction loop (reference train_rnn.py:247-281): forward, L1 loss on the
        # state sequence, BPTT, ONE flat gradient all-reduce over RCCL, Adam.
        # bf16: fp32 master weights, bf16 activations -> fused forward + fused BPTT (un-gated and time-gated cells);
        # f32 / f64: composed path
        from gated_gcrnns_amd.parallel import FlatGradAllReduce
        from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
        if args.dtype == 'bf16':
            cell = cell.float()
        target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32).to(dt)
        opt = torch.optim.Adam(cell.parameters(), lr=1e-3)
        sync_grads = FlatGradAllReduce(cell.parameters()) if dist_on else None

    runner = None
    if args.mode == 'fwd' and args.dtype == 'bf16' and args.hipgraph:
        from gated_gcrnns_amd.ops import FusedForwardGraph
        runner = FusedForwardGraph(cell, B, T)

    def step():
        if args.mode == 'fwd':
            if runner is not None:
                return runner(X, h0)          # copies X, h0 into the graph's static inputs, then ONE graph launch
            with torch.no_grad():
                return cell(X, h0)
        cell.zero_grad()
        loss = batchTimeL1Loss(cell(X, h0), target)      # the drivers' loss (reference miscTools.py:112-119), one fused pass
        loss.backward()
        if sync_grads is not None:
            sync_grads.all_reduce_()
        opt.step()
        return loss

    def sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    for _ in range(args.steps):
        step()
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)
    if dist_on:
        tw = torch.tensor([wall], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.MAX)
        wall = float(tw.item())
    ms_per_step = 1e3 * wall / args.steps
    value = world * B * args.steps / wall

    # ---- dominant kernel: the fused step kernel, timed live with HIP events on the stream it is launched on ----
    kern = None
    if args.dtype == 'bf16' and args.mode == 'fwd' and not args.time_gating:
        from gated_gcrnns_amd import ops
        with torch.no_grad():
            for _ in range(2):
                ops.fused_cell_forward(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, return_states=True)
            torch.cuda.synchronize()
            kern = ops.time_fused_step_kernel(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, reps=3)

    if rank == 0:
        abytes = algorithmic_bytes_per_seq(T, N, G, F, elt) * B           # per step (= per launch chain), per GPU
        step_s = (dev_ms / 1e3) / args.steps
        achieved = abytes / step_s / 1e9
        out = {
            'metric': 'sequences/sec (node), N=1000 K=5 T=32 F=64', 'value': value, 'unit': 'sequences/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': ms_per_step,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype,
            'data': 'synthetic',
            'config': {'workload': 'BASELINE configs[1]: synthetic k-step prediction, sparse SBM N=1000 nnz=%d, K=5 taps, '
                                   'T=32, G=%d, F=64, %s GGCRNNCell %s, h0=0' % (nnz, G, 'time-gated' if args.time_gating else 'un-gated',
                                                                              'forward' if args.mode == 'fwd' else 'training step'),
                       'batch_per_gpu': B, 'global_batch': world * B, 'mode': args.mode, 'hipgraph': bool(runner is not None), 'parallelism': 'dp%d' % world},
        }
        if kern is not None:
            # algorithmic bytes of ONE launch (one time step for the whole batch): read x_t, read h_{t-1}, write h_t
            kbytes = elt * N * (G + 2 * F) * B
            kach = kbytes / (kern['avg_us'] * 1e-6) / 1e9
            traffic = None
            tf = os.path.join(ROOT, 'profiles', 'step_kernel_traffic.json')
            if os.path.exists(tf):
                tj = json.load(open(tf))
                if tj.get('batch') == B:
                    traffic = tj['hbm_bytes_per_launch']
            out['roofline'] = {'bound': 'hbm', 'achieved': kach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': kach / HBM_PEAK_GBS, 'traffic': traffic,
                               'kernel': 'fused_step_kernel<5,2,%d> (one launch = one time step of the whole batch)' % (2 if G > 32 else 1),
                               'kernel_avg_us': kern['avg_us'], 'launches_timed': kern['launches'],
                               'algorithmic_bytes_per_launch': kbytes,
                               'gflop_per_launch': flops_per_seq(1, N, nnz, K, G, F) * B / 1e9,
                               'whole_step_GBps': achieved, 'device_ms_per_step': 1e3 * step_s}
        else:
            out['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                               'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                               'kernel': 'whole T-step recurrence (all launches of one step: %s)' % (
                                   'fused kernels' if args.dtype == 'bf16' else 'composed path'),
                               'algorithmic_bytes_per_step': abytes, 'device_ms_per_step': 1e3 * step_s,
                               'gflop_per_step': flops_per_seq(T, N, nnz, K, G, F) * B / 1e9}
        if not args.no_cpu_baseline and world == 1:          # the host baseline is a single-GPU-run item (rank 0, N = 1)
            out['cpu_baseline'] = cpu_baseline(S, params, T, G, F)
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


if __name__ == '__main__':
    main()
