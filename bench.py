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


def sbm_graph(N=1000, C=5, p_in=0.04, p_out=0.0025, seed=0, normalized=False):
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
    if normalized:      # reference Utils/graphTools.py:64 normalizeAdjacency: D^-1/2 W D^-1/2 -- a rank-1-weighted GSO (S[m][n] = d[m]^-1/2 d[n]^-1/2 on the support)
        d = W.sum(axis=1)
        W = W / np.sqrt(d)[:, None] / np.sqrt(d)[None, :]
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
    Bc = 8                                                     # bounded sample: ~0.6 s per pass, <= 5 passes -- the GPU region is not drowned by the host baseline
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


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=None, help='timed steps (default: 30; cfg5: 10)')
    ap.add_argument('--warmup', type=int, default=None, help='untimed warm-up steps (default: 5; cfg5: 3)')
    ap.add_argument('--batch', type=int, default=None, help='sequences per GPU per step (default: 256; cfg5: 8; cfg4: 100)')
    ap.add_argument('--dtype', default='bf16', choices=['bf16', 'f32', 'f64'])
    ap.add_argument('--mode', default='fwd', choices=['fwd', 'train'])
    ap.add_argument('--config', default='cfg2', choices=['cfg2', 'cfg4', 'cfg5'],
                    help='cfg2 = BASELINE configs[1] (the headline, default); cfg4 = configs[3] (seismic graph N=59, T=200: latency-bound); '
                         'cfg5 = configs[4] (N=100k, nnz=1e7 streaming CSR SpMM)')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-secondary', action='store_true', help='skip the secondary points of the default line (fp32-accurate forward, bf16 training step)')
    ap.add_argument('--time-gating', action='store_true', help='secondary point: the time-gated cell (fwd or train); '
                    'the headline workload is the un-gated cell')
    ap.add_argument('--spatial-gating', default=None, choices=['node', 'edge'], help='secondary point: node- / edge-gated cell')
    ap.add_argument('--gso', default='adjacency', choices=['adjacency', 'normalized'], help='secondary point: the GSO is the SBM adjacency / lambda_max '
                    '(the drivers, kStepPredGRNNs.py:768: one weight on every edge) or its normalised form D^-1/2 W D^-1/2 / lambda_max (graphTools.py:64)')
    ap.add_argument('--in-features', type=int, default=CFG['G'], help='secondary point: input features per node (the reference '
                    'drivers feed G = 1; the headline workload is G = F = 64)')
    ap.add_argument('--settle-ms', type=float, default=100.0, help='untimed steps for this long BEFORE the W warm-up steps (the chip\'s clock transient '
                                                                    'behind an idle period, profiles/r04_clock_transient.txt); 0 = none')
    ap.add_argument('--hipgraph', type=int, default=None, help='replay the fused forward as one captured hipGraph (bf16 fwd); default: 1 when the '
                    'capture succeeds (bit-identical replay, tests/test_fused.py), else eager')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'], help='process-group backend ("nccl" is RCCL on ROCm; '
                    'gloo only with --dry-run)')
    ap.add_argument('--dry-run', action='store_true', help='launcher / rendezvous / reduction plumbing only: no GPU, no kernels '
                    '(CPU test of the N > 1 entry)')
    return ap.parse_args(argv)


def free_port():
    import socket
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    p = s.getsockname()[1]
    s.close()
    return p


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no RANK in the environment: start the N ranks as CHILD processes
    (python -m torch.distributed.run, one rank per GPU, rendezvous on 127.0.0.1) before this process has made any GPU
    call, relay rank 0's JSON line, and exit with the children's status. The parent never touches the GPU and never
    re-executes itself."""
    import subprocess
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')        # dmabuf IPC only on this pool (RCCL needs it)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    lines = []
    for line in proc.stdout:
        s = line.strip()
        if s.startswith('{') and '"metric"' in s:
            lines.append(s)
        else:
            sys.stderr.write(line)
    rc = proc.wait()
    for s in lines:
        print(s, flush=True)
    if rc != 0 or not lines:
        sys.stderr.write('bench.py: the %d-rank launch failed (exit %d): %s\n' % (args.gpus, rc, ' '.join(cmd)))
        sys.exit(rc if rc != 0 else 1)
    sys.exit(0)


def dry_run(args, rank, world):
    """No GPU: rendezvous, one timed fake step per rank, MAX over ranks, rank 0 prints the JSON line."""
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group(args.backend if args.backend == 'gloo' else 'gloo')
    B = args.batch or 256
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3)
    wall = time.perf_counter() - t0
    if world > 1:
        tw = torch.tensor([wall], dtype=torch.float64)
        dist.all_reduce(tw, op=dist.ReduceOp.MAX)
        wall = float(tw.item())
    if rank == 0:
        print(json.dumps({'metric': 'sequences/sec (node), N=1000 K=5 T=32 F=64', 'value': world * B * args.steps / wall,
                          'unit': 'sequences/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
                          'ms_per_step': 1e3 * wall / args.steps, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
                          'dtype': args.dtype, 'data': 'dry-run (no kernels)',
                          'config': {'workload': 'dry-run', 'batch_per_gpu': B, 'global_batch': world * B, 'mode': args.mode,
                                     'parallelism': 'dp%d' % world}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    if args.steps is None:
        args.steps = 10 if args.config == 'cfg5' else 30
    if args.warmup is None:
        args.warmup = 3 if args.config == 'cfg5' else 5
    global SETTLE_MS
    SETTLE_MS = float(args.settle_ms)
    if args.gpus > 1 and 'RANK' not in os.environ:
        self_launch(args, argv)                       # does not return
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    if world != args.gpus:
        sys.exit('bench.py: --gpus %d but WORLD_SIZE=%d; run `python bench.py --gpus %d` (it starts the ranks itself) or '
                 '`python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 bench.py --gpus %d`'
                 % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if args.dry_run:
        return dry_run(args, rank, world)
    if args.backend != 'nccl':
        sys.exit('bench.py: --backend gloo is for --dry-run only (the product has no CPU path)')
    if args.batch is None:
        args.batch = {'cfg2': 256, 'cfg4': 100, 'cfg5': 8}[args.config]
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    dist_on = 'RANK' in os.environ and 'WORLD_SIZE' in os.environ          # launched by torch.distributed.run
    if dist_on:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        dist.init_process_group('nccl', device_id=dev)                      # "nccl" is RCCL on ROCm

    ctx = dict(args=args, rank=rank, world=world, dev=dev, dist_on=dist_on)
    if args.config == 'cfg5':
        out = run_cfg5(ctx)
    elif args.config == 'cfg4':
        out = run_cfg4(ctx)
    else:
        out = run_cfg2(ctx)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist_on:
        torch.distributed.destroy_process_group()


MFMA_BF16_PEAK_TFLOPS = 2500.0   # MI355X_MICROARCH.md: ~2.5 PF dense bf16
MFMA_F64_PEAK_TFLOPS = 78.6      # v_mfma_f64_16x16x4: 64 cycles per 16x16x4 MFMA per SIMD on MI355X = half the fp32 matrix rate


def timed_steps(ctx, step):
    """W untimed warm-up steps, then EXACTLY K steps bracketed by barrier + synchronize on both sides; wall = MAX over ranks.
    Returns (wall seconds, device milliseconds between HIP events on the current stream)."""
    args, dev, dist_on = ctx['args'], ctx['dev'], ctx['dist_on']

    def sync():
        if dist_on:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    # settle phase (untimed, BEFORE the W warm-up steps; --settle-ms 0 switches it off): behind an idle period the chip runs its first ~30 ms
    # of load 10-25 % slower than the steady state a server sees (clock transient: profiles/r04_clock_transient.txt), so a short window right
    # behind a cold start would measure the transient, not the kernels. Reported in config.settle_ms / settle_steps.
    import gc
    gc.collect()                  # like timeit: no cyclic-garbage pass of the interpreter inside the timed steps (its position
    gc_was_on = gc.isenabled()    # depends on the process's allocation history, not on the workload) -- and none between the warm-up
    gc.disable()                  # and the timed steps either: an idle gap there restarts the transient
    ctx['settle_steps'] = 0
    ctx['cold'] = None
    if args.settle_ms > 0:
        # the same K steps once right here, behind the idle period of the set-up (this rank's own clock, no barrier): the cold-start figure the
        # line would have carried without the settle phase -- reported next to `value` as config.cold_start_*
        torch.cuda.synchronize()
        tc = time.perf_counter()
        for _ in range(args.steps):
            step()
        torch.cuda.synchronize()
        ctx['cold'] = (time.perf_counter() - tc) / args.steps
        step()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); step(); e1.record()
        torch.cuda.synchronize()
        n = max(1, min(2000, int(args.settle_ms / max(e0.elapsed_time(e1), 1e-3))))
        for _ in range(n):        # queued back to back (no synchronisation inside: the load must be continuous), straight into the warm-up steps
            step()
        ctx['settle_steps'] = n + 2
    for _ in range(args.warmup):
        step()
    sync()
    if dist_on:
        # the collectives of the timing protocol itself, once, outside the timed region: RCCL sets up a communicator's channels and
        # loads its kernels on first use (seen on a cold box: one 40-60 ms stall inside a 5-step window, tools/experiments/dist_cold_probe.py)
        tw = torch.zeros(1, device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.MAX)
        sync()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()
    trace = [] if os.environ.get('GCRNN_BENCH_TRACE') else None       # diagnosis only: per-step host time and a device event
    for _ in range(args.steps):
        step()
        if trace is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            trace.append((time.perf_counter(), e))
    ev1.record()
    sync()
    wall = time.perf_counter() - t0
    if gc_was_on:
        gc.enable()
    dev_ms = ev0.elapsed_time(ev1)
    if trace:
        sys.stderr.write('bench trace host ms: %s\n' % ' '.join('%.1f' % (1e3 * (t - (trace[i - 1][0] if i else t0))) for i, (t, _) in enumerate(trace)))
        sys.stderr.write('bench trace dev  ms: %s\n' % ' '.join('%.1f' % (trace[i - 1][1] if i else ev0).elapsed_time(e) for i, (_, e) in enumerate(trace)))
    if dist_on:
        tw = torch.tensor([wall], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(tw, op=torch.distributed.ReduceOp.MAX)
        wall = float(tw.item())
    return wall, dev_ms


def base_line(ctx, value, wall, workload, B, extra=None):
    args, world = ctx['args'], ctx['world']
    cfg = {'workload': workload, 'batch_per_gpu': B, 'global_batch': world * B, 'mode': args.mode, 'parallelism': 'dp%d' % world,
           'settle_ms': args.settle_ms, 'settle_steps': ctx.get('settle_steps', 0),
           'settle_note': 'untimed steps before the W warm-up steps: the clock transient behind an idle period (profiles/r04_clock_transient.txt); '
                          'cold_start_*: the same K steps timed once BEFORE the settle phase, right behind the set-up'}
    if ctx.get('cold'):
        cfg['cold_start_ms_per_step'] = 1e3 * ctx['cold']
        cfg['cold_start_value'] = world * B / ctx['cold']
    cfg.update(extra or {})
    return {'metric': 'sequences/sec (node), N=1000 K=5 T=32 F=64', 'value': value, 'unit': 'sequences/s',
            'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': 1e3 * wall / args.steps,
            'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None, 'dtype': args.dtype, 'data': 'synthetic', 'config': cfg}


def profile_traffic(name, B):
    tf = os.path.join(ROOT, 'profiles', name)
    if os.path.exists(tf):
        tj = json.load(open(tf))
        if tj.get('batch') == B:
            return tj
    return None


def run_cfg2(ctx):
    args, rank, world, dev, dist_on = ctx['args'], ctx['rank'], ctx['world'], ctx['dev'], ctx['dist_on']
    import gated_gcrnns_amd.Utils.graphML as gml

    N, K, T, G, F = CFG['N'], CFG['K'], CFG['T'], args.in_features, CFG['F']
    B = args.batch
    dt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f64': torch.float64}[args.dtype]
    elt = {'bf16': 2, 'f32': 4, 'f64': 8}[args.dtype]
    S = sbm_graph(N, normalized=(args.gso == 'normalized'))
    nnz = int(np.count_nonzero(S))
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, args.time_gating, args.spatial_gating, 1, True)      # reference init U(+-1/sqrt(G*K))
    cell.addGSO(torch.tensor(S))
    params = {k: v.detach().numpy().copy() for k, v in cell.state_dict().items()}
    cell = cell.to(dev).to(dt)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(dt)
    h0 = torch.zeros(B, F, N, device=dev, dtype=dt)

    opt = None
    if args.mode == 'train':
        # one optimiser step of the k-step-prediction loop (reference train_rnn.py:247-281): zero_grad (one memset of the flat
        # gradient buffer), forward, L1 loss on the state sequence, BPTT, ONE flat gradient all-reduce over RCCL, Adam as one
        # kernel over the flat buffers.
        # bf16: fp32 master weights, bf16 activations -> fused forward + fused BPTT; f32 / f64: composed path
        from gated_gcrnns_amd.optim import FlatAdam
        from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
        if args.dtype == 'bf16':
            cell = cell.float()
        target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32).to(dt)
        opt = FlatAdam(cell.parameters(), lr=1e-3)

    runner = None
    if args.mode == 'fwd' and args.dtype == 'bf16' and args.hipgraph != 0:      # (node- / edge-gated cells too: the runner captures their own forward, tests/test_wide.py)
        from gated_gcrnns_amd.ops import FusedForwardGraph
        try:
            runner = FusedForwardGraph(cell, B, T, X=X, h0=h0)      # captured on the caller's own tensors: no staging copy
        except Exception as e:                                      # noqa: BLE001 -- default "auto": fall back to eager launches, say so
            if args.hipgraph:
                raise
            sys.stderr.write('bench.py: hipGraph capture failed (%s); eager launches\n' % e)
            runner = None

    def step():
        if args.mode == 'fwd':
            if runner is not None:
                return runner()               # ONE graph launch
            with torch.no_grad():
                return cell(X, h0)
        opt.zero_grad()
        loss = batchTimeL1Loss(cell(X, h0), target)      # the drivers' loss (reference miscTools.py:112-119), one fused pass
        loss.backward()
        if dist_on:
            opt.sync.all_reduce_()
        opt.step()
        return loss

    wall, dev_ms = timed_steps(ctx, step)
    value = world * B * args.steps / wall

    # ---- dominant kernel: the fused step kernel, timed live with HIP events on the stream it is launched on ----
    kern, native = None, None
    if args.dtype == 'bf16' and args.mode == 'fwd' and not args.time_gating and args.spatial_gating is None:
        from gated_gcrnns_amd import ops
        with torch.no_grad():
            for _ in range(2):
                ops.fused_cell_forward(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, return_states=True)
            torch.cuda.synchronize()
            # (10 back-to-back launches after a warm-up one: three catch the chip at a higher clock than a longer run holds -- a kernel trace of
            #  20 such forwards averaged 5 % above the 3-launch figure, profiles/r04_seq32_as_issued_only_kernel_stats.csv)
            KREPS = 30
            KWARM = 30 if args.settle_ms > 0 else 1      # untimed launches queued in front of the timed ones (the settle phase of the kernel timing)
            kern = ops.time_fused_step_kernel(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, reps=KREPS, warm=KWARM)
            if kern.get('inline_pack'):          # for comparison with earlier rounds: the same launches without the inline pack of x_{t+1}
                kern['bare_launch_avg_us'] = ops.time_fused_step_kernel(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, reps=KREPS, inline=False, warm=KWARM)['launch_avg_us']
            native = ops.time_fused_step_kernel(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, reps=KREPS, inline=False, user_layout=False, warm=KWARM)
    kern3 = None
    if args.dtype == 'f32' and args.mode == 'fwd' and not args.time_gating and args.spatial_gating is None:
        from gated_gcrnns_amd import ops
        with torch.no_grad():
            if cell._use_fused_x3(X, h0):
                kern3 = ops.time_fused_x3_kernel(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, reps=2)
    ar = None
    if args.mode == 'train':
        # the collective on its own: HIP events around back-to-back all-reduces of the flat gradient buffer
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        reps = 20
        e0.record()
        for _ in range(reps):
            opt.sync.all_reduce_(1.0)
        e1.record()
        torch.cuda.synchronize()
        ar = {'allreduce_bytes': opt.sync.nbytes(), 'allreduce_us': 1e3 * e0.elapsed_time(e1) / reps,
              'allreduce_note': 'one flat buffer per optimiser step (RCCL when n_gpus > 1; no collective at n_gpus = 1)'}
    dist_sec = None
    headline = (args.dtype == 'bf16' and args.mode == 'fwd' and not args.time_gating and args.spatial_gating is None and G == CFG['G'] and args.gso == 'adjacency')
    if dist_on and headline and not args.no_secondary:
        # launched under torch.distributed (the driver's --gpus N form): the forward line has no collective, so every rank also takes ONE
        # training point with the gradient all-reduce in it -- one driver command then measures inference scaling and the collective
        try:
            dist_sec = dist_train_point(ctx, S, params, B)
        except Exception as e:      # noqa: BLE001 -- never takes the headline down (all ranks raise or none: same code, same shapes)
            dist_sec = {'error': str(e)[:200]}
    if rank != 0:
        return None
    gating = ('time-gated' if args.time_gating else 'un-gated') + ('' if args.spatial_gating is None else ' + %s-gated' % args.spatial_gating)
    abytes = algorithmic_bytes_per_seq(T, N, G, F, elt) * B           # per step (= per launch chain), per GPU
    step_s = (dev_ms / 1e3) / args.steps
    achieved = abytes / step_s / 1e9
    out = base_line(ctx, value, wall, 'BASELINE configs[1]: synthetic k-step prediction, sparse SBM N=1000 nnz=%d%s, K=5 taps, '
                    'T=32, G=%d, F=64, %s GGCRNNCell %s, h0=0' % (nnz, ' (normalised adjacency)' if args.gso == 'normalized' else '', G, gating,
                                                                  'forward' if args.mode == 'fwd' else 'training step'),
                    B, {'hipgraph': bool(runner is not None)})
    if ar is not None:
        out['config'].update(ar)
    if kern is not None:
        # algorithmic bytes of ONE launch: per unit (one time step of one sequence) read x_t, read h_{t-1}, write h_t; a launch
        # of the sequence-resident persistent kernel covers T steps of the whole batch, one of the chunk-parallel kernel one step
        spl = kern['steps_per_launch']
        kbytes = elt * N * (G + 2 * F) * B * spl
        kach = kbytes / (kern['launch_avg_us'] * 1e-6) / 1e9
        tname = {'fused_seq32_kernel': 'seq32_kernel_traffic.json', 'fused_seq_kernel': 'seq_kernel_traffic.json'}.get(kern['kernel'], 'step_kernel_traffic.json')
        tj = profile_traffic(tname, B)
        Gp = 64 if G > 32 else 32
        mfma_flops = 2.0 * 1024 * K * F * (F + Gp) * B * spl           # executed on the matrix cores per launch (1024 padded node rows)
        kdesc = ('fused_seq32_kernel<5,2,%d,%d> (wide sequence-resident persistent kernel, 32-feature chunks: one launch = all T = %d time steps of the whole batch; '
                 'last template argument: bit 0 = inline pack, bit 1 = user-layout output -- native layout = <..,0>)' % (2 if G > 32 else 1, 3 if kern.get('inline_pack') else 2, spl)
                 if kern['kernel'] == 'fused_seq32_kernel' else
                 'fused_seq_kernel<5,2,%d,0> (sequence-resident persistent kernel: one launch = all T = %d time steps of the whole batch)' % (2 if G > 32 else 1, spl)
                 if kern['kernel'] == 'fused_seq_kernel' and spl > 1 else
                 '%s<5,2,%d> (one launch = one time step of the whole batch)' % (kern['kernel'], 2 if G > 32 else 1))
        out['roofline'] = {'bound': 'hbm', 'achieved': kach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': kach / HBM_PEAK_GBS, 'traffic': tj['hbm_bytes_per_launch'] if tj else None,
                           'kernel': kdesc, 'kernel_avg_us': kern['launch_avg_us'], 'launches_timed': kern['launches'],
                           'steps_per_launch': spl, 'us_per_time_step': kern['avg_us'],
                           'algorithmic_bytes_per_launch': kbytes,
                           'gflop_per_launch': flops_per_seq(1, N, nnz, K, G, F) * B * spl / 1e9,
                           'mfma_util': mfma_flops / (kern['launch_avg_us'] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                           'mfma_util_note': 'taps GEMM flops executed per launch / duration / 2.5 PF dense bf16 peak'
                                             + (' (PMC: SQ_VALU_MFMA_BUSY_CYCLES %.3g per launch)' % tj['mfma_busy_cycles'] if tj and 'mfma_busy_cycles' in tj else ''),
                           'whole_step_GBps': achieved, 'device_ms_per_step': 1e3 * step_s}
        if tj:
            # the PMC figures are NOT counted in this run (bench.py cannot run under rocprofv3's counter passes): they are replayed from a tracked
            # file, counted on the tree named inside it
            out['roofline']['traffic_source'] = {'file': 'profiles/' + tname, 'counted_on_tree': tj.get('counted_on_tree'), 'counted_by': tj.get('source'),
                                                 'note': 'PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, SQ_VALU_MFMA_BUSY_CYCLES) collected in separate rocprofv3 --pmc passes and '
                                                         'REPLAYED from the tracked file; not measured in this run'}
        out['roofline']['hop_state_image'] = ('bf16 rows, neighbour rows summed on the matrix cores (one-hot A operand; uniform-weight graph)'
                                              if ops.fused_img16_plan(cell.graph, False, None) is not None else 'fp32 rows, packed VALU sums')
        if kern.get('inline_pack'):
            # uniform-weight graph: every step ALSO lays out x_{t+1} (the former pack pass over X: 2 N G B more bytes per step that
            # `achieved` does not count). frac is that of the launch as issued; the bare kernel is timed next to it.
            out['roofline']['inline_pack'] = {'extra_bytes_per_launch': 2 * elt * N * G * B * spl, 'bare_kernel_avg_us': kern['bare_launch_avg_us'],
                                              'bare_frac': kbytes / (kern['bare_launch_avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                              'note': 'each step also lays out the next step input from the user layout (LDS-DMA during the '
                                                      'hops); the separate pack pass over X (0.5 ms per forward at B = 256) is gone'}
        if native is not None:
            # the same kernel on the sequence-major arrays alone (the cell's `native_layout` output is a view of the state
            # image): no user-layout copy of h_t, no lay-out of x_{t+1} -- the launch moves what the algorithm needs
            out['roofline_native_layout'] = {'bound': 'hbm', 'achieved': kbytes / (native['launch_avg_us'] * 1e-6) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                                             'frac': kbytes / (native['launch_avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                             'traffic': tj.get('native_hbm_bytes_per_launch') if tj else None,
                                             'kernel_avg_us': native['launch_avg_us'], 'us_per_time_step': native['avg_us'],
                                             'steps_per_launch': native['steps_per_launch'], 'algorithmic_bytes_per_launch': kbytes,
                                             'note': 'sequence-major X in, sequence-major H out (views): GGCRNNCell.forward_native'}
    elif kern3 is not None:
        # the fp32-accurate fused step (three bf16 planes per operand): algorithmic bytes = the fp32 tensors of the API
        kbytes = 4 * N * (G + 2 * F) * B
        kach = kbytes / (kern3['avg_us'] * 1e-6) / 1e9
        Gp = 64 if G > 32 else 32
        mfma_flops = 6 * 2.0 * 1024 * K * F * (F + Gp) * B              # six partial products per tap product
        out['roofline'] = {'bound': 'hbm', 'achieved': kach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': kach / HBM_PEAK_GBS,
                           'traffic': None,
                           'kernel': 'fused_step_x3_kernel<5,2,%d> (fp32-accurate: 3 bf16 planes per operand, 6 partial products; one launch = one time step)' % (2 if G > 32 else 1),
                           'kernel_avg_us': kern3['avg_us'], 'launches_timed': kern3['launches'], 'algorithmic_bytes_per_launch': kbytes,
                           'moved_bytes_per_launch_min': 6 * N * (G + 2 * F) * B + 4 * N * F * B,
                           'mfma_util': mfma_flops / (kern3['avg_us'] * 1e-6) / 1e12 / MFMA_BF16_PEAK_TFLOPS,
                           'mfma_util_note': 'bf16 MFMA flops executed per launch (6 x the tap GEMM) / duration / 2.5 PF',
                           'whole_step_GBps': achieved, 'device_ms_per_step': 1e3 * step_s}
    else:
        out['roofline'] = {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                           'frac': achieved / HBM_PEAK_GBS, 'traffic': None,
                           'kernel': 'whole T-step recurrence (all launches of one step: %s)' % (
                               'fused kernels' if args.dtype == 'bf16' else 'composed path'),
                           'algorithmic_bytes_per_step': abytes, 'device_ms_per_step': 1e3 * step_s,
                           'gflop_per_step': flops_per_seq(T, N, nnz, K, G, F) * B / 1e9}
    if (world == 1 and not args.no_secondary and args.dtype == 'bf16' and args.mode == 'fwd' and not args.time_gating
            and args.spatial_gating is None and G == CFG['G'] and args.gso == 'adjacency'):
        # (the headline is complete here: should a secondary point take the process down -- an asynchronous fault is not an exception --
        #  the line is already on stderr)
        sys.stderr.write('bench.py headline before the secondary points: %s\n' % json.dumps(out))
        sys.stderr.flush()
        del X, h0, runner, step
        out['secondary'] = secondary_points(ctx, S, params, B)
    if dist_sec is not None:
        out.setdefault('secondary', {})['train_bf16'] = dist_sec
    if not args.no_cpu_baseline and world == 1:          # the host baseline is a single-GPU-run item (rank 0, N = 1)
        out['cpu_baseline'] = cpu_baseline(S, params, T, G, F)
    return out


SETTLE_MS = 100.0      # set from --settle-ms by main(): untimed continuous load in front of a timed window (profiles/r04_clock_transient.txt)


def _timed(fn, steps, warmup):
    """(seconds per call) of fn: `warmup` untimed calls (+ the settle phase: calls queued back to back for about SETTLE_MS / 2, at most 40),
    then `steps` calls between two synchronisations."""
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    if SETTLE_MS > 0:
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        one = max(time.perf_counter() - t0, 1e-6)
        for _ in range(max(0, min(40, int(0.5e-3 * SETTLE_MS / one)))):
            fn()
        torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps


def dist_train_point(ctx, S, params, B, steps=5, warmup=2):
    """Run under torch.distributed (EVERY rank calls this; the driver's `--gpus N` line): one bf16 optimiser step of the same workload with
    the north_star's only collective in it -- zero_grad, forward, L1 loss, BPTT, ONE flat fp32 gradient all-reduce over RCCL
    (reference position: Modules/train_rnn.py:273 -> 276), FlatAdam -- timed with the bench's protocol (barrier + synchronize on both
    sides, MAX over ranks), and the collective alone between HIP events. Returns the point on rank 0, None elsewhere."""
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd.optim import FlatAdam
    from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
    dev, rank, world = ctx['dev'], ctx['rank'], ctx['world']
    dist = torch.distributed
    N, K, T, G, F = CFG['N'], CFG['K'], CFG['T'], CFG['G'], CFG['F']
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(torch.tensor(S))
    cell.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
    cell = cell.to(dev).float()
    gen = torch.Generator(device=dev); gen.manual_seed(4321 + rank)
    X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
    h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
    target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
    opt = FlatAdam(cell.parameters(), lr=1e-3)

    def tstep():
        opt.zero_grad()
        batchTimeL1Loss(cell(X, h0), target).backward()
        opt.sync.all_reduce_()
        opt.step()

    def sync():
        dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        tstep()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        tstep()
    sync()
    tw = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    dist.all_reduce(tw, op=dist.ReduceOp.MAX)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    sync()
    e0.record()
    for _ in range(reps):
        opt.sync.all_reduce_(1.0)
    e1.record()
    torch.cuda.synchronize()
    sync()
    if rank != 0:
        return None
    dt = float(tw.item()) / steps
    return {'value': world * B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': steps, 'dtype': 'bf16', 'n_gpus': world,
            'what': 'zero_grad, forward, L1 loss, BPTT (data chain + weight gradient), ONE flat fp32 gradient all-reduce, FlatAdam; fp32 master '
                    'weights; %d sequences per GPU (weak scaling), wall = MAX over ranks' % B,
            'allreduce_bytes': opt.sync.nbytes(), 'allreduce_us': 1e3 * e0.elapsed_time(e1) / reps, 'ranks': opt.sync.world,
            'allreduce_note': 'one flat buffer per optimiser step (RCCL when n_gpus > 1; a process group of one rank reduces over itself)'}


def secondary_points(ctx, S, params, B, steps=8, warmup=2):
    """Two more points of the SAME workload, timed in this process next to the headline (single GPU): the forward at the north_star's
    1e-5 fp32 tolerance (fused three-plane kernels) and one bf16 training step (fp32 master weights: forward, L1 loss, BPTT, Adam).
    At most `steps` timed calls each; the headline fields are unaffected."""
    import gc
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    from gated_gcrnns_amd.optim import FlatAdam
    from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
    dev = ctx['dev']
    N, K, T, G, F = CFG['N'], CFG['K'], CFG['T'], CFG['G'], CFG['F']
    sec = {}

    def fresh_cell(dtype):
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
        c.addGSO(torch.tensor(S))
        c.load_state_dict({k: torch.tensor(v) for k, v in params.items()})
        return c.to(dev).to(dtype)

    gen = torch.Generator(device=dev); gen.manual_seed(99)
    # ---- fp32-accurate forward (x3 kernels): <= 1e-5 against the fp64 oracle (tests/test_fused.py) ----
    try:
        cell = fresh_cell(torch.float32)
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.float32)
        with torch.no_grad():
            if cell._use_fused_x3(X, h0):
                dt = _timed(lambda: cell(X, h0), steps, warmup)
                k3 = ops.time_fused_x3_kernel(X, h0, cell.weight_A, cell.weight_B, cell.bias, cell.graph, reps=1)
                kb = 4 * N * (G + 2 * F) * B
                sec['f32_x3'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': steps, 'dtype': 'f32',
                                 'tolerance': '<= 1e-5 abs against the fp64 oracle (north_star)', 'kernel_avg_us': k3['avg_us'],
                                 'frac': kb / (k3['avg_us'] * 1e-6) / 1e9 / HBM_PEAK_GBS, 'algorithmic_bytes_per_launch': kb,
                                 'kernel': 'fused_step_x3_kernel (one launch = one time step; bytes = the fp32 tensors of the API)'}
            # ... and ONE optimiser step in the same precision: x3 forward, x3 data chain, exact-fp32 weight gradient (G11: <= 2e-5 of each
            # gradient's max against the reference's autograd)
            with torch.enable_grad():
                if cell._use_fused_x3_training(X, h0):
                    target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32, generator=gen)
                    opt = FlatAdam(cell.parameters(), lr=1e-3)

                    def t32():
                        opt.zero_grad()
                        batchTimeL1Loss(cell(X, h0), target).backward()
                        opt.step()

                    dt = _timed(t32, 3, 1)
                    sec['train_f32_x3'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 3, 'dtype': 'f32',
                                           'tolerance': 'gradients <= 2e-5 of their max against the reference autograd (tests G11)',
                                           'what': 'x3 forward + x3 BPTT data chain + exact-fp32 weight gradient + FlatAdam'}
                    del target, opt
        del cell, X, h0
    except Exception as e:      # noqa: BLE001 -- a secondary point never takes the headline line down
        sec['f32_x3'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    # ---- bf16 training step ----
    try:
        cell = fresh_cell(torch.float32)
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
        target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        opt = FlatAdam(cell.parameters(), lr=1e-3)

        def tstep():
            opt.zero_grad()
            batchTimeL1Loss(cell(X, h0), target).backward()
            opt.step()

        dt = _timed(tstep, steps, warmup)
        sec['train_bf16'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': steps, 'dtype': 'bf16',
                             'what': 'zero_grad, forward, L1 loss, BPTT (data chain + weight gradient), FlatAdam; fp32 master weights',
                             'frac': algorithmic_bytes_per_seq(T, N, G, F, 2) * B / dt / 1e9 / HBM_PEAK_GBS,
                             'frac_note': 'forward algorithmic bytes / step time (as --mode train reports it)'}
        del cell, X, h0, target, opt
    except Exception as e:      # noqa: BLE001
        sec['train_bf16'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    # ---- bf16 training step of the TIME-GATED cell (the reference's default cell, graphML.py:2196): both gate sub-networks trained too ----
    try:
        torch.manual_seed(0)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
        cell.addGSO(torch.tensor(S))
        cell = cell.to(dev).float()
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
        target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        opt = FlatAdam(cell.parameters(), lr=1e-3)

        def tgstep():
            opt.zero_grad()
            batchTimeL1Loss(cell(X, h0), target).backward()
            opt.step()

        dt = _timed(tgstep, max(3, steps // 3), 2)
        sec['train_timegated_bf16'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': max(3, steps // 3), 'dtype': 'bf16',
                                       'what': 'time-gated cell: zero_grad, forward (gate pre-pass + gated recurrence), L1 loss, BPTT (chain, gate gradients, three weight '
                                               'gradients, read-outs), FlatAdam; fp32 master weights'}
        del cell, X, h0, target, opt
    except Exception as e:      # noqa: BLE001
        sec['train_timegated_bf16'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    # ---- time-gated cell at the north_star's 1e-5 (x3 kernels: gates = x3 steps from h0 + read-out, recurrence = scaled x3 steps) ----
    try:
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
        c.addGSO(torch.tensor(S))
        c = c.to(dev).float()
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.float32)
        with torch.no_grad():
            if c._use_fused_x3(X, h0, time_gated=True):
                dt = _timed(lambda: c(X, h0), 3, 1)
                sec['fwd_timegated_f32_x3'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 3, 'dtype': 'f32',
                                               'tolerance': '<= 1e-5 abs against the fp64 oracle (tests/test_fused.py)',
                                               'what': 'GGCRNNCell(time_gating=True) forward on the fp32-accurate fused kernels'}
        # ... and ONE optimiser step of the time-gated cell in the same precision (what the reference's drivers train, in their precision:
        # train_rnn.py:247-281 under kStepPredGRNNs.py:44): G12, every gradient <= 2e-5 of its max against the reference's autograd
        with torch.enable_grad():
            if c._use_fused_x3_training(X, h0, time_gated=True):
                target = torch.randn(B, T, F, N, device=dev, dtype=torch.float32, generator=gen)
                opt = FlatAdam(c.parameters(), lr=1e-3)

                def tg32():
                    opt.zero_grad()
                    batchTimeL1Loss(c(X, h0), target).backward()
                    opt.step()

                dt = _timed(tg32, 2, 1)
                sec['train_timegated_f32_x3'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 2, 'dtype': 'f32',
                                                 'tolerance': 'gradients <= 2e-5 of their max against the reference autograd (tests G12)',
                                                 'what': 'gate cells + scaled recurrence (x3), gated x3 data chain, x3 filter pass (d gi), exact-fp32 '
                                                         'weight gradients of the cell and both gate cells, FlatAdam'}
                del target, opt
        del c, X, h0
    except Exception as e:      # noqa: BLE001
        sec['fwd_timegated_f32_x3'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    # ---- node-gated cell at the north_star's 1e-5 (round 5: gate cells as x3 steps, x3 filter passes, per-node gating in fp32; G13) ----
    try:
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, 'node', 1, True)
        c.addGSO(torch.tensor(S))
        c = c.to(dev).float()
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.float32)
        with torch.no_grad():
            if c._use_fused_x3_node(X, h0):
                dt = _timed(lambda: c(X, h0), 2, 1)
                sec['fwd_nodegated_f32_x3'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 2, 'dtype': 'f32',
                                               'tolerance': '<= 1e-5 abs against the reference fixture G13 and the fp64 oracle (tests/test_fused.py)',
                                               'what': 'GGCRNNCell(spatial_gating=node) forward on the fp32-accurate fused kernels (ops.fused_node_cell_forward_x3)'}
        del c, X, h0
    except Exception as e:      # noqa: BLE001
        sec['fwd_nodegated_f32_x3'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    # ---- edge-gated cell at 1e-5: both filters as x3 filter passes, the attentions on the fp32 edge-softmax kernels (G14); beside it the composed fp32 path ----
    try:
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, 'edge', 1, True)
        c.addGSO(torch.tensor(S))
        c = c.to(dev).float()
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.float32)
        with torch.no_grad():
            if c._use_fused_x3_edge(X, h0):
                dt = _timed(lambda: c(X, h0), 2, 1)
                os.environ['GCRNN_NO_X3_EDGE'] = '1'
                try:
                    dtc = _timed(lambda: c(X, h0), 1, 1)
                finally:
                    del os.environ['GCRNN_NO_X3_EDGE']
                sec['fwd_edgegated_f32_x3'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 2, 'dtype': 'f32',
                                               'composed_fp32_path_sequences_per_s': B / dtc,
                                               'tolerance': '<= 1e-5 abs against the reference fixture G14 and the fp64 oracle (tests/test_fused.py)',
                                               'what': 'GGCRNNCell(spatial_gating=edge) forward, filters on the fp32-accurate fused kernels (ops.fused_edge_cell_forward_x3)'}
        del c, X, h0
    except Exception as e:      # noqa: BLE001
        sec['fwd_edgegated_f32_x3'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    # ---- gated cells, bf16 forward (reference graphML.py:2357-2407, 2420-2423; random-init gate sub-networks of the reference's shapes) ----
    for name, tg, sg in (('fwd_timegated', True, None), ('fwd_nodegated', False, 'node'), ('fwd_edgegated', False, 'edge')):
        try:
            torch.manual_seed(0)
            c = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
            c.addGSO(torch.tensor(S))
            c = c.to(dev).to(torch.bfloat16)
            X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
            h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
            graphed = False
            with torch.no_grad():
                fn = lambda: c(X, h0)
                if ctx['args'].hipgraph != 0:      # as the headline: the forward replayed as ONE captured hipGraph (ops.FusedForwardGraph captures the gated cells' own forward)
                    try:
                        from gated_gcrnns_amd.ops import FusedForwardGraph
                        fn = FusedForwardGraph(c, B, T, X=X, h0=h0)
                        graphed = True
                    except Exception:      # noqa: BLE001
                        fn = lambda: c(X, h0)
                dt = _timed(fn, 5, 2)
            sec[name] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 5, 'dtype': 'bf16', 'hipgraph': graphed,
                         'what': 'GGCRNNCell(time_gating=%s, spatial_gating=%s) forward, same workload' % (tg, sg)}
            del c, X, h0, fn
        except Exception as e:      # noqa: BLE001
            sec[name] = {'error': str(e)[:200]}
        gc.collect(); torch.cuda.empty_cache()
    # ---- the same workload on the NORMALISED adjacency D^-1/2 W D^-1/2 / lambda_max (reference Utils/graphTools.py:64): a rank-1-weighted GSO ----
    try:
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
        c.addGSO(torch.tensor(sbm_graph(N, normalized=True)))
        c = c.to(dev).to(torch.bfloat16)
        X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            dt = _timed(lambda: c(X, h0), 5, 2)
        sec['fwd_normalized_adjacency'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 5, 'dtype': 'bf16',
                                           'kernel': 'fused_seq32_kernel<..,R1>' if c.graph.fused_plan_rank1() is not None else 'chunk-parallel weighted path',
                                           'what': 'un-gated forward, GSO = normalizeAdjacency(W) / lambda_max (S[m][n] = a[m] b[n] on the support), same workload'}
        del c
        # ... and the reference's default cell (time_gating=True, graphML.py:2196) on it: gate pair pre-pass + gated recurrence on the R1 variants
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, True, None, 1, True)
        c.addGSO(torch.tensor(sbm_graph(N, normalized=True)))
        c = c.to(dev).to(torch.bfloat16)
        with torch.no_grad():
            dt = _timed(lambda: c(X, h0), 5, 2)
        sec['fwd_timegated_normalized_adjacency'] = {'value': B / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 5, 'dtype': 'bf16',
                                                     'what': 'GGCRNNCell(time_gating=True) forward on the normalised adjacency, same workload'}
        del c, X, h0
    except Exception as e:      # noqa: BLE001
        sec.setdefault('fwd_normalized_adjacency', {'error': str(e)[:200]})
        sec.setdefault('fwd_timegated_normalized_adjacency', {'error': str(e)[:200]})
    gc.collect(); torch.cuda.empty_cache()
    # ---- beyond the fused kernels' 1024 nodes (short of configs[4]): the same cell on an SBM of N = 2048, Horner-form streaming path ----
    try:
        N2 = 2048
        torch.manual_seed(0)
        c = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
        c.addGSO(torch.tensor(sbm_graph(N2, p_in=0.02, p_out=0.00125)))        # same mean degree (~10) as the headline graph
        c = c.to(dev).to(torch.bfloat16)
        B2 = B // 2                                                            # the same number of (sequence, node) pairs as the headline batch
        X = torch.randn(B2, T, G, N2, device=dev, dtype=torch.float32, generator=gen).to(torch.bfloat16)
        h0 = torch.zeros(B2, F, N2, device=dev, dtype=torch.bfloat16)
        with torch.no_grad():
            assert not c._use_fused(X, h0)
            dt = _timed(lambda: c(X, h0), 3, 1)
        sec['fwd_n2048_streaming'] = {'value': B2 / dt, 'unit': 'sequences/s', 'ms_per_step': 1e3 * dt, 'steps': 3, 'dtype': 'bf16', 'batch': B2,
                                      'frac': algorithmic_bytes_per_seq(T, N2, G, F, 2) * B2 / dt / 1e9 / HBM_PEAK_GBS,
                                      'what': 'un-gated forward at N = 2048 nodes (the fused kernels hold N <= 1024): Horner-form streaming path, '
                                              'gcrnn_taps_bf16 + gcrnn_spmm per hop; frac on the compulsory bytes T s N (G + 2F) per sequence'}
        del c, X, h0
    except Exception as e:      # noqa: BLE001
        sec['fwd_n2048_streaming'] = {'error': str(e)[:200]}
    gc.collect(); torch.cuda.empty_cache()
    return sec


def run_cfg5(ctx):
    """BASELINE configs[4]: directed Erdos-Renyi graph N = 1e5, density 1e-3 (nnz ~ 1e7), K = 3 taps, T = 16, G = F = 32,
    B = 8 sequences per GPU, un-gated cell, h0 = 0 -- the streaming regime: every hop is one CSR SpMM pass over the
    [N][B F] state (GGCRNNCell._forward_horner). The reference cannot run it (dense 1e5 x 1e5 GSO = 40 GB)."""
    args, rank, world, dev = ctx['args'], ctx['rank'], ctx['world'], ctx['dev']
    import gated_gcrnns_amd.Utils.graphML as gml
    from gated_gcrnns_amd import ops
    from gated_gcrnns_amd.graph import erdos_renyi_csr, operator_from_csr
    N, K, T, G, F = 100000, 3, 16, 32, 32
    B = args.batch
    assert args.mode == 'fwd', 'cfg5 is an inference workload'
    dt = {'bf16': torch.bfloat16, 'f32': torch.float32, 'f64': torch.float64}[args.dtype]
    elt = {'bf16': 2, 'f32': 4, 'f64': 8}[args.dtype]
    rowptr, col, val = erdos_renyi_csr(N, 1e-3, seed=0)
    nnz = int(col.size)
    graph = operator_from_csr(rowptr, col, val, N, device=dev)
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, False, None, 1, True)
    cell.addGSO(graph)
    params = {k: v.detach().double().numpy().copy() for k, v in cell.state_dict().items()}
    cell = cell.to(dev).to(dt)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    X = torch.randn(B, T, G, N, device=dev, dtype=torch.float32, generator=gen).to(dt)
    h0 = torch.zeros(B, F, N, device=dev, dtype=dt)

    def step():
        with torch.no_grad():
            return cell(X, h0)

    wall, dev_ms = timed_steps(ctx, step)
    value = world * B * args.steps / wall
    # dominant kernel: spmm_stream_kernel, one launch = one hop of one time step for the whole batch; timed live with HIP events
    acc = torch.randn(1, N, B, F, device=dev).to(dt)
    dst = torch.randn(1, N, B, F, device=dev).to(dt)
    for _ in range(3):
        ops.spmm_raw(graph.fwd[0], acc, out=dst, accumulate=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    torch.cuda.synchronize()
    e0.record()
    for _ in range(reps):
        ops.spmm_raw(graph.fwd[0], acc, out=dst, accumulate=True)
    e1.record()
    torch.cuda.synchronize()
    hop_us = 1e3 * e0.elapsed_time(e1) / reps
    if rank != 0:
        return None
    csr_bytes = nnz * (4 + (8 if args.dtype == 'f64' else 4)) + 4 * (N + 1)
    hop_survey = csr_bytes + 2 * elt * N * B * (G + F)           # SURVEY 8d hop-streaming bytes of one hop: CSR + read + write of the (G+F)-channel operand
    hop_horner = csr_bytes + 3 * elt * N * B * F                 # what this launch must move at least: CSR + read acc + read u_k + write
    gathered = nnz * B * F * elt                                 # bytes the gather pulls through the caches (every non-zero fetches one B*F row)
    step_survey = (K - 1) * hop_survey + elt * N * B * (G + 2 * F)
    ach = hop_survey / (hop_us * 1e-6) / 1e9
    out = base_line(ctx, value, wall, 'BASELINE configs[4]: directed Erdos-Renyi N=%d nnz=%d (density 1e-3), K=3 taps, T=16, G=F=32, '
                    'un-gated GGCRNNCell forward on the streaming CSR path, h0=0' % (N, nnz), B)
    out['metric'] = 'sequences/sec (node), N=100k density 1e-3 K=3 T=16 F=32'
    tj = profile_traffic('cfg5_spmm_traffic.json', B)
    out['roofline'] = {'bound': 'hbm', 'achieved': ach, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': ach / HBM_PEAK_GBS,
                       'traffic': tj['hbm_bytes_per_launch'] if tj and tj.get('dtype') == args.dtype else None,
                       'kernel': 'spmm_stream_kernel (one launch = one graph hop of one time step, whole batch)',
                       'kernel_avg_us': hop_us, 'launches_timed': reps,
                       'algorithmic_bytes_per_launch': hop_survey,
                       'algorithmic_bytes_note': 'SURVEY 8d hop-streaming: CSR + 2 s N B (G+F); the Horner-form hop this kernel runs needs '
                                                 'CSR + 3 s N B F = %d bytes; the gather itself pulls %d bytes per launch through the caches '
                                                 '(every non-zero one B*F-wide row)' % (hop_horner, gathered),
                       'gathered_GBps': gathered / (hop_us * 1e-6) / 1e9,
                       'whole_step_GBps': step_survey * T / ((dev_ms / 1e3) / args.steps) / 1e9,
                       'ms_per_time_step': dev_ms / args.steps / T}
    if not args.no_cpu_baseline and world == 1:
        out['cpu_baseline'] = cpu_baseline_cfg5(rowptr, col, val, params, N, B, T, G, F)
    return out


def cpu_baseline_cfg5(rowptr, col, val, params, N, B, T, G, F):
    """The oracle's CSR restatement (the reference itself cannot hold this graph) on ONE time step of the batch, all nodes."""
    import scipy.sparse as sp
    from oracle import gcrnn_oracle as orc
    P = sp.csr_matrix((val, col, rowptr), shape=(N, N)).T.tocsr()
    rng = np.random.default_rng(1)
    x = rng.standard_normal((B, G, N)).astype(np.float32).astype(np.float64)
    h = np.tanh(rng.standard_normal((B, F, N)))
    t0 = time.perf_counter()
    orc.cell_step_rows_csr(params, P, x, h, np.arange(N))
    dt_ = time.perf_counter() - t0
    return {'value': B / (T * dt_), 'unit': 'sequences/s', 'cores': 1, 'kind': 'port',
            'sample': 'oracle/gcrnn_oracle.py cell_step_rows_csr (scipy CSR products, fp64, single thread) on ONE of the T=%d time steps '
                      'of the B=%d batch, all %d nodes: %.1f s; value = B / (T x that)' % (T, B, N, dt_)}


def run_cfg4(ctx):
    """BASELINE configs[3]: the seismograph graph of the reference (Adj.p: directed, N = 59, 10 in-neighbours per node), T = 200,
    K = 3 taps, G = 1, F = 20, B = 100, fp64 like the reference drivers -- latency-bound: 200 dependent steps of a tiny
    problem, run by the one-launch small-graph kernels (one workgroup per sequence on the fp64 matrix cores)."""
    args, rank, world, dev = ctx['args'], ctx['rank'], ctx['world'], ctx['dev']
    import gated_gcrnns_amd.Utils.graphML as gml
    A = np.load(os.path.join(ROOT, 'tests', 'golden', 'adj59.npy')).astype(np.float64)
    N, K, T, G, F = A.shape[0], 3, 200, 1, 20
    B = args.batch
    dt = {'bf16': torch.float64, 'f32': torch.float32, 'f64': torch.float64}[args.dtype]      # default dtype of this config: fp64
    S = (A / np.max(np.abs(np.linalg.eigvals(A)))).reshape(1, N, N)                              # epicenterEstimation.py:619
    torch.manual_seed(0)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, args.time_gating, None, 1, True)
    with torch.no_grad():
        cell.weight_B.mul_(0.25)            # as in the T=200 golden: the G=1 init makes the 20x20 state map chaotic
    cell.addGSO(torch.tensor(S))
    params = {k: v.detach().double().numpy().copy() for k, v in cell.state_dict().items()}
    cell = cell.to(dev).to(dt)
    gen = torch.Generator(device=dev); gen.manual_seed(1234 + rank)
    X = torch.randn(B, T, G, N, device=dev, dtype=torch.float64, generator=gen).to(dt)
    h0 = torch.zeros(B, F, N, device=dev, dtype=dt)
    if args.mode != 'fwd':
        raise SystemExit('bench.py: --config cfg4 times the forward only (training steps at the drivers\' sizes: tools/graphed_train_bench.py)')

    def step():
        with torch.no_grad():
            return cell(X, h0)

    wall, dev_ms = timed_steps(ctx, step)
    value = world * B * args.steps / wall
    if rank != 0:
        return None
    nnz = int(np.count_nonzero(S))
    us_per_time_step = 1e3 * dev_ms / args.steps / T
    # matrix-core work of one time step of ONE sequence (dense-S kernels): hops (K-1)(G+F) N^2 + taps K F (G+F) N, x2 flops
    flop_step = 2.0 * ((K - 1) * (G + F) * N * N + K * F * (G + F) * N)
    cus = torch.cuda.get_device_properties(dev).multi_processor_count
    ach = flop_step * min(B, cus) / (us_per_time_step * 1e-6) / 1e12          # one workgroup (sequence) per CU runs concurrently
    dname = 'f64' if dt == torch.float64 else 'f32'
    peak = MFMA_F64_PEAK_TFLOPS if dt == torch.float64 else 157.3
    out = base_line(ctx, value, wall, 'BASELINE configs[3]: seismograph graph N=59 nnz=%d (directed), T=200, K=3 taps, G=1, F=20, %s '
                    'GGCRNNCell forward, one launch for the whole recurrence, h0=0' % (nnz, 'time-gated' if args.time_gating else 'un-gated'), B)
    out['metric'] = 'sequences/sec (node), N=59 K=3 T=200 F=20'
    out['dtype'] = dname
    out['roofline'] = {'bound': 'mfma', 'achieved': ach, 'peak': peak, 'unit': 'TFLOP/s', 'frac': ach / peak, 'traffic': None,
                       'kernel': 'small_dense kernels: whole T-step recurrence in one launch, one workgroup per sequence',
                       'us_per_time_step': us_per_time_step,
                       'note': 'latency-bound configuration (SURVEY 8d): %d of %d CUs hold one sequence each and walk 200 dependent steps; '
                               'the figure to read is us_per_time_step against the ~1.2-1.5 us a dependent kernel boundary would cost per step '
                               '(MI355X_MICROARCH.md, row boundary) if the loop were launches' % (min(B, cus), cus)}
    if not args.no_cpu_baseline and world == 1:
        from oracle import gcrnn_oracle as orc
        Xc = X[:16].double().cpu().numpy()
        t0 = time.perf_counter()
        orc.ggcrnn_cell(params, S, Xc, np.zeros((16, F, N)), bool(args.time_gating), None)
        dt_ = time.perf_counter() - t0
        out['cpu_baseline'] = {'value': 16 / dt_, 'unit': 'sequences/s', 'cores': min(16, os.cpu_count() or 1), 'kind': 'port',
                               'sample': 'oracle ggcrnn_cell (dense x@S, numpy fp64) on 16 full T=200 sequences: %.2f s' % dt_}
    return out


if __name__ == '__main__':
    main()
