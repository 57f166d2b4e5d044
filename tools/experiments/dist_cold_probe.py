# Diagnosis (on the GPU box): per-step HOST and DEVICE times of the un-synchronised training loop under torch.distributed.run,
# first process on a cold box:   python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 tools/dist_cold_probe.py
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import bench
import torch.distributed as dist
local_rank = int(os.environ.get('LOCAL_RANK', '0'))
torch.cuda.set_device(local_rank)
dev = torch.device('cuda', local_rank)
use_dist = 'RANK' in os.environ
if use_dist:
    os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
    dist.init_process_group('nccl', device_id=dev)
args = bench.parse_args(['--mode', 'train', '--no-cpu-baseline'])
args.batch = 256
ctx = dict(args=args, rank=0, world=1, dev=dev, dist_on=use_dist)
orig = bench.timed_steps
def timed(ctx, step):
    for _ in range(2):
        step()
    if use_dist:
        dist.barrier()
    torch.cuda.synchronize()
    n = 24
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(n + 1)]
    host = [time.perf_counter()]
    evs[0].record()
    for i in range(n):
        step()
        evs[i + 1].record()
        host.append(time.perf_counter())
    torch.cuda.synchronize()
    print('host ms:', ' '.join('%.1f' % (1e3 * (host[i + 1] - host[i])) for i in range(n)), flush=True)
    print('dev  ms:', ' '.join('%.1f' % evs[i].elapsed_time(evs[i + 1]) for i in range(n)), flush=True)
    args.steps, args.warmup = 3, 0
    return orig(ctx, step)
bench.timed_steps = timed
bench.run_cfg2(ctx)
if use_dist:
    dist.destroy_process_group()
