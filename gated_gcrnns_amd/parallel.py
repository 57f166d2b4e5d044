"""Batch-sharded data parallelism for the GCRNN recurrence.

Sequences are independent in the forward pass and in BPTT; S and the parameters are replicated. The only
communication is ONE all-reduce of ONE flat fp32 gradient buffer per optimiser step (164 KB for the plain
cell at K=5, G=F=64; ~1.4 MB time-gated) -- latency-bound over xGMI, so it is a single contiguous RCCL call,
never a ring over per-parameter tensors. Inference needs no communication at all.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n, rank, world):
    """Contiguous, balanced [lo, hi) slice of n items for `rank` (sizes differ by at most one)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_batch(rank, world, *tensors):
    """Slice every tensor along dim 0 with the same shard_range."""
    lo, hi = shard_range(tensors[0].shape[0], rank, world)
    out = tuple(t[lo:hi] for t in tensors)
    return out if len(out) > 1 else out[0]


class FlatGradAllReduce(object):
    """Owns one flat fp32 buffer that mirrors the gradients of `params` (parameters without gradient, such as
    the reference's unused output gate GFL_out / MLP_out, contribute zeros so every rank reduces the same layout).

        sync = FlatGradAllReduce(model.parameters())
        loss.backward(); sync.all_reduce_(weights=local_batch / global_batch); optim.step()
    """

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.flat = torch.zeros(self.numel, dtype=torch.float32, device=dev)
        self.world = dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1

    def nbytes(self):
        return self.numel * 4

    def pack_(self, scale=1.0):
        off = 0
        for p in self.params:
            n = p.numel()
            if p.grad is None:
                self.flat[off:off + n].zero_()
            else:
                self.flat[off:off + n].copy_(p.grad.reshape(-1))
                if scale != 1.0:
                    self.flat[off:off + n].mul_(scale)
            off += n

    def unpack_(self):
        off = 0
        for p in self.params:
            n = p.numel()
            g = self.flat[off:off + n].view_as(p).to(p.dtype)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)
            off += n

    def all_reduce_(self, weight=None):
        """Sum of per-rank gradients, each pre-scaled by `weight` (default 1/world: equal local batches and
        mean-reduced local losses then give exactly the gradient of the global-batch mean loss)."""
        if weight is None:
            weight = 1.0 / self.world
        self.pack_(weight)
        if self.world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.unpack_()
        return self.flat
