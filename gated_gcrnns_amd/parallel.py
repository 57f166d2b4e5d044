"""Batch-sharded data parallelism for the GCRNN recurrence.

Sequences are independent in the forward pass and in BPTT; S and the parameters are replicated. The only
communication is ONE all-reduce of ONE flat gradient buffer per optimiser step (164 KB for the plain
cell at K=5, G=F=64; ~1.4 MB time-gated) -- latency-bound over xGMI, so it is a single contiguous RCCL call,
never a ring over per-parameter tensors. Inference needs no communication at all.

The parameters' `.grad` tensors ARE views into that flat buffer (autograd accumulates into them in place), so there is
no pack / unpack pass around the collective: zero_grad is one memset, the all-reduce reads the buffer where backward
left it, and the optimiser (optim.FlatAdam: one kernel over the flat buffers) reads it where the collective left it.

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
    """One flat gradient buffer (fp32; fp64 when the parameters are fp64) whose slices are the `.grad` of `params`.
    Parameters that never receive a gradient (the reference's unused output gate GFL_out / MLP_out) keep their zeros,
    so every rank reduces the same layout.

        sync = FlatGradAllReduce(model.parameters())
        sync.zero_grad(); loss.backward(); sync.all_reduce_(weight=local_batch / global_batch); optim.step()

    Parameters whose dtype differs from the buffer's (bf16 parameters) cannot alias it: they are copied in and out
    around the collective (`staged`), the only case that still costs per-parameter launches.
    """

    def __init__(self, params, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device('cpu')
        self.dtype = torch.float64 if any(p.dtype == torch.float64 for p in self.params) else torch.float32
        self.flat = torch.zeros(self.numel, dtype=self.dtype, device=dev)
        self.views, self.staged = [], []
        off = 0
        for p in self.params:
            n = p.numel()
            v = self.flat[off:off + n].view_as(p)
            self.views.append(v)
            if p.dtype == self.dtype:
                p.grad = v                              # autograd accumulates in place: the view survives backward()
            else:
                self.staged.append((p, v))
            off += n

    @property
    def world(self):
        """Read at every use, never cached: the object may be built (optim.FlatAdam does, at model-building time) before
        dist.init_process_group -- a world size frozen at 1 there would skip the collective and silently train on scaled gradients."""
        return dist.get_world_size(self.group) if dist.is_available() and dist.is_initialized() else 1

    def nbytes(self):
        return self.numel * self.flat.element_size()

    def attach_(self):
        """Re-point the `.grad` of every aliased parameter at its slice (after a zero_grad(set_to_none=True) elsewhere)."""
        for p, v in zip(self.params, self.views):
            if p.dtype == self.dtype and p.grad is not v:
                if p.grad is not None:
                    v.copy_(p.grad)
                p.grad = v

    def zero_grad(self):
        """One memset instead of one per parameter. Gradients that live OUTSIDE the buffer (a backward after someone set the grads to
        None) are dropped, not copied back in: zero_grad means zero."""
        self.flat.zero_()
        for p, _ in self.staged:
            p.grad = None
        for p, v in zip(self.params, self.views):
            if p.dtype == self.dtype and p.grad is not v:
                p.grad = v

    def all_reduce_(self, weight=None):
        """Sum of per-rank gradients, each pre-scaled by `weight` (default 1/world: equal local batches and
        mean-reduced local losses then give exactly the gradient of the global-batch mean loss)."""
        world = self.world
        if weight is None:
            weight = 1.0 / world
        elif world == 1 and weight != 1.0 and dist.is_available() and dist.is_initialized():
            raise RuntimeError('FlatGradAllReduce: a shard weight of %g in a process group of one rank' % weight)
        self.attach_()
        for p, v in self.staged:
            if p.grad is None:
                v.zero_()
            else:
                v.copy_(p.grad)
        if weight != 1.0:
            self.flat.mul_(weight)
        if world > 1:
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        for p, v in self.staged:
            p.grad = v.to(p.dtype)
        return self.flat
