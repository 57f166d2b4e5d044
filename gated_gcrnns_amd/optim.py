"""Optimiser of the GCRNN training loop on flat buffers.

The reference steps `torch.optim.Adam(lr, betas)` per model (kStepPredGRNNs.py:158-161, 794-796; stepped at
Modules/train_rnn.py:276). Here every parameter is a view into ONE flat parameter buffer and every `.grad` a view into
ONE flat gradient buffer (parallel.FlatGradAllReduce -- the buffer the data-parallel all-reduce reduces in place), so an
optimiser step is one HIP kernel over the flat buffers (C ABI `gcrnn_adam_flat`) instead of ~10 launches per tensor, and
the step counter lives on the device: the whole zero_grad -> forward -> loss -> BPTT -> Adam sequence is capturable as one
hipGraph with the all-reduce outside.
"""
import ctypes as C

import torch

from .parallel import FlatGradAllReduce


class FlatAdam(object):
    """Adam (no weight decay, no amsgrad -- the drivers' configuration) over the flat buffers of `params`.

        opt = FlatAdam(model.parameters(), lr=1e-3, betas=(0.9, 0.999))
        opt.zero_grad(); loss.backward(); opt.sync.all_reduce_(); opt.step()

    Parameters must share one dtype (fp32 or fp64: master weights) and one device. On a CPU tensor set (the gloo tests)
    the same update is evaluated with torch ops on the flat views -- the HIP kernel is the product path on a GPU."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, sync=None):
        self.sync = sync if sync is not None else FlatGradAllReduce(params)
        ps = self.sync.params
        assert ps, 'no trainable parameters'
        dt, dev = ps[0].dtype, ps[0].device
        assert all(p.dtype == dt and p.device == dev for p in ps), 'FlatAdam: one dtype and one device'
        assert dt in (torch.float32, torch.float64) and self.sync.dtype == dt
        self.lr, self.betas, self.eps = float(lr), (float(betas[0]), float(betas[1])), float(eps)
        self.flat_p = torch.empty(self.sync.numel, dtype=dt, device=dev)
        off = 0
        with torch.no_grad():
            for p in ps:
                n = p.numel()
                self.flat_p[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat_p[off:off + n].view_as(p)          # the module's tensors now alias the flat buffer
                off += n
        self.m = torch.zeros_like(self.flat_p)
        self.v = torch.zeros_like(self.flat_p)
        self.step_dev = torch.zeros(1, dtype=torch.int64, device=dev)

    def zero_grad(self, set_to_none=False):
        self.sync.zero_grad()

    def _check_alias(self):
        off = 0
        for p in self.sync.params:
            if p.data_ptr() != self.flat_p.data_ptr() + off * self.flat_p.element_size():
                raise RuntimeError('FlatAdam: a parameter no longer aliases the flat buffer (module moved or cast after the '
                                   'optimiser was built); build the optimiser after .to(device / dtype)')
            off += p.numel()

    @torch.no_grad()
    def step(self, grad_scale=1.0):
        self.sync.attach_()
        self._check_alias()
        g = self.sync.flat
        if self.flat_p.is_cuda:
            from . import _lib
            st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
            _lib.check(_lib.lib.gcrnn_adam_flat(_lib.dtype_code(self.flat_p.dtype), C.c_void_p(self.flat_p.data_ptr()),
                                                C.c_void_p(g.data_ptr()), C.c_void_p(self.m.data_ptr()),
                                                C.c_void_p(self.v.data_ptr()), self.flat_p.numel(), self.lr, self.betas[0],
                                                self.betas[1], self.eps, float(grad_scale), C.c_void_p(self.step_dev.data_ptr()), st),
                       'adam_flat')
            from . import ops
            ops.parameters_changed()      # (written through raw pointers: the parameters' version counters did not move -- cached packs of the old values must not answer)
            return
        # TEST-ONLY branch (CPU tensors): the same update in torch ops, so that the N > 1 logic around the optimiser (flat buffers,
        # sharded batches, the collective) can run over gloo in a container without a GPU (tests/test_parallel_gloo.py). The product
        # path -- the recurrence and everything on its data -- has no CPU implementation (GcrnnError on CPU tensors).
        b1, b2 = self.betas
        self.step_dev += 1
        t = float(self.step_dev.item())
        gs = g * grad_scale if grad_scale != 1.0 else g
        self.m.lerp_(gs, 1 - b1)
        self.v.mul_(b2).addcmul_(gs, gs, value=1 - b2)
        denom = (self.v.sqrt() / (1 - b2 ** t) ** 0.5).add_(self.eps)
        self.flat_p.addcdiv_(self.m, denom, value=-self.lr / (1 - b1 ** t))
        from . import ops
        ops.parameters_changed()

    def state_dict(self):
        return {'flat_p': self.flat_p.clone(), 'm': self.m.clone(), 'v': self.v.clone(), 'step': self.step_dev.clone(),
                'lr': self.lr, 'betas': self.betas, 'eps': self.eps}

    def load_state_dict(self, sd):
        self.flat_p.copy_(sd['flat_p']); self.m.copy_(sd['m']); self.v.copy_(sd['v']); self.step_dev.copy_(sd['step'])
        self.lr, self.betas, self.eps = sd['lr'], tuple(sd['betas']), sd['eps']
        from . import ops
        ops.parameters_changed()
