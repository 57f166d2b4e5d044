"""Loss / metric of the GCRNN training loop (counterpart of the reference's Utils/miscTools.py:112-130)."""
import torch


def batchTimeL1Loss(x, y):
    """Mean absolute error over every entry (reference miscTools.py:112-119; its view(-1,N,F) is a no-op for a mean)."""
    from .. import ops
    y = y.to(x.dtype)
    if x.shape != y.shape:
        x, y = torch.broadcast_tensors(x, y)
    return ops.l1_loss(x, y)


def batchTimeMSELoss(x, y):
    """Per flattened (N*F) column: sqrt(sum_rows (x-y)^2) / ||y column||_2, averaged over columns
    (reference miscTools.py:121-130). A metric (the drivers never back-propagate it): on the device without autograd it is
    two launches (gcrnn_batch_time_mse: column partial sums over row slabs, fixed-order finish); anything that needs a
    gradient takes the torch expression."""
    F, N = x.shape[-2], x.shape[-1]
    xv = x.reshape(-1, N * F)
    yv = y.to(x.dtype).reshape(-1, N * F)
    if x.is_cuda and not (torch.is_grad_enabled() and (x.requires_grad or y.requires_grad)) and \
            x.dtype in (torch.float32, torch.float64, torch.bfloat16):
        from .. import ops
        return ops.batch_time_mse(xv.contiguous(), yv.contiguous())
    num = torch.sqrt(torch.sum((xv - yv) ** 2, dim=0))
    return torch.mean(num / torch.norm(yv, dim=0))
