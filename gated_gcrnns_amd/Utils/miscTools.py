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
    (reference miscTools.py:121-130)."""
    F, N = x.shape[-2], x.shape[-1]
    xv = x.reshape(-1, N * F)
    yv = y.to(x.dtype).reshape(-1, N * F)
    num = torch.sqrt(torch.sum((xv - yv) ** 2, dim=0))
    return torch.mean(num / torch.norm(yv, dim=0))
