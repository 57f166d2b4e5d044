#!/usr/bin/env python3
"""API edge cases of GGCRNNCell on the GPU box, each against the same cell in fp64: non-contiguous X / h0 views, inputs that want gradients
(dX, dh0), B = 1, T = 1, torch.inference_mode, parameters updated in place between calls, last_only.   python3 tools/api_edge_cases.py"""
import copy
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gated_gcrnns_amd.Utils.graphML as gml
from shape_sweep import random_graph


def main():
    dev = torch.device('cuda:0')
    N, K = 1000, 5
    S = random_graph(N, seed=9)
    fails, n = [], 0
    for (tg, sg) in ((False, None), (True, None), (False, 'node'), (False, 'edge')):
        for dt in (torch.bfloat16, torch.float32):
            for case in ('noncontig', 'wants_dx', 'b1t1', 'inference_mode', 'inplace_update', 'expanded_h0', 'last_only'):
                F, G = 64, 64
                B, T = (1, 1) if case == 'b1t1' else ((8, 3) if sg == 'edge' else (100, 3))
                tag = 'tg=%s sg=%s %s %s' % (tg, sg, str(dt).split('.')[1], case)
                n += 1
                try:
                    torch.manual_seed(6)
                    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
                    cell.addGSO(torch.tensor(S))
                    cell = cell.to(dt).to(dev)
                    ref = copy.deepcopy(cell).double()
                    X = torch.randn(B, T, G, N, device=dev).to(dt)
                    h0 = (0.3 * torch.randn(B, F, N, device=dev)).to(dt)
                    tol = 6e-2 if dt == torch.bfloat16 else 2e-5
                    if case == 'noncontig':
                        X = X.transpose(2, 3).contiguous().transpose(2, 3)          # B x T x G x N view with N-major strides
                        h0 = h0.transpose(1, 2).contiguous().transpose(1, 2)
                        assert not X.is_contiguous() and not h0.is_contiguous()
                    if case == 'expanded_h0':
                        h0 = h0[:1].expand(B, F, N)                                  # stride 0 over the batch
                    if case == 'wants_dx':
                        X1, h1 = X.clone().requires_grad_(True), h0.clone().requires_grad_(True)
                        Xr, hr = X.double().requires_grad_(True), h0.double().requires_grad_(True)
                        tgt = torch.randn(B, T, F, N, device=dev)
                        H = cell(X1, h1)
                        (H.float() * tgt).sum().backward()
                        Hr = ref(Xr, hr)
                        (Hr * tgt.double()).sum().backward()
                        for a, b, nm in ((X1.grad, Xr.grad, 'dX'), (h1.grad, hr.grad, 'dh0')):
                            assert a is not None and a.shape == b.shape, nm + ' missing'
                            sc = float(b.abs().max())
                            d = float((a.double() - b).abs().max())
                            assert d <= (8e-2 if dt == torch.bfloat16 else 1e-4) * sc, '%s differs: %.3g of %.3g' % (nm, d, sc)
                    elif case == 'last_only':      # the classification model's call: B x 1 x F x N, the last state of the full forward
                        with torch.no_grad():
                            H = c_last = cell(X, h0, last_only=True)
                            Hr = ref(X.double(), h0.double())[:, -1:]
                            assert tuple(c_last.shape) == (B, 1, F, N)
                            assert float((c_last.double() - cell(X, h0)[:, -1:].double()).abs().max()) <= (4e-3 if dt == torch.bfloat16 else 1e-6), 'last_only differs from the full forward'
                    elif case == 'inference_mode':
                        with torch.inference_mode():
                            H = cell(X, h0)
                        with torch.no_grad():
                            Hr = ref(X.double(), h0.double())
                    elif case == 'inplace_update':
                        with torch.no_grad():
                            cell(X, h0)                                               # (packs cached)
                            for q, qr in zip(cell.parameters(), ref.parameters()):
                                q.mul_(0.5)
                                qr.copy_(q.double())
                            H = cell(X, h0)
                            Hr = ref(X.double(), h0.double())
                    else:
                        with torch.no_grad():
                            H = cell(X, h0)
                            Hr = ref(X.double(), h0.double())
                    d = float((H.detach().double() - Hr.detach()).abs().max())
                    assert H.shape == Hr.shape and torch.isfinite(H.float()).all() and d <= tol, 'H differs: %.3g' % d
                except Exception as e:      # noqa: BLE001
                    fails.append((tag, repr(e)[:300]))
                    print('FAIL', tag, repr(e)[:300], flush=True)
                    if os.environ.get('SWEEP_TRACE'):
                        traceback.print_exc()
    print('api edge cases: %d cases, %d failures' % (n, len(fails)))
    return n, fails


if __name__ == '__main__':
    sys.exit(1 if main()[1] else 0)
