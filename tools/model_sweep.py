#!/usr/bin/env python3
"""Robustness sweep at the model level (Modules/architectures.py): GatedGCRNNforRegression (oneMlp / multipMlp heads, one or two MLP layers) and
GatedGCRNNforClassification x gating x dtype (bf16 / f32 / f64) x {inference, training} x graph size (the drivers' N = 80, N = 1000) x state width
(the drivers' 20, 64) x batch, each against the SAME model in fp64 (deep copy): no exception, finite, outputs within the dtype's noise of fp64's, the gradient vector within 10 % (bf16) / 0.3 % (f32) in the L2 norm.
python3 tools/model_sweep.py [quick]"""
import copy
import itertools
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gated_gcrnns_amd.Modules.architectures as archit
from shape_sweep import random_graph


def main(quick=False):
    dev = torch.device('cuda:0')
    fails, n = [], 0
    gatings = ((False, None), (True, None), (False, 'node'), (True, 'node'), (False, 'edge'))
    heads = (('reg', 'multipMlp', [1]), ('reg', 'oneMlp', [40]), ('reg', 'multipMlp', [8, 1]), ('cls', None, [5]))
    for N in ((1000,) if quick else (80, 1000)):
        S = random_graph(N, seed=7)
        for (tg, sg), (kind, mlp, dims), F, B, dt, train in itertools.product(gatings, heads, (20,) if quick else (20, 64), (100,) if quick else (20, 100),
                                                                             (torch.bfloat16, torch.float32), (False, True)):
            if sg == 'edge' and (B > 20 and N > 80):
                continue
            G, K, T = 1, 5, 5
            tag = 'N=%d tg=%s sg=%s %s/%s%s F=%d B=%d %s %s' % (N, tg, sg, kind, mlp, dims, F, B, str(dt).split('.')[1], 'train' if train else 'infer')
            n += 1
            try:
                torch.manual_seed(4)
                if kind == 'reg':
                    m = archit.GatedGCRNNforRegression(G, F, K, K, torch.tanh, torch.nn.ReLU, dims, S, True, time_gating=tg, spatial_gating=sg, mlpType=mlp)
                else:
                    m = archit.GatedGCRNNforClassification(G, F, K, K, torch.tanh, torch.nn.ReLU, dims, S, True, time_gating=tg, spatial_gating=sg)
                with torch.no_grad():
                    m.stateGCRNN.weight_A.mul_(0.25)      # (one input feature: out of the noise-doubling regime, profiles/r04_shape_sweep.txt)
                m = m.to(dt).to(dev)
                ref = copy.deepcopy(m).double()
                X = torch.randn(B, T, G, N, device=dev).to(dt)
                h0 = torch.zeros(B, F, N, device=dev, dtype=dt) if B % 40 else (0.3 * torch.randn(B, F, N, device=dev)).to(dt)

                def run(c, X, h0):
                    if not train:
                        with torch.no_grad():
                            return c(X, h0).double(), {}
                    for q in c.parameters():
                        q.grad = None
                    y = c(X, h0)
                    w = torch.linspace(-1, 1, y.numel(), device=y.device, dtype=torch.float64).view(y.shape)
                    (y.double() * w).sum().backward()
                    return y.detach().double(), {k: q.grad.detach().double().clone() for k, q in c.named_parameters() if q.grad is not None}
                y1, g1 = run(m, X, h0)
                yr, gr = run(ref, X.double(), h0.double())
                assert y1.shape == yr.shape, 'shapes %s vs %s' % (tuple(y1.shape), tuple(yr.shape))
                assert torch.isfinite(y1).all(), 'non-finite output'
                ysc = max(float(yr.abs().max()), 1e-3)
                d = float((y1 - yr).abs().max())
                assert d <= (5e-2 if dt == torch.bfloat16 else 1e-4) * ysc, 'output differs: %.3g of %.3g' % (d, ysc)
                assert g1.keys() == gr.keys(), 'gradient sets differ: %s' % sorted(set(g1) ^ set(gr))
                # gradients: the whole gradient vector against fp64's in the L2 norm (single parameters whose gradient is a sum with heavy cancellation --
                # scalars, the second layer behind a ReLU under the signed weighting -- carry noise many times their own size in bf16)
                num = sum(float(((g1[k] - gr[k]) ** 2).sum()) for k in gr) ** 0.5
                den = sum(float((gr[k] ** 2).sum()) for k in gr) ** 0.5
                assert all(torch.isfinite(v).all() for v in g1.values()), 'non-finite gradient'
                kink = 3.0 if len(dims) > 1 else 1.0      # (a ReLU between two head layers: a pre-activation that changes sign under rounding moves the gradient by a finite amount)
                assert num <= kink * (0.1 if dt == torch.bfloat16 else 3e-3) * max(den, 1e-12), 'gradient differs: %.3g of %.3g (L2)' % (num, den)
            except Exception as e:      # noqa: BLE001
                fails.append((tag, repr(e)[:300]))
                print('FAIL', tag, repr(e)[:300], flush=True)
                if os.environ.get('SWEEP_TRACE'):
                    traceback.print_exc()
            if n % 50 == 0:
                print('%d combinations, %d failures' % (n, len(fails)), flush=True)
    print('model sweep: %d combinations, %d failures' % (n, len(fails)))
    for t, e in fails:
        print('  ', t, e)
    return n, fails


if __name__ == '__main__':
    sys.exit(1 if main('quick' in sys.argv[1:])[1] else 0)
