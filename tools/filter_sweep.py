#!/usr/bin/env python3
"""Stand-alone layers of Utils/graphML.py on the GPU box against the same layer in fp64: GraphFilter / LSIGF (E = 1, 2 edge features; with and
without bias; inputs shorter than N: the reference's zero padding, graphML.py:1181-1193) and GraphAttentional, N in {80, 1000, 2048}, bf16 / f32,
outputs and gradients.   python3 tools/filter_sweep.py [N ...]"""
import copy
import itertools
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import gated_gcrnns_amd.Utils.graphML as gml
from shape_sweep import random_graph


def main(Ns=(80, 1000, 2048)):
    dev = torch.device('cuda:0')
    fails, n = [], 0
    for N in Ns:
        S1 = random_graph(N, seed=5)
        S2 = np.concatenate([S1, random_graph(N, seed=6)], axis=0)
        for kind, E, G, F, K, bias, dt, B, short in itertools.product(('filter', 'attention'), (1, 2), (1, 32), (1, 64), (1, 4), (True, False),
                                                                     (torch.bfloat16, torch.float32), (7, 64), (False, True)):
            if kind == 'attention' and (not bias or short or E == 2 and N > 1000):
                continue
            tag = 'N=%d %s E=%d G=%d F=%d K=%d bias=%s %s B=%d short=%s' % (N, kind, E, G, F, K, bias, str(dt).split('.')[1], B, short)
            n += 1
            try:
                torch.manual_seed(2)
                layer = gml.GraphFilter(G, F, K, E, bias) if kind == 'filter' else gml.GraphAttentional(G, F, K, E)
                layer.addGSO(torch.tensor(S1 if E == 1 else S2))
                layer = layer.to(dt).to(dev)
                ref = copy.deepcopy(layer).double()
                Nin = N - 3 if short else N
                x = torch.randn(B, G, Nin, device=dev).to(dt).requires_grad_(True)
                xr = x.detach().double().requires_grad_(True)
                y = layer(x)
                yr = ref(xr)
                assert y.shape == yr.shape, 'shape %s vs %s' % (tuple(y.shape), tuple(yr.shape))
                w = torch.randn(yr.shape, device=dev, dtype=torch.float64)
                (y.double() * w).sum().backward()
                (yr * w).sum().backward()
                tol = 3e-2 if dt == torch.bfloat16 else 1e-4
                sc = max(float(yr.abs().max()), 1e-3)
                d = float((y.double() - yr).abs().max())
                assert d <= tol * sc, 'output differs: %.3g of %.3g' % (d, sc)
                pairs = [('dx', x.grad, xr.grad)] + [(k, p.grad, q.grad) for (k, p), (_, q) in zip(layer.named_parameters(), ref.named_parameters())]
                for k, a, b in pairs:
                    assert (a is None) == (b is None), k + ' gradient presence'
                    if a is None:
                        continue
                    sc = max(float(b.abs().max()), 1e-3)
                    d = float((a.double() - b).abs().max())
                    # (attention: a logit within rounding of 0 sits on the other side of the LeakyReLU kink in fp64 -- single entries of dx move by ~1 %, the parameter gradients (sums over B N deg terms in fp32) by up to 4e-3)
                    lim = 2e-2 if (kind == 'attention' and dt == torch.float32) else (3 * tol if a.numel() > 1 else 10 * tol)
                    assert d <= lim * sc, 'grad %s differs: %.3g of %.3g' % (k, d, sc)
            except Exception as e:      # noqa: BLE001
                fails.append((tag, repr(e)[:300]))
                print('FAIL', tag, repr(e)[:300], flush=True)
                if os.environ.get('SWEEP_TRACE'):
                    traceback.print_exc()
    print('filter sweep: %d combinations, %d failures' % (n, len(fails)))
    return n, fails


if __name__ == '__main__':
    sys.exit(1 if main(tuple(int(a) for a in sys.argv[1:]) or (80, 1000, 2048))[1] else 0)
