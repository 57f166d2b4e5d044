#!/usr/bin/env python3
"""Robustness sweep of the GGCRNNCell dispatch on the GPU box: every gating x {inference, training} x batch sizes / channel counts / step counts /
graph weightings at N = 1000 (sparse SBM), each run once on the default dispatch and once with the wide kernel and the inline layouts switched off
(GCRNN_SEQ32=0 GCRNN_NO_INLINE_PACK=1): no exception, finite results, and the two within bf16 noise of each other.
python3 tools/shape_sweep.py [quick]"""
import itertools
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml


def main(quick=False):
    dev = torch.device('cuda:0')
    N, K = 1000, 5
    graphs = {'uniform': bench.sbm_graph(N), 'normalized': bench.sbm_graph(N, normalized=True)}
    Bs = (100, 256) if quick else (3, 64, 100, 128, 160, 256)
    Gs = (1, 64) if quick else (1, 32, 64)
    Fs = (64,) if quick else (32, 64)
    Ts = (4,) if quick else (1, 5)
    gatings = ((False, None), (True, None), (False, 'node'), (True, 'node'), (False, 'edge'))
    SWITCHES = {'GCRNN_SEQ32': '0', 'GCRNN_NO_INLINE_PACK': '1'}
    fails, n = [], 0
    for gname, S in graphs.items():
        St = torch.tensor(S)
        for (tg, sg), F, G, T, B, train in itertools.product(gatings, Fs, Gs, Ts, Bs, (False, True)):
            if sg == 'edge' and (gname == 'normalized' or B > 128):      # (edge gates: the attention path is per edge, keep the sweep short)
                continue
            tag = '%s tg=%s sg=%s F=%d G=%d T=%d B=%d %s' % (gname, tg, sg, F, G, T, B, 'train' if train else 'infer')
            n += 1
            try:
                torch.manual_seed(1)
                cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
                cell.addGSO(St)
                cell = cell.to(torch.bfloat16).to(dev)
                X = torch.randn(B, T, G, N, device=dev).to(torch.bfloat16)
                h0 = (0.3 * torch.randn(B, F, N, device=dev)).to(torch.bfloat16) if (B % 2) else torch.zeros(B, F, N, dtype=torch.bfloat16, device=dev)
                tgt = torch.randn(B, T, F, N, device=dev).to(torch.bfloat16)

                def run():
                    if not train:
                        with torch.no_grad():
                            return cell(X, h0).float(), {}
                    for q in cell.parameters():
                        q.grad = None
                    H = cell(X, h0)
                    (H.float() * tgt.float()).sum().backward()
                    return H.detach().float(), {k: q.grad.detach().float().clone() for k, q in cell.named_parameters() if q.grad is not None}
                H1, g1 = run()
                old = {k: os.environ.get(k) for k in SWITCHES}
                os.environ.update(SWITCHES)
                for k in ('_fused_plan_rank1', '_fused_plan_rank1_adj'):
                    cell.graph.__dict__.pop(k, None)
                try:
                    H0, g0 = run()
                finally:
                    for k, v in old.items():
                        if v is None:
                            os.environ.pop(k, None)
                        else:
                            os.environ[k] = v
                assert torch.isfinite(H1).all() and torch.isfinite(H0).all(), 'non-finite state'
                d = (H1 - H0).abs()
                # (one input feature: input taps of size 1 / sqrt(K) drive the states into a regime where a step amplifies bf16 noise ~2x -- both kernel
                #  families sit 0.02-0.09 from the fp64 result by T = 5, tools/sweep_case.py; the wide kernel rounds w^k W_k where the others keep bf16 parameters exact)
                hmax = 0.2 if G == 1 else 6e-2
                assert float(d.max()) <= hmax and float(d.mean()) <= 3e-3, 'H differs: max %.3g mean %.3g' % (float(d.max()), float(d.mean()))
                assert g1.keys() == g0.keys(), 'gradient sets differ'
                for k in g1:
                    sc = float(g0[k].abs().max())
                    dd = float((g1[k] - g0[k]).abs().max())
                    tol = 0.4 if g1[k].numel() == 1 else 8e-2      # (a scalar's gradient is one sum with cancellation: its own size is no scale for its noise)
                    assert torch.isfinite(g1[k]).all() and dd <= tol * max(sc, 1e-6), 'grad %s differs: %.3g of %.3g' % (k, dd, sc)
            except Exception as e:      # noqa: BLE001
                fails.append((tag, repr(e)[:300]))
                print('FAIL', tag, repr(e)[:300], flush=True)
                if os.environ.get('SWEEP_TRACE'):
                    traceback.print_exc()
            if n % 50 == 0:
                print('%d combinations, %d failures' % (n, len(fails)), flush=True)
    print('shape sweep: %d combinations, %d failures' % (n, len(fails)))
    for t, e in fails:
        print('  ', t, e)
    return n, fails


if __name__ == '__main__':
    sys.exit(1 if main(len(sys.argv) > 1 and sys.argv[1] == 'quick')[1] else 0)
