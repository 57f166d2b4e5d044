#!/usr/bin/env python3
"""Batches beyond the bench's (B = 512 ... 2048 at N = 1000, T = 32, F = G = 64, bf16: tensors of 4-8 GB, past the 2^31-byte offsets some kernels index
with): the cell's result on the whole batch against its results on chunks of 256 sequences (sequences are independent), inference and training,
un-gated / time-gated / node-gated.   python3 tools/large_batch_check.py [B ...]"""
import os
import sys
import traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench
import gated_gcrnns_amd.Utils.graphML as gml


def main(Bs=(512, 1024, 2048)):
    dev = torch.device('cuda:0')
    N, K, F, G, T = 1000, 5, 64, 64, int(os.environ.get('T', 32))
    CH = int(os.environ.get('CHUNK', 256))      # (env T / CHUNK: long sequences, e.g. T=200 CHUNK=64 with B = 128)
    S = torch.tensor(bench.sbm_graph(N))
    fails, n = [], 0
    for (tg, sg) in (((False, 'edge'),) if os.environ.get('EDGE') else ((False, None), (True, None), (False, 'node'))):      # (env EDGE=1: the edge-gated cell alone)
        torch.manual_seed(3)
        cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, sg, 1, True)
        cell.addGSO(S)
        cell = cell.to(torch.bfloat16).to(dev)
        for B in Bs:
            for train in (False, True):
                tag = 'tg=%s sg=%s B=%d %s' % (tg, sg, B, 'train' if train else 'infer')
                n += 1
                try:
                    g = torch.Generator(device=dev).manual_seed(B)
                    X = torch.randn(B, T, G, N, device=dev, generator=g).to(torch.bfloat16)
                    h0 = torch.zeros(B, F, N, device=dev, dtype=torch.bfloat16)
                    tgt = torch.randn(B, T, F, N, device=dev, generator=g).to(torch.bfloat16)

                    def run(X, h0, tgt):
                        if not train:
                            with torch.no_grad():
                                return cell(X, h0), {}
                        for q in cell.parameters():
                            q.grad = None
                        H = cell(X, h0)
                        H.backward(tgt)                      # the loss sum(H * tgt) without a float copy of H
                        return H.detach(), {k: q.grad.detach().float().clone() for k, q in cell.named_parameters() if q.grad is not None}
                    Hb, gb = run(X, h0, tgt)
                    gc = {}
                    dmax = 0.0
                    for i in range(0, B, CH):
                        Hc, g1 = run(X[i:i + CH].contiguous(), h0[i:i + CH].contiguous(), tgt[i:i + CH].contiguous())
                        dmax = max(dmax, float((Hc.float() - Hb[i:i + CH].float()).abs().max()))
                        for k, v in g1.items():
                            gc[k] = gc.get(k, 0) + v
                    assert torch.isfinite(Hb.float()).all(), 'non-finite state'
                    assert dmax <= 2.5e-2, 'H differs from the chunked run: %.3g' % dmax
                    for k in gc:
                        sc = float(gc[k].abs().max())
                        dd = float((gb[k] - gc[k]).abs().max())
                        assert dd <= (0.3 if gc[k].numel() == 1 else 6e-2) * max(sc, 1e-6), 'grad %s differs: %.3g of %.3g' % (k, dd, sc)
                    print('ok  ', tag, 'max |H - chunked| = %.3g' % dmax, flush=True)
                except Exception as e:      # noqa: BLE001
                    fails.append((tag, repr(e)[:300]))
                    print('FAIL', tag, repr(e)[:300], flush=True)
                    if os.environ.get('SWEEP_TRACE'):
                        traceback.print_exc()
                torch.cuda.empty_cache()
    print('large batch check: %d cases, %d failures' % (n, len(fails)))
    return n, fails


if __name__ == '__main__':
    sys.exit(1 if main(tuple(int(a) for a in sys.argv[1:]) or (512, 1024, 2048))[1] else 0)
