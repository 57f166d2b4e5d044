"""Where do two runs of the same graphed trajectory part? parameter checksums after the warm-up, the capture and every replay (GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Modules import train_rnn
from gated_gcrnns_amd.Modules.train_rnn import GraphedTrainStep
from gated_gcrnns_amd.optim import FlatAdam
from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
from shape_sweep import random_graph
dev = torch.device('cuda:0')
N, tg, sg, B, F = 80, True, None, 256, 64
S = random_graph(N, seed=7)
def cks(opt): return '%.9f %.9f %.9f step=%d' % (float(opt.flat_p.double().abs().sum()), float(opt.m.double().abs().sum()), float(opt.sync.flat.double().abs().sum()), int(opt.step_dev.item()))
orig_eager = GraphedTrainStep._eager
def run(tag):
    torch.manual_seed(8)
    G, K, T = 1, 5, 5
    m = archit.GatedGCRNNforRegression(G, F, K, K, torch.tanh, torch.nn.ReLU, [1], S, True, time_gating=tg, spatial_gating=sg, mlpType='multipMlp').float().to(dev)
    xs = [torch.randn(B, T, G, N, device=dev) for _ in range(2)]
    ys = [(0.5 * x).contiguous() for x in xs]
    opt = FlatAdam(m.parameters(), lr=5e-3)
    print(tag, 'init      ', cks(opt))
    n = [0]
    def eager(self):
        r = orig_eager(self)
        torch.cuda.synchronize()
        n[0] += 1
        print(tag, 'warm-up %d ' % n[0], cks(opt), 'loss %.6f' % float(self.loss))
        return r
    GraphedTrainStep._eager = eager
    step = GraphedTrainStep(m, batchTimeL1Loss, opt, xs[0], ys[0], F)
    GraphedTrainStep._eager = orig_eager
    torch.cuda.synchronize()
    print(tag, 'captured  ', cks(opt))
    for i in range(3):
        loss, _ = step(xs[i % 2], ys[i % 2])
        torch.cuda.synchronize()
        print(tag, 'replay %d  ' % i, cks(opt), 'loss %.6f' % float(loss))
for r in range(4):
    run('run%d' % r)
