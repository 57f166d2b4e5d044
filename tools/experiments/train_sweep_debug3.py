"""Is a replay self-consistent, and which of the two outcomes equals the eager forward? (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Modules.train_rnn import GraphedTrainStep
from gated_gcrnns_amd.optim import FlatAdam
from gated_gcrnns_amd.Utils.miscTools import batchTimeL1Loss
from shape_sweep import random_graph
dev = torch.device('cuda:0')
N, tg, sg, B, F = 80, True, None, 256, 64
S = random_graph(N, seed=7)
def run(tag):
    torch.manual_seed(8)
    G, K, T = 1, 5, 5
    m = archit.GatedGCRNNforRegression(G, F, K, K, torch.tanh, torch.nn.ReLU, [1], S, True, time_gating=tg, spatial_gating=sg, mlpType='multipMlp').float().to(dev)
    xs = [torch.randn(B, T, G, N, device=dev) for _ in range(2)]
    ys = [(0.5 * x).contiguous() for x in xs]
    opt = FlatAdam(m.parameters(), lr=5e-3)
    step = GraphedTrainStep(m, batchTimeL1Loss, opt, xs[0], ys[0], F)
    torch.cuda.synchronize()
    snap = (opt.flat_p.clone(), opt.m.clone(), opt.v.clone(), opt.step_dev.clone())
    def restore():
        opt.flat_p.copy_(snap[0]); opt.m.copy_(snap[1]); opt.v.copy_(snap[2]); opt.step_dev.copy_(snap[3])
    h0 = torch.zeros(B, F, N, device=dev)
    with torch.no_grad():
        y_e = m(xs[0], h0)
        le = float(batchTimeL1Loss(y_e, ys[0]))
    # eager with grad (the training dispatch)
    y_t = m(xs[0], h0)
    lt = float(batchTimeL1Loss(y_t, ys[0]))
    out = []
    for i in range(3):
        restore()
        loss, yh = step(xs[0], ys[0])
        torch.cuda.synchronize()
        out.append((float(loss), float((yh - y_t.detach()).abs().max()), float((yh - y_e).abs().max())))
    print(tag, 'eager no_grad %.6f  eager train-dispatch %.6f  replays' % (le, lt), ['%.6f dy_train %.3g dy_infer %.3g' % o for o in out], flush=True)
for r in range(4):
    run('run%d' % r)
