#!/usr/bin/env python3
"""One gating variant's training step at the reference driver's default sizes (N=80 SBM, taps 5, T=5, F=20, batch 100, fp64),
for rocprofv3:  python3 tools/variant_train_probe.py {none|time|node|edge|cfg4|cfg4time} [steps]
(cfg4 = BASELINE configs[3]: seismic graph N=59, K=3, T=200, classification head)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Modules.train_rnn import train_step
from gated_gcrnns_amd.Utils import dataTools, miscTools

variant = sys.argv[1] if len(sys.argv) > 1 else 'edge'
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device('cuda:0')
torch.set_default_dtype(torch.float64)
if variant.startswith('cfg4'):
    adj = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'adj59.npy'))
    S4 = dataTools.normalised_gso(adj)
    x = torch.randn(100, 200, 1, 59, device=dev)
    y = torch.randint(0, 11, (100,), device=dev)
    torch.manual_seed(0)
    m = archit.GatedGCRNNforClassification(1, 20, 3, 3, torch.tanh, torch.nn.ReLU, [11], S4, True,
                                           time_gating=(variant == 'cfg4time')).to(dev)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    ce = torch.nn.CrossEntropyLoss()
    for _ in range(3):
        train_step(m, ce, opt, x, y, 20)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps):
        l = train_step(m, ce, opt, x, y, 20)[0]
    torch.cuda.synchronize()
    print('%s: %.2f ms/step, loss %.5f' % (variant, 1e3 * (time.perf_counter() - t0) / steps, float(l)))
    sys.exit(0)
tg, sg = {'none': (False, None), 'time': (True, None), 'node': (False, 'node'), 'edge': (False, 'edge')}[variant]
rng = np.random.default_rng(0)
W = dataTools.sbm_adjacency(80, 5, 0.8, 0.2, rng)
S = dataTools.normalised_gso(W)
data = dataTools.KStepPrediction(W, 5, 400, 10, 10, horizon=10, rng=rng)
xT, yT = data.getSamples('train')
x = xT[:100].view(100, 5, 1, 80).to(dev); y = yT[:100].view(100, 5, 1, 80).to(dev)
torch.manual_seed(0)
m = archit.GatedGCRNNforRegression(1, 20, 5, 5, torch.tanh, torch.nn.ReLU, [1], S, True, time_gating=tg, spatial_gating=sg,
                                   mlpType='multipMlp').to(dev)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for _ in range(3):
    train_step(m, miscTools.batchTimeL1Loss, opt, x, y, 20)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(steps):
    l = train_step(m, miscTools.batchTimeL1Loss, opt, x, y, 20)[0]
torch.cuda.synchronize()
print('%s: %.2f ms/step, loss %.5f' % (variant, 1e3 * (time.perf_counter() - t0) / steps, float(l)))
