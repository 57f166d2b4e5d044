#!/usr/bin/env python3
"""cfg4-sized small-graph forward (+ optional backward) only, for rocprofv3: python3 tools/small_probe.py [f64|f32] [train]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gated_gcrnns_amd.Utils.graphML as gml
from gated_gcrnns_amd.Utils import dataTools
dt = torch.float32 if len(sys.argv) > 1 and sys.argv[1] == 'f32' else torch.float64
train = len(sys.argv) > 2 and sys.argv[2] == 'train'
dev = torch.device('cuda:0')
adj = np.load(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'adj59.npy'))
S = dataTools.normalised_gso(adj)
torch.manual_seed(0)
cell = gml.GGCRNNCell(1, 20, 3, 3, torch.tanh, False, None, 1, True)
cell.addGSO(torch.tensor(S[None]))
cell = cell.to(dev).to(dt)
x = torch.randn(100, 200, 1, 59, device=dev, dtype=dt)
h0 = torch.zeros(100, 20, 59, device=dev, dtype=dt)
for _ in range(4):
    if train:
        cell.zero_grad(); cell(x, h0).sum().backward()
    else:
        with torch.no_grad(): cell(x, h0)
torch.cuda.synchronize()
