#!/usr/bin/env python3
"""K-step prediction with gated GCRNNs on MI355X -- counterpart of the reference driver kStepPredGRNNs.py
(graph -> S = W/lambda_max -> data -> models -> train -> test; its lines 598-1677), reduced to the GCRNN models.

    python examples/kstep_prediction.py [--nodes 80] [--taps 5] [--seq 5] [--epochs 1] [--dtype f64]
    python examples/kstep_prediction.py --nodes 1000 --features 64 --seq 32 --dtype bf16 --ntrain 1024 --batch 256 --sparse

Defaults follow the reference driver (N=80 SBM 0.8/0.2, 5 taps, K=seqLen=5, F=20, batch 100, Adam 1e-3). --dtype bf16 feeds
bf16 batches to fp32 master weights: all four variants (un-gated, time-, node- and edge-gated) then train on the fused kernels
(N <= 1024, F in {32, 64}; other shapes: composed path in fp32). --sparse draws the BASELINE configs[1] graph (mean degree ~10).
"""
import argparse
import os
import sys
import tempfile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

import gated_gcrnns_amd.Modules.architectures as archit
from gated_gcrnns_amd.Modules.train_rnn import MultipleModels, TrainableModel
from gated_gcrnns_amd.Utils import dataTools, miscTools
from gated_gcrnns_amd.optim import FlatAdam


def main(argv=None):
    """Returns {model name: dict(loss=[per-step training loss], score=test metric, ms=median ms per batch)}."""
    ap = argparse.ArgumentParser()
    ap.add_argument('--nodes', type=int, default=80)
    ap.add_argument('--taps', type=int, default=5)
    ap.add_argument('--seq', type=int, default=5)
    ap.add_argument('--features', type=int, default=20)
    ap.add_argument('--epochs', type=int, default=1)
    ap.add_argument('--batch', type=int, default=100)
    ap.add_argument('--ntrain', type=int, default=2000)
    ap.add_argument('--dtype', default='f64', choices=['f32', 'f64', 'bf16'])
    ap.add_argument('--sparse', action='store_true', help='SBM with p_in 0.04 / p_out 0.0025 (BASELINE configs[1]) instead of 0.8 / 0.2')
    ap.add_argument('--models', default='GCRNNMLP,TimeGCRNNMLP,NodeGCRNNMLP,EdgeGCRNNMLP')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--optim', default='flat', choices=['flat', 'torch'], help='flat: optim.FlatAdam (one kernel over the flat '
                    'parameter / gradient buffers); torch: torch.optim.Adam as in the reference driver')
    args = ap.parse_args(argv)
    dt = torch.float64 if args.dtype == 'f64' else torch.float32       # parameter dtype (bf16: fp32 master weights)
    data_dt = torch.bfloat16 if args.dtype == 'bf16' else dt
    torch.set_default_dtype(dt)                                   # the reference driver runs in float64 (line 44)
    dev = torch.device('cuda:0')
    rng = np.random.default_rng(args.seed)
    torch.manual_seed(args.seed)
    W = dataTools.sbm_adjacency(args.nodes, 5, 0.04, 0.0025, rng) if args.sparse else dataTools.sbm_adjacency(args.nodes, 5, 0.8, 0.2, rng)
    S = dataTools.normalised_gso(W)
    K = args.seq
    data = dataTools.KStepPrediction(W, K, args.ntrain, 200, 200, horizon=2 * K, rng=rng, dataType=dt)
    saveDir = tempfile.mkdtemp(prefix='kstep_')
    models = {}
    for name, tg, sg in (('GCRNNMLP', False, None), ('TimeGCRNNMLP', True, None), ('NodeGCRNNMLP', False, 'node'),
                         ('EdgeGCRNNMLP', False, 'edge')):
        if name not in args.models.split(','):
            continue
        m = archit.GatedGCRNNforRegression(1, args.features, args.taps, args.taps, torch.tanh, torch.nn.ReLU, [1], S, True,
                                           time_gating=tg, spatial_gating=sg, mlpType='multipMlp').to(dev)
        if args.optim == 'flat':
            opt = FlatAdam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))          # kStepPredGRNNs.py:158-161
        else:
            opt = torch.optim.Adam(m.parameters(), lr=1e-3, betas=(0.9, 0.999))
        models[name] = TrainableModel(m, miscTools.batchTimeL1Loss, opt, name, saveDir)
    xT, yT = data.getSamples('train')
    xV, yV = data.getSamples('valid')
    out = MultipleModels(models, xT, yT, xV, yV, args.epochs, args.batch, data.seqLen, args.features,
                         data.evaluate, validationInterval=5, rng=rng, doPrint=False, dataType=data_dt)
    xE, yE = data.getSamples('test')
    xE = xE.view(xE.shape[0], data.seqLen, -1).to(dev, data_dt).unsqueeze(2)
    yE = yE.view(yE.shape[0], data.seqLen, -1).to(dev, data_dt).unsqueeze(2)
    result = {}
    for name, tm in models.items():
        tm.load('Best')
        with torch.no_grad():
            h0 = torch.zeros(xE.shape[0], args.features, args.nodes, device=dev, dtype=data_dt)
            score = float(data.evaluate(tm.archit(xE, h0).to(yE.dtype), yE))
        t = np.median(out['timeTrain'][name])
        print('%-14s test RMSE-metric %.4f   loss %.4f -> %.4f   median %.1f ms/batch (%.0f seq/s)' % (
            name, score, out['lossTrain'][name][0], out['lossTrain'][name][-1], 1e3 * t, args.batch / t))
        result[name] = dict(loss=list(out['lossTrain'][name]), score=score, ms=1e3 * t)
    return result


if __name__ == '__main__':
    main()
