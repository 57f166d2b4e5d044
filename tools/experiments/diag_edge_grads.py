import sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, 'tests'))
import numpy as np, torch
from test_fused import random_graph, bf16_round
import gated_gcrnns_amd.Utils.graphML as gml
dev = torch.device('cuda:0')
for (N, F, G, K, B, T, tg) in [(600, 64, 1, 3, 3, 3, False), (1000, 64, 64, 2, 2, 2, True), (1000, 64, 64, 5, 4, 4, False)]:
    S = random_graph(N, min(0.5, 10.0 / N), 59)
    rng = np.random.default_rng(16)
    X = bf16_round(rng.standard_normal((B, T, G, N)))
    h0 = bf16_round(0.4 * rng.standard_normal((B, F, N)))
    dH = bf16_round(rng.standard_normal((B, T, F, N)))
    torch.manual_seed(23)
    cell = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'edge', 1, True)
    cell.addGSO(torch.tensor(S))
    cell = cell.to(torch.bfloat16).to(torch.float32)
    refs = []
    for dt in (torch.float32, torch.float64):
        ref = gml.GGCRNNCell(G, F, K, K, torch.tanh, tg, 'edge', 1, True)
        ref.addGSO(torch.tensor(S))
        ref.load_state_dict(cell.state_dict())
        ref = ref.to(dev).to(dt)
        Hr = ref(torch.tensor(X, dtype=dt, device=dev), torch.tensor(h0, dtype=dt, device=dev))
        (Hr * torch.tensor(dH, dtype=dt, device=dev)).sum().backward()
        refs.append(ref)
    cell = cell.to(dev)
    Xd = torch.tensor(X, dtype=torch.bfloat16, device=dev); hd = torch.tensor(h0, dtype=torch.bfloat16, device=dev)
    H = cell(Xd, hd)
    (H.float() * torch.tensor(dH, dtype=torch.float32, device=dev)).sum().backward()
    print('case', N, F, G, K, B, T, tg)
    got = dict(cell.named_parameters()); r32 = dict(refs[0].named_parameters())
    for n, p in refs[1].named_parameters():
        if p.grad is None: continue
        gr = p.grad.double().cpu().numpy(); g = got[n].grad.double().cpu().numpy(); g32 = r32[n].grad.double().cpu().numpy()
        sc = np.abs(gr).max()
        print('  %-28s max|gr| %.3e  fused: max %.4f mean %.4f l2rel %.4f cos %.5f | fp32 composed: max %.2e' % (
            n, sc, np.abs(g - gr).max() / sc, np.abs(g - gr).mean() / sc, np.linalg.norm(g - gr) / np.linalg.norm(gr),
            float((g * gr).sum() / (np.linalg.norm(g) * np.linalg.norm(gr))), np.abs(g32 - gr).max() / sc))
