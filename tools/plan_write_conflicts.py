# Host-side model of the LDS cycles of a fused plan (runs on the CPU):   python tools/plan_write_conflicts.py [adjoint]
# gathers: the plan's own gcrnn_ell_conflict_cycles; state write-backs (ds_write_b128: 8 groups of 8 contiguous lanes, 32 banks):
# a group = 8 consecutive slots of a tile at one quad q, its cycles = the largest number of slots on one bank quad.
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
import bench
from gated_gcrnns_amd.graph import GraphOperator, as_operator

def write_cycles(plan):
    slots = plan['tile_slots'].cpu().numpy().astype(np.int64) & 0xFFFF
    row, swz = slots >> 6, (slots >> 4) & 3
    wkey = ((row & 1) << 2) | swz
    tiles = wkey.reshape(-1, 2, 8)
    m = np.zeros(tiles.shape[:2], dtype=np.int64)
    for k in range(8):
        m = np.maximum(m, (tiles == k).sum(-1))
    return int(4 * m.sum()), int(4 * m.size), np.bincount(m.reshape(-1), minlength=9)

def scatter_cycles(plan, row_dwords, rotate):
    """2-byte scatter of a tile into a transposed [feature][node] bf16 image (user-layout output of the step kernels, du_k image of the
    weight gradient): ds_write_b16, two groups of 32 lanes (16 slots x 2 quads), 32 banks; lane (slot, quad q) writes feature row
    4 q + (j ^ (2 (q & 1)) if rotate else j) in instruction j. Returns (LDS cycles, conflict-free cycles) per image."""
    nodes = (plan['tile_slots'].cpu().numpy().astype(np.int64) >> 16).reshape(-1, 16)
    tot = 0
    for tile in nodes:
        for j in range(4):
            for qp in (0, 2):
                seen = {}
                for q in (qp, qp + 1):
                    row = 4 * q + ((j ^ (2 * (q & 1))) if rotate else j)
                    for n in tile:
                        dw = row * row_dwords + (n >> 1)
                        seen.setdefault(dw % 32, set()).add(dw)
                tot += max(len(v) for v in seen.values())
    return tot, nodes.shape[0] * 8


if __name__ == '__main__':
    adjoint = len(sys.argv) > 1 and sys.argv[1] == 'adjoint'
    S = bench.sbm_graph()
    op = as_operator(torch.from_numpy(S).float())
    plan = op.fused_plan(adjoint=adjoint)
    cyc, ideal, hist = write_cycles(plan)
    ent = plan['entries']
    for rot in (False, True):
        c, ideal = scatter_cycles(plan, 520, rot)
        print('2-byte scatter of the output tile (row stride 2080 B), rotate=%s: %d LDS cycles per item (conflict-free %d)' % (rot, c, ideal))
    print('entries %d  gather cycles %d (conflict-free %d)  write-back LDS cycles per hop %d (conflict-free %d)  half-tile multiplicity histogram %s'
          % (ent, plan['gather_cycles'], ent * 4, cyc, ideal, hist.tolist()))
