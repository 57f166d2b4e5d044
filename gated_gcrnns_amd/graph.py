"""Graph shift operator (GSO) preparation: dense E x N x N  ->  device CSR.

The reference keeps S dense and shifts with x @ S (Utils/graphML.py:116-123).
Here each edge-feature slice S_e becomes
  fwd  = CSR(S_e^T)   the shift on node-major data  ((x S)[:, n] = sum_m S[m, n] x[:, m])
  adj  = CSR(S_e)     its adjoint (backward pass)
  mask = CSR pattern and values of S_e + I restricted to |.| > 1e-9: the attention
         support of graphAttention (Utils/graphML.py:577, 611-613), row m -> columns n.
Index arrays are int32 and are built by the C ABI's host routines
(gcrnn_csr_count / gcrnn_csr_fill), so they are exact by construction.
"""
import ctypes as C

import numpy as np
import torch

from . import _lib

ZERO_TOLERANCE = 1e-9


class CSR(object):
    """rowptr/col int32 + values (float64 master copy, cast per dtype on demand), all on one device."""

    def __init__(self, rowptr, col, val, device):
        self.N = int(rowptr.size - 1)
        self.nnz = int(col.size)
        self.rowptr = torch.from_numpy(rowptr).to(device)
        self.col = torch.from_numpy(col).to(device)
        self._val64 = torch.from_numpy(val).to(device)
        self._vals = {torch.float64: self._val64}
        self._rows = None

    def val(self, dtype):
        if dtype == torch.bfloat16:
            dtype = torch.float32
        v = self._vals.get(dtype)
        if v is None:
            v = self._val64.to(dtype)
            self._vals[dtype] = v
        return v

    def rows(self):
        """Expanded row index per non-zero (int64), for index-based composed ops."""
        if self._rows is None:
            counts = (self.rowptr[1:] - self.rowptr[:-1]).long()
            self._rows = torch.repeat_interleave(torch.arange(self.N, device=self.rowptr.device), counts)
        return self._rows

    def to(self, device):
        out = CSR.__new__(CSR)
        out.N, out.nnz = self.N, self.nnz
        out.rowptr, out.col = self.rowptr.to(device), self.col.to(device)
        out._val64 = self._val64.to(device)
        out._vals = {torch.float64: out._val64}
        out._rows = None
        return out


def csr_from_dense(S, transpose=False, add_identity=False, tol=0.0):
    """Host CSR arrays (rowptr int32[N+1], col int32[nnz], val float64[nnz]) via the C ABI."""
    S = np.ascontiguousarray(np.asarray(S, dtype=np.float64))
    assert S.ndim == 2 and S.shape[0] == S.shape[1]
    N = S.shape[0]
    nnz = C.c_int64(0)
    sp = S.ctypes.data_as(C.c_void_p)
    _lib.check(_lib.lib.gcrnn_csr_count(sp, N, int(transpose), int(add_identity), float(tol), C.byref(nnz)), 'csr_count')
    rowptr = np.empty(N + 1, dtype=np.int32)
    col = np.empty(nnz.value, dtype=np.int32)
    val = np.empty(nnz.value, dtype=np.float64)
    _lib.check(_lib.lib.gcrnn_csr_fill(sp, N, int(transpose), int(add_identity), float(tol),
                                       rowptr.ctypes.data_as(C.c_void_p), col.ctypes.data_as(C.c_void_p),
                                       val.ctypes.data_as(C.c_void_p)), 'csr_fill')
    return rowptr, col, val


def degree_order(rowptr):
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int32)
    order = np.empty(rowptr.size - 1, dtype=np.int32)
    _lib.check(_lib.lib.gcrnn_degree_order(rowptr.ctypes.data_as(C.c_void_p), rowptr.size - 1,
                                           order.ctypes.data_as(C.c_void_p)), 'degree_order')
    return order


def spread_tile_classes(slots, deg, nreal, pad=4):
    """slots [ntiles][16]: node ids in descending-degree order (padding rows last). The kernels scatter a tile's results 2 bytes at a time
    into transposed [feature][node] LDS images (user-layout output of the step kernels, du_k image of the weight gradient); with the
    rows rotated per quad, such a scatter is bank-conflict-free exactly when the 16 nodes of a tile differ in (node >> 1) & 15.
    Exchange nodes between tiles until (nearly) every tile has that property, never raising a tile's depth (a node only moves to a tile
    at least as deep as its degree), so the ELL keeps its size. Padding rows stay where they are. Deterministic."""
    slots = slots.copy()
    ntiles = slots.shape[0]
    dg = np.concatenate([deg, np.zeros(ntiles * 16 - deg.size, dtype=deg.dtype)]).astype(np.int64)
    depth = ((dg[slots].max(axis=1) + pad - 1) // pad) * pad
    tile_of = np.repeat(np.arange(ntiles), 16).reshape(ntiles, 16)
    for _ in range(6):
        cls = (slots >> 1) & 15
        count = np.zeros((ntiles, 16), dtype=np.int64)
        np.add.at(count, (tile_of.reshape(-1), cls.reshape(-1)), 1)
        moved = False
        for t in range(ntiles):
            for r in range(16):
                a = int(slots[t, r])
                c = (a >> 1) & 15
                if count[t, c] < 2 or a >= nreal:
                    continue
                missing = np.nonzero(count[t] == 0)[0]
                if missing.size == 0:
                    continue
                fcls = (slots >> 1) & 15
                ok = np.isin(fcls, missing) & (dg[slots] <= depth[t]) & (depth[:, None] >= dg[a]) & (slots < nreal) & (tile_of != t)
                if not ok.any():
                    continue
                # what the exchange does to the other tile: it loses a node of class m (good if m was doubled there), gains one of class c
                du = -(count[tile_of, fcls] > 1).astype(np.int64) + (count[:, c][:, None] >= 1).astype(np.int64)
                du = np.where(ok, du, 9)
                u, ru = np.unravel_index(int(np.argmin(du)), du.shape)
                if du[u, ru] > 0:
                    continue                                  # would only move the conflict
                b = int(slots[u, ru])
                m = (b >> 1) & 15
                slots[t, r], slots[u, ru] = b, a
                count[t, c] -= 1; count[t, m] += 1
                count[u, m] -= 1; count[u, c] += 1
                moved = True
        if not moved:
            break
    return slots


class GraphOperator(object):
    """Device-resident sparse form of a GSO  S: E x N x N  (numpy array or torch tensor)."""

    def __init__(self, S, device=None):
        if isinstance(S, torch.Tensor):
            if device is None:
                device = S.device
            S = S.detach().cpu().numpy()
        S = np.asarray(S, dtype=np.float64)
        assert S.ndim == 3, 'GSO must be E x N x N'
        assert S.shape[1] == S.shape[2]
        self.E, self.N = int(S.shape[0]), int(S.shape[1])
        self.device = torch.device(device if device is not None else 'cpu')
        self.fwd = [CSR(*csr_from_dense(S[e], transpose=True), device=self.device) for e in range(self.E)]
        self.adj = [CSR(*csr_from_dense(S[e], transpose=False), device=self.device) for e in range(self.E)]
        # attention support: mask = sum_e |S_e + I| > 1e-9 (graphML.py:577, 611-613); values S_e + I on it
        Splus = S + np.eye(self.N).reshape(1, self.N, self.N)
        mask = (np.abs(Splus).sum(axis=0) > ZERO_TOLERANCE).astype(np.float64)
        rp, col, _ = csr_from_dense(mask)
        self.mask = CSR(rp, col, np.ones(col.size), device=self.device)
        rows = np.repeat(np.arange(self.N), np.diff(rp))
        self.mask_vals = [torch.from_numpy(np.ascontiguousarray(Splus[e][rows, col])).to(self.device) for e in range(self.E)]
        self.nnz = sum(c.nnz for c in self.fwd)

    def dense(self, dtype, e=0):
        """Dense S_e ([N][N], row m, column n) on the device, for the small-graph matrix-core kernels."""
        cache = self.__dict__.setdefault('_dense', {})
        key = (dtype, e)
        if key not in cache:
            c = self.adj[e]                                            # CSR(S): row m lists the columns n of S[m][:]
            d = torch.zeros((self.N, self.N), dtype=dtype, device=self.device)
            d[c.rows(), c.col.long()] = c.val(dtype)
            cache[key] = d
        return cache[key]

    def mask_transposed(self):
        """Transposed attention support: (t_rowptr int32 [N+1], t_row int32 [nnz], t_pos int32 [nnz], edge_row int32 [nnz])
        -- for column n the rows m with (m, n) in the support and the position of that edge in mask.col / mask_vals;
        edge_row = the row of every edge in mask order."""
        mt = self.__dict__.get('_mask_t')
        if mt is None:
            rp = self.mask.rowptr.cpu().numpy().astype(np.int64)
            col = self.mask.col.cpu().numpy()
            rows = np.repeat(np.arange(self.N, dtype=np.int32), np.diff(rp))
            order = np.lexsort((rows, col))                          # by (column, row)
            t_rowptr = np.zeros(self.N + 1, dtype=np.int64)
            np.add.at(t_rowptr, col.astype(np.int64) + 1, 1)
            t_rowptr = np.cumsum(t_rowptr).astype(np.int32)
            mt = tuple(torch.from_numpy(np.ascontiguousarray(a)).to(self.device)
                       for a in (t_rowptr, rows[order].astype(np.int32), order.astype(np.int32), rows))
            self._mask_t = mt
        return mt

    def edge_plan(self, e=0):
        """Packed attention support for the fused edge gate (gcrnn_fused_edge_attention_bf16 and its backward): support rows m
        (rowptr, r_edge[j] = {n, bits of (S+I)[m][n]}), support columns n (t_rowptr, t_edge[q] = {m, bits of (S+I)[m][n]}) and
        t_pos[q] = the position of column-ordered edge q in the row order, t_order / r_order = the nodes by descending in- / out-degree;
        all int32 on the device."""
        cache = self.__dict__.setdefault('_edge_plan', {})
        if e not in cache:
            trp, trow, tpos, _ = self.mask_transposed()
            bits = self.mask_vals[e].to(torch.float32).contiguous().view(torch.int32)
            r_edge = torch.stack([self.mask.col.to(torch.int32), bits], dim=1).contiguous()
            t_edge = torch.stack([trow, bits[tpos.long()]], dim=1).contiguous()
            indeg = (trp[1:] - trp[:-1]).long()
            t_order = torch.argsort(indeg, descending=True, stable=True).to(torch.int32).contiguous()
            outdeg = (self.mask.rowptr[1:] - self.mask.rowptr[:-1]).long()
            r_order = torch.argsort(outdeg, descending=True, stable=True).to(torch.int32).contiguous()
            cache[e] = {'rowptr': self.mask.rowptr, 'r_edge': r_edge, 't_rowptr': trp, 't_edge': t_edge, 't_pos': tpos,
                        't_order': t_order, 'r_order': r_order, 'nnz': int(self.mask.nnz),
                        'max_out_degree': int(outdeg.max().item()) if outdeg.numel() else 0}
        return cache[e]

    def rank1_factors(self, adjoint=False):
        """(a, b) with S[m][n] = a[m] b[n] on the support of S (E = 1), or None: the graphs a uniform adjacency turns into under the
        usual normalisations -- D^-1/2 A D^-1/2 (a = b = d^-1/2; reference Utils/graphTools.py:64 normalizeAdjacency), the random-walk forms
        D^-1 A / A D^-1, any of them divided by an eigenvalue. Found by breadth-first propagation of a[m] = S[m][n] / b[n] over the bipartite
        graph of rows and columns (numpy over whole frontiers; depth = the graph's diameter, so rounding does not pile up along a path),
        polished by two alternating averaging sweeps, and verified on EVERY non-zero to 8 eps of the precision the GSO was given in (fp32-
        representable values: 9.5e-7; fp64: 1.8e-15) -- an almost-rank-1 operator is NOT replaced by a rank-1 approximation (the
        fp32-accurate path's whole budget is 1e-5). adjoint: the factors of S^T (a and b swap)."""
        key = '_rank1_adj' if adjoint else '_rank1'
        if key in self.__dict__:
            return self.__dict__[key]
        res = None
        if self.E == 1 and self.nnz > 0:
            c = self.adj[0]                                            # CSR(S): row m lists the columns n of S[m][:]
            rp = c.rowptr.cpu().numpy().astype(np.int64)
            col = c.col.cpu().numpy().astype(np.int64)
            val = c.val(torch.float64).cpu().numpy()
            N = self.N
            rows = np.repeat(np.arange(N), np.diff(rp))
            ok = bool(np.all(val != 0)) and not bool(np.all(val == val[0]))
            if ok:
                eps = 2.0 ** -23 if np.array_equal(val.astype(np.float32).astype(np.float64), val) else 2.0 ** -52
                a = np.zeros(N); b = np.zeros(N)
                seen_a = np.zeros(N, dtype=bool); seen_b = np.zeros(N, dtype=bool)
                has_row = np.diff(rp) > 0
                while True:
                    todo = np.flatnonzero(has_row & ~seen_a)
                    if todo.size == 0:
                        break
                    a[todo[0]] = 1.0; seen_a[todo[0]] = True
                    new_rows = np.zeros(N, dtype=bool); new_rows[todo[0]] = True
                    while new_rows.any():
                        e = np.flatnonzero(new_rows[rows] & ~seen_b[col])          # edges from the row frontier to unseen columns
                        new_cols = np.zeros(N, dtype=bool)
                        if e.size:
                            n_u, first = np.unique(col[e], return_index=True)
                            b[n_u] = val[e[first]] / a[rows[e[first]]]
                            seen_b[n_u] = True; new_cols[n_u] = True
                        e = np.flatnonzero(new_cols[col] & ~seen_a[rows])          # ... and back to unseen rows
                        new_rows = np.zeros(N, dtype=bool)
                        if e.size:
                            m_u, first = np.unique(rows[e], return_index=True)
                            a[m_u] = val[e[first]] / b[col[e[first]]]
                            seen_a[m_u] = True; new_rows[m_u] = True
                with np.errstate(divide='ignore', invalid='ignore'):
                    for _ in range(2):                                            # polish: each factor = the mean of what its non-zeros say
                        cnt_a = np.bincount(rows, minlength=N); cnt_b = np.bincount(col, minlength=N)
                        a_new = np.bincount(rows, weights=val / b[col], minlength=N) / np.maximum(cnt_a, 1)
                        a = np.where(cnt_a > 0, a_new, a)
                        b_new = np.bincount(col, weights=val / a[rows], minlength=N) / np.maximum(cnt_b, 1)
                        b = np.where(cnt_b > 0, b_new, b)
                    good = np.all(np.isfinite(a)) and np.all(np.isfinite(b)) and np.all(np.abs(a[rows] * b[col] - val) <= 8.0 * eps * np.abs(val))
                if good:
                    # balance the two factors (any split works; this one keeps both in fp32's comfortable range)
                    na, nb_ = np.abs(a[seen_a]).max(), np.abs(b[seen_b]).max()
                    g = np.sqrt(nb_ / na) if na > 0 and nb_ > 0 else 1.0
                    res = (a * g, b / g)
        if res is not None and adjoint:
            res = (res[1], res[0])
        self.__dict__[key] = res
        return res

    def fused_plan_rank1(self, adjoint=False):
        """The bf16-image plan (as fused_plan_img16) of the 0/1 PATTERN of a rank-1-weighted graph plus its two factor tables, for the wide
        sequence-resident kernel: a hop is b[n] * sum_{m in N(n)} (a[m] v[m]) -- the image holds a (.) v, the sums are scaled by b before the
        tap is added (csrc/gcrnn_fused_seq32.h, R1). None for uniform graphs (they have their own plan) and graphs that are not rank-1."""
        key = '_fused_plan_rank1_adj' if adjoint else '_fused_plan_rank1'
        if key in self.__dict__:
            return self.__dict__[key]
        res = None
        import os
        fac = None if os.environ.get('GCRNN_NO_RANK1') else self.rank1_factors(adjoint=False)
        if fac is not None:
            pat = self.__dict__.get('_pattern_operator')
            if pat is None:
                Sd = np.zeros((1, self.N, self.N))
                c = self.adj[0]
                Sd[0, c.rows().cpu().numpy(), c.col.cpu().numpy().astype(np.int64)] = 1.0
                pat = GraphOperator(Sd, device=self.device)
                self._pattern_operator = pat
            p16 = pat.fused_plan_img16(adjoint=adjoint)
            if p16 is not None:
                a, b = (fac[1], fac[0]) if adjoint else fac
                npad = p16['npad']
                ta = np.zeros(npad, dtype=np.float32); tb = np.zeros(npad, dtype=np.float32)
                ta[:self.N] = a; tb[:self.N] = b
                res = dict(p16)
                res.update(rank1_a=torch.from_numpy(ta).to(self.device), rank1_b=torch.from_numpy(tb).to(self.device), uniform_w=1.0, rank1=True)
        self.__dict__[key] = res
        return res

    def fused_plan_x3(self, adjoint=False):
        """The plan the fp32-accurate ("x3") kernels run on: fused_plan(adjoint) for uniform-weight graphs; for RANK-1-weighted graphs
        (rank1_factors: normalised adjacencies, Utils/graphTools.py:64) the plan of the 0/1 pattern (uniform_w = 1) plus 'rank1_x3', the
        [4][NPad] fp32 table a | a b | 1 / b | b of include/gcrnn.h (a = source factor, b = destination factor of this direction; 1 where b = 0).
        Other graphs: fused_plan(adjoint) as it is (uniform_w = 0: the x3 predicates say no)."""
        key = '_fused_plan_x3_adj' if adjoint else '_fused_plan_x3'
        if key in self.__dict__:
            return self.__dict__[key]
        res = self.fused_plan(adjoint=adjoint)
        import os
        if res.get('uniform_w', 0.0) == 0.0 and not os.environ.get('GCRNN_NO_RANK1'):
            fac = self.rank1_factors(adjoint=False)
            if fac is not None:
                pat = self.__dict__.get('_pattern_operator')
                if pat is None:
                    Sd = np.zeros((1, self.N, self.N))
                    c = self.adj[0]
                    Sd[0, c.rows().cpu().numpy(), c.col.cpu().numpy().astype(np.int64)] = 1.0
                    pat = GraphOperator(Sd, device=self.device)
                    self._pattern_operator = pat
                pp = pat.fused_plan(adjoint=adjoint)
                if pp.get('uniform_w', 0.0) == 1.0:
                    a, b = (fac[1], fac[0]) if adjoint else (fac[0], fac[1])
                    npad = pp['npad']
                    tab = np.zeros((4, npad), dtype=np.float64)
                    tab[2:] = 1.0
                    nz = b != 0
                    tab[0, :self.N] = a
                    tab[1, :self.N] = a * np.where(nz, b, 1.0)
                    tab[2, :self.N] = np.where(nz, 1.0 / np.where(nz, b, 1.0), 1.0)
                    tab[3, :self.N] = np.where(nz, b, 1.0)
                    res = dict(pp)
                    res['rank1_x3'] = torch.from_numpy(tab.astype(np.float32)).to(self.device).contiguous()
        self.__dict__[key] = res
        return res

    def fused_plan(self, adjoint=False, kernel='step'):
        """Degree-sorted sliced ELL of CSR(S^T) (forward shift) or, with adjoint=True, of CSR(S) (the shift of the
        backward pass) for the fused step kernels (E = 1): device tensors order (int32 [N]), tile_off
        (int32 [ntiles+1]), ell_col (int32 [entries*16]), ell_val (fp32) and the packed LDS image."""
        waves = int(_lib.lib.gcrnn_fused_wgrad_waves() if kernel == 'wgrad' else _lib.lib.gcrnn_fused_step_waves())
        key = ('_fused_plan_adj' if adjoint else '_fused_plan') + '_w%d' % waves
        plan = self.__dict__.get(key)
        if plan is not None:
            return plan
        assert self.E == 1
        npad = int(_lib.lib.gcrnn_fused_padded_nodes())
        assert self.N <= npad
        ntiles = npad // 16
        c = self.adj[0] if adjoint else self.fwd[0]
        rowptr = np.ascontiguousarray(c.rowptr.cpu().numpy())
        col = np.ascontiguousarray(c.col.cpu().numpy())
        val = np.ascontiguousarray(c.val(torch.float64).cpu().numpy())
        order = degree_order(rowptr)
        # tiles of 16 slots in degree order; the kernel's wave w owns `per` consecutive STORAGE tiles (contiguous, so its hop
        # is one stream of ELL groups) -- deal the degree-ranked tiles round-robin over the waves to balance them.
        slots = np.concatenate([order, np.arange(self.N, npad, dtype=np.int32)]).astype(np.int32).reshape(ntiles, 16)
        import os
        if not os.environ.get('GCRNN_PLAN_NO_CLASS_SPREAD'):                 # env: A/B switch
            slots = spread_tile_classes(slots, (rowptr[1:] - rowptr[:-1]), self.N)
        per = ntiles // waves
        storage = np.empty_like(slots)
        if os.environ.get('GCRNN_PLAN_ROUND_ROBIN'):                          # env: A/B switch (the dealing up to round 3)
            for w in range(waves):
                for i in range(per):
                    storage[w * per + i] = slots[i * waves + w]
        else:
            # A wave's hop is as long as the ELL groups of its tiles (a tile has ceil(max degree / 4) groups) and the workgroup waits for
            # its slowest wave at every hop's barrier. Round-robin over the degree ranking hands wave 0 the largest tile of every round
            # (bench graph: 26 groups against 20 for the last wave, mean 22.9); longest-processing-time dealing -- largest tile first,
            # to the wave with the fewest groups that still has a free tile slot -- levels them (23 / 22).
            deg_pad = np.concatenate([rowptr[1:] - rowptr[:-1], np.zeros(npad - self.N, dtype=rowptr.dtype)])
            groups = (deg_pad[slots].max(axis=1) + 3) // 4
            load = np.zeros(waves, dtype=np.int64)
            fill = np.zeros(waves, dtype=np.int64)
            # GCRNN_PLAN_YOUNG_COST=c (experiment): a group costs the second half of the waves c times what it costs the first -- the younger
            # wave of each SIMD loses issue arbitration, so with every wave running the same program (gcrnn_fused_seq32p.h) waves 4..7 leave
            # their streams ~30 % after waves 0..3 (profiles/r05_p32_stamps_*.txt); fewer groups for them levels the finish times
            cost = np.ones(waves)
            cost[waves // 2:] = float(os.environ.get('GCRNN_PLAN_YOUNG_COST', '1.0'))
            for t in np.argsort(-groups, kind='stable'):
                free = np.flatnonzero(fill < per)
                w = free[np.argmin((load[free] + groups[t]) * cost[free])]
                storage[w * per + fill[w]] = slots[t]
                load[w] += groups[t]
                fill[w] += 1
        order_full = storage.reshape(-1)                                   # node id (or padding row id >= N) per slot
        nent = C.c_int64(0)
        vp = lambda a: a.ctypes.data_as(C.c_void_p)
        # padding rows (ids >= N) have no edges: extend rowptr so that they can be addressed like nodes
        rowptr_pad = np.concatenate([rowptr, np.full(npad - self.N, rowptr[-1], dtype=np.int32)]).astype(np.int32)
        _lib.check(_lib.lib.gcrnn_ell_size(vp(rowptr_pad), npad, vp(order_full), 16, 4, ntiles, C.byref(nent)), 'ell_size')
        tile_off = np.zeros(ntiles + 1, dtype=np.int32)
        ell_col = np.zeros(max(nent.value, 1) * 16, dtype=np.int32)
        ell_val = np.zeros(max(nent.value, 1) * 16, dtype=np.float32)
        # LDS row + quad swizzle of every node, chosen so that the gathers of every tile spread over the 16 bank quads
        node_addr = np.zeros(npad, dtype=np.int32)
        # Uniform-weight graph (all non-zeros equal -- the reference drivers' W / lambda_max of an unweighted adjacency,
        # kStepPredGRNNs.py:768) with at least 16 padding rows: the padding rows (state always zero) take the 16 bank keys in
        # turn and every padding entry points at one of them, so the forward kernels may drop the weight image (uniform_w).
        v32 = val.astype(np.float32)
        uniform_w = 0.0
        import os
        if val.size and np.all(v32 == v32[0]) and v32[0] != 0 and npad - self.N >= 16 and not os.environ.get('GCRNN_NO_UNIFORM'):   # env: A/B switch
            uniform_w = float(v32[0])
        zero_from = self.N if uniform_w != 0.0 else -1
        _lib.check(_lib.lib.gcrnn_ell_assign_rows_z(vp(rowptr_pad), vp(col), npad, vp(order_full), 4, ntiles, zero_from,
                                                    vp(node_addr)), 'ell_assign_rows')
        _lib.check(_lib.lib.gcrnn_ell_fill_z(vp(rowptr_pad), vp(col), vp(val), npad, vp(order_full), 16, 4, ntiles,
                                             vp(node_addr), zero_from, vp(tile_off), vp(ell_col), vp(ell_val)), 'ell_fill')
        cyc = C.c_int64(0)
        _lib.check(_lib.lib.gcrnn_ell_conflict_cycles(vp(ell_col), nent.value, vp(node_addr), C.byref(cyc)),
                   'ell_conflict_cycles')
        val4 = np.zeros(max(nent.value, 4) * 16, dtype=np.float32)
        col4 = np.zeros(max(nent.value, 4) * 16, dtype=np.uint16)
        _lib.check(_lib.lib.gcrnn_ell_pack_lds(vp(ell_col), vp(ell_val), nent.value, vp(node_addr), vp(val4), vp(col4)),
                   'ell_pack_lds')
        dev = self.device
        tile_nodes = order_full
        tile_slots = ((tile_nodes.astype(np.int64) << 16) | node_addr[tile_nodes]).astype(np.int32)   # what the kernels take
        ell_addr = node_addr[ell_col].astype(np.int32)
        plan = dict(npad=npad, order=torch.from_numpy(order).to(dev), tile_nodes=torch.from_numpy(tile_nodes).to(dev),
                    tile_slots=torch.from_numpy(tile_slots).to(dev), node_addr=torch.from_numpy(node_addr).to(dev),
                    tile_off=torch.from_numpy(tile_off).to(dev),
                    ell_col=torch.from_numpy(ell_col).to(dev), ell_addr=torch.from_numpy(ell_addr).to(dev),
                    ell_val=torch.from_numpy(ell_val).to(dev),
                    ell_val4=torch.from_numpy(val4).to(dev), ell_col4=torch.from_numpy(col4.view(np.int16)).to(dev),
                    entries=int(nent.value), gather_cycles=int(cyc.value), uniform_w=uniform_w)
        setattr(self, key, plan)
        return plan

    def fused_plan_img16(self, adjoint=False):
        """The forward plan re-addressed for a bf16 hop image (32-byte state rows; the un-gated step kernel then sums the gathered
        rows on the matrix cores, csrc GCRNN_HOP_ASM_UNI16_STREAM), or None when the graph has no uniform-weight plan.
        A row's two 16-byte halves are XOR-swizzled by `hswz`; node_addr16 = row16 << 5 | hswz << 4. The gather key of the fp32 image
        (swz << 2 | row & 3, slots 4..11 flipped by 4) maps one-to-one onto the new one ((row16 & 7) << 1 | hswz, flipped by 1) with
        hswz = swz & 1 and row16 & 7 a bijection of (row & 3, swz >> 1), so the conflict-free schedule of the entries carries over as it is.
        Column words per slot and group of four entries: (entry 0, entry 2, entry 1, entry 3) as four uint16 neighbour addresses."""
        key = '_fused_plan_img16_adj' if adjoint else '_fused_plan_img16'
        plan16 = self.__dict__.get(key, False)
        if plan16 is not False:
            return plan16
        plan = self.fused_plan(adjoint=adjoint)
        plan16 = None
        if plan.get('uniform_w', 0.0) != 0.0 and plan['entries'] > 0:
            npad = plan['npad']
            na = plan['node_addr'].cpu().numpy().astype(np.int64)
            row, swz = na >> 6, (na >> 4) & 3
            # row16 & 7 = row bit 0 | swz bit 1 << 1 | row bit 1 << 2: bits 0..1 and hswz are then the three bits the fp32 image's
            # write-aware search makes a permutation in every half tile -> every (row16 & 3, hswz) exactly twice per tile, the best a
            # 16-lane ds_write_b64 group can have (2-way); bit 2 is that search's free bit
            cls8 = (row & 1) | ((swz >> 1) << 1) | (((row >> 1) & 1) << 2)
            hswz = swz & 1
            cap = npad // 8
            row16 = np.zeros(npad, dtype=np.int64)
            refs = np.bincount(plan['ell_col'].cpu().numpy().astype(np.int64), minlength=npad)
            order = np.argsort(-refs, kind='stable')         # the most-gathered rows (the zero rows of the padding entries first) keep their class
            left = []
            cnt = np.zeros(8, dtype=np.int64)
            for n in order:
                c = int(cls8[n])
                if cnt[c] < cap:
                    row16[n] = c + 8 * cnt[c]; cnt[c] += 1
                else:
                    left.append(int(n))
            for n in left:                                   # class full: the sibling class (other value of the free row bit), else any
                c = int(cls8[n]) ^ 4
                if cnt[c] >= cap:
                    c = int(np.argmin(cnt))
                row16[n] = c + 8 * cnt[c]; cnt[c] += 1
            addr16 = ((row16 << 5) | (hswz << 4)).astype(np.int64)
            tile_nodes = plan['tile_nodes'].cpu().numpy().astype(np.int64)
            tile_slots = ((tile_nodes << 16) | addr16[tile_nodes]).astype(np.int32)
            ent = plan['entries']
            nb = addr16[plan['ell_col'].cpu().numpy().astype(np.int64)].reshape(ent // 4, 4, 16)          # [group][entry][slot]
            col4 = np.ascontiguousarray(nb[:, [0, 2, 1, 3], :].transpose(0, 2, 1)).astype(np.uint16)     # [group][slot][e0, e2, e1, e3]
            dev = self.device
            plan16 = dict(plan)
            plan16.update(tile_slots=torch.from_numpy(tile_slots).to(dev),
                          ell_col4=torch.from_numpy(col4.reshape(-1).view(np.int16)).to(dev),
                          node_addr16=torch.from_numpy(addr16.astype(np.int32)).to(dev), img16=True, img16_moved=len(left))
        setattr(self, key, plan16)
        return plan16

    def to(self, device):
        device = torch.device(device)
        if device.type == 'cuda' and device.index is None:
            device = torch.device('cuda', torch.cuda.current_device())
        if device == self.device:
            return self
        moved = self.__dict__.setdefault('_moved', {})
        if device in moved:
            return moved[device]
        out = GraphOperator.__new__(GraphOperator)
        out.E, out.N, out.nnz, out.device = self.E, self.N, self.nnz, device
        out.fwd = [c.to(device) for c in self.fwd]
        out.adj = [c.to(device) for c in self.adj]
        out.mask = self.mask.to(device)
        out.mask_vals = [v.to(device) for v in self.mask_vals]
        moved[device] = out
        out._moved = {self.device: self}
        return out

    def __repr__(self):
        return 'GraphOperator(E=%d, N=%d, nnz=%d, device=%s)' % (self.E, self.N, self.nnz, self.device)


_CACHE = {}


def as_operator(S):
    """Accept a GraphOperator or a dense E x N x N tensor. Dense tensors are converted once and cached on
    (storage address, shape, version, device, dtype); the cache entry holds a reference to the tensor so that its
    address cannot be recycled for a different matrix while the entry is alive. Strides and storage offset are part of the
    key: S.transpose(1, 2) shares address, shape and version counter with S but is a different operator. A GSO edited in
    place through a path that does not bump `_version` (numpy-shared memory, `.data`) must be passed to addGSO again."""
    if isinstance(S, GraphOperator):
        return S
    assert isinstance(S, torch.Tensor), 'GSO must be a torch.Tensor or a GraphOperator'
    assert S.dim() == 3, 'GSO must be E x N x N'
    if S.is_inference():                   # (made under torch.inference_mode: no version counter to key on -- converted, not cached)
        return GraphOperator(S, device=S.device)
    key = (S.data_ptr(), tuple(S.shape), tuple(S.stride()), S.storage_offset(), S._version, str(S.device), S.dtype)
    hit = _CACHE.get(key)
    if hit is not None and hit[0] is S:
        return hit[1]
    if hit is not None and hit[0].data_ptr() == S.data_ptr() and hit[0]._version == S._version and \
            hit[0].stride() == S.stride() and hit[0].storage_offset() == S.storage_offset():
        return hit[1]                      # the same view (address, strides, offset) of the same live storage
    if len(_CACHE) > 16:
        _CACHE.clear()
    op = GraphOperator(S, device=S.device)
    _CACHE[key] = (S, op)
    return op


def operator_from_csr(rowptr, col, val, N, device=None):
    """GraphOperator for graphs that cannot be held densely (BASELINE config 5: N = 1e5): `rowptr/col/val` is the CSR of
    S itself (row m lists S[m, :]). The forward shift needs CSR(S^T), built here by a counting transpose; the attention
    support (edge gate) is not built for such graphs."""
    rowptr = np.ascontiguousarray(rowptr, dtype=np.int64)
    col = np.ascontiguousarray(col, dtype=np.int32)
    val = np.ascontiguousarray(val, dtype=np.float64)
    assert rowptr.size == N + 1 and col.size == val.size == rowptr[-1]
    rows = np.repeat(np.arange(N, dtype=np.int32), np.diff(rowptr))
    order = np.argsort(col, kind='stable')               # entries are row-sorted already: stable by column = (column, row) order
    t_rowptr = np.concatenate([[0], np.cumsum(np.bincount(col, minlength=N))]).astype(np.int64)
    op = GraphOperator.__new__(GraphOperator)
    op.E, op.N = 1, int(N)
    op.device = torch.device(device if device is not None else 'cpu')
    op.adj = [CSR(rowptr.astype(np.int32), col, val, device=op.device)]
    op.fwd = [CSR(t_rowptr.astype(np.int32), rows[order].astype(np.int32), val[order], device=op.device)]
    op.mask, op.mask_vals = None, None
    op.nnz = int(col.size)
    return op


def erdos_renyi_csr(N, density, seed=0):
    """Directed Erdos-Renyi graph generated directly in CSR (never dense) -- the synthetic graph of BASELINE configs[4]
    (N = 1e5, p = 1e-3, nnz ~ 1e7): per-row degrees ~ Binomial(N, p), columns uniform without
    repetition, weights U(0, 1) scaled by 1 / (max row sum) as a cheap spectral bound. Returns (rowptr int64[N+1],
    col int32[nnz], val float64[nnz]) of S (row m lists S[m, :], columns ascending)."""
    rng = np.random.default_rng(seed)
    deg = rng.binomial(N, density, size=N).astype(np.int64)
    rows = np.repeat(np.arange(N, dtype=np.int64), deg)
    cols = rng.integers(0, N, size=rows.size, dtype=np.int64)
    key = np.unique(rows * N + cols)                      # sorted by (row, column); a repeated column in a row is dropped
    rows, cols = key // N, (key % N).astype(np.int32)
    rowptr = np.concatenate([[0], np.cumsum(np.bincount(rows, minlength=N))]).astype(np.int64)
    val = rng.random(cols.size)
    rowsum = np.bincount(rows, weights=val, minlength=N)
    val /= rowsum.max()
    return rowptr, cols, val
