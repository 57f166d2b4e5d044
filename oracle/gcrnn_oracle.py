"""CPU oracle for the gated-GCRNN hot path (TEST INFRASTRUCTURE, NOT PRODUCT).

This file is a plain-numpy restatement of the arithmetic of the reference's hot
path (luanaruiz9/gated_gcrnns, Utils/graphML.py and Modules/architectures.py).
It is written from the equations, not from the
reference's code, and keeps the reference's *dense* graph shift operator so its
cost model is the reference's own (dense x@S per hop).

Who may import it: only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg, and only as the checker. The product (gated_gcrnns_amd/)
never imports anything from oracle/.

Parity status: PINNED by golden vectors generated from the imported reference
in the build container (tests/golden/make_golden.py -> tests/golden/*.npz); the
reference itself ships no tests or known-answer vectors.

Conventions (all follow the reference):
  * h (filter taps)  : F x E x K x G                (graphML.py:87)
  * S (GSO)          : E x N x N, row-vector shift x@S  (graphML.py:116-123)
  * x                : B x G x N ; sequences X : B x T x G x N (graphML.py:2186)
  * parameters are passed as a dict keyed by the reference's state_dict keys
    relative to the cell (weight_A, weight_B, bias, GFL_in.weight_A, ...).
"""
import numpy as np

ZERO_TOLERANCE = 1e-9      # graphML.py:42
INFINITE_NUMBER = 1e12     # graphML.py:43


def sigmoid(v):
    return 1.0 / (1.0 + np.exp(-v))


def lsigf(h, S, x, b=None):
    """y[b,f,n] = sum_e sum_k sum_g h[f,e,k,g] (x S_e^k)[b,g,n] + b[f]   (graphML.py:47-140).

    k = 0 term is x itself for every e (graphML.py:118); shift is x @ S_e
    (graphML.py:123); contraction order of the flattened axis is (e, k, g)
    (graphML.py:134-135); bias broadcast over nodes (graphML.py:138-139).
    """
    F, E, K, G = h.shape
    assert S.shape[0] == E and S.shape[1] == S.shape[2]
    N = S.shape[1]
    B = x.shape[0]
    assert x.shape[1] == G and x.shape[2] == N
    y = np.zeros((B, F, N), dtype=np.result_type(h.dtype, x.dtype))
    for e in range(E):
        z = x
        for k in range(K):
            if k > 0:
                z = z @ S[e]                           # B x G x N, row-vector shift
            # y[b,f,n] += sum_g h[f,e,k,g] z[b,g,n]
            y = y + np.matmul(h[:, e, k, :], z)                # [F x G] @ [B x G x N] -> B x F x N (BLAS)
    if b is not None:
        y = y + b.reshape(1, F, -1)
    return y


def graph_filter(weight, bias, S, x):
    """GraphFilter.forward incl. the Nin < N zero-pad / trim path (graphML.py:1175-1194)."""
    N = S.shape[1]
    Nin = x.shape[2]
    if Nin < N:
        x = np.concatenate([x, np.zeros((x.shape[0], x.shape[1], N - Nin), dtype=x.dtype)], axis=2)
    u = lsigf(weight, S, x, bias)
    if Nin < N:
        u = u[:, :, :Nin]
    return u


def graph_attention(x, a, W, S, negative_slope=0.2):
    """Dense GAT used as the edge gate (graphML.py:521-627). Returns B x K x F x N."""
    B, G, N = x.shape
    K, E = a.shape[0], a.shape[1]
    F = W.shape[2]
    assert a.shape[2] == 2 * F
    S = S + np.eye(N, dtype=S.dtype).reshape(1, N, N)              # :577
    Wx = np.einsum('kefg,bgn->bkefn', W, x)                        # :588  B x K x E x F x N
    a1, a2 = a[:, :, :F], a[:, :, F:]                              # :591-592
    a1Wx = np.einsum('kef,bkefn->bken', a1, Wx)                    # :593
    a2Wx = np.einsum('kef,bkefn->bken', a2, Wx)                    # :594
    # aWx[m, n] = a1.Wx[:, n] + a2.Wx[:, m]                          :597
    aWx = a1Wx[:, :, :, None, :] + a2Wx[:, :, :, :, None]
    eij = np.where(aWx >= 0, aWx, negative_slope * aWx)            # :603
    mask = (np.sum(np.abs(S), axis=0) > ZERO_TOLERANCE).astype(x.dtype)   # :611-613
    logits = eij * mask - (1.0 - mask) * INFINITE_NUMBER           # :615-618
    logits = logits - logits.max(axis=4, keepdims=True)
    ex = np.exp(logits)
    aij = ex / ex.sum(axis=4, keepdims=True)                       # softmax over last axis (n)
    aij = aij * mask                                               # :622
    y = np.einsum('bkefm,bkemn->bkefn', Wx, S.reshape(1, 1, E, N, N) * aij)   # :625
    return y.sum(axis=2)                                           # :627


def graph_attentional(mixer, weight, S, x):
    """GraphAttentional.forward with concatenate=True and ReLU (graphML.py:2084-2116)."""
    B = x.shape[0]
    y = graph_attention(x, mixer, weight, S)                       # B x K x F x N
    y = np.maximum(y, 0.0)                                         # :2101
    K, F, N = y.shape[1], y.shape[2], y.shape[3]
    return y.reshape(B, K * F, N)                                  # :2105-2107 (k-major, then f)


def _sub(params, prefix):
    plen = len(prefix)
    return {k[plen:]: v for k, v in params.items() if k.startswith(prefix)}


def _plain_step(p, S, x, h, sigma):
    """One un-gated step sigma(LSIGF(A,S,x,b) + LSIGF(B,S,h,b)); bias added twice (graphML.py:2420-2423)."""
    b = p.get('bias', None)
    return sigma(lsigf(p['weight_A'], S, x, b) + lsigf(p['weight_B'], S, h, b))


def ggcrnn_cell(params, S, X, h0, time_gating=False, spatial_gating=None, sigma=np.tanh):
    """GGCRNNCell.forward (graphML.py:2336-2428). X: B x T x G x N, h0: B x F x N -> H: B x T x F x N.

    Every gate is computed from (x_t, h0) -- the INITIAL state (graphML.py:2362, 2370, 2383, 2393).
    """
    B, T = X.shape[0], X.shape[1]
    assert h0.shape[0] == B
    A, Bw = params['weight_A'], params['weight_B']
    b = params.get('bias', None)
    F = A.shape[0]
    N = S.shape[1]
    H = []
    h = h0
    for t in range(T):
        x = X[:, t]
        gi = np.ones((B, 1, 1), dtype=X.dtype)
        gf = np.ones((B, 1, 1), dtype=X.dtype)
        if time_gating:
            ci = _plain_step(_sub(params, 'GFL_in.'), S, x, h0, sigma).reshape(B, F * N)       # :2362-2364
            gi = sigmoid(ci @ params['MLP_in.0.weight'].T + params.get('MLP_in.0.bias', 0.0)).reshape(B, 1, 1)
            cf = _plain_step(_sub(params, 'GFL_forget.'), S, x, h0, sigma).reshape(B, F * N)   # :2370-2372
            gf = sigmoid(cf @ params['MLP_forget.0.weight'].T + params.get('MLP_forget.0.bias', 0.0)).reshape(B, 1, 1)
        ya = lsigf(A, S, x, b)
        yb = lsigf(Bw, S, h, b)
        if spatial_gating == 'node':
            di = _plain_step(_sub(params, 'GRNN_node_in.'), S, x, h0, sigma)                    # :2383
            ni = sigmoid(graph_filter(params['GFL_node_in.0.weight'], params.get('GFL_node_in.0.bias'), S, di))
            df = _plain_step(_sub(params, 'GRNN_node_forget.'), S, x, h0, sigma)                # :2393
            nf = sigmoid(graph_filter(params['GFL_node_forget.0.weight'], params.get('GFL_node_forget.0.bias'), S, df))
            h = sigma(gi * (ni * ya) + gf * (nf * yb))                                          # :2402-2407
        elif spatial_gating == 'edge':
            ya = graph_attentional(params['input_attention.mixer'], params['input_attention.weight'], S, ya)
            yb = graph_attentional(params['forget_attention.mixer'], params['forget_attention.weight'], S, yb)
            h = sigma(gi * ya + gf * yb)                                                        # :2411-2416
        else:
            h = sigma(gi * ya + gf * yb)                                                        # :2420-2423
        H.append(h)
    return np.stack(H, axis=1)


def _mlp(params, prefix, x, nonlin):
    """nn.Sequential of Linear / nonlinearity / Linear ... as built at architectures.py:1543-1567."""
    idx = sorted({int(k[len(prefix):].split('.')[0]) for k in params if k.startswith(prefix)})
    for j, i in enumerate(idx):
        w = params['%s%d.weight' % (prefix, i)]
        bb = params.get('%s%d.bias' % (prefix, i), None)
        if j > 0:
            x = nonlin(x)
        x = x @ w.T
        if bb is not None:
            x = x + bb
    return x


def relu(v):
    return np.maximum(v, 0.0)


def gated_gcrnn_regression(params, S, x, h0, time_gating=False, spatial_gating=None,
                           mlp_type='oneMlp', sigma=np.tanh, rho=relu):
    """GatedGCRNNforRegression.forward with an MLP head (architectures.py:1607-1636).

    params keys are the full-model state_dict keys (stateGCRNN.*, outputNN.*).
    Returns B x T x 1 x (N*out) ('multipMlp': per-node perceptron, out features
    major then node -- flatY.transpose(1,2) at :1626).
    """
    B, T = x.shape[0], x.shape[1]
    cell = _sub(params, 'stateGCRNN.')
    H = ggcrnn_cell(cell, S, x, h0, time_gating, spatial_gating, sigma)
    F, N = H.shape[2], H.shape[3]
    flat = H.reshape(B * T, F, N)
    if mlp_type == 'multipMlp':
        hn = flat.transpose(0, 2, 1)                       # (BT) x N x F      :1618
        yn = _mlp(params, 'outputNN.', hn, rho)            # (BT) x N x out    :1620-1625
        flatY = yn.transpose(0, 2, 1)                      # (BT) x out x N    :1626
    else:
        flatY = _mlp(params, 'outputNN.', flat.reshape(B * T, F * N), rho)   # :1629-1630
    return flatY.reshape(B, T, -1)[:, :, None, :]          # :1634-1635


def gated_gcrnn_classification(params, S, x, h0, time_gating=False, spatial_gating=None,
                               sigma=np.tanh, rho=relu):
    """GatedGCRNNforClassification.forward with an MLP head (architectures.py:1841-1850)."""
    cell = _sub(params, 'stateGCRNN.')
    H = ggcrnn_cell(cell, S, x, h0, time_gating, spatial_gating, sigma)
    h = H[:, -1]                                           # :1844
    return _mlp(params, 'outputNN.', h.reshape(h.shape[0], -1), rho)   # :1846-1847


def batch_time_l1_loss(x, y):
    """miscTools.batchTimeL1Loss (miscTools.py:112-119): plain mean absolute error."""
    return np.mean(np.abs(x - y))


def batch_time_mse_loss(x, y):
    """miscTools.batchTimeMSELoss (miscTools.py:121-130).

    Rows = everything but the last two axes; per flattened (N*F) column:
    sqrt(sum_rows (x-y)^2) / ||y column||_2 ; mean over columns.
    """
    F, N = x.shape[-2], x.shape[-1]
    xv = x.reshape(-1, N * F)
    yv = y.reshape(-1, N * F)
    num = np.sqrt(np.sum((xv - yv) ** 2, axis=0))
    den = np.sqrt(np.sum(yv ** 2, axis=0))
    return np.mean(num / den)


# ---------------------------------------------------------------------------
# CSR helpers (index work: must be bit-exact with the product's CSR builder)
# ---------------------------------------------------------------------------

def csr_from_dense(M, tol=0.0):
    """CSR of a dense N x N matrix keeping entries with |v| > tol, columns ascending."""
    N = M.shape[0]
    rowptr = np.zeros(N + 1, dtype=np.int64)
    cols, vals = [], []
    for i in range(N):
        nz = np.nonzero(np.abs(M[i]) > tol)[0]
        cols.append(nz.astype(np.int32))
        vals.append(M[i, nz])
        rowptr[i + 1] = rowptr[i] + nz.size
    col = np.concatenate(cols) if cols else np.zeros(0, np.int32)
    val = np.concatenate(vals) if vals else np.zeros(0, M.dtype)
    return rowptr, col.astype(np.int32), val


def csr_matvec_rows(rowptr, col, val, Xn):
    """Y[n, :] = sum_j val[j] * Xn[col[j], :] -- node-major SpMM used by the product layout."""
    N = rowptr.size - 1
    Y = np.zeros_like(Xn)
    for n in range(N):
        s, e = rowptr[n], rowptr[n + 1]
        if e > s:
            Y[n] = val[s:e] @ Xn[col[s:e]]
    return Y


# ---------------------------------------------------------------------------
# graphs that cannot be held densely (BASELINE configs[4]: N = 1e5): the same arithmetic, LSIGF (graphML.py:116-139) and the
# un-gated cell step (graphML.py:2420-2423), with the shift x @ S written as a sparse product and evaluated only on the
# node rows that are asked for. Checked against the dense functions above on small graphs (tests/test_oracle_golden.py).
# ---------------------------------------------------------------------------

def lsigf_rows_csr(h, P, x, b, rows):
    """LSIGF output on the nodes `rows` only. h: F x 1 x K x G, P: scipy.sparse CSR of S^T (row n lists S[:, n]),
    x: B x G x N, b: F x 1 or None  ->  B x F x len(rows).
    (x S^k)[:, :, n] is row n of P^k applied to the node-major matrix Z0 = x^T ([N][B*G]); z_k is needed on `rows` for
    every k, hence z_{k-1} on rows + their in-neighbours, and so on down to z_0 which is known everywhere."""
    F, E, K, G = h.shape
    assert E == 1
    B, _, N = x.shape
    rows = np.asarray(rows)
    need = [None] * K
    need[K - 1] = np.unique(rows)
    for k in range(K - 1, 0, -1):
        need[k - 1] = np.union1d(need[k], P[need[k]].indices)
    Z = np.ascontiguousarray(np.transpose(x, (2, 0, 1)).reshape(N, B * G))          # z_0, valid on every row
    y = np.zeros((B, F, rows.size), dtype=np.result_type(h.dtype, x.dtype))
    for k in range(K):
        if k > 0:
            nxt = np.zeros_like(Z)
            nxt[need[k]] = P[need[k]] @ Z                                               # z_k on the rows still needed
            Z = nxt
        zk = Z[rows].reshape(rows.size, B, G)                                           # rows x B x G
        y = y + np.einsum('fg,nbg->bfn', h[:, 0, k, :], zk)
    if b is not None:
        y = y + b.reshape(1, F, 1)
    return y


def cell_step_rows_csr(params, P, x_t, h_prev, rows):
    """One step of the un-gated cell on the nodes `rows`: tanh(LSIGF(A, S, x_t, b) + LSIGF(B, S, h_prev, b)) -- the one bias
    enters through both filters (graphML.py:2420-2423). x_t: B x G x N, h_prev: B x F x N -> B x F x len(rows).
    With equal tap counts the two filters are ONE filter on the stacked signal [h; x] with taps [B | A] (the sum of two
    contractions over disjoint channel blocks), which shares the sparse products between them."""
    b = params.get('bias')
    A, Bw = params['weight_A'], params['weight_B']
    if A.shape[2] == Bw.shape[2]:
        y = lsigf_rows_csr(np.concatenate([Bw, A], axis=3), P, np.concatenate([h_prev, x_t], axis=1), None, rows)
        return np.tanh(y + (2.0 * b.reshape(1, -1, 1) if b is not None else 0.0))
    ya = lsigf_rows_csr(A, P, x_t, b, rows)
    yb = lsigf_rows_csr(Bw, P, h_prev, b, rows)
    return np.tanh(ya + yb)
