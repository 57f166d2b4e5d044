import os
#!/usr/bin/env python3
"""Generator of gated_gcrnns_amd/csrc/gcrnn_hop_asm.inc: the graph-hop gather stream of the fused kernels as ONE inline-asm
block per hop (python3 tools/gen_hop_asm.py > gated_gcrnns_amd/csrc/gcrnn_hop_asm.inc).

Why one block: hipcc treats an asm `ds_read` destination as written at the statement, so any copy it inserts between the read
and our counted wait captures stale data (cdna_hip_programming.md 5.7); inside one block every register is ours. That
makes a THREE-deep stream safe: the gathers of groups n+1 and n+2 (12 LDS reads) are in flight while group n is consumed,
one `s_waitcnt lgkmcnt(12)` per trip. Per trip n (p = n mod 3 selects the register set):
    S1(n+2): 4 gather addresses = 16-bit column fields ^ (q << 4) (v_xor_b32_sdwa), weights + 4 x ds_read_b128 -> set (p+2)%3
    S0(n+5): ds_read_b64 of the column words of group n+5 -> the column slot S1 has just consumed
    wait   : all but the 12 youngest reads have landed  =>  S1(n) and S0(n+3)
    FMA    : 8 x v_pk_fma_f32 of set p into the CURRENT tile's accumulator halves
A wave's 8 tiles are stored back to back, so the stream runs across tile boundaries; only the accumulator operand changes,
hence the trip body exists once per (tile, phase) with the phase carried through the tile switches.
Operands: %0..%15 = lo/hi halves (f32x2) of the 8 tile accumulators (in/out); %16..%23 = end group of every tile (SGPR);
%24 = first group, %25 = last valid group (SGPR); %26 = column base + 8 r, %27 = weight base + 16 r, %28 = q << 4 (VGPR).
Clobbers: v[186:253], s[88:90], scc. The dynamic LDS segment must start at LDS address 0 (the kernels use no static LDS)."""
import sys

BASE = 186
NT = 8


def W(p, half):            # weights of set p: v[b:b+3]; half 0 -> [0:1], 1 -> [2:3]
    b = BASE + 20 * p
    return 'v[%d:%d]' % (b + 2 * half, b + 2 * half + 1)


def Wfull(p):
    b = BASE + 20 * p
    return 'v[%d:%d]' % (b, b + 3)


def X(p, e, half=None):
    b = BASE + 20 * p + 4 + 4 * e
    if half is None:
        return 'v[%d:%d]' % (b, b + 3)
    return 'v[%d:%d]' % (b + 2 * half, b + 2 * half + 1)


def Xa(p, e):
    return 'v%d' % (BASE + 20 * p + 4 + 4 * e)


def Cw(p, hi):
    return 'v%d' % (BASE + 60 + 2 * p + hi)


def Cfull(p):
    return 'v[%d:%d]' % (BASE + 60 + 2 * p, BASE + 60 + 2 * p + 1)


VA, CA = 'v%d' % (BASE + 66), 'v%d' % (BASE + 67)
SG, ST = 's88', 's89'
COLB, VALB, QX, GBEG, GLAST = '%26', '%27', '%28', '%24', '%25'


def s1(q, goff, lines):
    """issue weights + gathers of group (g + goff) into set q (its columns are in C[q])"""
    lines += ['s_add_i32 %s, %s, %d' % (ST, SG, goff), 's_min_i32 %s, %s, %s' % (ST, ST, GLAST),
              'v_lshl_add_u32 %s, %s, 8, %s' % (VA, ST, VALB)]
    for e in range(4):
        lines.append('v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
                     % (Xa(q, e), Cw(q, e >> 1), QX, e & 1))
    lines.append('ds_read_b128 %s, %s' % (Wfull(q), VA))
    for e in range(4):
        lines.append('ds_read_b128 %s, %s' % (X(q, e), Xa(q, e)))


def s0(q, goff, lines):
    lines += ['s_add_i32 %s, %s, %d' % (ST, SG, goff), 's_min_i32 %s, %s, %s' % (ST, ST, GLAST),
              'v_lshl_add_u32 %s, %s, 7, %s' % (CA, ST, COLB), 'ds_read_b64 %s, %s' % (Cfull(q), CA)]


def fma(p, tile, lines):
    lo, hi = '%%%d' % (2 * tile), '%%%d' % (2 * tile + 1)
    for e in range(4):
        sel = 'op_sel_hi:[0,1,1]' if (e & 1) == 0 else 'op_sel:[1,0,0]'
        lines.append('v_pk_fma_f32 %s, %s, %s, %s %s' % (lo, W(p, e >> 1), X(p, e, 0), lo, sel))
        lines.append('v_pk_fma_f32 %s, %s, %s, %s %s' % (hi, W(p, e >> 1), X(p, e, 1), hi, sel))


def gen():
    L = []
    L.append('s_mov_b32 %s, %s' % (SG, GBEG))
    # prologue: columns of g, g+1, g+2; then S1(g), S0(g+3), S1(g+1), S0(g+4)
    for p in range(3):
        s0(p, p, L)
    L.append('s_waitcnt lgkmcnt(0)')
    s1(0, 0, L); s0(0, 3, L)
    s1(1, 1, L); s0(1, 4, L)
    for t in range(NT):
        for p in range(3):
            L.append('L_T%d_P%d_%%=:' % (t, p))
            L.append('s_cmp_ge_i32 %s, %%%d' % (SG, 16 + t))
            L.append('s_cbranch_scc1 L_T%d_P%d_%%=' % (t + 1, p))
            q = (p + 2) % 3
            s1(q, 2, L)
            s0(q, 5, L)
            L.append('s_waitcnt lgkmcnt(12)')
            fma(p, t, L)
            L.append('s_add_i32 %s, %s, 1' % (SG, SG))
        L.append('s_branch L_T%d_P0_%%=' % t)
    for p in range(3):
        L.append('L_T%d_P%d_%%=:' % (NT, p))
    L.append('s_waitcnt lgkmcnt(0)')
    return L


# ---- uniform-weight variant: no weight image; a tile's gathered rows are summed (v_pk_add_f32) and enter its accumulator once,
# acc = init + w * sum, when the stream leaves the tile. Operands: %0..%15 accumulator halves, %16..%23 tile ends, %24 first
# group, %25 last valid group (SGPR), %26 column base + 8 r, %27 q << 4, %28 = {w, w} (VGPR pair). Clobbers v[194:253].
UB = 194


def UX(p, e, half=None):
    b = UB + 16 * p + 4 * e
    if half is None:
        return 'v[%d:%d]' % (b, b + 3)
    return 'v[%d:%d]' % (b + 2 * half, b + 2 * half + 1)


def UXa(p, e):
    return 'v%d' % (UB + 16 * p + 4 * e)


def UCw(p, hi):
    return 'v%d' % (UB + 48 + 2 * p + hi)


def UCfull(p):
    return 'v[%d:%d]' % (UB + 48 + 2 * p, UB + 48 + 2 * p + 1)


UCA = 'v%d' % (UB + 54)
USUM = ['v[%d:%d]' % (UB + 56, UB + 57), 'v[%d:%d]' % (UB + 58, UB + 59)]
UQX, UWP = '%27', '%28'


MFMA_EXP = bool(os.environ.get('GCRNN_HOP_EXPERIMENT_MFMA'))     # timing experiment only (wrong results): bf16 state image (two 16-byte
# gathers per 4 entries: a lane fetches half of ONE neighbour's 32-byte row) summed on the matrix cores by a one-hot A operand
# (v_mfma_f32_16x16x32_bf16: D[feature][node] += sum over 2 entries) instead of 8 v_pk_add_f32. A operand = v[UB+8:UB+11] (free here).
UMA = 'v[%d:%d]' % (UB + 8, UB + 11)
USUM4 = 'v[%d:%d]' % (UB + 56, UB + 59)


def us1(q, goff, lines):
    if MFMA_EXP:
        for e in range(2):
            lines.append('v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
                         % (UXa(q, e), UCw(q, 0), UQX, e & 1))
        for e in range(2):
            lines.append('ds_read_b128 %s, %s' % (UX(q, e), UXa(q, e)))
        return
    for e in range(4):
        lines.append('v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
                     % (UXa(q, e), UCw(q, e >> 1), UQX, e & 1))
    for e in range(4):
        if os.environ.get('GCRNN_HOP_EXPERIMENT_B64'):      # timing experiment only (wrong results): what would 8-byte gathers of a bf16 state image buy?
            lines.append('ds_read_b64 %s, %s' % (UX(q, e, 0), UXa(q, e)))
        else:
            lines.append('ds_read_b128 %s, %s' % (UX(q, e), UXa(q, e)))


def us0(q, goff, lines):
    lines += ['s_add_i32 %s, %s, %d' % (ST, SG, goff), 's_min_i32 %s, %s, %s' % (ST, ST, GLAST),
              'v_lshl_add_u32 %s, %s, 7, %s' % (UCA, ST, COLB), 'ds_read_b64 %s, %s' % (UCfull(q), UCA)]


def gen_uniform():
    L = ['s_mov_b32 %s, %s' % (SG, GBEG)]
    if MFMA_EXP:
        for r in range(4):
            L.append('v_mov_b32 v%d, 0x3f803f80' % (UB + 8 + r))
    for r in range(4):
        L.append('v_mov_b32 v%d, 0' % (UB + 56 + r))
    for p in range(3):
        us0(p, p, L)
    L.append('s_waitcnt lgkmcnt(0)')
    us1(0, 0, L); us0(0, 3, L)
    us1(1, 1, L); us0(1, 4, L)
    for t in range(NT):
        for p in range(3):
            L.append('L_T%d_P%d_%%=:' % (t, p))
            L.append('s_cmp_ge_i32 %s, %%%d' % (SG, 16 + t))
            L.append('s_cbranch_scc1 L_X%d_P%d_%%=' % (t, p))
            q = (p + 2) % 3
            us1(q, 2, L)
            us0(q, 5, L)
            if MFMA_EXP:
                L.append('s_waitcnt lgkmcnt(6)')
                for e in range(2):
                    L.append('v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s' % (USUM4, UMA, UX(p, e), USUM4))
            else:
                L.append('s_waitcnt lgkmcnt(10)')
                for e in range(4):
                    L.append('v_pk_add_f32 %s, %s, %s' % (USUM[0], USUM[0], UX(p, e, 0)))
                    L.append('v_pk_add_f32 %s, %s, %s' % (USUM[1], USUM[1], UX(p, e, 1)))
            L.append('s_add_i32 %s, %s, 1' % (SG, SG))
        L.append('s_branch L_T%d_P0_%%=' % t)
        for p in range(3):                    # leaving tile t in phase p: acc_t += w * sum, sum = 0
            L.append('L_X%d_P%d_%%=:' % (t, p))
            L.append('v_pk_fma_f32 %%%d, %s, %s, %%%d' % (2 * t, UWP, USUM[0], 2 * t))
            L.append('v_pk_fma_f32 %%%d, %s, %s, %%%d' % (2 * t + 1, UWP, USUM[1], 2 * t + 1))
            for r in range(4):
                L.append('v_mov_b32 v%d, 0' % (UB + 56 + r))
            L.append('s_branch L_T%d_P%d_%%=' % (t + 1, p))
    for p in range(3):
        L.append('L_T%d_P%d_%%=:' % (NT, p))
    L.append('s_waitcnt lgkmcnt(0)')
    return L


# ---- uniform-weight variant on a bf16 state image (32-byte rows), summed on the matrix cores -----------------------------------
# A lane (slot r, quad q) gathers 16 bytes = features 8 (q & 1) .. + 7 of ONE neighbour: of entry (q >> 1) for the first MFMA of a
# trip, of entry 2 + (q >> 1) for the second. v_mfma_f32_16x16x32_bf16 with the one-hot A operand A[i][k] = (i == k mod 16) then adds
# both neighbours' rows to D[feature][slot], whose register layout (lane (r, q): features 4q .. 4q+3 of slot r) is the accumulators'.
# Column words per slot and trip: dword 0 = addr(entry 0) | addr(entry 2) << 16, dword 1 = addr(entry 1) | addr(entry 3) << 16.
# Operands: %0..%15 accumulator halves, %16..%23 tile ends, %24 first group, %25 last valid group (SGPR), %26 column base + 8 r,
# (+ 4 (q >> 1): a lane reads only its own dword), %27 = (q & 1) << 4, %28 = {w, w}, %29 = the lane's A operand (4 VGPRs).
# Clobbers v[218:253] (UB16 ..), s[88:90], scc, vcc.
VD = int(os.environ.get('GCRNN_HOP16_DEPTH', '3'))      # groups in flight (register sets): 3 = as the fp32 streams; the two 16-byte gathers of a
# set leave room for 5 sets inside v[194:253] -- the trips are short now, so the LDS latency needs more of them in flight


UB16 = 218          # the bf16-image stream's own register window v[218:253]: 36 registers (5 gather sets would need v[UB16:UB16+39] + 12: D <= 3 here)
SPA = 212           # sparse variant (v_smfmac): compressed one-hot A operand v[212:215], index register v216, scratch v217 -- window v[212:253]


def VX(p, e):
    b = UB16 + 8 * p + 4 * e
    return 'v[%d:%d]' % (b, b + 3)


def VXa(p, e):
    return 'v%d' % (UB16 + 8 * p + 4 * e)


NOCLAMP = False     # (gen_uniform16(sums=True) sets it: see vs0p)
IMMOFF = None
IMGOFF = 0          # LDS byte offset of the hop image the gathers read (the summing variants have a second image: gen_uniform16(img_off=..))
VCOFF = 24          # column-word slots behind three gather sets (the summing variants pack their window: 8 D)


def VC(p):
    return 'v%d' % (UB16 + VCOFF + p)


VCA = 'v%d' % (UB16 + 31)
VP5, VPCL = 'v%d' % (UB16 + 29), 'v%d' % (UB16 + 30)               # running column pointer of group g + 2 D - 1, and its clamp (last valid group)
VSUMB = [UB16 + 32]                                               # the D accumulator (4 VGPRs)
VSUM4 = ['v[%d:%d]' % (b_, b_ + 3) for b_ in VSUMB]
VSUMH = [['v[%d:%d]' % (b_, b_ + 1), 'v[%d:%d]' % (b_ + 2, b_ + 3)] for b_ in VSUMB]


def vs1(q, lines):
    X = os.environ.get('GCRNN_HOP16_EXPERIMENT_TRIP', '')      # timing experiments (wrong results): which of a trip's instructions cost its time?
    for e in range(2):
        if 'plainxor' in X:                                    # a plain VALU op instead of the SDWA one (in-range, aligned, wrong addresses)
            lines.append('v_and_b32 %s, 0x7ff0, %s' % (VXa(q, e), VC(q)))
        elif 'noxor' not in X:
            lines.append('v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD'
                         % (VXa(q, e), VC(q), UQX, e))
    for e in range(2):
        if 'nogather' not in X:
            lines.append('ds_read_b128 %s, %s' % (VX(q, e), (VXa(q, e) if 'noxor' not in X else VPCL)) + (' offset:%d' % IMGOFF if IMGOFF else ''))


def vs0(q, goff, lines):      # the lane's OWN column dword of group g + goff (column base operand = base + 8 r + 4 (q >> 1))
    lines += ['s_add_i32 %s, %s, %d' % (ST, SG, goff), 's_min_i32 %s, %s, %s' % (ST, ST, GLAST),
              'v_lshl_add_u32 %s, %s, 7, %s' % (VCA, ST, COLB), 'ds_read_b32 %s, %s' % (VC(q), VCA)]


def vs0p(q, lines):            # steady state: the same read through the running pointer -- no scalar arithmetic in the trip
    if 'nocol' in os.environ.get('GCRNN_HOP16_EXPERIMENT_TRIP', ''):
        return
    if NOCLAMP:
        # (summing variants) no clamp of the running column pointer: past its last group a wave reads the next wave's column words, the last
        # wave GCRNN_HOP_COLUMN_PAD bytes of zeros the kernels keep behind the column image -- valid, aligned gather addresses, unused sums.
        # IMMOFF (position of the trip in the unrolled loop body, or None): the group's offset rides in the instruction, the pointer moves
        # once per loop iteration (and by the positions skipped at a tile exit)
        if IMMOFF is None:
            lines += ['ds_read_b32 %s, %s' % (VC(q), VP5), 'v_add_u32 %s, 128, %s' % (VP5, VP5)]
        else:
            lines += ['ds_read_b32 %s, %s' % (VC(q), VP5) + (' offset:%d' % (128 * IMMOFF) if IMMOFF else '')]
        return
    lines += ['v_min_u32 %s, %s, %s' % (VCA, VP5, VPCL), 'ds_read_b32 %s, %s' % (VC(q), VCA), 'v_add_u32 %s, 128, %s' % (VP5, VP5)]


IMAGE_B_OFFSET = 33792      # second hop image of the sequence-resident kernel: behind image A / the transposed tile / bias / flags (gcrnn_fused_seq.h LDS map)
SUMS_D = int(os.environ.get('GCRNN_HOP16_SUMS_DEPTH', '2'))      # groups in flight of the summing variants (2 and 3 measured equal: profiles/r03_hop16_depth_ab.txt)
SUMS_UB16 = (254 - (9 * SUMS_D + 3)) & ~1                        # their packed register window v[SUMS_UB16:253] (even base: the 8-register B tuples) ...
SUMS_SPA = SUMS_UB16 - 6                                         # ... sparse: v[SUMS_SPA:253]


def gen_uniform16(sparse=False, sums=False, img_off=0):
    """sums: the stream only SUMS -- tile t's gathered rows go straight into its own accumulator tuple (operand %t, zeroed by the caller),
    the caller applies acc = init + w * sum after the block. A tile exit then only switches tiles: no matrix-core -> VALU wait states, no
    packed FMAs, no re-zeroing per tile (timing experiment GCRNN_HOP16_EXPERIMENT_CHEAP_EXIT: the exits cost 6.5 % of the launch). Operands:
    %0..%7 the tiles' D tuples (4 VGPRs each), %8..%15 tile ends, %16 first group, %17 last valid group, %18 column base, %19 = (q & 1) << 4,
    dense variant: %20 = the lane's A operand.
    sparse: ONE v_smfmac_f32_16x16x64_bf16 per trip instead of two v_mfma_f32_16x16x32_bf16 -- the 2:4-sparse instruction takes a dense
    B of K = 64 = [first gather | second gather] (probed: tools/probes/smfmac_probe.hip, profiles/r03_smfmac_layout_probe.txt: B elements
    0..7 of lane (j, kg) are k = 8 kg + e, elements 8..15 are k = 32 + 8 kg + e, i.e. the two dense operands stacked along K) and a
    compressed A: lane (i, sg) holds the kept values of dense k = 16 sg .. 16 sg + 15 (4 groups of 4, two kept per group, 2-bit positions
    in the index register). The one-hot A[i][k] = (i == k mod 16) has ONE non-zero per lane: group i >> 2, position i & 3. Both are built
    from the lane id in the block's prologue (the asm statement is at its 30-operand limit: the dense variant's A operand %29 goes away)."""
    D = SUMS_D if sums else VD
    assert 2 <= D <= 3      # (the register window holds three gather sets; 2 / 3 / 4 / 5 sets measured equal)
    SC = 's90'                                # (group - tile end) of the current tile, counted up: the carry of its increment ends the tile
    global COLB, GBEG, GLAST, UQX, UB16, SPA, VCOFF, VCA, VP5, VPCL, IMGOFF, NOCLAMP, IMMOFF
    saved = (COLB, GBEG, GLAST, UQX, UB16, SPA, VCOFF, VCA, VP5, VPCL)
    IMGOFF = img_off
    NOCLAMP = sums and not os.environ.get('GCRNN_HOP16_CLAMP')
    PIPEADR = sums and sparse and NOCLAMP and bool(os.environ.get('GCRNN_HOP16_PIPELINED_ADDRESSES'))      # A/B: 120.3-120.5k vs 120.3-121.0k seq/s, no gain (profiles/r03_hop16_pipelined_addresses_ab.txt) -- off
    ADR = None
    TEND0, AOP = 16, '%29'
    if sums:
        GBEG, GLAST, COLB, UQX, TEND0, AOP = '%16', '%17', '%18', '%19', 8, '%20'
        # packed window ending at v253: D gather sets, D column words, pointer, clamp, address (sparse: + A operand, index, scratch below)
        UB16, SPA, VCOFF = SUMS_UB16, SUMS_SPA, 8 * D
        VP5, VPCL, VCA = 'v%d' % (UB16 + 9 * D), 'v%d' % (UB16 + 9 * D + 1), 'v%d' % (UB16 + 9 * D + 2)
        ADR = [VPCL, 'v%d' % (UB16 + 9 * D + 3)]                # (pipelined gather addresses: the clamp register is free without the clamp; + the window's last register)
        assert UB16 + 9 * D + 3 <= 253
    L = ['s_mov_b32 %s, %s' % (SG, GBEG)]
    if sums:                                  # the first column words are requested before the A operand is built: its ~25 VALU instructions cover their latency
        for p in range(D):
            vs0(p, p, L)
    if sparse:
        A0, IDX, TMP = SPA, SPA + 4, SPA + 5
        vM, vP, vVal, vNib = 'v%d' % A0, 'v%d' % (A0 + 1), 'v%d' % (A0 + 2), 'v%d' % (A0 + 3)      # temporaries first live in the A registers
        L += ['v_mbcnt_lo_u32_b32 v%d, -1, 0' % TMP, 'v_mbcnt_hi_u32_b32 v%d, -1, v%d' % (TMP, TMP),
              'v_and_b32 v%d, 15, v%d' % (TMP, TMP),                      # i = lane & 15
              'v_and_b32 v%d, 3, v%d' % (IDX, TMP),                       # pos = i & 3
              'v_lshrrev_b32 v%d, 2, v%d' % (TMP, TMP),                   # m = i >> 2: the lane's group with the non-zero
              'v_cmp_eq_u32 vcc, 3, v%d' % IDX,
              'v_or_b32 v%d, 12, v%d' % (IDX, IDX),                       # nibble: first kept at pos, second kept at 3 ...
              'v_cndmask_b32 v%d, v%d, 12, vcc' % (IDX, IDX),              # ... pos == 3: first kept at 0 (a zero), second kept at 3 (the one)
              'v_mov_b32 v%d, 0x3f80' % (A0 + 3),
              'v_mov_b32 v%d, 0x3f800000' % (A0 + 2),
              'v_cndmask_b32 v%d, v%d, v%d, vcc' % (A0 + 3, A0 + 3, A0 + 2),   # the group's register: 1.0 in its low (first kept) or high (second kept) half
              'v_xor_b32 v%d, 4, v%d' % (IDX, IDX),                        # index = 0x4444 with nibble m replaced
              'v_lshlrev_b32 v%d, 2, v%d' % (A0 + 2, TMP),
              'v_lshlrev_b32 v%d, v%d, v%d' % (IDX, A0 + 2, IDX),
              'v_xor_b32 v%d, 0x4444, v%d' % (IDX, IDX),
              'v_mov_b32 v%d, v%d' % (A0 + 2, A0 + 3)]                     # vVal
        # A registers: group g' holds vVal iff g' == m (set 3, 1, 0 first: they do not hold vVal; then 2)
        for g in (3, 1, 0):
            L += ['v_cmp_eq_u32 vcc, %d, v%d' % (g, TMP), 'v_cndmask_b32 v%d, 0, v%d, vcc' % (A0 + g, A0 + 2)]
        L += ['v_cmp_eq_u32 vcc, 2, v%d' % TMP, 'v_cndmask_b32 v%d, 0, v%d, vcc' % (A0 + 2, A0 + 2)]
    if not sums:
        for r in range(4):
            L.append('v_mov_b32 v%d, 0' % (VSUMB[0] + r))
    if not sums:
        for p in range(D):
            vs0(p, p, L)
    L.append('s_waitcnt lgkmcnt(0)')
    for p in range(D - 1):                    # gathers of groups 0 .. D-2; their column slots take groups D .. 2D-2
        vs1(p, L); vs0(p, D + p, L)
    L += ['s_add_i32 %s, %s, %d' % (ST, SG, 2 * D - 1), 'v_lshl_add_u32 %s, %s, 7, %s' % (VP5, ST, COLB),
          'v_lshl_add_u32 %s, %s, 7, %s' % (VPCL, GLAST, COLB)]
    if PIPEADR:
        # (the clamp register doubles as an address register: nothing reads the clamp without the clamp) the addresses of group D - 1, for trip 0
        for e in range(2):
            L.append('v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD' % (ADR[e], VC(D - 1), UQX, e))
    L += ['s_sub_u32 %s, %s, %%%d' % (SC, GBEG, TEND0), 's_cmp_eq_u32 %s, 0' % SC]
    R = int(os.environ.get('GCRNN_HOP16_UNROLL', '2'))           # the D phases are laid out R times before the loop branches back
    for t in range(NT):
        IMM = NOCLAMP and bool(os.environ.get('GCRNN_HOP16_IMMEDIATE_OFFSETS'))      # column reads with immediate group offsets, pointer moved once per loop iteration: A/B 118.8-120.1k vs 119.3-119.8k seq/s, no gain (profiles/r03_hop16_immediate_offsets_ab.txt) -- off
        for pp in range(D * R):
            p = pp % D
            L.append('L_T%d_P%d_%%=:' % (t, pp))
            L.append('s_cbranch_scc1 L_X%d_P%d_%%=' % (t, pp if IMM else p))
            IMMOFF = pp if IMM else None
            for _ in range(int(os.environ.get('GCRNN_HOP16_EXPERIMENT_EXTRA_BRANCHES', '0'))):      # (timing experiment: what does a not-taken branch cost? ~6 cycles)
                L.append('s_cbranch_scc1 L_X%d_P%d_%%=' % (t, p))
            for _ in range(int(os.environ.get('GCRNN_HOP16_EXPERIMENT_EXTRA_SALU', '0'))):          # (... and a scalar instruction? ~3 cycles)
                L.append('s_mov_b32 s89, s88')
            if sums and not os.environ.get('GCRNN_HOP16_COUNTER_LAST'):
                # the counter's increment right behind the branch that consumed its carry: the next trip's branch then finds SCC long settled
                # (at the end of the trip the scalar result -> branch latency was exposed every trip; nothing in between writes SCC)
                L.append('s_add_u32 %s, %s, 1' % (SC, SC))
            q = (p + D - 1) % D
            if PIPEADR:
                # gathers from addresses formed at the END of the previous trip (no address instruction right in front of its gather: in-order
                # issue waits for the VALU result there -- what the clamp in front of the column read cost too); this trip forms the next one's
                L += ['ds_read_b128 %s, %s' % (VX(q, 0), ADR[0]) + (' offset:%d' % IMGOFF if IMGOFF else ''),
                      'ds_read_b128 %s, %s' % (VX(q, 1), ADR[1]) + (' offset:%d' % IMGOFF if IMGOFF else '')]
                vs0p(q, L)
                L.append('s_waitcnt lgkmcnt(%d)' % (3 * (D - 1)))
                L.append('v_smfmac_f32_16x16x64_bf16 %s, v[%d:%d], v[%d:%d], v%d' % ('%%%d' % t, SPA, SPA + 3, UB16 + 8 * p, UB16 + 8 * p + 7, SPA + 4))
                for e in range(2):
                    L.append('v_xor_b32_sdwa %s, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD' % (ADR[e], VC(p), UQX, e))
                continue
            if sums and os.environ.get('GCRNN_HOP16_TRIP_ORDER_SPREAD') and not os.environ.get('GCRNN_HOP16_EXPERIMENT_TRIP'):      # (A/B: 116.3-116.5k vs 116.0-117.4k seq/s, no gain: profiles/r03_hop16_trip_order_ab.txt)
                # the same seven instructions with no instruction right behind the one it depends on (in-order issue: a gather behind its
                # address XOR, the column read behind its clamp, each waited for the VALU result): clamp, XORs, column read, pointer, gathers
                a_, b_ = [], []
                vs1(q, a_); vs0p(q, b_)                           # a_ = xor, xor, gather, gather; b_ = v_min, ds_read_b32, v_add
                L += [b_[0], a_[0], a_[1], b_[1], b_[2], a_[2], a_[3]]
            else:
                vs1(q, L)                                         # (the set's registers were B operands of the PREVIOUS trip's MFMAs: read long ago)
                vs0p(q, L)
            if not os.environ.get('GCRNN_HOP16_EXPERIMENT_NO_WAIT'):      # (timing experiments, wrong results: where does a trip stall?)
                L.append('s_waitcnt lgkmcnt(%d)' % (3 * (D - 1)))
            acc = ('%%%d' % t) if sums else VSUM4[0]
            if sparse and not os.environ.get('GCRNN_HOP16_EXPERIMENT_NO_MFMA'):
                L.append('v_smfmac_f32_16x16x64_bf16 %s, v[%d:%d], v[%d:%d], v%d' % (acc, SPA, SPA + 3, UB16 + 8 * p, UB16 + 8 * p + 7, SPA + 4))
            for e in range(0 if sparse else (1 if os.environ.get('GCRNN_HOP16_EXPERIMENT_ONE_MFMA') else 2)):      # (timing experiment, wrong results: what would ONE
                # matrix instruction per four entries -- a 2:4-sparse v_smfmac_f32_16x16x64_bf16 with the one-hot A -- buy?)
                # one accumulator (two, so that an MFMA never waits for its predecessor: slower, the exits pay more)
                L.append('v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s' % (acc, AOP, VX(p, e), acc))
            if not sums or os.environ.get('GCRNN_HOP16_COUNTER_LAST'):
                L.append('s_add_u32 %s, %s, 1' % (SC, SC))        # SCC = carry = this was the tile's last group
        IMMOFF = None
        if IMM:
            L.append('v_add_u32 %s, %d, %s' % (VP5, 128 * D * R, VP5))
        L.append('s_branch L_T%d_P0_%%=' % t)
        for px in range(D * R if IMM else D):                    # leaving tile t in phase p: acc_t += w * sum, sum = 0
            p = px % D
            L.append('L_X%d_P%d_%%=:' % (t, px))
            if IMM and px >= D:                                   # positions of the loop body this exit skips: the next tile enters at position p
                L.append('v_add_u32 %s, %d, %s' % (VP5, 128 * (px - p), VP5))
            if not sums and not os.environ.get('GCRNN_HOP16_EXPERIMENT_CHEAP_EXIT'):      # (timing experiment, wrong results: what do the tile exits cost?)
                L += ['s_nop 15', 's_nop 7'] if sparse else ['s_nop 11']       # matrix-core result -> VALU read: 11 wait states after the 8-pass dense MFMA; the sparse one is given the 16-pass distance
                L.append('v_pk_fma_f32 %%%d, %s, %s, %%%d' % (2 * t, UWP, VSUMH[0][0], 2 * t))
                L.append('v_pk_fma_f32 %%%d, %s, %s, %%%d' % (2 * t + 1, UWP, VSUMH[0][1], 2 * t + 1))
                for r in range(4):
                    L.append('v_mov_b32 v%d, 0' % (VSUMB[0] + r))
            if t + 1 < NT:                                        # the next tile runs from this tile's end to its own
                L += ['s_sub_u32 %s, %%%d, %%%d' % (SC, TEND0 + t, TEND0 + t + 1), 's_cmp_eq_u32 %s, 0' % SC]
            if not sums:
                L.append('s_nop 1')                               # VALU write -> matrix-core read of the accumulator
            L.append('s_branch L_T%d_P%d_%%=' % (t + 1, p))
    for p in range(D):
        L.append('L_T%d_P%d_%%=:' % (NT, p))
    if sums:                                                      # matrix-core results -> the caller's VALU reads: once per block (16-pass distance),
        L += ['s_nop 15', 's_nop 7']                              # while the last (unused, clamped) gathers drain
    L.append('s_waitcnt lgkmcnt(0)')
    COLB, GBEG, GLAST, UQX, UB16, SPA, VCOFF, VCA, VP5, VPCL = saved
    IMGOFF = 0
    NOCLAMP = False
    return L


# ---- two-plane ("wide") summing stream of the 32-feature-chunk sequence-resident kernel (gcrnn_fused_seq32.h) ---------------------
# The hop image is TWO bf16 planes of 32-byte rows (plane h = output features half h of the chunk), plane 1 WIDE_PLANE bytes behind
# plane 0, both addressed by the SAME column words (the img16 plan, graph.fused_plan_img16): one column read and two address XORs per
# trip serve four 16-byte gathers (plane 1 through the instruction's offset field) and two v_smfmac_f32_16x16x64_bf16 -- the per-trip
# instruction overhead and the per-stream fixed cost are paid once for twice the bytes.
# Operands: %0..%15 the tiles' D tuples (tile t: %(2t) = half 0, %(2t+1) = half 1; 4 VGPRs each, zeroed by the caller), %16..%23 tile
# ends, %24 first group, %25 last valid group, %26 column base (+ 8 r + 4 (q >> 1)), %27 = (q & 1) << 4.
# A set (16 VGPRs) = [plane-1 gather of entry pair word 0 | plane-1 word 1 | plane-0 word 0 | plane-0 word 1]; the address registers are
# the first registers of the plane-0 tuples, so the plane-1 gathers (which read them) are issued BEFORE the plane-0 gathers overwrite them.
WIDE_PLANE = IMAGE_B_OFFSET
WIDE_D = int(os.environ.get('GCRNN_WIDE_DEPTH', '2'))
WIDE_EXTRA = int(os.environ.get('GCRNN_WIDE_EXTRA_CLOBBER', '0'))      # (experiment: registers a stream with in-stream tap MFMAs would need for half a tap's fragments -- do the kernels still allocate?)
WIDE_WIN = 16 * WIDE_D + WIDE_D + 2 + 6 + WIDE_EXTRA  # gather sets, column words, pointer, prologue address; sparse A (4) + index + scratch
WIDE_BASE = (254 - WIDE_WIN) & ~1


def gen_wide32():
    D = WIDE_D
    assert 2 <= D <= 3
    A0 = WIDE_BASE
    IDX, TMP = A0 + 4, A0 + 5
    UB = A0 + 6
    assert UB % 2 == 0 and UB + 16 * D + D + 2 <= 254
    VCw = lambda p: 'v%d' % (UB + 16 * D + p)
    VP, VA = 'v%d' % (UB + 16 * D + D), 'v%d' % (UB + 16 * D + D + 1)
    SGr, STr, SCr = 's88', 's89', 's90'
    GB, GL, CB, QX, TE = '%24', '%25', '%26', '%27', 16
    Y = lambda p, e: UB + 16 * p + 4 * e              # plane-1 tuple of word e
    Xr = lambda p, e: UB + 16 * p + 8 + 4 * e         # plane-0 tuple of word e (its first register holds the gather address)
    tup = lambda b: 'v[%d:%d]' % (b, b + 3)

    def col_clamped(p, goff, L):
        L += ['s_add_i32 %s, %s, %d' % (STr, SGr, goff), 's_min_i32 %s, %s, %s' % (STr, STr, GL),
              'v_lshl_add_u32 %s, %s, 7, %s' % (VA, STr, CB), 'ds_read_b32 %s, %s' % (VCw(p), VA)]

    def gathers(q, L):
        for e in range(2):
            L.append('v_xor_b32_sdwa v%d, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD' % (Xr(q, e), VCw(q), QX, e))
        for e in range(2):
            L.append('ds_read_b128 %s, v%d offset:%d' % (tup(Y(q, e)), Xr(q, e), WIDE_PLANE))
        for e in range(2):
            L.append('ds_read_b128 %s, v%d' % (tup(Xr(q, e)), Xr(q, e)))

    L = ['s_mov_b32 %s, %s' % (SGr, GB)]
    for p in range(D):
        col_clamped(p, p, L)
    # the compressed one-hot A operand of v_smfmac (one non-zero per lane: group (lane & 15) >> 2, position lane & 3) and its index register
    L += ['v_mbcnt_lo_u32_b32 v%d, -1, 0' % TMP, 'v_mbcnt_hi_u32_b32 v%d, -1, v%d' % (TMP, TMP),
          'v_and_b32 v%d, 15, v%d' % (TMP, TMP),
          'v_and_b32 v%d, 3, v%d' % (IDX, TMP),
          'v_lshrrev_b32 v%d, 2, v%d' % (TMP, TMP),
          'v_cmp_eq_u32 vcc, 3, v%d' % IDX,
          'v_or_b32 v%d, 12, v%d' % (IDX, IDX),
          'v_cndmask_b32 v%d, v%d, 12, vcc' % (IDX, IDX),
          'v_mov_b32 v%d, 0x3f80' % (A0 + 3),
          'v_mov_b32 v%d, 0x3f800000' % (A0 + 2),
          'v_cndmask_b32 v%d, v%d, v%d, vcc' % (A0 + 3, A0 + 3, A0 + 2),
          'v_xor_b32 v%d, 4, v%d' % (IDX, IDX),
          'v_lshlrev_b32 v%d, 2, v%d' % (A0 + 2, TMP),
          'v_lshlrev_b32 v%d, v%d, v%d' % (IDX, A0 + 2, IDX),
          'v_xor_b32 v%d, 0x4444, v%d' % (IDX, IDX),
          'v_mov_b32 v%d, v%d' % (A0 + 2, A0 + 3)]
    for g in (3, 1, 0):
        L += ['v_cmp_eq_u32 vcc, %d, v%d' % (g, TMP), 'v_cndmask_b32 v%d, 0, v%d, vcc' % (A0 + g, A0 + 2)]
    L += ['v_cmp_eq_u32 vcc, 2, v%d' % TMP, 'v_cndmask_b32 v%d, 0, v%d, vcc' % (A0 + 2, A0 + 2)]
    L.append('s_waitcnt lgkmcnt(0)')
    for p in range(D - 1):
        gathers(p, L); col_clamped(p, D + p, L)
    L += ['s_add_i32 %s, %s, %d' % (STr, SGr, 2 * D - 1), 'v_lshl_add_u32 %s, %s, 7, %s' % (VP, STr, CB)]
    L += ['s_sub_u32 %s, %s, %%%d' % (SCr, GB, TE), 's_cmp_eq_u32 %s, 0' % SCr]
    R = 2
    Aop, Iop = 'v[%d:%d]' % (A0, A0 + 3), 'v%d' % IDX
    for t in range(NT):
        for pp in range(D * R):
            p = pp % D
            q = (p + D - 1) % D
            L.append('L_T%d_P%d_%%=:' % (t, pp))
            L.append('s_cbranch_scc1 L_X%d_P%d_%%=' % (t, p))
            L.append('s_add_u32 %s, %s, 1' % (SCr, SCr))
            gathers(q, L)
            L += ['ds_read_b32 %s, %s' % (VCw(q), VP), 'v_add_u32 %s, 128, %s' % (VP, VP)]
            L.append('s_waitcnt lgkmcnt(%d)' % (5 * (D - 1)))
            L.append('v_smfmac_f32_16x16x64_bf16 %%%d, %s, v[%d:%d], %s' % (2 * t + 1, Aop, Y(p, 0), Y(p, 0) + 7, Iop))
            L.append('v_smfmac_f32_16x16x64_bf16 %%%d, %s, v[%d:%d], %s' % (2 * t, Aop, Xr(p, 0), Xr(p, 0) + 7, Iop))
        L.append('s_branch L_T%d_P0_%%=' % t)
        for p in range(D):
            L.append('L_X%d_P%d_%%=:' % (t, p))
            if t + 1 < NT:
                L += ['s_sub_u32 %s, %%%d, %%%d' % (SCr, TE + t, TE + t + 1), 's_cmp_eq_u32 %s, 0' % SCr]
            L.append('s_branch L_T%d_P%d_%%=' % (t + 1, p))
    for p in range(D):
        L.append('L_T%d_P%d_%%=:' % (NT, p))
    L += ['s_nop 15', 's_nop 7', 's_waitcnt lgkmcnt(0)']
    return L


# ---- round 5: the WHOLE hop of the wide kernel as one hand-allocated block: gather stream + the hop's tap MFMAs + (last hop of a step) the
# next step's operand requests (gcrnn_fused_seq32p.h). Round 4's stamps: a wave ran stream THEN taps (or taps then stream), so the LDS
# array idled at both ends of every hop, and the step boundary's 192 KB of operand requests (170-190 of ~1,100 units per step) could not
# start before the last tap had read the old operand. Here every register is pinned ("{v[a:b]}" constraints on 32-register tuples):
#   v[0:127]   operand: k-step s, tile i = v[32 s + 4 i .. + 3]          (B fragments of the tap MFMAs)
#   v[128:191] accumulators: half h, tile i = v[128 + 32 h + 4 i .. + 3]  (D of the stream's v_smfmac AND of the taps)
#   v[192:253] the block's own: sparse A operand (4) + index + scratch | two gather sets (32) | column words (2) | pointer, address |
#              two weight-fragment buffers (8) | tile node ids (8) | column base, q << 4, fragment base, 16 q
# Tap MFMAs: the hop's 2 KS weight fragments (h, s) are issued one per TILE EXIT of the stream (8 exits per wave and hop, each passed exactly
# once whatever the trip counts): exit e reads fragment e + 1 from LDS, issues fragment e's 8 MFMAs (one per tile; the tile the stream has
# just left LAST, the one it enters next FIRST) while the next tile's first gathers are in flight, then waits for everything it has issued.
# Fragments are ordered k-step-major, input k-steps first, the state k-step that is handed over in registers last: in the LAST hop of a step
# k-step s of the operand is dead behind the exit that issues its half-1 fragment, and the requests of the NEXT step's k-step s (8
# buffer_load_dwordx4 per wave, straight into the operand registers) follow it, `lpe` per exit -- dribbled through the stream instead of
# 24 per wave at once in the epilogue (a wave blocked issuing loads issues nothing else).
# Operands: %0 %1 = accumulator tuples (in/out); loads variant: %2 .. = the KS operand tuples (in/out); then the scalars: 8 tile ends, first
# group, last valid group, LDS address of the column image, LDS address of the hop's tap in the weight fragments; loads variant: buffer
# resources of h_t / x_{t+1} (128-bit scalars), their sequence offsets, LDS address of this wave's slot table (8 tiles x 16 words). Plain
# variant: the KS operand tuples follow as inputs (never named by number: every vector register is named by its pinned position).
def frag_order(HS, XS):
    ks = list(range(HS, HS + XS)) + list(range(0, HS - 1)) + [HS - 1]
    return [(s_, h) for s_ in ks for h in (0, 1)]


def gen_wide32_taps(HS, XS, loads=False, lpe=4, fpe=1, depth=None, taps=True):
    """Registers (62: v192..v253): sparse A operand (4) + its index | D gather sets of 16 | D column words | running column pointer | q << 4 |
    ONE weight-fragment buffer (4) whose registers double as the prologue's scratch (lane id, addresses) and, at every exit, as the address of
    the next fragment's read (re-derived from the lane id: nothing fragment-related lives across the trips) | loads variant: 8 tile node
    ids, 16 q, an address register. D = 3 fits the plain variant exactly; the loads variant (one hop of a step) runs two-deep."""
    KS = HS + XS
    D = depth if depth is not None else (2 if loads else int(os.environ.get('GCRNN_P_DEPTH', '2')))
    # exit style: 'A' = two fragment buffers, the next fragment's read IN FRONT of the MFMAs and everything waited for behind them (the first
    # form; needs D == 2); 'B' = one buffer, the read behind the MFMAs, a wait only when the tile was shorter than D - 1 trips
    # 'C' = the tap MFMAs in groups of GSZ, one group per TRIP (a subroutine the trip calls between issuing its gathers and waiting for the
    # previous ones; groups of equal byte size, addressed base + index * size), the rest behind the stream -- no bursts: a wave's partner on
    # the SIMD never finds the matrix pipe taken for 8 MFMAs in a row; two fragment buffers (D == 2)
    style = os.environ.get('GCRNN_P_EXIT', 'A' if D == 2 else 'B')
    if not taps:
        style = 'B'
    if loads and style == 'C':
        style = 'A'
    GSZ = int(os.environ.get('GCRNN_P_GROUP', '3'))
    assert style == 'B' or D == 2
    FO = frag_order(HS, XS) if taps else []      # taps=False: the stream alone (the split form: a wave's tap MFMAs run before or behind its stream)
    NF = len(FO)
    assert NF <= NT * fpe and 2 <= D <= 3
    nxt = [192]

    def alloc(n, even=False):
        if even and nxt[0] % 2:
            nxt[0] += 1
        r = nxt[0]
        nxt[0] += n
        return r
    A0 = alloc(4, True)
    UB = alloc(16 * D, True)
    WF0 = alloc(4, True)
    WF1 = alloc(4, True) if style in ('A', 'C') else WF0
    WFBn = alloc(1) if style in ('A', 'C') else None
    IDX = alloc(1)
    VCW0 = alloc(D)
    VPn, QXn = alloc(1), alloc(1)
    TMP, VAn, VCBn = WF0 + 1, WF0 + 2, WF0 + 3      # prologue scratch inside the fragment buffer (the first fragment is read behind their last use)
    if loads:
        ND0 = alloc(8)
        Q16n, LAD = alloc(1), alloc(1)
    assert nxt[0] <= 254, nxt[0]
    VCw = lambda p: 'v%d' % (VCW0 + p)
    VP, VA, VCB, QX = 'v%d' % VPn, 'v%d' % VAn, 'v%d' % VCBn, 'v%d' % QXn
    OP = lambda s_, i: 32 * s_ + 4 * i
    ACC = lambda h, i: 128 + 32 * h + 4 * i
    SGr, STr, SCr = 's88', 's89', 's90'
    SB = 2 + KS if loads else 2          # first scalar operand: behind the in/out tuples (plain variant: the operand tuples are inputs, listed last)
    TE = SB
    GB, GL, COLS, WOFS = ('%%%d' % (SB + 8 + n) for n in range(4))
    RH, RX, SOH, SOX, SLOT = ('%%%d' % (SB + 12 + n) for n in range(5))
    Y = lambda p, e: UB + 16 * p + 4 * e
    Xr = lambda p, e: UB + 16 * p + 8 + 4 * e
    tup = lambda b_: 'v[%d:%d]' % (b_, b_ + 3)
    foff = lambda f: (FO[f][1] * KS + FO[f][0]) * 1024
    WFt = tup(WF0)

    def col_clamped(p, goff, L):
        L += ['s_add_i32 %s, %s, %d' % (STr, SGr, goff), 's_min_i32 %s, %s, %s' % (STr, STr, GL),
              'v_lshl_add_u32 %s, %s, 7, %s' % (VA, STr, VCB), 'ds_read_b32 %s, %s' % (VCw(p), VA)]

    def gathers(q, L):
        for e in range(2):
            L.append('v_xor_b32_sdwa v%d, %s, %s dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_%d src1_sel:DWORD' % (Xr(q, e), VCw(q), QX, e))
        for e in range(2):
            L.append('ds_read_b128 %s, v%d offset:%d' % (tup(Y(q, e)), Xr(q, e), WIDE_PLANE))
        for e in range(2):
            L.append('ds_read_b128 %s, v%d' % (tup(Xr(q, e)), Xr(q, e)))

    def frag_read(f, L):
        """this lane's 16 bytes of weight fragment f -> the fragment buffer (address = lane * 16 + the tap's place, formed in the buffer's first register)"""
        L += ['v_mbcnt_lo_u32_b32 v%d, -1, 0' % WF0, 'v_mbcnt_hi_u32_b32 v%d, -1, v%d' % (WF0, WF0),
              'v_lshlrev_b32 v%d, 4, v%d' % (WF0, WF0), 'v_add_u32 v%d, %s, v%d' % (WF0, WOFS, WF0),
              'ds_read_b128 %s, v%d offset:%d' % (WFt, WF0, foff(f))]

    L = ['s_mov_b32 %s, %s' % (SGr, GB)]
    # lane id -> q << 4 (the half a lane gathers), this lane's column dword
    L += ['v_mbcnt_lo_u32_b32 v%d, -1, 0' % TMP, 'v_mbcnt_hi_u32_b32 v%d, -1, v%d' % (TMP, TMP),
          'v_lshrrev_b32 %s, 4, v%d' % (VA, TMP),
          'v_and_b32 %s, 1, %s' % (QX, VA), 'v_lshlrev_b32 %s, 4, %s' % (QX, QX)]
    if loads:
        L.append('v_lshlrev_b32 v%d, 4, %s' % (Q16n, VA))
    L += ['v_lshrrev_b32 %s, 1, %s' % (VA, VA), 'v_lshlrev_b32 %s, 2, %s' % (VA, VA),
          'v_and_b32 %s, 15, v%d' % (VP, TMP),
          'v_lshl_add_u32 %s, %s, 3, %s' % (VCB, VP, VA), 'v_add_u32 %s, %s, %s' % (VCB, COLS, VCB)]
    if loads:
        L += ['v_lshlrev_b32 %s, 2, %s' % (VA, VP), 'v_add_u32 %s, %s, %s' % (VA, SLOT, VA)]
        for i in range(NT):
            L.append('ds_read_b32 v%d, %s offset:%d' % (ND0 + i, VA, 64 * i))
    for p in range(D):
        col_clamped(p, p, L)
    # the compressed one-hot A operand of v_smfmac (one non-zero per lane: group (lane & 15) >> 2, position lane & 3) and its index register
    L += ['v_and_b32 v%d, 15, v%d' % (TMP, TMP),
          'v_and_b32 v%d, 3, v%d' % (IDX, TMP),
          'v_lshrrev_b32 v%d, 2, v%d' % (TMP, TMP),
          'v_cmp_eq_u32 vcc, 3, v%d' % IDX,
          'v_or_b32 v%d, 12, v%d' % (IDX, IDX),
          'v_cndmask_b32 v%d, v%d, 12, vcc' % (IDX, IDX),
          'v_mov_b32 v%d, 0x3f80' % (A0 + 3),
          'v_mov_b32 v%d, 0x3f800000' % (A0 + 2),
          'v_cndmask_b32 v%d, v%d, v%d, vcc' % (A0 + 3, A0 + 3, A0 + 2),
          'v_xor_b32 v%d, 4, v%d' % (IDX, IDX),
          'v_lshlrev_b32 v%d, 2, v%d' % (A0 + 2, TMP),
          'v_lshlrev_b32 v%d, v%d, v%d' % (IDX, A0 + 2, IDX),
          'v_xor_b32 v%d, 0x4444, v%d' % (IDX, IDX),
          'v_mov_b32 v%d, v%d' % (A0 + 2, A0 + 3)]
    for g in (3, 1, 0):
        L += ['v_cmp_eq_u32 vcc, %d, v%d' % (g, TMP), 'v_cndmask_b32 v%d, 0, v%d, vcc' % (A0 + g, A0 + 2)]
    L += ['v_cmp_eq_u32 vcc, 2, v%d' % TMP, 'v_cndmask_b32 v%d, 0, v%d, vcc' % (A0 + 2, A0 + 2)]
    L.append('s_waitcnt lgkmcnt(0)')
    if loads:
        for i in range(NT):
            L.append('v_lshrrev_b32 v%d, 16, v%d' % (ND0 + i, ND0 + i))
    for p in range(D - 1):
        gathers(p, L); col_clamped(p, D + p, L)
    L += ['s_add_i32 %s, %s, %d' % (STr, SGr, 2 * D - 1), 'v_lshl_add_u32 %s, %s, 7, %s' % (VP, STr, VCB)]
    if not taps:
        pass
    elif style in ('A', 'C'):
        L += ['v_mbcnt_lo_u32_b32 v%d, -1, 0' % WFBn, 'v_mbcnt_hi_u32_b32 v%d, -1, v%d' % (WFBn, WFBn),
              'v_lshlrev_b32 v%d, 4, v%d' % (WFBn, WFBn), 'v_add_u32 v%d, %s, v%d' % (WFBn, WOFS, WFBn),
              'ds_read_b128 %s, v%d offset:%d' % (WFt, WFBn, foff(0)), 's_waitcnt lgkmcnt(0)']
        # (style A waits for the first fragment here; the prologue's gathers of the sets in flight are issued behind it)
    else:
        frag_read(0, L)      # (the prologue's scratch registers are dead)
    L += ['s_sub_u32 %s, %s, %%%d' % (SCr, GB, TE), 's_cmp_eq_u32 %s, 0' % SCr]
    R = 2 if D == 2 else 1
    Aop, Iop = 'v[%d:%d]' % (A0, A0 + 3), 'v%d' % IDX

    # the operand requests of the loads variant, in the order their registers die
    queue = []          # (exit behind which the k-step is dead, k-step, tile)
    for f, (s_, h) in enumerate(FO):
        if h == 1 and s_ != HS - 1:
            queue += [(f // fpe, s_, i) for i in range(NT)]
    if not loads:
        queue = []
    nloads = len(queue)

    def burst(t, L, pending):
        """tile exit t: fragment(s) t * fpe .. of the hop's taps, the next fragment's read, operand requests"""
        if style == 'C':
            return
        if style == 'A':
            for f in range(t * fpe, min((t + 1) * fpe, NF)):
                wf = WF0 if f % 2 == 0 else WF1
                if f + 1 < NF:
                    L.append('ds_read_b128 %s, v%d offset:%d' % (tup(WF1 if f % 2 == 0 else WF0), WFBn, foff(f + 1)))
                s_, h = FO[f]
                for i in [(t + 1 + d) % NT for d in range(NT)]:
                    L.append('v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s' % (tup(ACC(h, i)), tup(wf), tup(OP(s_, i)), tup(ACC(h, i))))
                if f + 1 < NF:
                    L.append('s_waitcnt lgkmcnt(0)')
        for f in (range(t * fpe, min((t + 1) * fpe, NF)) if style == 'B' else ()):
            # The fragment was read at the previous exit (or in the prologue). Every trip waits for all but the 5 (D - 1) youngest LDS
            # operations, so after D - 1 trips of this tile it has landed; a tile with fewer trips waits for it here.
            first = (f == t * fpe)
            if first:
                L += ['s_sub_u32 %s, %%%d, %s' % (STr, TE + t, ('%%%d' % (TE + t - 1)) if t > 0 else GB),
                      's_cmp_ge_u32 %s, %d' % (STr, D - 1), 's_cbranch_scc1 L_W%d_%%=' % t, 's_waitcnt lgkmcnt(0)',
                      'L_W%d_%%=:' % t]
            else:
                L.append('s_waitcnt lgkmcnt(0)')
            s_, h = FO[f]
            for i in [(t + 1 + d) % NT for d in range(NT)]:
                L.append('v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s' % (tup(ACC(h, i)), WFt, tup(OP(s_, i)), tup(ACC(h, i))))
            if f + 1 < NF:
                L.append('s_nop 1')      # (the fragment buffer's first register becomes an address right behind the MFMAs that read it)
                frag_read(f + 1, L)
        if loads:
            n = 0
            while pending and pending[0][0] <= t and (n < lpe or t == NT - 1):
                _, s_, i = pending.pop(0)
                row_log2 = {1: 6, 2: 7}[HS if s_ < HS else XS]
                L.append('v_lshl_add_u32 v%d, v%d, %d, v%d' % (LAD, ND0 + i, row_log2, Q16n))
                if s_ < HS:
                    L.append('buffer_load_dwordx4 %s, v%d, %s, %s offen offset:%d' % (tup(OP(s_, i)), LAD, RH, SOH, 64 * s_))
                else:
                    L.append('buffer_load_dwordx4 %s, v%d, %s, %s offen offset:%d' % (tup(OP(s_, i)), LAD, RX, SOX, 64 * (s_ - HS)))
                n += 1

    SRET, SGRP, SCG, SSV = 's[92:93]', 's[94:95]', 's91', 's96'
    groups, GBYTES, NG = [], 0, 0
    if style == 'C':
        order = [(f, i) for f in range(NF) for i in range(NT)]
        NG = (len(order) + GSZ - 1) // GSZ
        GBYTES = 8 * GSZ + 8 + 4 + 4      # GSZ MFMAs, at most one fragment read + s_nop, s_setpc_b64
        for g in range(NG):
            G_ = []
            nbytes = 0
            for (f, i) in order[g * GSZ:(g + 1) * GSZ]:
                s_, h = FO[f]
                wf = WF0 if f % 2 == 0 else WF1
                if i == 0 and f + 1 < NF:      # the next fragment's read, behind the last MFMA that read its buffer
                    G_ += ['s_nop 1', 'ds_read_b128 %s, v%d offset:%d' % (tup(WF1 if f % 2 == 0 else WF0), WFBn, foff(f + 1))]
                    nbytes += 12
                G_.append('v_mfma_f32_16x16x32_bf16 %s, %s, %s, %s' % (tup(ACC(h, i)), tup(wf), tup(OP(s_, i)), tup(ACC(h, i))))
                nbytes += 8
            G_.append('s_setpc_b64 %s' % SRET)
            nbytes += 4
            assert nbytes <= GBYTES and (GBYTES - nbytes) % 4 == 0
            G_ += ['s_nop 0'] * ((GBYTES - nbytes) // 4)      # (behind the return: never executed, keeps the groups one size)
            groups.append(G_)
        # prologue: the address of group 0, the group counter
        L += ['s_getpc_b64 %s' % SGRP, 'L_PC_%=:', 's_add_u32 s94, s94, L_G0_%=-L_PC_%=', 's_addc_u32 s95, s95, 0', 's_mov_b32 %s, 0' % SCG]
        # (the trip loop keeps its exit condition in scc across the body: re-established below)
        L += ['s_sub_u32 %s, %s, %%%d' % (SCr, GB, TE), 's_cmp_eq_u32 %s, 0' % SCr]

    ngc = [0]

    def call_group(L):
        """one tap group, if any is left (scc is the trip loop's: saved and put back)"""
        L += ['s_cselect_b32 %s, 1, 0' % SSV,
              's_cmp_ge_u32 %s, %d' % (SCG, NG), 's_cbranch_scc1 L_NG%d_%%=' % ngc[0],
              's_swappc_b64 %s, %s' % (SRET, SGRP),
              's_add_u32 s94, s94, %d' % GBYTES, 's_addc_u32 s95, s95, 0', 's_add_u32 %s, %s, 1' % (SCG, SCG),
              'L_NG%d_%%=:' % ngc[0],
              's_cmp_lg_u32 %s, 0' % SSV]
        ngc[0] += 1

    pend = list(queue)
    exits = []
    for t in range(NT):
        per_phase = []
        save = list(pend)
        for p in range(D):
            pend = list(save)
            ex = []
            burst(t, ex, pend)
            per_phase.append(ex)
        exits.append(per_phase)
    assert not pend
    for t in range(NT):
        for pp in range(D * R):
            p = pp % D
            q = (p + D - 1) % D
            L.append('L_T%d_P%d_%%=:' % (t, pp))
            L.append('s_cbranch_scc1 L_X%d_P%d_%%=' % (t, p))
            L.append('s_add_u32 %s, %s, 1' % (SCr, SCr))
            gathers(q, L)
            L += ['ds_read_b32 %s, %s' % (VCw(q), VP), 'v_add_u32 %s, 128, %s' % (VP, VP)]
            if style == 'C':
                call_group(L)      # this trip's share of the hop's tap MFMAs, while its gathers are in flight
            L.append('s_waitcnt lgkmcnt(%d)' % (5 * (D - 1)))
            L.append('v_smfmac_f32_16x16x64_bf16 %s, %s, v[%d:%d], %s' % (tup(ACC(1, t)), Aop, Y(p, 0), Y(p, 0) + 7, Iop))
            L.append('v_smfmac_f32_16x16x64_bf16 %s, %s, v[%d:%d], %s' % (tup(ACC(0, t)), Aop, Xr(p, 0), Xr(p, 0) + 7, Iop))
        L.append('s_branch L_T%d_P0_%%=' % t)
        for p in range(D):
            L.append('L_X%d_P%d_%%=:' % (t, p))
            # (labels inside an exit must be unique per copy)
            L += [ln.replace('L_W%d_%%=' % t, 'L_W%dp%d_%%=' % (t, p)) for ln in exits[t][p]]
            if t + 1 < NT:
                L += ['s_sub_u32 %s, %%%d, %%%d' % (SCr, TE + t, TE + t + 1), 's_cmp_eq_u32 %s, 0' % SCr]
            L.append('s_branch L_T%d_P%d_%%=' % (t + 1, p))
    for p in range(D):
        L.append('L_T%d_P%d_%%=:' % (NT, p))
    if style == 'C':
        # the groups the trips did not reach (a wave with fewer trips than groups), back to back; then skip over the group bodies
        L += ['L_TAIL_%=:', 's_cmp_ge_u32 %s, %d' % (SCG, NG), 's_cbranch_scc1 L_TAILEND_%=', 's_waitcnt lgkmcnt(0)',
              's_swappc_b64 %s, %s' % (SRET, SGRP), 's_add_u32 s94, s94, %d' % GBYTES, 's_addc_u32 s95, s95, 0',
              's_add_u32 %s, %s, 1' % (SCG, SCG), 's_branch L_TAIL_%=', 'L_TAILEND_%=:', 's_branch L_END_%=', 'L_G0_%=:']
        for G_ in groups:
            L += G_
        L.append('L_END_%=:')
    L += ['s_nop 15', 's_nop 7', 's_waitcnt lgkmcnt(0)']
    # this wave's LDS-DMA pieces (issued in front of the block) have landed: everything but the operand requests behind them
    L.append('s_waitcnt vmcnt(%d)' % (nloads if loads else 0))
    return L, nloads


def emit(name, lines):
    print('#define %s \\' % name)
    for ln in lines:
        print('  "%s\\n\\t" \\' % ln)
    print('  ""')


def main():
    lines = gen()
    print('// GENERATED by tools/gen_hop_asm.py -- do not edit. The hop gather stream as one asm block (see the generator).')
    print('#define GCRNN_HOP_ASM_TEXT \\')
    for ln in lines:
        print('  "%s\\n\\t" \\' % ln)
    print('  ""')
    regs = ', '.join('"v%d"' % r for r in range(BASE, BASE + 68))
    print('#define GCRNN_HOP_ASM_CLOBBERS %s, "s88", "s89", "s90", "scc", "memory"' % regs)
    emit('GCRNN_HOP_ASM_UNI_TEXT', gen_uniform())
    emit('GCRNN_HOP_ASM_UNI16_TEXT', gen_uniform16())
    emit('GCRNN_HOP_ASM_UNI16_SPARSE_TEXT', gen_uniform16(sparse=True))
    emit('GCRNN_HOP_ASM_UNI16_SUMS_TEXT', gen_uniform16(sparse=False, sums=True))
    emit('GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_TEXT', gen_uniform16(sparse=True, sums=True))
    # the same two reading the SECOND hop image (sequence-resident kernel: hops alternate between two images, one barrier per hop)
    print('#define GCRNN_HOP_IMAGE_B_OFFSET %d' % IMAGE_B_OFFSET)
    print('#define GCRNN_HOP_COLUMN_PAD 512      /* bytes of zeros behind the column image: the summing streams read up to three groups past a wave\'s last one */')
    emit('GCRNN_HOP_ASM_UNI16_SUMS_TEXT_B', gen_uniform16(sparse=False, sums=True, img_off=IMAGE_B_OFFSET))
    emit('GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_TEXT_B', gen_uniform16(sparse=True, sums=True, img_off=IMAGE_B_OFFSET))
    emit('GCRNN_HOP_ASM_WIDE32_TEXT', gen_wide32())
    print('#define GCRNN_HOP_WIDE_PLANE %d' % WIDE_PLANE)
    lpe, fpe = int(os.environ.get('GCRNN_P_LPE', '4')), int(os.environ.get('GCRNN_P_FPE', '1'))
    sdepth = int(os.environ.get('GCRNN_P_SDEPTH', '3'))
    for (hs, xs) in ((2, 2), (2, 1), (1, 1)):
        emit('GCRNN_HOP_ASM_P32_TEXT_%d_%d' % (hs, xs), gen_wide32_taps(hs, xs)[0])
        emit('GCRNN_HOP_ASM_P32_STREAM_TEXT_%d_%d' % (hs, xs), gen_wide32_taps(hs, xs, taps=False, depth=sdepth)[0])
        ll, nl = gen_wide32_taps(hs, xs, loads=True, lpe=lpe, fpe=fpe)
        emit('GCRNN_HOP_ASM_P32_LOADS_TEXT_%d_%d' % (hs, xs), ll)
        print('#define GCRNN_HOP_ASM_P32_NLOADS_%d_%d %d' % (hs, xs, nl))
    print('#define GCRNN_HOP_ASM_P32_CLOBBERS %s, "s88", "s89", "s90", "s91", "s92", "s93", "s94", "s95", "s96", "scc", "vcc", "memory"' % ', '.join('"v%d"' % r for r in range(192, 254)))
    print('#define GCRNN_HOP_ASM_WIDE32_CLOBBERS %s, "s88", "s89", "s90", "scc", "vcc", "memory"' % ', '.join('"v%d"' % r for r in range(WIDE_BASE, 254)))
    regs = ', '.join('"v%d"' % r for r in range(UB, UB + 60))
    print('#define GCRNN_HOP_ASM_UNI_CLOBBERS %s, "s88", "s89", "s90", "scc", "memory"' % regs)
    regs16 = ', '.join('"v%d"' % r for r in range(UB16, UB16 + 36))
    print('#define GCRNN_HOP_ASM_UNI16_CLOBBERS %s, "s88", "s89", "s90", "scc", "vcc", "memory"' % regs16)
    regs16s = ', '.join('"v%d"' % r for r in range(SPA, UB16 + 36))
    print('#define GCRNN_HOP_ASM_UNI16_SPARSE_CLOBBERS %s, "s88", "s89", "s90", "scc", "vcc", "memory"' % regs16s)
    # the summing variants do not use the stream's own D accumulator (the last four registers of the window)
    print('#define GCRNN_HOP_ASM_UNI16_SUMS_CLOBBERS %s, "s88", "s89", "s90", "scc", "vcc", "memory"' % ', '.join('"v%d"' % r for r in range(SUMS_UB16, 254)))
    print('#define GCRNN_HOP_ASM_UNI16_SUMS_SPARSE_CLOBBERS %s, "s88", "s89", "s90", "scc", "vcc", "memory"' % ', '.join('"v%d"' % r for r in range(SUMS_SPA, 254)))


if __name__ == '__main__':
    main()
