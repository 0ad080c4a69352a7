"""The operand-layout algebra behind csrc/field.hip's native-layout kernels (k_field_forward_nat / k_field_backward_nat), checked on
the CPU with an emulation of v_mfma_f32_32x32x16_f16's register layouts (the ones tools/ubench/mfma_probe.hip validated on hardware):

    A [32 x 16]: lane (i, hi) holds A[i][8 hi .. 8 hi + 7]          B [16 x 32]: lane (n, hi) holds B[8 hi .. 8 hi + 7][n]
    D [32 x 32]: element r of lane (n, hi) is D[(r & 3) + 8 (r >> 2) + 4 hi][n]

Claim: if a layer's weight fragments have their K columns in the order phi(t, hi, j) below, the packed result of
one layer IS the B operand of the next — no cross-lane exchange — and W1^T's result comes out as whole levels per lane.
This file restates k_field_pack's second fragment set and the kernels' slot arithmetic in numpy and runs the whole forward and
backward chain of the 32-64-64-4 MLP through the emulated MFMA against plain matrix products. (The kernels themselves are tested
on the GPU against the torch module: tests/test_gpu_parity.py.)"""
import numpy as np

K_IN, K_HID, K_OUT = 32, 64, 4


def phi_enc(t, hi, j):      # encoder features: lane half hi holds levels 8 hi .. 8 hi + 7; K step t covers 4 of them
    return 2 * (8 * hi + 4 * t + (j >> 1)) + (j & 1)


def phi_h(t, hi, j):        # hidden features: where the previous layer's MFMA result left them
    return 32 * (t >> 1) + 4 * hi + 8 * (2 * (t & 1) + (j >> 2)) + (j & 3)


def d_row(r, hi):           # row of D element r in lane half hi
    return (r & 3) + 8 * (r >> 2) + 4 * hi


def mfma(A, B, acc):
    """A [64 lanes, 8], B [64 lanes, 8], acc [64 lanes, 16] -> acc + A . B in the layouts of the docstring."""
    Am = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for lane in range(64):
        i, hi = lane & 31, lane >> 5
        Am[i, 8 * hi:8 * hi + 8] = A[lane]
        Bm[8 * hi:8 * hi + 8, i] = B[lane]
    D = Am @ Bm
    out = acc.copy()
    for lane in range(64):
        n, hi = lane & 31, lane >> 5
        for r in range(16):
            out[lane, r] += D[d_row(r, hi), n]
    return out


def frag(W, rows, cols_of):
    """A-operand fragment: lane (m, hi) slot j = W[rows(m)][cols_of(hi, j)] (zero where rows / cols are None)."""
    F = np.zeros((64, 8))
    for lane in range(64):
        m, hi = lane & 31, lane >> 5
        r = rows(m)
        for j in range(8):
            c = cols_of(hi, j)
            if r is not None and c is not None:
                F[lane, j] = W[r, c]
    return F


def layer(frags, x_words, ks):
    """One 32-row block: sum over K steps; x_words [64 lanes, 8 ks] = this lane's slots in K-step order."""
    acc = np.zeros((64, 16))
    for t in range(ks):
        acc = mfma(frags[t], x_words[:, 8 * t:8 * t + 8], acc)
    return acc


def to_words(accs):
    """Two 32-row blocks' D registers -> the lane's 32 hidden slots in K-step order (word q = elements 2q, 2q + 1)."""
    return np.concatenate([accs[0], accs[1]], axis=1)      # [64, 32]: slot index = 16 mb + r = 8 t + j with t = 2 mb + (r >> 3)


def test_native_layout_chain_equals_matrix_products():
    rng = np.random.default_rng(0)
    W1, W2, W3 = rng.normal(size=(K_HID, K_IN)), rng.normal(size=(K_HID, K_HID)), rng.normal(size=(K_OUT, K_HID))
    X = rng.normal(size=(K_IN, 32))                        # 32 samples = one column block
    # the lane's encoder slots: lane (n, hi), K step t, slot j = feature phi_enc(t, hi, j) of sample n
    e = np.zeros((64, 16))
    for lane in range(64):
        n, hi = lane & 31, lane >> 5
        for t in range(2):
            for j in range(8):
                e[lane, 8 * t + j] = X[phi_enc(t, hi, j), n]

    # slot r of block mb of a layer result is feature 32 mb + d_row(r, hi): the claim is that this IS phi_h(2 mb + (r >> 3), hi, r & 7)
    for mb in range(2):
        for hi in range(2):
            for r in range(16):
                assert 32 * mb + d_row(r, hi) == phi_h(2 * mb + (r >> 3), hi, r & 7)

    relu = lambda a: np.maximum(a, 0)
    F1 = [[frag(W1, lambda m, mb=mb: 32 * mb + m, lambda hi, j, t=t: phi_enc(t, hi, j)) for t in range(2)] for mb in range(2)]
    h1 = to_words([relu(layer(F1[mb], e, 2)) for mb in range(2)])
    F2 = [[frag(W2, lambda m, mb=mb: 32 * mb + m, lambda hi, j, t=t: phi_h(t, hi, j)) for t in range(4)] for mb in range(2)]
    h2 = to_words([relu(layer(F2[mb], h1, 4)) for mb in range(2)])
    F3 = [frag(W3, lambda m: m if m < K_OUT else None, lambda hi, j, t=t: phi_h(t, hi, j)) for t in range(4)]
    out = layer(F3, h2, 4)

    H1, H2 = relu(W1 @ X), None
    H2 = relu(W2 @ H1)
    Y = W3 @ H2
    for n in range(32):                                    # the 4 outputs of sample n are elements 0..3 of its hi = 0 lane
        assert np.allclose(out[n, :4], Y[:, n])
    for lane in range(64):                                 # hidden slots hold the features the map says
        n, hi = lane & 31, lane >> 5
        for t in range(4):
            for j in range(8):
                assert np.isclose(h1[lane, 8 * t + j], H1[phi_h(t, hi, j), n]) and np.isclose(h2[lane, 8 * t + j], H2[phi_h(t, hi, j), n])

    # ---- backward: dh3 lives in slots 0..3 of the hi = 0 lanes' first K step ----
    G3 = rng.normal(size=(K_OUT, 32))
    d3 = np.zeros((64, 8))
    d3[:32, :4] = G3.T
    F3T = [frag(W3.T, lambda m, mb=mb: 32 * mb + m, lambda hi, j: (8 * hi + j) if 8 * hi + j < K_OUT else None) for mb in range(2)]
    g2 = to_words([layer([F3T[mb]], d3, 1) for mb in range(2)]) * (h2 > 0)
    F2T = [[frag(W2.T, lambda m, mb=mb: 32 * mb + m, lambda hi, j, t=t: phi_h(t, hi, j)) for t in range(4)] for mb in range(2)]
    g1 = to_words([layer(F2T[mb], g2, 4) for mb in range(2)]) * (h1 > 0)
    F1T = [frag(W1.T, lambda m: m, lambda hi, j, t=t: phi_h(t, hi, j)) for t in range(4)]
    dx = layer(F1T, g1, 4)

    G2 = (W3.T @ G3) * (H2 > 0)
    G1 = (W2.T @ G2) * (H1 > 0)
    DX = W1.T @ G1
    for lane in range(64):
        n, hi = lane & 31, lane >> 5
        for q in range(8):                                 # word q of lane half hi is level (q & 1) + 4 (q >> 1) + 2 hi
            level = (q & 1) + 4 * (q >> 1) + 2 * hi
            assert np.allclose(dx[lane, 2 * q:2 * q + 2], DX[2 * level:2 * level + 2, n])
    levels = sorted((q & 1) + 4 * (q >> 1) + 2 * hi for hi in range(2) for q in range(8))
    assert levels == list(range(16))                       # the two lane halves write every level exactly once


def test_staging_rows_cover_every_feature_once():
    """word p of lane half hi holds features f, f + 1 with f = 32 (p >> 3) + 4 hi + 8 ((p & 7) >> 1) + 2 (p & 1) (what stage_hidden_tr
    relies on: words 8 b + 2 m, + 1 are the four consecutive features 32 b + 4 hi + 8 m .. + 3)."""
    rows = sorted(f + d for hi in range(2) for p in range(16) for d in (0, 1)
                  for f in [32 * (p >> 3) + 4 * hi + 8 * ((p & 7) >> 1) + 2 * (p & 1)])
    assert rows == list(range(64))
    for hi in range(2):
        for p in range(16):
            f = 32 * (p >> 3) + 4 * hi + 8 * ((p & 7) >> 1) + 2 * (p & 1)
            t, j = (2 * p) >> 3, (2 * p) & 7                # slot pair (2p, 2p + 1) in K-step order
            assert f == phi_h(t, hi, j) and f + 1 == phi_h(t, hi, j + 1)


# ---- round 3: operands staged as [sample][32 features] blocks, read back through ds_read_b64_tr_b16 -------------------------------
TR_PITCH, TS = 72, 128


def tr_read(lds, addr):
    """ds_read_b64_tr_b16 as tools/ubench/tr_read.hip found it on the hardware: lane c of a 16-lane group receives, for j = 0..3,
    element (c & 3) of the 8-byte piece addressed by lane 4 j + (c >> 2) of its group. lds: uint16 view; addr [64] byte addresses."""
    out = np.zeros((64, 4), lds.dtype)
    for lane in range(64):
        g, c = lane >> 4, lane & 15
        for j in range(4):
            src = 16 * g + 4 * j + (c >> 2)
            out[lane, j] = lds[addr[src] // 2 + (c & 3)]
    return out


def tr_lane_offset(lane, pitch, with_features):
    g, i = lane >> 4, lane & 15
    j, q = i >> 2, i & 3
    return (8 * (g >> 1) + j) * pitch + ((16 * (g & 1) + 4 * q) * 2 if with_features else 0)


def frag_tr(lds, off, step, pitch):
    a0 = np.array([off[l] + step * 16 * pitch for l in range(64)])
    return np.concatenate([tr_read(lds, a0), tr_read(lds, a0 + 4 * pitch)], axis=1)      # [64, 8]


def test_probe_pattern_of_the_transposing_read():
    """the first pattern of tools/ubench/tr_read.hip (lane l addresses byte 8 l, LDS holds its own element indices): lane c of group g
    read 64 g + c + 16 j"""
    lds = np.arange(4096)
    got = tr_read(lds, [8 * l for l in range(64)])
    for lane in range(64):
        assert list(got[lane]) == [64 * (lane >> 4) + (lane & 15) + 16 * j for j in range(4)]


def test_transposing_read_staging_contracts_over_samples():
    """stage_hidden_tr + frag_tr + the MFMA layouts give dW[i][j] = sum over the tile's 128 samples of g[i][s] h[j][s] for every 32 x 32
    block pair of two staged 64-feature quantities, and the 4-row d h3 block (8-byte pitch, every lane of a group addressing the
    row's only piece) against a 32-row h block with the operand rows >= 4 zeroed."""
    rng = np.random.default_rng(1)
    H, G = rng.normal(size=(K_HID, TS)), rng.normal(size=(K_HID, TS))
    G3 = rng.normal(size=(K_OUT, TS))
    kFB = TS * TR_PITCH
    lds = np.zeros((5 * kFB + TS * 8) // 2)

    def stage_hidden(blk, M):     # every lane (n, hi) of the four waves writes its 16 words as eight 8-byte pieces
        for wave in range(4):
            for lane in range(64):
                n, hi = lane & 31, lane >> 5
                col = 32 * wave + n
                for b in range(2):
                    for m in range(4):
                        f0 = 32 * b + 4 * hi + 8 * m                      # words 8 b + 2 m, + 1 = features f0 .. f0 + 3 (test above)
                        byte = (blk + b) * kFB + col * TR_PITCH + (8 * m + 4 * hi) * 2
                        lds[byte // 2:byte // 2 + 4] = M[f0:f0 + 4, col]

    stage_hidden(0, H)
    stage_hidden(2, G)
    for s in range(TS):
        lds[(5 * kFB + s * 8) // 2:(5 * kFB + s * 8) // 2 + 4] = G3[:, s]
    lane_off = [tr_lane_offset(l, TR_PITCH, True) for l in range(64)]
    lane_off_d3 = [tr_lane_offset(l, 8, False) for l in range(64)]

    def contract(a_base, a_pitch, a_lane, b_base, a_rows_valid=32):
        acc = np.zeros((64, 16))
        for step in range(TS // 16):
            A = frag_tr(lds, [a_base + o for o in a_lane], step, a_pitch)
            B = frag_tr(lds, [b_base + o for o in lane_off], step, TR_PITCH)
            for lane in range(64):
                if (lane & 31) >= a_rows_valid:
                    A[lane] = 0
            acc = mfma(A, B, acc)
        D = np.zeros((32, 32))
        for lane in range(64):
            n, hi = lane & 31, lane >> 5
            for r in range(16):
                D[d_row(r, hi), n] = acc[lane, r]
        return D

    W = G @ H.T
    for bi in range(2):
        for bj in range(2):
            D = contract((2 + bi) * kFB, TR_PITCH, lane_off, bj * kFB)
            assert np.allclose(D, W[32 * bi:32 * bi + 32, 32 * bj:32 * bj + 32])
    W3 = G3 @ H.T
    for bj in range(2):
        D = contract(5 * kFB, 8, lane_off_d3, bj * kFB, a_rows_valid=K_OUT)
        assert np.allclose(D[:K_OUT], W3[:, 32 * bj:32 * bj + 32]) and np.all(D[K_OUT:] == 0)
