"""CPU check of the NSF_CL tile gradient kernel's index tables and lane conventions (mnf_nsf_bwd_tile.hip).

Runs without a GPU: the operand / flush tables come from the library's host functions, the MFMAs are emulated lane by
lane in float64 (v_mfma_f32_16x16x16_f16: A lane (i, kq) holds A[i][4 kq + e], B lane (n, kq) holds B[4 kq + e][n],
C lane (n, q) register r is C[4 q + r][n]; the K = 32 form holds 8 k per lane), and every stage of one half-step is
compared with plain matrix arithmetic on the flat parameters: hidden layers, the 3K-1 parameters per element, W4^T g,
the deltas, the conditioning half's cotangent and -- through the flush table -- every parameter gradient.

usage: python tools/emulate_nsf_tile.py [dim K h0 h1 h2]"""
import ctypes
import sys

import numpy as np

sys.path.insert(0, ".")
from torch_mnf_amd import _lib  # noqa: E402

LO = 1 << 30


def mfma16(A, B):
    """A, B: [64][4] -> C [64][4]"""
    C = np.zeros((64, 4))
    for lane in range(64):
        n, q = lane & 15, lane >> 4
        for r in range(4):
            m = 4 * q + r
            C[lane, r] = sum(A[m + 16 * kq, e] * B[n + 16 * kq, e] for kq in range(4) for e in range(4))
    return C


def mfma32(A, B):
    """A, B: [64][8] -> C [64][4]"""
    C = np.zeros((64, 4))
    for lane in range(64):
        n, q = lane & 15, lane >> 4
        for r in range(4):
            m = 4 * q + r
            C[lane, r] = sum(A[m + 16 * kq, e] * B[n + 16 * kq, e] for kq in range(4) for e in range(8))
    return C


def main(dim=32, K=8, hidden=(8, 8, 8)):
    lib = _lib.load()
    hid = _lib.int_array(hidden)
    ns, npl, npar = ctypes.c_int64(), ctypes.c_int64(), ctypes.c_int64()
    _lib.check("layout", lib.mnf_nsf_cl_bwd_tile_layout(dim, K, 3, hid, ctypes.byref(ns), ctypes.byref(npl), ctypes.byref(npar)))
    ns, npl, npar = ns.value, npl.value, npar.value
    idx = (ctypes.c_int32 * (2 * ns + npl))()
    flush = (ctypes.c_int32 * npar)()
    _lib.check("index", lib.mnf_nsf_cl_bwd_tile_index(dim, K, 3, hid, idx, flush))
    idx, flush = np.array(idx), np.array(flush)
    hr, P = dim // 2, 3 * K - 1
    H = 16 if hr <= 16 else 32
    NH = 8 if max(hidden) <= 8 else 16
    G, S, NB = H // 16, H // 4, (P + 3) // 4
    n_ops = 2 * G + 4 + 2 * S * NB
    assert ns == 2 * n_ops * 256, (ns, n_ops)
    rng = np.random.default_rng(0)
    flat = rng.normal(size=npar)
    per_net = npar // 2
    sizes = [hr, *hidden, P * hr]

    def net_params(nn):
        o, Ws, bs = nn * per_net, [], []
        for l in range(4):
            Ws.append(flat[o:o + sizes[l + 1] * sizes[l]].reshape(sizes[l + 1], sizes[l]))
            o += sizes[l + 1] * sizes[l]
            bs.append(flat[o:o + sizes[l + 1]])
            o += sizes[l + 1]
        return Ws, bs, nn * per_net

    # image values (head words only; check that the residual entry names the same source)
    words = idx[:2 * ns].reshape(ns, 2)
    ops = np.zeros((2, n_ops, 64, 4))
    for nn in range(2):
        for op in range(n_ops):
            for lane in range(64):
                for e in range(4):
                    base = nn * n_ops * 256 + (op * 64 + lane) * 4
                    hi = words[base + (e >> 1), e & 1]
                    lo = words[base + 2 + (e >> 1), e & 1]
                    assert (hi < 0) == (lo < 0)
                    if hi >= 0:
                        assert lo == hi | LO, (hi, lo)
                        ops[nn, op, lane, e] = flat[hi]
    plain = idx[2 * ns:]
    bias = np.where(plain >= 0, flat[np.maximum(plain, 0)], 0.0).reshape(2, 3 + S * NB, 16)
    OP_F1, OP_F2, OP_F4 = 0, G, G + 2
    OP_T4 = OP_F4 + S * NB
    OP_T3 = OP_T4 + S * NB
    OP_T2, OP_T1 = OP_T3 + 1, OP_T3 + 2
    T_W3 = S * NB
    T_W2, T_W1 = T_W3 + 1, T_W3 + 2
    T_B = T_W1 + G
    biascol = NH <= 8
    tiles = T_B + (3 + (0 if biascol else S * NB) + 15) // 16

    def c_layout(M):  # M [16 features][16 rows] -> [64][4]
        out = np.zeros((64, 4))
        for lane in range(64):
            out[lane] = M[4 * (lane >> 4):4 * (lane >> 4) + 4, lane & 15]
        return out

    def from_c(C):
        M = np.zeros((16, 16))
        for lane in range(64):
            M[4 * (lane >> 4):4 * (lane >> 4) + 4, lane & 15] = C[lane]
        return M

    def bias_c(nn, tile):
        return np.array([bias[nn, tile, 4 * (lane >> 4):4 * (lane >> 4) + 4] for lane in range(64)])

    ident = np.array([[1.0 if 4 * (lane >> 4) + e == (lane & 15) else 0.0 for e in range(4)] for lane in range(64)])

    def transpose(C):
        return mfma16(C, ident)

    leaky = lambda v: np.maximum(v, 0.2 * v)
    worst = 0.0
    for nn in range(2):
        Ws, bs, base = net_params(nn)
        x = np.zeros((16, H))
        x[:, :hr] = rng.normal(size=(16, hr))  # [row][feature]
        cond = [c_layout(x[:, 16 * g:16 * g + 16].T) for g in range(G)]
        # ---- forward
        acc = bias_c(nn, 0)
        for g in range(G):
            acc = acc + mfma16(ops[nn, OP_F1 + g], cond[g])
        h = [leaky(acc)]
        for l in range(2):
            h.append(leaky(bias_c(nn, 1 + l) + mfma16(ops[nn, OP_F2 + l], h[l])))
        ref_h = [x[:, :hr]]
        for l in range(3):
            ref_h.append(leaky(ref_h[-1] @ Ws[l].T + bs[l]))
        for l in range(3):
            got = from_c(h[l])[:hidden[l]].T
            worst = max(worst, np.abs(got - ref_h[l + 1]).max())
            assert np.allclose(got, ref_h[l + 1]), ("hidden", l)
            assert np.allclose(from_c(h[l])[hidden[l]:], 0), "structural zeros"
        ref_p = (ref_h[3] @ Ws[3].T + bs[3]).reshape(16, hr, P)  # [row][elem][param]
        gp = rng.normal(size=(16, hr, P))
        red = np.zeros((tiles, 64, 4))
        ones_col = np.array([[1.0] * 4 if (lane & 15) == 8 else [0.0] * 4 for lane in range(64)])

        def act_ops(C, ones):
            t = transpose(C)
            if ones:
                t = np.where(ones_col > 0, 1.0, t)
            return np.concatenate([t, t], axis=1)

        def delta_op(C):
            t = transpose(C)
            return np.concatenate([t, np.zeros_like(t)], axis=1)  # (residual part: zero in this exact emulation)

        def onehot(c):
            return np.array([[1.0] * 8 if (lane & 15) == c else [0.0] * 8 for lane in range(64)])

        h3ops = act_ops(h[2], biascol)
        y = np.zeros((64, 4))
        for s in range(S):
            p = np.zeros((64, 4 * NB))
            for kb in range(NB):
                p[:, 4 * kb:4 * kb + 4] = bias_c(nn, 3 + s * NB + kb) + mfma16(ops[nn, OP_F4 + s * NB + kb], h[2])
            for lane in range(64):
                row, q = lane & 15, lane >> 4
                elem = 16 * (s >> 2) + 4 * q + (s & 3)
                if elem < hr:
                    assert np.allclose(p[lane, :P], ref_p[row, elem]), ("params", s, lane)
            for kb in range(NB):
                gt = np.zeros((64, 4))
                for lane in range(64):
                    row, q = lane & 15, lane >> 4
                    elem = 16 * (s >> 2) + 4 * q + (s & 3)
                    for r in range(4):
                        if 4 * kb + r < P and elem < hr:
                            gt[lane, r] = gp[row, elem, 4 * kb + r]
                        else:
                            gt[lane, r] = rng.normal() if 4 * kb + r < P else 0.0  # dead element: garbage must not matter
                y = y + mfma16(ops[nn, OP_T4 + s * NB + kb], gt)
                d = delta_op(gt)
                t = s * NB + kb
                red[t] += mfma32(d, h3ops)
                if not biascol:
                    red[T_B + (3 + t) // 16] += mfma32(d, onehot((3 + t) & 15))
        ref_y = gp.reshape(16, hr * P) @ Ws[3]  # [row][u]
        assert np.allclose(from_c(y)[:hidden[2]].T, ref_y), "W4^T g"
        assert np.allclose(from_c(y)[hidden[2]:], 0)
        masks = [(ref_h[l + 1] > 0) * 0.8 + 0.2 for l in range(3)]
        ref_d = [None, None, ref_y * masks[2]]

        def masked(C, l):
            M = from_c(C)
            m = np.ones((16, 16))
            m[:hidden[l]] = masks[l].T
            return c_layout(M * m)

        d3 = masked(y, 2)
        red[T_W3] += mfma32(delta_op(d3), act_ops(h[1], biascol))
        if not biascol:
            red[T_B] += mfma32(delta_op(d3), onehot(2))
        d2 = masked(mfma16(ops[nn, OP_T3], d3), 1)
        ref_d[1] = (ref_d[2] @ Ws[2]) * masks[1]
        assert np.allclose(from_c(d2)[:hidden[1]].T, ref_d[1]), "delta 2"
        red[T_W2] += mfma32(delta_op(d2), act_ops(h[0], biascol))
        if not biascol:
            red[T_B] += mfma32(delta_op(d2), onehot(1))
        d1 = masked(mfma16(ops[nn, OP_T2], d2), 0)
        ref_d[0] = (ref_d[1] @ Ws[1]) * masks[0]
        assert np.allclose(from_c(d1)[:hidden[0]].T, ref_d[0]), "delta 1"
        red[T_B] += mfma32(delta_op(d1), onehot(0))
        for g in range(G):
            red[T_W1 + g] += mfma32(delta_op(d1), act_ops(cond[g], False))
            gc = from_c(mfma16(ops[nn, OP_T1 + g], d1))
            ref_gc = (ref_d[0] @ Ws[0])[:, 16 * g:16 * g + 16]
            w = min(16, hr - 16 * g)
            assert np.allclose(gc[:w].T, ref_gc[:, :w]), "g_cond"
        # ---- the flush table against plain outer products
        ref_grad = np.zeros(per_net)
        o = 0
        ins = [ref_h[0], ref_h[1], ref_h[2], ref_h[3]]
        deltas = [ref_d[0], ref_d[1], ref_d[2], gp.reshape(16, hr * P)]
        for l in range(4):
            dW = deltas[l].T @ ins[l]
            ref_grad[o:o + dW.size] = dW.reshape(-1)
            o += dW.size
            ref_grad[o:o + sizes[l + 1]] = deltas[l].sum(0)
            o += sizes[l + 1]
        flat_red = red.reshape(-1)
        got = flat_red[flush[base:base + per_net]]
        bad = np.flatnonzero(~np.isclose(got, ref_grad))
        assert bad.size == 0, ("flush", nn, bad[:10], got[bad[:10]], ref_grad[bad[:10]])
        assert np.all(flush[base:base + per_net] < tiles * 256)
    print(f"dim={dim} K={K} hidden={hidden}: tables and lane conventions OK")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        a = [int(v) for v in sys.argv[1:]]
        main(a[0], a[1], tuple(a[2:5]))
    else:
        main()
        main(32, 5, (8, 8, 8))
        main(32, 8, (6, 6, 6))
        main(32, 8, (3, 5, 8))
        main(16, 8, (8, 8, 8))
        main(24, 5, (4, 4, 4))
        main(32, 8, (16, 16, 16))
        main(32, 5, (12, 16, 9))
        main(64, 8, (8, 8, 8))
        main(64, 5, (16, 16, 16))
        main(48, 8, (8, 8, 8))
