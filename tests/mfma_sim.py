"""numpy model of v_mfma_f32_32x32x16_bf16's lane layouts (cdna_hip_programming.md section 3) used to
check the attention kernel's index math on CPU.  A: lane l holds A[l&31][8*(l>>5)+j], B: lane l holds
B[8*(l>>5)+j][l&31], D: lane l reg r holds D[(r&3)+8*(r>>2)+4*(l>>5)][l&31]."""
import numpy as np


def mfma_32x32x16(a_frag, b_frag, c_frag):
    """a_frag,b_frag [64,8], c_frag [64,16] -> [64,16]"""
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for l in range(64):
        for j in range(8):
            A[l & 31, 8 * (l >> 5) + j] = a_frag[l, j]
            Bm[8 * (l >> 5) + j, l & 31] = b_frag[l, j]
    D = A @ Bm
    out = c_frag.copy()
    for l in range(64):
        for r in range(16):
            out[l, r] += D[(r & 3) + 8 * (r >> 2) + 4 * (l >> 5), l & 31]
    return out


def attention_wave_sim(Q, K, V, scale):
    """Emulates one wave of k_attn_fwd<DP> (DP = D, multiple of 32): Q [32,D], K [Skv,D], V [Skv,D].
    Follows the kernel's staging/permutation/fragment code path literally; returns O [32,D]."""
    D = Q.shape[1]; Skv = K.shape[0]
    KSTEPS, DT = D // 16, D // 32
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    qf = np.zeros((KSTEPS, 64, 8))
    for kk in range(KSTEPS):
        for l in range(64):
            qf[kk, l] = Q[l31[l], 16 * kk + 8 * hi[l]: 16 * kk + 8 * hi[l] + 8]
    o = np.zeros((DT, 64, 16)); m_run = np.full(64, -np.inf); l_run = np.zeros(64)
    sl2 = scale * 1.4426950408889634
    n_tiles = (Skv + 63) // 64
    for j in range(n_tiles):
        kv0 = 64 * j
        # LDS images
        kt = np.zeros((64, D))
        for row in range(64):
            if kv0 + row < Skv:
                kt[row] = K[kv0 + row]
        vt = np.zeros((D, 64))  # permuted kv order within 16-groups
        for d in range(D):
            for kc in range(8):
                src = np.zeros(8)
                for e in range(8):
                    kv = kv0 + kc * 8 + e
                    if kv < Skv:
                        src[e] = V[kv, d]
                base = (kc >> 1) * 16 + (kc & 1) * 4      # element offsets: bytes/2
                vt[d, base: base + 4] = src[:4]
                vt[d, base + 8: base + 12] = src[4:]
        s = np.zeros((2, 64, 16))
        for t in range(2):
            for kk in range(KSTEPS):
                kf = np.zeros((64, 8))
                for l in range(64):
                    kf[l] = kt[32 * t + l31[l], 16 * kk + 8 * hi[l]: 16 * kk + 8 * hi[l] + 8]
                s[t] = mfma_32x32x16(kf, qf[kk], s[t])
        for t in range(2):
            for r in range(16):
                kv = kv0 + 32 * t + (r & 3) + 8 * (r >> 2) + 4 * hi
                s[t][kv >= Skv, r] = -np.inf
        mx = np.maximum(s[0].max(1), s[1].max(1))
        mx = np.maximum(mx, mx[lanes ^ 32])
        m_new = np.maximum(m_run, mx * sl2)
        alpha = np.exp2(m_run - m_new)
        p = np.exp2(s * sl2 - m_new[None, :, None])
        l_run = l_run * alpha + p.sum((0, 2))
        m_run = m_new
        o *= alpha[None, :, None]
        for dt in range(DT):
            for ks in range(4):
                t, u = ks >> 1, ks & 1
                pf = p[t][:, 8 * u: 8 * u + 8]
                vf = np.zeros((64, 8))
                for l in range(64):
                    vf[l] = vt[32 * dt + l31[l], 16 * ks + 8 * hi[l]: 16 * ks + 8 * hi[l] + 8]
                o[dt] = mfma_32x32x16(vf, pf, o[dt])
    l_tot = l_run + l_run[lanes ^ 32]
    O = np.zeros((32, D))
    for dt in range(DT):
        for g in range(4):
            for l in range(64):
                d = 32 * dt + 8 * g + 4 * hi[l]
                O[l31[l], d: d + 4] = o[dt][l, 4 * g: 4 * g + 4] / l_tot[l]
    return O


def _swap23(r):
    return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1)


def attention_wave_sim_v3(Q, K, V, scale, prescale=True, thr=8.0):
    """Emulates one wave of k_attn_fwd_v3<DP, PRESCALE> (csrc/attention.hip): K rows stored in LDS with bits 2/3 of the
    row index swapped (so the S^T accumulator order is the natural k order of P.V and the V^T fragments are plain
    16-byte rows), the running maximum entering through the MFMA's C operand, deferred re-basing with threshold `thr`
    and the forced first-tile re-base.  Returns O [32, D] and the number of tiles that took the re-base branch."""
    D = Q.shape[1]; Skv = K.shape[0]
    KSTEPS, DT = D // 16, D // 32
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    sl2 = scale * 1.4426950408889634
    Qs = Q * sl2 if prescale else Q
    qf = np.zeros((KSTEPS, 64, 8))
    for kk in range(KSTEPS):
        for l in range(64):
            qf[kk, l] = Qs[l31[l], 16 * kk + 8 * hi[l]: 16 * kk + 8 * hi[l] + 8]
    o = np.zeros((DT, 64, 16)); m_run = np.zeros(64); l_run = np.zeros(64)
    cinit = np.zeros((64, 16))
    n_tiles = (Skv + 63) // 64
    n_rebase = 0
    for j in range(n_tiles):
        kv0 = 64 * j
        first = j == 0
        kt = np.zeros((64, D))                              # LDS image of the K tile: row i <- K[kv0 + swap23(i)]
        for row in range(64):
            src = kv0 + _swap23(row)
            if src < Skv:                                   # rows past the end are outside num_records: zero
                kt[row] = K[src]
        vt = np.zeros((D, 64))                              # V^T tile, natural kv order
        for kv in range(64):
            if kv0 + kv < Skv:
                vt[:, kv] = V[kv0 + kv]
        s = np.zeros((2, 64, 16))
        for t in range(2):
            acc = cinit.copy()
            for kk in range(KSTEPS):
                kf = np.zeros((64, 8))
                for l in range(64):
                    kf[l] = kt[32 * t + l31[l], 16 * kk + 8 * hi[l]: 16 * kk + 8 * hi[l] + 8]
                acc = mfma_32x32x16(kf, qf[kk], acc)
            s[t] = acc
        if (j + 1) * 64 > Skv:
            for t in range(2):
                for r in range(16):
                    kv = kv0 + 32 * t + 16 * (r >> 3) + 8 * hi + 4 * ((r >> 2) & 1) + (r & 3)
                    s[t][kv >= Skv, r] = -np.inf
        mx = np.maximum(s[0].max(1), s[1].max(1))
        mx = np.maximum(mx, mx[lanes ^ 32])
        if prescale:
            if first or not np.all(mx <= thr):
                n_rebase += 1
                delta = mx if first else np.maximum(mx, 0.0)
                alpha = np.ones(64) if first else np.exp2(-delta)
                m_run = m_run + delta
                l_run = l_run * alpha
                o *= alpha[None, :, None]
                s -= delta[None, :, None]
                cinit = np.repeat(-m_run[:, None], 16, 1)
            p = np.exp2(s)
        else:
            mx = mx * sl2
            if first or not np.all(mx <= m_run + thr):
                n_rebase += 1
                m_new = mx if first else np.maximum(m_run, mx)
                alpha = np.ones(64) if first else np.exp2(m_run - m_new)
                m_run = m_new
                l_run = l_run * alpha
                o *= alpha[None, :, None]
            p = np.exp2(s * sl2 - m_run[None, :, None])
        l_run = l_run + p.sum((0, 2))
        for dt in range(DT):
            for ks in range(4):
                t, u = ks >> 1, ks & 1
                pf = p[t][:, 8 * u: 8 * u + 8]
                vf = np.zeros((64, 8))
                for l in range(64):
                    vf[l] = vt[32 * dt + l31[l], 16 * ks + 8 * hi[l]: 16 * ks + 8 * hi[l] + 8]
                o[dt] = mfma_32x32x16(vf, pf, o[dt])
    l_tot = l_run + l_run[lanes ^ 32]
    O = np.zeros((32, D))
    for dt in range(DT):
        for g in range(4):
            for l in range(64):
                d = 32 * dt + 8 * g + 4 * hi[l]
                O[l31[l], d: d + 4] = o[dt][l, 4 * g: 4 * g + 4] / l_tot[l]
    return O, n_rebase


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for D, Skv in ((64, 128), (64, 77), (32, 200)):
        Q = rng.standard_normal((32, D)); K = rng.standard_normal((Skv, D)); V = rng.standard_normal((Skv, D))
        sc = D ** -0.5
        S = Q @ K.T * sc
        P = np.exp(S - S.max(1, keepdims=True)); P /= P.sum(1, keepdims=True)
        ref = P @ V
        got = attention_wave_sim(Q, K, V, sc)
        got3, nb = attention_wave_sim_v3(Q, K, V, sc)
        print(D, Skv, np.abs(got - ref).max(), np.abs(got3 - ref).max(), nb)


# ---------------------------------------------------------------------------------------------- attention backward
def _mfma_fast(a_frag, b_frag, c_frag):
    """mfma_32x32x16 with the lane loops vectorised (same layouts)."""
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    A = np.zeros((32, 16)); Bm = np.zeros((16, 32))
    for j in range(8):
        A[l31, 8 * hi + j] = a_frag[:, j]
        Bm[8 * hi + j, l31] = b_frag[:, j]
    Dm = A @ Bm
    out = c_frag.copy()
    for r in range(16):
        out[:, r] += Dm[(r & 3) + 8 * (r >> 2) + 4 * hi, l31]
    return out


def _stage_rows(X, r0, DP):
    """load_rows + write_rowmajor of csrc/attn_bwd.hip: a [64, DP] LDS image, zeros past the end of X / past D."""
    img = np.zeros((64, DP))
    n = max(0, min(64, X.shape[0] - r0))
    img[:n, :X.shape[1]] = X[r0:r0 + n]
    return img


def _tr_a_frag(img, ks, dt):
    """tr_frag of csrc/attn_bwd.hip: A-operand fragment of img^T (img = row-major [64, DP] tile) for k-step ks / column tile
    dt, through the transposing LDS read with the kernel's per-lane source addresses (pitch = DP + 8 elements)."""
    DP = img.shape[1]
    P = DP + 8
    flat = np.zeros(64 * P)
    for r in range(64):
        flat[r * P: r * P + DP] = img[r]
    lanes = np.arange(64); hi = lanes >> 5; i16 = lanes & 15; g1 = (lanes >> 4) & 1
    addr = (16 * ks + 4 * hi + (i16 >> 2)) * P + 32 * dt + 16 * g1 + 4 * (lanes & 3)
    return np.concatenate([ds_read_tr16_b64(flat, addr), ds_read_tr16_b64(flat, addr + 8 * P)], 1)


def _own_frags(X, rows, ok, DP):
    """load_own: B-operand fragments [KSTEPS, 64, 8] of the row each lane owns."""
    lanes = np.arange(64); hi = lanes >> 5
    f = np.zeros((DP // 16, 64, 8))
    for kk in range(DP // 16):
        for l in range(64):
            d = 16 * kk + 8 * hi[l]
            if ok[l] and d < X.shape[1]:
                f[kk, l] = X[rows[l], d: d + 8]
    return f


def _a_frag(img, row0, col0):
    """A operand read: lane l takes img[row0 + (l & 31)][col0 + 8 (l >> 5) .. + 7]."""
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    return np.stack([img[row0 + l31[l], col0 + 8 * hi[l]: col0 + 8 * hi[l] + 8] for l in range(64)])


def _store_own(acc, scale, D, DP):
    """store_own: [32, D] from the lanes' accumulator columns."""
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    out = np.zeros((32, DP))
    for dt in range(DP // 32):
        for g in range(4):
            for l in range(64):
                d = 32 * dt + 8 * g + 4 * hi[l]
                out[l31[l], d: d + 4] = acc[dt][l, 4 * g: 4 * g + 4] * scale
    return out[:, :D]


def attention_bwd_dq_wave_sim(Q, K, V, O, dO, lse2, scale, DP):
    """One wave of k_attn_bwd_dq<DP> (csrc/attn_bwd.hip): its 32 query rows Q, O, dO [<=32, D] against all of K, V
    [Skv, D]; lse2 = rowmax + log2(rowsum) of the scaled scores in the log2 domain.  Returns dQ [n, D] and delta [n]."""
    n, D = Q.shape
    Skv = K.shape[0]
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    ok = l31 < n
    rows = np.minimum(l31, n - 1)
    qf, dof, of = (_own_frags(X, rows, ok, DP) for X in (Q, dO, O))
    part = (dof * of).sum((0, 2))
    delta_q = part + part[lanes ^ 32]
    lse_q = np.where(ok, lse2[rows], 0.0)
    sl2 = scale * 1.4426950408889634
    DT, KSTEPS = DP // 32, DP // 16
    dqT = np.zeros((DT, 64, 16))
    for j in range((Skv + 63) // 64):
        kb, vb = _stage_rows(K, 64 * j, DP), _stage_rows(V, 64 * j, DP)
        dsf = [None] * 4
        for t in range(2):
            s = np.zeros((64, 16)); dp = np.zeros((64, 16))
            for kk in range(KSTEPS):
                s = _mfma_fast(_a_frag(kb, 32 * t, 16 * kk), qf[kk], s)
            for kk in range(KSTEPS):
                dp = _mfma_fast(_a_frag(vb, 32 * t, 16 * kk), dof[kk], dp)
            p = np.exp2(s * sl2 - lse_q[:, None])
            ds = p * (dp - delta_q[:, None])
            dsf[2 * t], dsf[2 * t + 1] = ds[:, :8], ds[:, 8:]
        for dt in range(DT):
            for ks in range(4):
                dqT[dt] = _mfma_fast(_tr_a_frag(kb, ks, dt), dsf[ks], dqT[dt])
    return _store_own(dqT, scale, D, DP)[:n], delta_q[:n]


def attention_bwd_dkv_wave_sim(Q, K, V, dO, lse2, delta, scale, DP):
    """One wave of k_attn_bwd_dkv<DP>: its 32 key rows K, V [<=32, D] against all query rows Q, dO [Sq, D]."""
    n, D = K.shape
    Sq = Q.shape[0]
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31
    ok = l31 < n
    rows = np.minimum(l31, n - 1)
    kf, vf = _own_frags(K, rows, ok, DP), _own_frags(V, rows, ok, DP)
    sl2 = scale * 1.4426950408889634
    DT, KSTEPS = DP // 32, DP // 16
    dvT = np.zeros((DT, 64, 16)); dkT = np.zeros((DT, 64, 16))
    for j in range((Sq + 63) // 64):
        q0 = 64 * j
        qb, dob = _stage_rows(Q, q0, DP), _stage_rows(dO, q0, DP)
        lds_l = np.array([lse2[q0 + r] if q0 + r < Sq else np.inf for r in range(64)])
        lds_d = np.array([delta[q0 + r] if q0 + r < Sq else 0.0 for r in range(64)])
        pf, dsf = [None] * 4, [None] * 4
        for t in range(2):
            s = np.zeros((64, 16)); dp = np.zeros((64, 16))
            for kk in range(KSTEPS):
                s = _mfma_fast(_a_frag(qb, 32 * t, 16 * kk), kf[kk], s)
            for kk in range(KSTEPS):
                dp = _mfma_fast(_a_frag(dob, 32 * t, 16 * kk), vf[kk], dp)
            for g in range(4):
                for e in range(4):
                    row = 32 * t + 8 * g + 4 * hi + e
                    p = np.exp2(s[:, 4 * g + e] * sl2 - lds_l[row])
                    s[:, 4 * g + e] = p
                    dp[:, 4 * g + e] = p * (dp[:, 4 * g + e] - lds_d[row])
            pf[2 * t], pf[2 * t + 1] = s[:, :8], s[:, 8:]
            dsf[2 * t], dsf[2 * t + 1] = dp[:, :8], dp[:, 8:]
        for dt in range(DT):
            for ks in range(4):
                dvT[dt] = _mfma_fast(_tr_a_frag(dob, ks, dt), pf[ks], dvT[dt])
                dkT[dt] = _mfma_fast(_tr_a_frag(qb, ks, dt), dsf[ks], dkT[dt])
    return _store_own(dkT, scale, D, DP)[:n], _store_own(dvT, 1.0, D, DP)[:n]


# ---------------------------------------------------------------------------------------------- conv weight gradient
def ds_read_tr16_b64(lds, addr):
    """ds_read_b64_tr_b16 as measured on gfx950 (tools/tr_probe.cpp): lane 16 g + i receives, for j = 0..3, element (i & 3)
    of the 4 consecutive 16-bit elements addressed by lane 16 g + 4 j + (i >> 2).  lds: flat array, addr [64] element indices."""
    out = np.zeros((64, 4))
    for l in range(64):
        g, i = l >> 4, l & 15
        for j in range(4):
            out[l, j] = lds[addr[16 * g + 4 * j + (i >> 2)] + (i & 3)]
    return out


def conv_wgrad_workgroup_sim(x, dy, stride, co0=0, ci0=0):
    """One workgroup of k_conv3x3_wgrad (csrc/conv_wgrad.hip) over all chunks: x [B,H,W,Cin], dy [B,Ho,Wo,Cout] ->
    dW[co0:co0+64, 3, 3, ci0:ci0+64].  Follows the kernel's staging, per-lane transposing-read addresses and fragment order."""
    B, H, W, Cin = x.shape
    _, Ho, Wo, Cout = dy.shape
    PITCH = 96                                   # elements (192 bytes)
    rpc = 64 // Wo
    rows_in, cols_in = stride * (rpc - 1) + 3, stride * (Wo - 1) + 3
    lanes = np.arange(64); hi = lanes >> 5; l31 = lanes & 31; i16 = lanes & 15; g1 = (lanes >> 4) & 1
    out = np.zeros((64, 9, 64))
    for wave in range(4):
        wm, wn = wave >> 1, wave & 1
        off_a = np.zeros((4, 2, 64), dtype=np.int64); off_b = np.zeros((4, 2, 64), dtype=np.int64)
        for ks in range(4):
            for r in range(2):
                p = 16 * ks + 8 * hi + 4 * r + (i16 >> 2)
                off_a[ks, r] = p * PITCH + wm * 32 + 16 * g1 + 4 * (i16 & 3)
                yl, xl = p // Wo, p % Wo
                off_b[ks, r] = ((yl * stride) * cols_in + xl * stride) * PITCH + wn * 32 + 16 * g1 + 4 * (i16 & 3)
        acc = np.zeros((9, 64, 16))
        for c in range(B * Ho // rpc):
            b, y0 = c // (Ho // rpc), (c % (Ho // rpc)) * rpc
            dy_img = np.zeros(64 * PITCH); x_img = np.zeros(rows_in * cols_in * PITCH)
            flat = dy[b].reshape(Ho * Wo, Cout)
            for px in range(64):
                dy_img[px * PITCH: px * PITCH + 64] = flat[y0 * Wo + px, co0:co0 + 64]
            for pix in range(rows_in * cols_in):
                ry, rx = pix // cols_in, pix % cols_in
                yin, xin = stride * y0 - 1 + ry, rx - 1
                if 0 <= yin < H and 0 <= xin < W:
                    x_img[pix * PITCH: pix * PITCH + 64] = x[b, yin, xin, ci0:ci0 + 64]
            af = [np.concatenate([ds_read_tr16_b64(dy_img, off_a[ks, 0]), ds_read_tr16_b64(dy_img, off_a[ks, 1])], 1)
                  for ks in range(4)]
            for t in range(9):
                tap = ((t // 3) * cols_in + (t % 3)) * PITCH
                for ks in range(4):
                    bf = np.concatenate([ds_read_tr16_b64(x_img, tap + off_b[ks, 0]), ds_read_tr16_b64(x_img, tap + off_b[ks, 1])], 1)
                    acc[t] = _mfma_fast(af[ks], bf, acc[t])
        for t in range(9):
            for r in range(16):
                co = wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * hi
                out[co, t, wn * 32 + l31] = acc[t][:, r]
    return out.reshape(64, 3, 3, 64)
