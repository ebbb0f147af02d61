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
