#!/usr/bin/env python3
"""Lane-level numpy model of the wave FFT used by csrc/k_frontend.hip.

Validates the index math of the decomposition before it is written in HIP
(there is no GPU in the build container):
  Nc = 256*T_LO complex points per frame, L = 16*T_LO lanes per frame,
  stage A: radix-16 over registers (stride L) + twiddle W_Nc^(t*k1)
  LDS exchange, lane u = (k1 = u // T_LO, t_lo = u % T_LO), reg m -> t = t_lo + T_LO*m
  stage B: radix-16 over registers + twiddle W_L^(t_lo*k2a)
  stage C: radix-T_LO across the T_LO adjacent lanes (xor butterflies, bit-reversed k2b)
  real-FFT post-process: pairs (k, Nc-k) -> power bins k and Nc-k.
"""
import numpy as np


def dft16(v):
    """4x4 decomposition, natural-order output (mirrors the device dft16)."""
    W16 = np.exp(-2j * np.pi * np.arange(16) / 16)
    y = np.zeros((4, 4), complex)
    for j0 in range(4):
        x0, x1, x2, x3 = v[j0], v[j0 + 4], v[j0 + 8], v[j0 + 12]
        a, b, c, d = x0 + x2, x0 - x2, x1 + x3, (x1 - x3) * (-1j)
        y[j0] = [a + c, b + d, a - c, b - d]
    out = np.zeros(16, complex)
    for ka in range(4):
        t = [y[j0][ka] * W16[(j0 * ka) % 16] for j0 in range(4)]
        a, b, c, d = t[0] + t[2], t[0] - t[2], t[1] + t[3], (t[1] - t[3]) * (-1j)
        out[ka + 0], out[ka + 4], out[ka + 8], out[ka + 12] = a + c, b + d, a - c, b - d
    return out


def wave_fft(z, T_LO):
    Nc, L = 256 * T_LO, 16 * T_LO
    assert z.shape == (Nc,)
    WN = lambda p: np.exp(-2j * np.pi * p / Nc)
    WL = lambda p: np.exp(-2j * np.pi * p / L)
    lds = np.zeros((16, L), complex)
    for t in range(L):                                   # stage A, per lane
        a = dft16(np.array([z[t + L * j] for j in range(16)]))
        for k1 in range(16):
            lds[k1, t] = a[k1] * WN(t * k1)
    regs = np.zeros((L, 16), complex)
    for u in range(L):                                   # stage B
        k1, t_lo = u // T_LO, u % T_LO
        b = dft16(np.array([lds[k1, t_lo + T_LO * m] for m in range(16)]))
        for k2a in range(16):
            regs[u, k2a] = b[k2a] * WL(t_lo * k2a)
    Z = np.zeros(Nc, complex)
    for u in range(L):                                   # stage C (xor butterflies)
        pass
    cur = regs.copy()
    if T_LO == 4:
        nxt = np.zeros_like(cur)
        for u in range(L):                               # round 1: partner u^2
            t_lo, p = u % 4, cur[u ^ 2]
            if t_lo < 2:
                nxt[u] = cur[u] + p
            else:
                nxt[u] = (p - cur[u]) * (-1j if t_lo == 3 else 1)
        cur, nxt = nxt, np.zeros_like(cur)
        for u in range(L):                               # round 2: partner u^1
            t_lo, p = u % 4, cur[u ^ 1]
            nxt[u] = cur[u] + p if (t_lo & 1) == 0 else p - cur[u]
        cur = nxt
        k2b_of = lambda t_lo: ((t_lo & 1) << 1) | (t_lo >> 1)
    elif T_LO == 2:
        nxt = np.zeros_like(cur)
        for u in range(L):
            t_lo, p = u % 2, cur[u ^ 1]
            nxt[u] = cur[u] + p if t_lo == 0 else p - cur[u]
        cur = nxt
        k2b_of = lambda t_lo: t_lo
    else:
        k2b_of = lambda t_lo: 0
    for u in range(L):
        k1, t_lo = u // T_LO, u % T_LO
        for k2a in range(16):
            Z[k1 + 16 * (k2a + 16 * k2b_of(t_lo))] = cur[u, k2a]
    return Z


def real_power(x, T_LO):
    """x real, length 2*Nc -> |rfft(x)|^2 (Nc+1 bins) through the packed complex FFT."""
    Nc = 256 * T_LO
    z = x[0::2] + 1j * x[1::2]
    Z = wave_fft(z, T_LO)
    P = np.zeros(Nc + 1)
    for k in range(Nc // 2 + 1):
        zk, zc = Z[k], np.conj(Z[(Nc - k) % Nc])
        xe, xo = 0.5 * (zk + zc), -0.5j * (zk - zc)
        t = np.exp(-2j * np.pi * k / (2 * Nc)) * xo
        P[k] = abs(xe + t) ** 2
        P[Nc - k] = abs(xe - t) ** 2
    return P


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    v = rng.standard_normal(16) + 1j * rng.standard_normal(16)
    print("dft16 err", np.abs(dft16(v) - np.fft.fft(v)).max())
    for T_LO in (1, 2, 4):
        Nc = 256 * T_LO
        z = rng.standard_normal(Nc) + 1j * rng.standard_normal(Nc)
        print("T_LO", T_LO, "fft err", np.abs(wave_fft(z, T_LO) - np.fft.fft(z)).max())
        x = rng.standard_normal(2 * Nc)
        print("       power err", np.abs(real_power(x, T_LO) - np.abs(np.fft.rfft(x)) ** 2).max())


# ---------------------------------------------------------------------------------------------------------------------
# Second data flow (round 2): fewer, wider LDS operations.  Stage A as above.  After the exchange lane u = (k1 = u // T_LO,
# q = u % T_LO) reads 16 CONTIGUOUS points t = 16 q + m of row k1; the radix-T_LO across the T_LO adjacent lanes comes
# FIRST (on raw values), then the twiddle W_L^(m kq), then the radix-16 in registers: lane (k1, q) register r ends up with
# bin k = k1 + 16 (kq + T_LO r), kq = bitrev(q).  The mirror bin Nc - k of every one of a lane's 16 bins sits in ONE other
# lane (register 15 - r): each lane writes its 16 values as one contiguous row, reads its mirror lane's row, and computes the
# power of its own 16 bins - the real-FFT split needs no scattered LDS traffic.
def mirror_lane(u, T_LO):
    L = 16 * T_LO
    k1, q = u // T_LO, u % T_LO
    br = (lambda x: ((x & 1) << 1) | (x >> 1)) if T_LO == 4 else (lambda x: x)
    kq = br(q)
    if k1 != 0:
        return (16 - k1) * T_LO + (T_LO - 1 - q)
    if kq != 0:
        return br(T_LO - kq)            # lane of row k1 = 0 holding kq' = T_LO - kq
    return 0                            # bins 16 T_LO r: mirror register (16 - r) % 16 of the lane itself


def wave_power_v2(x, T_LO):
    Nc, L = 256 * T_LO, 16 * T_LO
    z = x[0::2] + 1j * x[1::2]
    WN = lambda p: np.exp(-2j * np.pi * p / Nc)
    WL = lambda p: np.exp(-2j * np.pi * p / L)
    lds = np.zeros((16, L), complex)
    for t in range(L):                                   # stage A, per lane (unchanged)
        a = dft16(np.array([z[t + L * j] for j in range(16)]))
        for k1 in range(16):
            lds[k1, t] = a[k1] * WN(t * k1)
    br = (lambda v: ((v & 1) << 1) | (v >> 1)) if T_LO == 4 else (lambda v: v)
    cur = np.zeros((L, 16), complex)
    for u in range(L):                                   # contiguous read: t = 16 q + m
        k1, q = u // T_LO, u % T_LO
        cur[u] = lds[k1, 16 * q:16 * q + 16]
    if T_LO == 4:                                        # radix-4 across the quad, xor butterflies -> kq = bitrev(q)
        nxt = np.zeros_like(cur)
        for u in range(L):
            q, p = u % 4, cur[u ^ 2]
            nxt[u] = cur[u] + p if q < 2 else (p - cur[u]) * (-1j if q == 3 else 1)
        cur, nxt = nxt, np.zeros_like(cur)
        for u in range(L):
            q, p = u % 4, cur[u ^ 1]
            nxt[u] = cur[u] + p if (q & 1) == 0 else p - cur[u]
        cur = nxt
    else:
        nxt = np.zeros_like(cur)
        for u in range(L):
            q, p = u % 2, cur[u ^ 1]
            nxt[u] = cur[u] + p if q == 0 else p - cur[u]
        cur = nxt
    X = np.zeros((L, 16), complex)                       # lane rows: X[u][r] = bin k1 + 16 (kq + T_LO r)
    for u in range(L):
        k1, q = u // T_LO, u % T_LO
        kq = br(q)
        X[u] = dft16(np.array([cur[u, m] * WL(m * kq) for m in range(16)]))
    # check the ownership map against numpy
    ref = np.fft.fft(z)
    for u in range(L):
        k1, q = u // T_LO, u % T_LO
        for r in range(16):
            assert abs(X[u, r] - ref[k1 + 16 * (br(q) + T_LO * r)]) < 1e-9
    P = np.zeros(Nc + 1)
    for u in range(L):                                   # split: own row + the mirror lane's row
        k1, q = u // T_LO, u % T_LO
        kq = br(q)
        mu_ = mirror_lane(u, T_LO)
        for r in range(16):
            k = k1 + 16 * (kq + T_LO * r)
            b = X[mu_, (16 - r) % 16] if (k1 == 0 and kq == 0) else X[mu_, 15 - r]
            assert abs(b - ref[(Nc - k) % Nc]) < 1e-9, (u, r)
            a = X[u, r]
            xe, xo = 0.5 * (a + np.conj(b)), -0.5j * (a - np.conj(b))
            w = np.exp(-2j * np.pi * k / (2 * Nc))
            P[k] = abs(xe + w * xo) ** 2
            if k == 0:
                P[Nc] = abs(xe - xo) ** 2
    return P


if __name__ == "__main__":
    rng = np.random.default_rng(1)
    for T_LO in (2, 4):
        x = rng.standard_normal(512 * T_LO)
        print("v2 T_LO", T_LO, "power err", np.abs(wave_power_v2(x, T_LO) - np.abs(np.fft.rfft(x)) ** 2).max(),
              "mirror is an involution:", all(mirror_lane(mirror_lane(u, T_LO), T_LO) == u for u in range(16 * T_LO)))
