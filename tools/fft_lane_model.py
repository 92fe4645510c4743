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
