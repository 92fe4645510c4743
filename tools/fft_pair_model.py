"""Lane / register model of the pair-of-frames STFT kernel (k_stft_pair, csrc/k_frontend_pair.hip).

Two consecutive real frames a, b of one row (n_fft = 2048) are transformed as ONE complex 2048-point FFT of
z = w (a + i b) on one 64-lane wave, 32 complex points per lane.  The model moves values exactly as the kernel does
(which lane, which register, which LDS word) and checks the result against numpy's rfft of both frames; it was written
before the kernel.  Run: python tools/fft_pair_model.py

Index maps (N = 2048, lane l, p(l) = (l & 15) + 16 (l >> 5), b4 = (l >> 4) & 1, dw(l) = p + 32 b4, col(l) = 2 p + b4):
  load     lane l, dword j' (0..15): samples n = 128 j' + 2 dw(l) + e, e = 0, 1    -> seq[e][m = j']
           b4 = 0: n = 64 (2 j') + (2 p + e)        : the EVEN-j half of column 2 p + e
           b4 = 1: n = 64 (2 j' + 1) + (2 p + e)    : the ODD-j half of column 2 p + e
  dft16    per sequence; then v_permlane16_swap(seq[0] of the odd 16-lane rows <-> seq[1] of the even rows):
           every lane holds E (seq 0) and O (seq 1) of ONE column col(l); lanes 0..31 own columns 0..31 (conflict-free
           exchange stores; the first version paired lanes l, l + 32: a half wave owned the even columns only)
  combine  Y[k + 16 p] = E[k] +- W32^k O[k]                                         -> arr[p][POS(k)]
  twiddle  Y'[k1] = Y[k1] W_2048^(col(l) k1)
  exchange T[k1][col]; lane 2 k1' + h reads T[k1'][32 h + mm], mm = 0..31 (b128 rows)
  stage 2  radix-2 DIF across the lane pair (h), twiddle W64^mm on h = 1, DFT32 over mm:
           X[k1' + 32 h + 64 k2'], k2' = k + 16 p at arr[p][POS(k)]
  split    lane' = k1' + 32 h owns bins k = lane' + 64 i (i = 0..15, arr[0]); partners N - k come from arr[1] of
           the lane with lane' -> (64 - lane') & 63 through a 17 x 64 tile
"""
import numpy as np

N = 2048
HOP = 384


def pos16(k):
    return ((k >> 2) & 3) | ((k & 3) << 2)


def dft16_inplace(v):
    """v[16] complex -> v[pos16(k)] = X[k] (the register placement of the kernel's dft16)."""
    X = np.fft.fft(v)
    out = np.empty(16, complex)
    for k in range(16):
        out[pos16(k)] = X[k]
    return out


def dft32_arrays(a0, a1):
    """a[p][m] = x[2 m + p] -> (arr0, arr1) with Y[k + 16 p] at arr[p][pos16(k)]."""
    E = dft16_inplace(a0)
    O = dft16_inplace(a1)
    r0 = np.empty(16, complex)
    r1 = np.empty(16, complex)
    for k in range(16):
        q = pos16(k)
        t = np.exp(-2j * np.pi * k / 32) * O[q]
        r0[q] = E[q] + t
        r1[q] = E[q] - t
    return r0, r1


def pair_of_lane(l):
    return (l & 15) + 16 * (l >> 5)


def col_of_lane(l):
    return 2 * pair_of_lane(l) + ((l >> 4) & 1)


def wave_pair_power(row, s0):
    """power spectra (x4) of frames starting at s0 and s0 + HOP of `row` (already padded), as the wave computes them."""
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
    # ---- load + window
    seq = np.zeros((64, 2, 16), complex)
    for l in range(64):
        for jp in range(16):
            for e in range(2):
                n = 128 * jp + 2 * (pair_of_lane(l) + 32 * ((l >> 4) & 1)) + e
                seq[l, e, jp] = w[n] * (row[s0 + n] + 1j * row[s0 + HOP + n])
    # ---- dft16 per sequence
    d = np.zeros_like(seq)
    for l in range(64):
        for e in range(2):
            d[l, e] = dft16_inplace(seq[l, e])
    # ---- permlane16_swap(vdst = seq0, src0 = seq1): vdst's odd rows of 16 lanes <-> src0's even rows
    E = d[:, 0].copy()
    O = d[:, 1].copy()
    for r in (0, 2):
        for i in range(16):
            E[16 * (r + 1) + i], O[16 * r + i] = d[16 * r + i, 1].copy(), d[16 * (r + 1) + i, 0].copy()
    # now lane l: E = even-j half, O = odd-j half of column col_of_lane(l)
    arr = np.zeros((64, 2, 16), complex)
    for l in range(64):
        for k in range(16):
            q = pos16(k)
            t = np.exp(-2j * np.pi * k / 32) * O[l][q]
            arr[l, 0, q] = E[l][q] + t
            arr[l, 1, q] = E[l][q] - t
    # ---- twiddle + exchange tile T[k1][col]
    T = np.zeros((32, 64), complex)
    for l in range(64):
        c = col_of_lane(l)
        for p in range(2):
            for k in range(16):
                k1 = k + 16 * p
                T[k1, c] = arr[l, p, pos16(k)] * np.exp(-2j * np.pi * ((c * k1) % N) / N)
    # ---- stage 2
    out = np.zeros((64, 2, 16), complex)
    for l in range(64):
        k1p, h = l >> 1, l & 1
        y = T[k1p, 32 * h:32 * h + 32]                       # own values y[mm]
        part = T[k1p, 32 * (1 - h):32 * (1 - h) + 32]        # the lane pair partner's
        u = (y + part) if h == 0 else (part - y)             # pv + sgn * v
        tw = np.exp(-2j * np.pi * np.arange(32) / 64) if h == 1 else np.ones(32)
        u = u * tw
        a0, a1 = u[0::2], u[1::2]                            # mm = 2 m + p
        out[l, 0], out[l, 1] = dft32_arrays(a0, a1)
    # ---- check: X[k1' + 32 h + 64 k2']
    z = w * (row[s0:s0 + N] + 1j * row[s0 + HOP:s0 + HOP + N])
    Z = np.fft.fft(z)
    for l in range(64):
        k1p, h = l >> 1, l & 1
        for p in range(2):
            for k in range(16):
                kk = k1p + 32 * h + 64 * (k + 16 * p)
                assert abs(out[l, p, pos16(k)] - Z[kk]) < 1e-9 * (1 + abs(Z).max()), (l, p, k)
    # ---- split: partner tile rows r = k2' - 16 (0..15) + row 16 = Z[0] of lane' 0
    Zp = np.zeros((17, 64), complex)
    for l in range(64):
        lp = (l >> 1) + 32 * (l & 1)
        for r in range(16):
            Zp[r, lp] = out[l, 1, pos16(r)]
        if lp == 0:
            Zp[16, 0] = out[l, 0, pos16(0)]
    Pa = np.zeros(N // 2 + 1)
    Pb = np.zeros(N // 2 + 1)
    for l in range(64):
        lp = (l >> 1) + 32 * (l & 1)
        colp = (64 - lp) & 63
        rofs = 1 if lp == 0 else 0
        for i in range(16):
            zk = out[l, 0, pos16(i)]
            pk = Zp[15 - i + rofs, colp]
            ar, ai = zk.real + pk.real, zk.imag - pk.imag
            br, bi = zk.imag + pk.imag, pk.real - zk.real
            Pa[lp + 64 * i] = ar * ar + ai * ai
            Pb[lp + 64 * i] = br * br + bi * bi
        if lp == 0:                                        # bin 1024: its own partner
            zk = out[l, 1, pos16(0)]
            Pa[1024] = 4 * zk.real ** 2
            Pb[1024] = 4 * zk.imag ** 2
    return Pa, Pb


def main():
    rng = np.random.default_rng(0)
    row = rng.standard_normal(N + 4 * HOP)
    for s0 in (0, 128, 768):
        Pa, Pb = wave_pair_power(row, s0)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)
        ra = np.abs(np.fft.rfft(w * row[s0:s0 + N])) ** 2
        rb = np.abs(np.fft.rfft(w * row[s0 + HOP:s0 + HOP + N])) ** 2
        ea = np.abs(Pa / 4 - ra).max() / ra.max()
        eb = np.abs(Pb / 4 - rb).max() / rb.max()
        print(f"s0={s0}: max |P/4 - rfft power| / max = {ea:.2e} (frame a), {eb:.2e} (frame b)")
        assert ea < 1e-12 and eb < 1e-12
    print("pair model OK")


if __name__ == "__main__":
    main()


# ----------------------------------------------------------------------------------------------------------------------
# n_fft = 1024 (k_stft_pair<..., LP = 32>): a pair of frames is one complex 1024-point FFT on a HALF wave (32 lanes x 32 points),
# a wave transforms two pairs.  lane' l (0..31): samples 64 j' + 2 l + e; l < 16 even-j halves of columns 2 l + e, l >= 16 the odd-j
# halves of columns 2 (l - 16) + e; v_permlane16_swap pairs them; col(l) = 2 (l & 15) + (l >> 4).  Stage 2 needs no cross-lane step:
# lane k1' reads the whole exchange row T[k1'][0..31] and transforms it: X[k1' + 32 k2'].  Partners N - k: lane (32 - k1') & 31,
# register 31 - i (k1' >= 1) / 32 - i (k1' = 0).
def half_wave_pair_power(row, s0, hop=256):
    n_ = 1024
    w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(n_) / n_)
    seq = np.zeros((32, 2, 16), complex)
    for l in range(32):
        for jp in range(16):
            for e in range(2):
                n = 64 * jp + 2 * l + e
                seq[l, e, jp] = w[n] * (row[s0 + n] + 1j * row[s0 + hop + n])
    d = np.zeros_like(seq)
    for l in range(32):
        for e in range(2):
            d[l, e] = dft16_inplace(seq[l, e])
    E, O = d[:, 0].copy(), d[:, 1].copy()
    for i in range(16):                                   # v_permlane16_swap(vdst = seq0, src0 = seq1): vdst[16 + i] <-> src0[i]
        E[16 + i], O[i] = d[i, 1].copy(), d[16 + i, 0].copy()
    T = np.zeros((32, 32), complex)
    for l in range(32):
        c = 2 * (l & 15) + (l >> 4)
        r0, r1 = np.empty(16, complex), np.empty(16, complex)
        for k in range(16):
            q = pos16(k)
            t = np.exp(-2j * np.pi * k / 32) * O[l][q]
            r0[q], r1[q] = E[l][q] + t, E[l][q] - t
        for p, arr in enumerate((r0, r1)):
            for k in range(16):
                k1 = k + 16 * p
                T[k1, c] = arr[pos16(k)] * np.exp(-2j * np.pi * ((c * k1) % n_) / n_)
    out = np.zeros((32, 2, 16), complex)
    for l in range(32):
        u = T[l]
        out[l, 0], out[l, 1] = dft32_arrays(u[0::2], u[1::2])
    z = w * (row[s0:s0 + n_] + 1j * row[s0 + hop:s0 + hop + n_])
    Z = np.fft.fft(z)
    for l in range(32):
        for p in range(2):
            for k in range(16):
                assert abs(out[l, p, pos16(k)] - Z[l + 32 * (k + 16 * p)]) < 1e-9 * (1 + abs(Z).max())
    Zp = np.zeros((17, 32), complex)
    for l in range(32):
        for r in range(16):
            Zp[r, l] = out[l, 1, pos16(r)]
        if l == 0:
            Zp[16, 0] = out[0, 0, pos16(0)]
    Pa, Pb = np.zeros(n_ // 2 + 1), np.zeros(n_ // 2 + 1)
    for l in range(32):
        colp, rofs = (32 - l) & 31, (1 if l == 0 else 0)
        for i in range(16):
            zk, pk = out[l, 0, pos16(i)], Zp[15 - i + rofs, colp]
            Pa[l + 32 * i] = (zk.real + pk.real) ** 2 + (zk.imag - pk.imag) ** 2
            Pb[l + 32 * i] = (zk.imag + pk.imag) ** 2 + (pk.real - zk.real) ** 2
        if l == 0:
            zk = out[0, 1, pos16(0)]
            Pa[512], Pb[512] = 4 * zk.real ** 2, 4 * zk.imag ** 2
    return Pa, Pb


def main2():
    rng = np.random.default_rng(1)
    row = rng.standard_normal(1024 + 8 * 256)
    for s0 in (0, 512, 1280):
        Pa, Pb = half_wave_pair_power(row, s0)
        w = 0.5 - 0.5 * np.cos(2 * np.pi * np.arange(1024) / 1024)
        ra = np.abs(np.fft.rfft(w * row[s0:s0 + 1024])) ** 2
        rb = np.abs(np.fft.rfft(w * row[s0 + 256:s0 + 256 + 1024])) ** 2
        ea, eb = np.abs(Pa / 4 - ra).max() / ra.max(), np.abs(Pb / 4 - rb).max() / rb.max()
        print(f"n_fft 1024, s0={s0}: {ea:.2e} (frame a), {eb:.2e} (frame b)")
        assert ea < 1e-12 and eb < 1e-12
    print("half-wave pair model OK")


if __name__ == "__main__":
    main2()
