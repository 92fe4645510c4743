#!/usr/bin/env python3
"""Accuracy model of k_stft_pair's FFT with its two DFT stages on the f16 matrix cores (docs/HISTORY.md, round 4: sized, not built).

A pair of frames a, b is one complex 2048-point FFT of z = w (a + i b), 2048 = 32 x 64 (n = 64 n1 + n2, k = k1 + 32 k2):
  stage 1   Y[k1][n2] = sum_n1 F32[k1][n1] z[n1][n2]                 (real form: a 64 x 64 matrix times [Zr; Zi])
  twiddle   Y'[k1][n2] = Y[k1][n2] W_2048^(k1 n2)                    (fp32 VALU)
  stage 2   X[k1 + 32 k2] = sum_n2 Y'[k1][n2] F64[n2][k2]            (real form again)
Each matrix product as the MFMA form would do it: both operands split x = h + 2^-11 l (h = fp16(x), l = fp16((x - h) 2^11)), the three
term pairs h h', h l', l h' accumulated in fp32 (products of fp16 values are exact in fp32; numpy's fp32 matmul stands in for the MFMA's
accumulation order), the cross terms scaled once.  Compared with the float64 FFT and with an fp32 FFT (what the kernel does today)
on synthetic PCM16 chunks: relative L2 error of the power spectrum rows and of the log-mel-like quantity 10 log10(max(P, 1e-10 max P))."""
import numpy as np

rng = np.random.default_rng(0)
N, N1, N2 = 2048, 32, 64


def split2(x):
    h = x.astype(np.float16)
    l = ((x - h.astype(np.float32)) * np.float32(2048.0)).astype(np.float16)
    return h.astype(np.float32), l.astype(np.float32)


def mm_split(a, b):                       # a @ b with both operands as two fp16 terms, fp32 accumulation
    ah, al = split2(a.astype(np.float32))
    bh, bl = split2(b.astype(np.float32))
    return (ah @ bh) + np.float32(1.0 / 2048.0) * ((ah @ bl) + (al @ bh))


def cmm_split(Ar, Ai, Br, Bi):            # complex product in real form: [Ar -Ai; Ai Ar] [Br; Bi]
    A = np.block([[Ar, -Ai], [Ai, Ar]]).astype(np.float32)
    B = np.concatenate([Br, Bi], 0).astype(np.float32)
    C = mm_split(A, B)
    return C[: Ar.shape[0]], C[Ar.shape[0]:]


def dft(n):
    k = np.arange(n)
    w = np.exp(-2j * np.pi * np.outer(k, k) / n)
    return w.real, w.imag


F32r, F32i = dft(N1)
F64r, F64i = dft(N2)
k1, n2 = np.arange(N1)[:, None], np.arange(N2)[None, :]
TW = np.exp(-2j * np.pi * k1 * n2 / N).astype(np.complex64)


def fft_mfma(z):                          # z: complex64 [2048]
    Z = z.reshape(N1, N2)                 # [n1][n2]
    Yr, Yi = cmm_split(F32r, F32i, Z.real, Z.imag)                       # [k1][n2]
    Y = (Yr + 1j * Yi).astype(np.complex64) * TW
    # stage 2: X[k1][k2] = sum_n2 Y'[k1][n2] F64[n2][k2]  ->  transpose: (F64^T)(Y'^T)
    Xr, Xi = cmm_split(F64r.T, F64i.T, Y.real.T.copy(), Y.imag.T.copy())   # [k2][k1]
    return (Xr + 1j * Xi).astype(np.complex64).reshape(-1)              # index k2 * 32 + k1 = k


def synth(n, kind):
    t = np.arange(n) / 22050.0
    if kind == 0:
        x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.01 * rng.standard_normal(n)
    elif kind == 1:
        x = 0.2 * rng.standard_normal(n)
    else:
        x = 0.5 * np.sin(2 * np.pi * (500 + 3000 * t) * t) * np.exp(-3 * t) + 0.001 * rng.standard_normal(n)
    return np.clip(np.rint(x * 32767), -32768, 32767).astype(np.int16)


win = ((0.5 - 0.5 * np.cos(2 * np.pi * np.arange(N) / N)) / 32768.0).astype(np.float32)      # (the kernel's window table carries the 2^-15: |z| <= 1, the
# stage-1 sums stay below 32 - without it they pass fp16's range and the second split overflows)
errs = {"mfma": [], "fp32": []}
dberr = {"mfma": [], "fp32": []}
for kind in range(3):
    for rep in range(8):
        pcm = synth(N + 384, kind)
        a, b = pcm[:N].astype(np.float32), pcm[384:384 + N].astype(np.float32)
        z = (win * a + 1j * (win * b)).astype(np.complex64)
        ref = np.fft.fft(z.astype(np.complex128))
        cand = {"mfma": fft_mfma(z), "fp32": np.fft.fft(z).astype(np.complex64) if False else None}
        # an fp32 FFT: numpy computes in double, so emulate with scipy's float32 path
        import scipy.fft as sf
        cand["fp32"] = sf.fft(z)
        for name, Zc in cand.items():
            def powers(Zc):
                Zc = Zc.astype(np.complex128)
                k = np.arange(1, N // 2)
                A = (Zc[k] + np.conj(Zc[N - k])) / 2
                B = (Zc[k] - np.conj(Zc[N - k])) / 2j
                return np.abs(A) ** 2, np.abs(B) ** 2
            Pa, Pb = powers(Zc)
            Ra, Rb = powers(ref)
            errs[name].append(np.sqrt(((Pa - Ra) ** 2).sum() / (Ra ** 2).sum()))
            db = lambda P, M: 10 * np.log10(np.maximum(P, 1e-8 * M))
            dberr[name].append(np.abs(db(Pa, Ra.max()) - db(Ra, Ra.max())).max())
for name in errs:
    print("%-5s power spectrum rel. L2 error: median %.2e  max %.2e | max |dB error| down to -80 dB: median %.2e  max %.2e" %
          (name, np.median(errs[name]), np.max(errs[name]), np.median(dberr[name]), np.max(dberr[name])))
