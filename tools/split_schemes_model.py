#!/usr/bin/env python3
"""Representation error of the split-operand GEMM schemes on a conv-like product (CPU, numpy): three bf16 terms per operand with six
products (k_conv_gemm_f3s / _f4s / _g4 / _g5) against two fp16 terms with three products (k_conv_gemm_s3 / _s3w since round 3:
x = h + 2^-11 l, x w = h h' + 2^-11 (h l' + l h')), both against the float64 product of the same fp32 operands, beside the rounding
error of a plain fp32 matmul.  Accumulation is exact here: the number isolates what the split itself loses."""
import numpy as np

rng = np.random.default_rng(0)
K, M = 1152, 512
x = np.maximum(rng.standard_normal((M, K)).astype(np.float32), 0) * rng.uniform(0.01, 3, (M, 1)).astype(np.float32)
w = (rng.standard_normal((K, 64)) * 0.05).astype(np.float32)
ref = x.astype(np.float64) @ w.astype(np.float64)


def bf16(a):
    u = a.astype(np.float32).view(np.uint32).astype(np.uint64)
    u = u + 0x7FFF + ((u >> 16) & 1)
    return ((u >> 16) << 16).astype(np.uint32).view(np.float32)


def split3(a):
    h = bf16(a); r = a - h; m = bf16(r); return h, m, bf16(r - m)


def split2h(a):
    h = a.astype(np.float16).astype(np.float32)
    return h, ((a - h) * 2048).astype(np.float16).astype(np.float32)


d = lambda a, b: a.astype(np.float64) @ b.astype(np.float64)
rel = lambda v: np.linalg.norm(v - ref, axis=1).max() / np.linalg.norm(ref, axis=1).mean()
xh, xm, xl = split3(x); wh, wm, wl = split3(w)
print("bf16 x 3, six products  :", rel(d(xh, wh) + d(xh, wm) + d(xm, wh) + d(xm, wm) + d(xh, wl) + d(xl, wh)))
a, b = split2h(x); c, e = split2h(w)
print("fp16 x 2, three products:", rel(d(a, c) + (d(a, e) + d(b, c)) / 2048))
print("plain fp32 matmul       :", rel((x @ w).astype(np.float64)))
print("largest |x|", float(x.max()), "- fp16 holds 65504; below 6.1e-5 h is subnormal (absolute 6e-8), l picks up the rest to 3e-11")
