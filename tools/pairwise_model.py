#!/usr/bin/env python3
"""numpy model of the K0 reduction in csrc/k_frontend.hip: numpy's float32 summation
(np.add.reduce on a contiguous array) evaluated as a parallel tree.

  the array is consumed in ufunc-buffer chunks of 8192 elements; each chunk is summed
  pairwise (below) and the chunk sums are accumulated sequentially: ((s0 + s1) + s2) ...
  (checked here against np.sum / np.mean for n up to 960 000).
  within a chunk:

  node(n): n <= 128 -> leaf; else left = n/2 - (n/2)%8, right = n - left
  every leaf slot s in [0, 2^Dmax) walks the tree from the root by its bits; a leaf
  reached at level l < Dmax is owned by the slot whose remaining bits are zero.
  leaf value: 8 strided accumulators r[j] += a[i+j] in increasing i, combined as
  ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)), then the n%8 tail sequentially  (n >= 8)
  heap[(1<<l)+path] holds node values; internal nodes = left + right, deepest level first.
"""
import numpy as np

f = np.float32


def depth_max(n):
    d = 0
    while n > 128:
        n2 = n // 2
        n2 -= n2 % 8
        n = n - n2          # the right child is never smaller
        d += 1
    return d


def walk(n, path, level):
    """(offset, size, reached_level) of the node at `level` along `path` (MSB first), stopping at leaves."""
    off = 0
    for l in range(level):
        if n <= 128:
            return off, n, l
        n2 = n // 2
        n2 -= n2 % 8
        if (path >> (level - 1 - l)) & 1:
            off, n = off + n2, n - n2
        else:
            n = n2
    return off, n, level


def leaf_sum(a):
    n = a.shape[0]
    if n < 8:
        r = f(-0.0)
        for v in a:
            r = f(r + v)
        return r
    r = [f(a[j]) for j in range(8)]
    i = 8
    while i < n - (n % 8):
        for j in range(8):
            r[j] = f(r[j] + a[i + j])
        i += 8
    res = f(f(f(r[0] + r[1]) + f(r[2] + r[3])) + f(f(r[4] + r[5]) + f(r[6] + r[7])))
    while i < n:
        res = f(res + a[i])
        i += 1
    return res


def numpy_sum(a, chunk=8192):
    """np.sum(a) for contiguous float32 `a`, any length."""
    res = None
    for i in range(0, a.shape[0], chunk):
        s = tree_sum(a[i:i + chunk])
        res = s if res is None else f(res + s)
    return f(0.0) if res is None else res


def tree_sum(a):
    """pairwise sum of one chunk (n <= 8192) as leaves + heap."""
    n = a.shape[0]
    D = depth_max(n)
    heap = np.zeros(2 << D, dtype=np.float32)
    for s in range(1 << D):                       # one "slot" per potential deepest leaf
        off, sz, lvl = walk(n, s, D)
        rem = D - lvl
        if s & ((1 << rem) - 1):
            continue                              # an early leaf is owned by the slot with zero remaining bits
        heap[(1 << lvl) + (s >> rem)] = leaf_sum(a[off:off + sz])
    for lvl in range(D - 1, -1, -1):
        for p in range(1 << lvl):
            off, sz, reached = walk(n, p, lvl)
            if reached == lvl and sz > 128:       # internal node that exists
                heap[(1 << lvl) + p] = f(heap[(2 << lvl) + 2 * p] + heap[(2 << lvl) + 2 * p + 1])
    return heap[1]


if __name__ == "__main__":
    rng = np.random.default_rng(0)
    for n in [1, 5, 8, 100, 128, 129, 255, 256, 257, 1000, 3712, 8192, 8193, 22050, 33000, 65536, 144000, 240000, 960000]:
        ok = True
        for t in range(3):
            y = (rng.standard_normal(n) * 0.1).astype(np.float32)
            a = y * y
            ok &= numpy_sum(a) == np.sum(a)
            ok &= f(numpy_sum(a) / f(n)) == np.mean(y ** 2)
        print(n, "chunk depth", depth_max(min(n, 8192)), "OK" if ok else "MISMATCH")
