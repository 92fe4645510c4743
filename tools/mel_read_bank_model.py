#!/usr/bin/env python3
"""LDS bank model of the mel phase of k_stft_pair (csrc/k_frontend.hip): 16 filters per round on 16 lane quads, 16 bins per quad and step,
the pair's power spectra as P2[k] = (|A[k]|^2, |B[k]|^2) 8-byte elements, weights zero padded per filter (amph_api.hip build_frontend).
Counts the ds_read_b128 array cycles per pair for (a) the kernel's lane -> bin map (lane q of a quad: bins 4q..4q+3 as two reads) and
(b) a map whose reads are contiguous per quad (bins 2q, 2q+1 | 8+2q, 8+2q+1) and (c) that map with the filters placed on the quads and their
start bins moved so that the four quads of a lane group tile the bank row - the search of build_pair_mel (csrc/amph_api.hip), restated -
with the real Slaney filter bounds of the bench preset.  CPU only."""
import itertools
import sys
import numpy as np

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS += [[l + 32 for l in g] for g in B128_GROUPS]


def b128_cycles(addr):
    tot = 0
    for g in B128_GROUPS:
        per_bank = {}
        for l in g:
            a = addr[l]
            if a is None:
                continue
            for d in range(4):
                per_bank.setdefault((a // 4 + d) % 64, set()).add(a // 4 + d)
        tot += max((len(v) for v in per_bank.values()), default=1)
    return tot


def slaney_bounds(sr, n_fft, n_mels, fmin, fmax):
    def hz_to_mel(f):
        f = np.asarray(f, dtype=np.float64)
        m = f / (200.0 / 3)
        lin = f >= 1000.0
        return np.where(lin, 15.0 + np.log(np.maximum(f, 1e-9) / 1000.0) / (np.log(6.4) / 27.0), m)

    def mel_to_hz(m):
        m = np.asarray(m, dtype=np.float64)
        return np.where(m >= 15.0, 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)), m * (200.0 / 3))
    pts = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    freqs = np.arange(1 + n_fft // 2) * sr / n_fft
    out = []
    for m in range(n_mels):
        nz = np.nonzero((freqs > pts[m]) & (freqs < pts[m + 2]))[0]
        out.append((int(nz[0]), int(nz[-1])) if len(nz) else (-1, -1))
    return out


GROUP_QUADS = [[0, 3, 5, 6], [1, 2, 4, 7], [8, 11, 13, 14], [9, 10, 12, 15]]


def place_round(fl, rs):
    """16 (first, last) bins -> per filter (quad, start) with every lane group's starts / 4 mod 8 = {0, 2, 4, 6} or {1, 3, 5, 7}; None: no such placement"""
    cands = []
    for f, l in fl:
        lo = max(0, l - 16 * rs + 1)
        cands.append(sorted({(st // 4) % 8 for st in range(f & ~3, lo - 1, -4)}))

    def solve(rem, groups):
        if not rem:
            return groups
        a = rem[0]
        for trio in itertools.combinations(rem[1:], 3):
            g = (a,) + trio
            for par in (0, 1):
                for perm in itertools.permutations([par, par + 2, par + 4, par + 6]):
                    if all(perm[j] in cands[g[j]] for j in range(4)):
                        got = solve([x for x in rem if x not in g], groups + [(g, perm)])
                        if got is not None:
                            return got
                        break
        return None
    sol = solve(list(range(16)), [])
    if sol is None:
        return None
    out = [None] * 16
    for gi, (g, perm) in enumerate(sol):
        for j in range(4):
            f = fl[g[j]][0]
            st = f & ~3
            while (st // 4) % 8 != perm[j]:
                st -= 4
            out[g[j]] = (GROUP_QUADS[gi][j], st)
    return out


def main(fmin=150.0, fmax=None):
    sr, n_fft, n_mels = 22050, 2048, 64          # PRESETS["bench-1s"]: fmin 150, fmax sr / 2
    fmax = sr / 2 if fmax is None else fmax
    F = 1 + n_fft // 2
    bounds = slaney_bounds(sr, n_fft, n_mels, fmin, fmax)
    start, steps, wofs, wn = [], [], [], 0
    for m0 in range(0, n_mels, 16):
        rs = max((bounds[m][1] - (bounds[m][0] & ~3) + 16) // 16 for m in range(m0, min(m0 + 16, n_mels)) if bounds[m][0] >= 0)
        for m in range(m0, min(m0 + 16, n_mels)):
            start.append(0 if bounds[m][0] < 0 else bounds[m][0] & ~3)
            steps.append(rs)
            wofs.append(wn)
            wn += 16 * rs + (16 if rs % 2 == 0 else 0)
    res = {}
    for name, xmap in (("kernel (bins 4q .. 4q+3: reads at 32 q and 32 q + 16)", lambda q: (32 * q, 32 * q + 16)),
                       ("contiguous per quad (bins 2q, 2q+1 | 8+2q, 8+2q+1: reads at 16 q and 64 + 16 q)", lambda q: (16 * q, 64 + 16 * q))):
        w_c = x_c = n_rd = 0
        for r in range(n_mels // 16):
            for i in range(steps[16 * r]):
                m = [16 * r + (l >> 2) for l in range(64)]
                w_c += b128_cycles([4 * (wofs[m[l]] + 16 * i + 4 * (l & 3)) for l in range(64)])
                for part in range(2):
                    x_c += b128_cycles([8 * (start[m[l]] + 16 * i) + xmap(l & 3)[part] for l in range(64)])
                n_rd += 1
        res[name] = (w_c, x_c, n_rd)
        print(f"{name}: {n_rd} steps per pair; weight reads {w_c} array cycles (conflict-free {4 * n_rd}), power reads {x_c} (conflict-free {8 * n_rd})")
    # (c) the placed arrangement
    w_c = x_c = n_rd = 0
    wn = 0
    for r in range(n_mels // 16):
        rs = steps[16 * r]
        placed = place_round(bounds[16 * r:16 * r + 16], rs)
        assert placed is not None, f"round {r}: no conflict-free placement"
        st_q, wo_q = [0] * 16, [0] * 16
        for j, (quad, st) in enumerate(placed):
            st_q[quad] = st
        for t in range(16):
            wo_q[t] = wn
            wn += 16 * rs + (16 if rs % 2 == 0 else 0)
        for i in range(rs):
            w_c += b128_cycles([4 * (wo_q[l >> 2] + 16 * i + 4 * (l & 3)) for l in range(64)])
            for part in range(2):
                x_c += b128_cycles([8 * (st_q[l >> 2] + 16 * i) + 16 * (l & 3) + 64 * part for l in range(64)])
            n_rd += 1
    print(f"contiguous per quad + placed filters (build_pair_mel): weight reads {w_c} array cycles (conflict-free {4 * n_rd}), power reads {x_c} (conflict-free {8 * n_rd})")
    res["placed"] = (w_c, x_c, n_rd)
    assert x_c == 8 * n_rd and w_c == 4 * n_rd
    print("steps per round:", [steps[16 * r] for r in range(n_mels // 16)], " filter starts of round 0:", start[:16])


    return res


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 150.0, float(sys.argv[2]) if len(sys.argv) > 2 else None)
