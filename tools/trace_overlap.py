#!/usr/bin/env python3
"""What the lanes' kernels do to each other: from a rocprofv3 --kernel-trace CSV of the 4-lane batch loop, per kernel the mean duration
under contention, its workgroups, and the CU-time it holds (duration x min(workgroups x share of a CU, 256) / 256), plus the batch period.
usage: trace_overlap.py <kernel_trace.csv> [skip_first_n_batches]"""
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
def short(n):
    for k in ("k_stft_pair", "k_conv_gemm_f4s", "k_conv_gemm_g4", "k_conv_gemm_s3w", "k_conv_gemm_s3", "k_head_reduce", "k_rms_chunk_sums_q"):
        if k in n: return k
    return n[:30]
ev = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(1, int(r["Workgroup_Size_X"])) * max(1, int(r.get("Grid_Size_Y", 1) or 1))) for r in rows]
ev.sort()
st = [e for e in ev if e[2] == "k_stft_pair" and e[3] >= 32]
st = st[skip:]
t0, t1 = st[0][0], st[-1][0]
period = (t1 - t0) / (len(st) - 1) / 1e3
print("batches %d, period %.1f us" % (len(st), period))
agg = collections.defaultdict(lambda: [0, 0.0, 0])
for s, e, n, g in ev:
    if s < t0 or s >= t1: continue
    a = agg[(n, g)]; a[0] += 1; a[1] += (e - s) / 1e3
for (n, g), (c, d, _) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print("  %-22s wgs %5d  n %4d  mean %7.1f us   per batch %6.1f us x wgs = %8.0f wg.us" % (n, g, c, d / c, d / (len(st) - 1), d / (len(st) - 1) * g))
# concurrency: time-average of the number of kernels in flight and of the workgroups they ask for
pts = []
for s, e, n, g in ev:
    if s < t0 or s >= t1: continue
    pts.append((s, 1, g)); pts.append((e, -1, -g))
pts.sort()
k = w = 0; last = pts[0][0]; ak = aw = 0.0; idle = 0.0
for t, dk, dg in pts:
    ak += k * (t - last); aw += w * (t - last)
    if k == 0: idle += t - last
    last = t; k += dk; w += dg
T = pts[-1][0] - pts[0][0]
print("  kernels in flight %.2f, workgroups asked for %.0f (time averages); nothing in flight %.1f %% of the time" % (ak / T, aw / T, 100 * idle / T))
