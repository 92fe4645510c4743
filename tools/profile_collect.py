#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (written by tools/profile_round.sh on the GPU box) into the files kept under profiles/:

    python tools/profile_collect.py v3 r01

  profiles/<round>_bench_<tag>.json                      un-profiled bench line (default lanes) + the 1-lane line
  profiles/<round>_bench_kernel_stats_<tag>_lanes{1,4}.csv   rocprofv3 --kernel-trace --stats summaries
  profiles/<round>_pmc_<tag>.txt                         mean counter values per kernel and grid size (separate --pmc passes)
  profiles/<round>_traffic_<tag>.json                    per stage: FETCH_SIZE / WRITE_SIZE KB per launch -> bytes, L2 hit rate
                                                         (bench.py copies the dominant kernel's figure into roofline.traffic)
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
tag, rnd = (sys.argv[1] if len(sys.argv) > 1 else "v3"), (sys.argv[2] if len(sys.argv) > 2 else "r01")
src, dst = ROOT / "gpurun_out" / f"prof_{tag}", ROOT / "profiles"


def stage_of(kernel: str, grid: int) -> str:
    k = kernel.split("(")[0].replace("void amph::", "").replace("amph::", "")
    if k.startswith("k_rms_fold_apply"): return "rms_apply"             # second kernel of the chunked K0 path (bench's stage "rms_scale" = both)
    if k.startswith("k_rms_scale") or k.startswith("k_rms_chunk_sums"): return "rms_scale"     # (round 4: the chunk sums are the whole stage - fold + apply live in k_stft_pair)
    if k.startswith("k_stft_mel") or k.startswith("k_stft_pair"): return "stft_mel"
    if k.startswith("k_dbnorm"): return "dbnorm"
    if k.startswith("k_conv_gemm_f3") or k.startswith("k_conv_gemm_f4") or (k.startswith("k_conv_gemm<") and "true" in k): return "conv2"
    if k.startswith("k_conv_gemm_g4<64"): return "conv3"
    if k.startswith("k_conv_gemm_g4<128"): return "conv4"
    if k.startswith("k_conv_gemm_s3w"): return {32768: "conv4_wide"}.get(grid, f"gemm_wide_grid{grid}")   # 128 x 128 tiles (multi-lane calls)
    if k.startswith("k_conv_gemm_p3") or k.startswith("k_conv_gemm_s3") or k.startswith("k_conv_gemm<"):
        return {131072: "conv3", 65536: "conv4", 16384: "head_gemm"}.get(grid, f"gemm_grid{grid}")
    if k.startswith("k_head_reduce"): return "head_reduce"
    if k.startswith("k_conv_first"): return "conv1"
    return k


def counters(d):
    by = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(str(d / "*" / "*_counter_collection.csv")):
        for r in csv.DictReader(open(f)):
            if "rocclr" in r["Kernel_Name"]:
                continue
            by[(r["Kernel_Name"], int(r["Grid_Size"]))][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return by


lines, traffic = [], collections.defaultdict(dict)
for name in ("pmc_fetch", "pmc_write", "pmc_l2", "pmc_sq", "pmc_mfma", "pmc_mfma_wide"):
    by = counters(src / name)
    if name == "pmc_mfma_wide":                          # the same pass with --knob wide_gemm=1: only the kernel it adds
        by = {kg: cs for kg, cs in by.items() if stage_of(*kg) == "conv4_wide"}
        if not by:
            continue
    lines.append(f"== {name}   (rocprofv3 --kernel-trace --pmc ..., python3 bench.py --steps 2 --warmup 1 --chunks 5120 --cpu-seconds 0 --lanes 1)")
    # full-size launches only: the largest grid of each stage (the last batch of 5120 chunks is a full 256 too)
    best = {}
    for (k, g), cs in by.items():
        st = stage_of(k, g)
        n = len(next(iter(cs.values())))
        if st not in best or n > best[st][2]:
            best[st] = (k, g, n, cs)
    for st, (k, g, n, cs) in sorted(best.items()):
        short = k.split("(")[0].replace("void amph::", "").replace("amph::", "")
        vals = {c: sum(v) / len(v) for c, v in sorted(cs.items())}
        lines.append(f"  {st:12s} {short:28s} grid {g:>8d} x{n:<3d} " + "  ".join(f"{c}={v:.4g}" for c, v in vals.items()))
        for c, v in vals.items():
            traffic[st][c] = v
(dst / f"{rnd}_pmc_{tag}.txt").write_text("\n".join(lines) + "\n")

out = {"_about": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (separate passes, KB per launch of a 256-chunk "
                 "batch; bench configuration, 1 lane).  traffic_bytes = (FETCH_SIZE + WRITE_SIZE) * 1024 as counted. gfx950 "
                 "under-counts wide (16 B/lane) streaming reads by 2x (MI355X_MICROARCH.md, HBM): fetch_bytes_x2_upper_bound "
                 "is the figure with every fetched byte doubled.", "kernels": {}}
for st, c in sorted(traffic.items()):
    if "FETCH_SIZE" in c and "WRITE_SIZE" in c:
        e = {"fetch_kb": round(c["FETCH_SIZE"], 1), "write_kb": round(c["WRITE_SIZE"], 1),
             "traffic_bytes": int((c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024),
             "fetch_bytes_x2_upper_bound": int((2 * c["FETCH_SIZE"] + c["WRITE_SIZE"]) * 1024)}
        if "TCC_HIT_sum" in c:
            e["l2_hit_rate"] = round(c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"]), 4)
        out["kernels"][st] = e
(dst / f"{rnd}_traffic_{tag}.json").write_text(json.dumps(out, indent=1) + "\n")

# matrix-pipe work per launch (bench.py's mixed_roofline): MFMA instructions, the cycles the pipe was busy, the launch's GPU-active cycles
mf = {"_about": "rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE (one pass; per launch of a "
                "256-chunk batch, bench configuration, 1 lane).  SQ_INSTS_MFMA counts wave instructions; every MFMA of the default path is a "
                "v_mfma_f32_32x32x16_f16 on two-term fp16 splits of fp32 operands (32 matrix-pipe cycles on its SIMD); SQ_VALU_MFMA_BUSY_CYCLES is summed over the chip's 1024 SIMDs; "
                "GRBM_GUI_ACTIVE is summed over the 8 XCDs.", "kernels": {}}
for st, c in sorted(traffic.items()):
    if "SQ_INSTS_MFMA" in c:
        mf["kernels"][st] = {"mfma_insts": round(c["SQ_INSTS_MFMA"], 1), "mfma_busy_cycles": round(c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0), 1),
                             "mfma_mops_f32": round(c.get("SQ_INSTS_VALU_MFMA_MOPS_F32", 0.0), 1), "gui_active_sum": round(c.get("GRBM_GUI_ACTIVE", 0.0), 1)}
(dst / f"{rnd}_mfma_{tag}.json").write_text(json.dumps(mf, indent=1) + "\n")
mf["_source"] = f"profiles/{rnd}_mfma_{tag}.json"
(dst / "mfma_latest.json").write_text(json.dumps(mf, indent=1) + "\n")

shutil.copy(dst / f"{rnd}_traffic_{tag}.json", dst / "traffic_latest.json")
tl = json.loads((dst / "traffic_latest.json").read_text())
tl["_source"] = f"profiles/{rnd}_traffic_{tag}.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes; corrected = FETCH_SIZE x 2 + WRITE_SIZE)"
(dst / "traffic_latest.json").write_text(json.dumps(tl, indent=1) + "\n")

# per-stage kernel durations of FULL-SIZE launches from the 1-lane kernel trace (what bench.py's isolated HIP-event figure must agree with)
dur = collections.defaultdict(lambda: collections.defaultdict(list))
wg_threads = {}
for f in glob.glob(str(src / "stats_lanes1" / "*" / "*_kernel_trace.csv")):
    for r in csv.DictReader(open(f)):
        if "rocclr" in r["Kernel_Name"]:
            continue
        g = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
        st_ = stage_of(r["Kernel_Name"], g)
        dur[st_][g].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
        wg_threads[(st_, g)] = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
avg_us, n_l, wgs = {}, {}, {}
for st, by_g in dur.items():
    if st.startswith("gemm_grid"):                          # GEMM launches of the partial batch
        continue
    g = max(by_g, key=lambda k: len(by_g[k]))            # the full 256-chunk batches (one partial batch per step has its own grid)
    avg_us[st] = round(sum(by_g[g]) / len(by_g[g]), 3)
    n_l[st] = len(by_g[g])
    wgs[st] = {"workgroups": g // max(1, wg_threads[(st, g)]), "threads": wg_threads[(st, g)]}
if "rms_apply" in avg_us and "rms_scale" in avg_us:      # the stage bench.py times is the pair
    avg_us["rms_scale_sum_kernel"] = avg_us["rms_scale"]
    avg_us["rms_scale"] = round(avg_us["rms_scale"] + avg_us["rms_apply"], 3)
kd = {"_source": f"profiles/{rnd}_bench_kernel_stats_{tag}_lanes1.csv run (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 5 --warmup 2 "
                 f"--cpu-seconds 0 --lanes 1): mean End - Start of the full-size launches of each stage, microseconds",
      "avg_us": avg_us, "launches": n_l, "grid": wgs}
(dst / f"{rnd}_kernel_durations_{tag}.json").write_text(json.dumps(kd, indent=1) + "\n")
(dst / "kernel_durations_latest.json").write_text(json.dumps(kd, indent=1) + "\n")

for lanes in (1, 4):
    fs = glob.glob(str(src / f"stats_lanes{lanes}" / "*" / "*_kernel_stats.csv"))
    if fs:
        shutil.copy(fs[0], dst / f"{rnd}_bench_kernel_stats_{tag}_lanes{lanes}.csv")
    f = src / f"bench_rocprof_lanes{lanes}.json"
    if f.exists():
        shutil.copy(f, dst / f"{rnd}_bench_under_rocprof_{tag}_lanes{lanes}.json")
shutil.copy(src / "bench.json", dst / f"{rnd}_bench_{tag}.json")
if (src / "bench_lanes1.json").exists():
    shutil.copy(src / "bench_lanes1.json", dst / f"{rnd}_bench_{tag}_lanes1.json")
print(open(dst / f"{rnd}_pmc_{tag}.txt").read())
print(open(dst / f"{rnd}_traffic_{tag}.json").read())
