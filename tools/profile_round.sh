#!/bin/bash
# Profile the bench configuration on the GPU box (run through gpurun from the repo root):
#   bash tools/profile_round.sh <tag>
# Writes everything under gpurun_out/prof_<tag>/; tools/profile_collect.py turns it into profiles/ files.
# rocprofv3 passes are separate (kernel stats; FETCH_SIZE; WRITE_SIZE; L2 hit/miss; SQ busy/MFMA) and never
# combine --pmc with a trace domain other than --kernel-trace.
set -u
TAG=${1:-v3}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
python3 bench.py --steps 20 --warmup 5 > "$OUT/bench.json" 2> "$OUT/bench.err" || exit 1
echo "bench done: $(cut -c1-200 "$OUT/bench.json")"
python3 bench.py --lanes 1 --cpu-seconds 0 --no-extra-legs > "$OUT/bench_lanes1.json" 2>> "$OUT/bench.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_lanes1" -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extra-legs --lanes 1 > "$OUT/bench_rocprof_lanes1.json" 2> "$OUT/stats_lanes1.err" || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/stats_lanes4" -- python3 bench.py --steps 5 --warmup 2 --cpu-seconds 0 --no-extra-legs > "$OUT/bench_rocprof_lanes4.json" 2> "$OUT/stats_lanes4.err" || exit 1
echo "kernel stats done"
PM="python3 bench.py --steps 2 --warmup 1 --prewarm 0 --chunks 5120 --cpu-seconds 0 --no-extra-legs --lanes 1"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$OUT/pmc_fetch" -- $PM > /dev/null 2> "$OUT/pmc_fetch.err" || exit 1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$OUT/pmc_write" -- $PM > /dev/null 2> "$OUT/pmc_write.err" || exit 1
rocprofv3 --kernel-trace --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d "$OUT/pmc_l2" -- $PM > /dev/null 2> "$OUT/pmc_l2.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d "$OUT/pmc_sq" -- $PM > /dev/null 2> "$OUT/pmc_sq.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma" -- $PM > /dev/null 2> "$OUT/pmc_mfma.err" || exit 1
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d "$OUT/pmc_mfma_wide" -- $PM --knob wide_gemm=1 > /dev/null 2> "$OUT/pmc_mfma_wide.err" || exit 1
echo "pmc passes done"
python3 tools/pmc_summary.py "$OUT/pmc_fetch" "$OUT/pmc_write" "$OUT/pmc_l2" "$OUT/pmc_sq" "$OUT/pmc_mfma" > "$OUT/pmc_summary.txt" 2>&1
find "$OUT" -name "*kernel_stats.csv" | head
