#!/bin/bash
# BASELINE configs[3] on ITS OWN preset (bench-1s-128: n_fft 1024 / hop 256 / 128 mel bands, 96 x 128 image), bf16 encoder and fp32:
# bench lines, rocprofv3 kernel stats (1 lane) and PMC passes (MFMA busy, FETCH / WRITE).  GPU box, through gpurun:
#   bash tools/profile_config4.sh <tag>      -> gpurun_out/prof_config4_<tag>/
set -u
TAG=${1:-x}
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/prof_config4_$TAG
mkdir -p $OUT
B="python3 bench.py --preset bench-1s-128 --cpu-seconds 0 --no-detect"
$B --bf16 > $OUT/bench_bf16.json 2> $OUT/bench.err || exit 1
$B > $OUT/bench_f32.json 2>> $OUT/bench.err || exit 1
echo "bench: bf16 $(python3 -c "import json;print(json.load(open('$OUT/bench_bf16.json'))['value'])") f32 $(python3 -c "import json;print(json.load(open('$OUT/bench_f32.json'))['value'])")"
PM="$B --bf16 --steps 2 --warmup 1 --prewarm 0 --chunks 5120 --lanes 1"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_bf16 -- $B --bf16 --steps 5 --warmup 2 --lanes 1 > /dev/null 2> $OUT/stats_bf16.err || echo "stats failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_f32 -- $B --steps 5 --warmup 2 --lanes 1 > /dev/null 2> $OUT/stats_f32.err || echo "stats f32 failed"
rocprofv3 --kernel-trace --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_mfma -- $PM > /dev/null 2> $OUT/pmc_mfma.err || echo "mfma failed"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- $PM > /dev/null 2> $OUT/pmc_fetch.err || echo "fetch failed"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- $PM > /dev/null 2> $OUT/pmc_write.err || echo "write failed"
python3 tools/pmc_summary.py $OUT/pmc_mfma $OUT/pmc_fetch $OUT/pmc_write > $OUT/pmc_summary.txt 2>&1
find $OUT/stats_bf16 -name "*kernel_stats.csv" -exec head -12 {} \;
