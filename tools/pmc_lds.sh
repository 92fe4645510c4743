set -u
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_lds
mkdir -p $OUT
PM="python3 bench.py --steps 2 --warmup 1 --prewarm 0 --chunks 5120 --cpu-seconds 0 --lanes 1"
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL --output-format csv -d $OUT/a -- $PM > /dev/null 2> $OUT/a.err || echo "pass a failed"
rocprofv3 --kernel-trace --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD --output-format csv -d $OUT/b -- $PM > /dev/null 2> $OUT/b.err || echo "pass b failed"
rocprofv3 --kernel-trace --pmc SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/c -- $PM > /dev/null 2> $OUT/c.err || echo "pass c failed"
python3 tools/pmc_summary.py $OUT/a $OUT/b $OUT/c > $OUT/summary.txt 2>&1
grep -i "stft\|==" $OUT/summary.txt
tail -3 $OUT/a.err $OUT/b.err $OUT/c.err | grep -i "error\|invalid\|not" | head
