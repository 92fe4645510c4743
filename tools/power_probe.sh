#!/bin/bash
# Samples power / clocks with rocm-smi while the headline loop runs (tools/sweep.py): is the batch pipeline power- or clock-limited?
# usage (on the GPU box): bash tools/power_probe.sh "<sweep.py arguments>"
OUT=gpurun_out/power_probe.txt
( for i in $(seq 1 40); do rocm-smi --showpower --showclocks --showuse 2>/dev/null | grep -E "Power|sclk|mclk|GPU use|busy" | tr '\n' ';'; echo; sleep 0.5; done ) > $OUT.samples &
SAMP=$!
python tools/sweep.py $1 > $OUT.sweep 2>&1
kill $SAMP 2>/dev/null
wait $SAMP 2>/dev/null
tail -4 $OUT.sweep
sort $OUT.samples | uniq -c | sort -rn | head -12
