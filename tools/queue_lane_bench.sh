#!/bin/bash
# bench.py under (GPU_MAX_HW_QUEUES, lanes) pairs, plain and under torchrun (RCCL initialised), with the host-buffer and from-files legs:
# which stream -> hardware-queue packing the encode lanes want.  GPU box:  bash tools/queue_lane_bench.sh
line() { python3 -c "import sys,json; d=json.loads(sys.stdin.readline()); print(round(d['value']), d['step_ms']['median'], {k: round(d[k]['value']) for k in ('pcie','from_files') if k in d and d[k]})"; }
B="--steps 20 --warmup 5 --cpu-seconds 0 --no-detect"
for cfg in "8 4" "4 8" "4 12" "8 4" "4 8"; do set -- $cfg; echo "queues $1 lanes $2"; GPU_MAX_HW_QUEUES=$1 python3 bench.py $B --lanes $2 2>/dev/null | line; done
for cfg in "8 4" "4 8"; do set -- $cfg; echo "torchrun, queues $1 lanes $2"
  GPU_MAX_HW_QUEUES=$1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port $((29500 + $1 + $2)) bench.py --gpus 1 $B --lanes $2 2>/dev/null | line; done
for cfg in "8 4" "4 8"; do set -- $cfg; echo "pcie + files legs, queues $1 lanes $2"; GPU_MAX_HW_QUEUES=$1 python3 bench.py $B --lanes $2 --pcie --files 2000 2>/dev/null | line; done
