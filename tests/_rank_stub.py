"""A stand-in for bench.py's ranks on the CPU (tests/test_distributed_cpu.py): every rank joins a gloo group, runs one
"step" (an all-gather of its rank number) and rank 0 prints ONE JSON line - what launcher.launch_ranks must forward."""
import json
import os
import sys

import torch
import torch.distributed as dist

rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
fail_rank = int(sys.argv[1]) if len(sys.argv) > 1 else -1
if rank == fail_rank:
    sys.exit(7)
dist.init_process_group("gloo", rank=rank, world_size=world)
got = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
dist.all_gather(got, torch.tensor([rank], dtype=torch.int64))
print(f"rank {rank} chatter on stdout")            # only rank 0's stdout may reach the parent's stdout
if rank == 0:
    print(json.dumps({"world_size": dist.get_world_size(), "ranks": [int(t.item()) for t in got],
                      "local_rank": int(os.environ["LOCAL_RANK"]), "local_world_size": int(os.environ["LOCAL_WORLD_SIZE"])}), flush=True)
dist.barrier()
dist.destroy_process_group()
