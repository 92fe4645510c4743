"""One process per GPU, started from a plain `python script.py --gpus N`.

The folder job of 08_fit_radial_detector.py:461-520 shards over the GPUs of one node
(SURVEY 8(e)): every rank is its own process with RANK / LOCAL_RANK / WORLD_SIZE /
MASTER_ADDR / MASTER_PORT in its environment, exactly what `python -m
torch.distributed.run --nproc-per-node N` would give it, so a script works under either.
The parent must call this BEFORE anything touches the GPU (it never re-execs itself: the
ranks are child processes, and the parent only waits for them).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
import threading
import time
from typing import Dict, List, Optional, Sequence


def under_launcher(env: Optional[Dict[str, str]] = None) -> bool:
    """True inside a rank (started by launch_ranks or torch.distributed.run)."""
    env = os.environ if env is None else env
    return "RANK" in env and "WORLD_SIZE" in env


def free_port(addr: str = "127.0.0.1") -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind((addr, 0))
        return s.getsockname()[1]


def rank_env(rank: int, world: int, addr: str, port: int, base: Optional[Dict[str, str]] = None) -> Dict[str, str]:
    env = dict(os.environ if base is None else base)
    env.update({"RANK": str(rank), "LOCAL_RANK": str(rank), "WORLD_SIZE": str(world), "LOCAL_WORLD_SIZE": str(world),
                "MASTER_ADDR": addr, "MASTER_PORT": str(port)})
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")      # dmabuf IPC: what RCCL needs on this driver
    env.setdefault("GPU_MAX_HW_QUEUES", "8")               # the 4 encode lanes + RCCL's streams (include/amph.h, "lanes")
    return env


def launch_ranks(cmd: Sequence[str], n: int, *, master_addr: str = "127.0.0.1", master_port: Optional[int] = None,
                 env: Optional[Dict[str, str]] = None, out=None, err=None, poll_s: float = 0.05) -> int:
    """Runs `cmd` n times, rank r with LOCAL_RANK = r.  Rank 0's stdout is forwarded to `out` (default: this process's
    stdout) line by line, the other ranks' stdout goes to `err` with their stderr, so the job still prints what one
    process would.  The first rank that exits non-zero ends the job: the remaining ranks (exactly the PIDs started here)
    are terminated and its code is returned; 0 when every rank returned 0."""
    if n < 1:
        raise ValueError("launch_ranks: n must be >= 1")
    out = sys.stdout if out is None else out
    err = sys.stderr if err is None else err
    port = free_port(master_addr) if master_port is None else int(master_port)
    procs: List[subprocess.Popen] = []
    pumps: List[threading.Thread] = []

    def pump(src, dst):
        for line in iter(src.readline, b""):
            dst.write(line.decode("utf-8", "replace"))
            dst.flush()
        src.close()

    try:
        for r in range(n):
            p = subprocess.Popen(list(cmd), env=rank_env(r, n, master_addr, port, env), stdout=subprocess.PIPE,
                                 stderr=subprocess.PIPE)
            procs.append(p)
            for src, dst in ((p.stdout, out if r == 0 else err), (p.stderr, err)):
                t = threading.Thread(target=pump, args=(src, dst), daemon=True)
                t.start()
                pumps.append(t)
        rc = 0
        live = set(range(n))
        while live and rc == 0:
            for r in sorted(live):
                code = procs[r].poll()
                if code is None:
                    continue
                live.discard(r)
                if code != 0:
                    rc = code if code > 0 else 128 - code
                    err.write(f"[launcher] rank {r} exited with {code}; stopping the other ranks\n")
                    break
            if live and rc == 0:
                time.sleep(poll_s)
        return rc
    finally:
        for p in procs:                    # only reached with live children when a rank failed or the parent is interrupted
            if p.poll() is None:
                p.terminate()
        deadline = time.time() + 10.0
        for p in procs:
            try:
                p.wait(timeout=max(0.1, deadline - time.time()))
            except subprocess.TimeoutExpired:
                p.kill()
                p.wait()
        for t in pumps:
            t.join(timeout=5.0)
