"""Folder-level batch encode, sharded across the GPUs of one node.

Replaces the encode loop of latent_space_exploration/08_fit_radial_detector.py:461-520
(species in config order, name-sorted WAVs, optional seeded `random.sample`,
per-file failures counted not fatal, `cache_npz/Z_<root>_<species>.npz` cache).
The reference is a batch-1 single-process loop; here the ordered chunk list is cut
into contiguous per-rank slices (one process per GPU), each rank decodes its WAVs
with libamph's host thread pool, encodes them with the HIP kernels, and ONE
all-gather (RCCL over xGMI when the process group's backend is nccl) concatenates
the latent matrix.  Every chunk is encoded independently of its batch, so the
sharded result is bit-identical to the single-GPU one.
"""
from __future__ import annotations

import math
import os
import random
from pathlib import Path
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.distributed as dist

from . import wavio
from ._native import AMPH_ERR_IO as N_IO
from ._native import AMPH_ERR_UNSUPPORTED as N_UNSUPPORTED
from .config import FrontendConfig


def list_species_wavs(chunks_dir: Path, species_list: Sequence[str], max_per_class: int = 0,
                      skip: Sequence[str] = ()) -> Dict[str, List[Path]]:
    """08_fit_radial_detector.py:461-484.  The caller seeds `random` once
    (`random.seed(seed)`, :374); species in `skip` (served from cache) draw nothing
    from the RNG, exactly as in the reference."""
    out: Dict[str, List[Path]] = {}
    for sp in species_list:
        sp_dir = (Path(chunks_dir) / sp).resolve()
        if sp in skip or not sp_dir.exists():
            continue
        wavs = sorted(sp_dir.glob("*.wav"))
        if not wavs:
            continue
        if max_per_class and len(wavs) > max_per_class:
            wavs = random.sample(wavs, max_per_class)
        out[sp] = wavs
    return out


def plan_shards(n_items: int, world_size: int) -> Tuple[List[Tuple[int, int]], int]:
    """Contiguous slices [r*per, (r+1)*per) with per = ceil(n/world); the tail ranks
    may be short or empty (they pad with dummy rows for the all-gather)."""
    per = max(1, math.ceil(n_items / max(1, world_size)))
    return [(min(n_items, r * per), min(n_items, (r + 1) * per)) for r in range(world_size)], per


def _dist_info(group=None) -> Tuple[int, int]:
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _cgroup_cpus() -> Optional[float]:
    """CPUs' worth of time this process's cgroup may use per period (cgroup v2 cpu.max, v1 cfs quota), None when unlimited / unknown."""
    try:
        txt = Path("/sys/fs/cgroup/cpu.max").read_text().split()
        if txt and txt[0] != "max":
            return float(txt[0]) / float(txt[1])
    except (OSError, ValueError, IndexError):
        pass
    try:
        q = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read_text())
        per = float(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
        if q > 0 and per > 0:
            return q / per
    except (OSError, ValueError):
        pass
    return None


def default_loader_threads(world: int = 1) -> int:
    """Host threads this rank's WAV reader uses when the caller names none: the CPUs this process may actually use - its affinity mask,
    cut to its cgroup's CPU quota where there is one - divided by the ranks that share the node (LOCAL_WORLD_SIZE under
    torch.distributed.run, else the group's size), between 1 and 16.  More threads than CPUs' worth of quota is far worse than too few:
    the 1-GPU box shows 256 CPUs in the mask and gives the job about 16 - 340 k files/s at 8 - 16 threads, 50 k at 32 (the cgroup is
    throttled for the rest of every period), and eight ranks of one node each taking the whole mask only fight over the same cores."""
    import os
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    quota = _cgroup_cpus()
    if quota is not None:
        n_aff = max(1, min(n_aff, int(quota)))
    try:
        local = int(os.environ.get("LOCAL_WORLD_SIZE", "0"))
    except ValueError:
        local = 0
    return max(1, min(16, n_aff // max(1, local if local > 0 else world)))


def encode_paths(encoder, paths: Sequence[Path], cfg: FrontendConfig, *, apply_rms: bool = False,
                 loader_threads: Optional[int] = None, file_batch: int = 4096) -> Tuple[torch.Tensor, np.ndarray]:
    """Decode + encode a list of WAVs on this rank's GPU.
    Returns (Z [n, D] float32 device tensor, status [n] int32; 0 = ok)."""
    if loader_threads is None:
        loader_threads = default_loader_threads(_dist_info()[1])
    sess = encoder.session(cfg)
    n = len(paths)
    Z = torch.zeros((n, sess.latent_dim), dtype=torch.float32, device=sess.device)
    status = np.zeros(n, dtype=np.int32)
    # with RMS normalisation the whole file contributes to the RMS (00 normalises the file,
    # 07 truncates afterwards): read up to 4x the chunk length, enough for any chunked dataset
    max_samples = cfg.n_samples * (4 if apply_rms else 1)
    parr = wavio.PathArray(paths)                            # the C array of paths, once for the whole list
    if apply_rms and n:
        # ... but no more than the longest file holds: a folder of chunk-length files then moves 1x, not 4x, the bytes
        # through memset / parse / PCIe (file size bounds the sample count of a PCM16 file from above).  Sizes by the native
        # thread pool: a Python stat() per file was 5 us x n before the first byte was read
        longest = int(max(0, wavio.file_sizes(parr, n_threads=loader_threads).max())) // 2
        max_samples = max(cfg.n_samples, min(max_samples, (longest + 7) // 8 * 8))      # rows stay 16-byte aligned
    # two reusable pinned slots: a host thread parses batch b + 1 (the C call releases the GIL) while batch b is
    # copied to the device and encoded
    import threading
    # batches ramp up (512, 1024, ... file_batch): the first read is the one nothing overlaps, so it is short
    offs, o, step = [], 0, min(512, file_batch)
    while o < n:
        offs.append(o)
        o += step
        step = min(file_batch, 2 * step)
    ends = offs[1:] + [n]
    results = {}

    with torch.cuda.device(sess.device):                  # pinned slots are allocated by THIS thread, on this rank's device
        slots = [wavio.PINNED.rows(k, min(file_batch, n), max_samples) for k in range(min(2, len(offs)))]

    def read(bi):
        n_ = ends[bi] - offs[bi]
        rows = slots[bi & 1][:n_]
        try:
            results[bi] = (rows,) + wavio.read_many_into(rows, parr, cfg.sr, n_threads=loader_threads, offset=offs[bi])
        except Exception as e:  # noqa: BLE001
            results[bi] = e

    reader = threading.Thread(target=read, args=(0,)) if offs else None
    if reader:
        reader.start()
    for bi, off in enumerate(offs):
        part = paths[off:ends[bi]]
        reader.join()
        got = results.pop(bi)
        if isinstance(got, Exception):
            raise got
        pcm, lengths, st = got
        if bi + 1 < len(offs):
            reader = threading.Thread(target=read, args=(bi + 1,))
            reader.start()
        status[off:off + len(part)] = st
        dev_pcm = torch.from_numpy(pcm).to(sess.device, non_blocking=True)
        sess.encode(dev_pcm, lengths=lengths, apply_rms=apply_rms, out_mu=Z[off:off + len(part)])
        torch.cuda.current_stream(sess.device).synchronize()      # the slot is reused two batches later
        # files the native reader declined (stereo, another subtype, another sample rate): the general loader
        # (channel mean, libsndfile scaling, GPU resampler) one file at a time, as librosa.load would
        for i in np.nonzero(st == N_UNSUPPORTED)[0]:
            try:
                y = wavio.load_float_mono(part[i], cfg.sr, session=sess)[:max_samples]
                mu, _, _ = sess.encode(y[None], lengths=np.array([y.shape[0]], dtype=np.int32), apply_rms=apply_rms)
                Z[off + i] = mu[0]
                status[off + i] = 0
            except Exception:  # noqa: BLE001  (the reference counts the file as failed and continues, 08:504-506)
                status[off + i] = N_IO
    return Z, status


def _pad_block(local: torch.Tensor, per: int) -> torch.Tensor:
    """A rank's rows padded with dummy (zero) rows to the common block height of the all-gather."""
    if local.shape[0] == per:
        return local.contiguous()
    pad = torch.zeros((per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    pad[:local.shape[0]] = local
    return pad


def gather_rows(local: torch.Tensor, per: int, n_total: int, group=None) -> torch.Tensor:
    """All-gather equal-size row blocks (padded to `per` rows) and drop the padding."""
    rank, world = _dist_info(group)
    if not (dist.is_available() and dist.is_initialized()):
        return local[:n_total]
    # (a one-rank group still runs the collective: the exchange step is then exercised on a single-GPU box too)
    full = torch.empty((world * per,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(full, _pad_block(local, per), group=group)
    return full[:n_total]


def encode_resident_shards(sess, pcm: torch.Tensor, n_shards: int, *, apply_rms: bool = True,
                           lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """BASELINE configs[2] on ONE GPU: the HBM-resident chunk matrix `pcm` [n, n_in] (rows in the reference's list
    order) is cut into `n_shards` contiguous slices exactly as `encode_file_list` cuts the path list (plan_shards:
    ceil(n / shards) rows each, tail shards short or empty); every slice is encoded by its own call, as its rank would,
    into a block padded to the common height; the blocks are laid out as all_gather_into_tensor lays them out and the
    padding is dropped.  No collective runs: this is the data path of the 8-GPU job with the exchange step replaced by
    the memory layout it produces.  -> Z [n, D] float32 on the device."""
    n = int(pcm.shape[0])
    shards, per = plan_shards(n, n_shards)
    full = torch.empty((n_shards * per, sess.latent_dim), dtype=torch.float32, device=sess.device)
    for r, (lo, hi) in enumerate(shards):
        if hi > lo:
            local, _, _ = sess.encode(pcm[lo:hi], lengths=lengths[lo:hi] if lengths is not None else None, apply_rms=apply_rms)
        else:
            local = torch.zeros((0, sess.latent_dim), dtype=torch.float32, device=sess.device)
        full[r * per:(r + 1) * per] = _pad_block(local, per)
    return full[:n]


def encode_file_list(encoder, paths: Sequence[Path], cfg: FrontendConfig, *, apply_rms: bool = False, group=None,
                     loader_threads: Optional[int] = None,
                     encode_fn: Optional[Callable[[Sequence[Path]], Tuple[torch.Tensor, np.ndarray]]] = None
                     ) -> Tuple[np.ndarray, np.ndarray]:
    """Sharded encode of an ordered path list -> (Z [n, D] float32, status [n] int32) on every rank.
    `encode_fn(paths) -> (Z tensor, status)` replaces the HIP encode in host-logic tests
    (gloo on CPU); it has no default other than the HIP path."""
    rank, world = _dist_info(group)
    n = len(paths)
    shards, per = plan_shards(n, world)
    lo, hi = shards[rank]
    mine = list(paths[lo:hi])
    if encode_fn is None:
        Zl, st = encode_paths(encoder, mine, cfg, apply_rms=apply_rms, loader_threads=loader_threads)
    else:
        Zl, st = encode_fn(mine)
    stl = torch.from_numpy(np.ascontiguousarray(st, dtype=np.int32)).to(Zl.device)
    Z = gather_rows(Zl, per, n, group)
    S = gather_rows(stl, per, n, group)
    return Z.cpu().numpy().astype(np.float32, copy=False), S.cpu().numpy()


def encode_folder(encoder, chunks_dir: Path, species_list: Sequence[str], cfg: FrontendConfig, *,
                  max_per_class: int = 0, seed: int = 123, cache_dir: Optional[Path] = None, apply_rms: bool = False,
                  group=None, loader_threads: Optional[int] = None, encode_fn=None, verbose: bool = True
                  ) -> Tuple[Dict[str, np.ndarray], Dict[str, int]]:
    """08_fit_radial_detector.py:374,461-520 -> ({species: Z [N_sp, D]}, {species: n_failed}).
    With cache_dir, per-species `Z_<chunks_dir.name>_<sp>.npz` files (keys Z, failed, root)
    are read when present and written (by rank 0) otherwise: the reference's cache format.
    Sharded runs: rank 0 alone decides which species are served from the cache and which files are encoded, and
    broadcasts that plan - every rank then builds the same path list (the all-gather needs equal shapes) whatever its own
    view of the cache directory is; cache files are written under a temporary name and renamed, and a barrier behind
    the writes means a following call on any rank (08 then 08b share cache_npz) finds all of them complete."""
    rank, world = _dist_info(group)
    chunks_dir = Path(chunks_dir).resolve()
    random.seed(seed)
    np.random.seed(seed)
    say = print if (verbose and rank == 0) else (lambda *a, **k: None)

    Z_by: Dict[str, np.ndarray] = {}
    failed_by: Dict[str, int] = {}
    cached: List[str] = []
    wavs_by: Dict[str, List[Path]] = {}
    if cache_dir is not None:
        cache_dir = Path(cache_dir)
    if rank == 0:
        if cache_dir is not None:
            cached = [sp for sp in species_list
                      if (cache_dir / f"Z_{chunks_dir.name}_{sp}.npz").exists() and (chunks_dir / sp).exists()]
        wavs_by = list_species_wavs(chunks_dir, species_list, max_per_class, skip=cached)
    if world > 1:
        plan = [cached, {sp: [str(p) for p in v] for sp, v in wavs_by.items()}]
        dist.broadcast_object_list(plan, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
        cached, wavs_by = plan[0], {sp: [Path(p) for p in v] for sp, v in plan[1].items()}
    for sp in cached:
        cp = cache_dir / f"Z_{chunks_dir.name}_{sp}.npz"
        data = np.load(cp)
        Z_by[sp] = data["Z"].astype(np.float32)
        failed_by[sp] = int(data["failed"]) if "failed" in data else 0
        say(f"🧊 {sp}: cargado cache {cp.name} -> N={Z_by[sp].shape[0]}")
    for sp in species_list:
        if sp not in cached and sp not in wavs_by:
            say(f"⚠️ {sp}: carpeta no existe o sin wavs en {chunks_dir / sp} (se omite).")

    order = [sp for sp in species_list if sp in wavs_by]
    flat: List[Path] = [p for sp in order for p in wavs_by[sp]]
    if flat:
        Z, status = encode_file_list(encoder, flat, cfg, apply_rms=apply_rms, group=group,
                                     loader_threads=loader_threads, encode_fn=encode_fn)
        pos = 0
        for sp in order:
            k = len(wavs_by[sp])
            zs, ss = Z[pos:pos + k], status[pos:pos + k]
            for p, s in zip(wavs_by[sp], ss):
                if s != 0:
                    say(f"⚠️ {sp}: fallo {p.name}: status {int(s)}")
            pos += k
            good = ss == 0
            if not good.any():
                say(f"❌ {sp}: no se pudo codificar nada (se omite).")
                continue
            Zm = np.ascontiguousarray(zs[good], dtype=np.float32)
            Z_by[sp] = Zm
            failed_by[sp] = int((~good).sum())
            say(f"🧪 {sp}: encoded N={Zm.shape[0]} (failed={failed_by[sp]})")
            if cache_dir is not None and rank == 0:
                cache_dir.mkdir(parents=True, exist_ok=True)
                final = cache_dir / f"Z_{chunks_dir.name}_{sp}.npz"
                tmp = cache_dir / f".{final.name}.{os.getpid()}.tmp"
                with open(tmp, "wb") as f:                            # a reader never sees a half-written npz
                    np.savez_compressed(f, Z=Zm, failed=failed_by[sp], root=str(chunks_dir))
                os.replace(tmp, final)
    if world > 1 and cache_dir is not None:
        dist.barrier(group=group)                                     # the cache is complete before ANY rank returns
    # keep config order
    Z_by = {sp: Z_by[sp] for sp in species_list if sp in Z_by}
    return Z_by, failed_by
