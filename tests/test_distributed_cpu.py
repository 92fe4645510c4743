"""N>1 path on CPU: world_size-2 gloo process group, folder sharding + all-gather.
The HIP encode is replaced by an injected encode_fn (the CPU oracle on real WAV files),
which is the only place the product accepts one; the sharded result must equal the
single-process result bit-for-bit."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

from amphibian_vae_latent_detector_amd import PRESETS, SPECS  # noqa: E402
from amphibian_vae_latent_detector_amd import folder as pfolder  # noqa: E402
from amphibian_vae_latent_detector_amd import synth  # noqa: E402
from amphibian_vae_latent_detector_amd.encoder import random_init_state  # noqa: E402

CFG = PRESETS["bench-1s"]
SPECIES = list(synth.SPECIES)


def _oracle_encode_fn():
    from oracle import encoder as oenc
    from oracle import frontend as ofe
    spec = SPECS["tiny"](CFG.target_frames, CFG.n_mels)
    tenc = oenc.build_torch_encoder(spec.to_dict(), random_init_state(spec, seed=0))
    torch.set_num_threads(1)

    def fn(paths):
        Z = np.zeros((len(paths), spec.latent_dim), np.float32)
        st = np.zeros(len(paths), np.int32)
        for i, p in enumerate(paths):
            try:
                Z[i] = oenc.encode_mel_to_latent(tenc, ofe.wav_to_mel(p, **CFG.mel_kwargs()))
            except Exception:  # noqa: BLE001   (per-file failure -> flagged row, batch continues)
                st[i] = 4
        return torch.from_numpy(Z), st
    return fn


def _worker(rank, world, port, root, out_dir, tag=""):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        kw = dict(max_per_class=3, seed=123, cache_dir=Path(out_dir) / "cache", encode_fn=_oracle_encode_fn(), verbose=False)
        Z, failed = pfolder.encode_folder(None, Path(root), SPECIES, CFG, **kw)
        np.savez(Path(out_dir) / f"rank{rank}{tag}.npz", failed=np.array([failed[s] for s in Z]), **Z)
        # straight away again on the same cache_dir (08 then 08b share cache_npz): every rank must find the complete
        # cache rank 0 has just written - no rank may have returned from the first call before the files were in place
        Z2, failed2 = pfolder.encode_folder(None, Path(root), SPECIES, CFG, **kw)
        np.savez(Path(out_dir) / f"rank{rank}{tag}_again.npz", failed=np.array([failed2[s] for s in Z2]), **Z2)
    finally:
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(300)
def test_two_rank_sharded_folder_encode_equals_single_process(tmp_path):
    root = tmp_path / "train_chunks_norm"
    synth.write_synth_folder(root, 19, CFG.n_samples, CFG.sr)        # 5,5,5,4 files; sampling keeps 3 each
    (root / SPECIES[1] / "chunk_000001.wav").write_bytes(b"broken")   # one unreadable file in the sample pool
    single, failed1 = pfolder.encode_folder(None, root, SPECIES, CFG, max_per_class=3, seed=123,
                                            encode_fn=_oracle_encode_fn(), verbose=False)
    assert list(single) == SPECIES and sum(z.shape[0] for z in single.values()) + sum(failed1.values()) == 12
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_worker, args=(2, _free_port(), str(root), str(out)), nprocs=2, join=True)
    r0, r1 = np.load(out / "rank0.npz"), np.load(out / "rank1.npz")
    for sp in SPECIES:
        assert np.array_equal(r0[sp], single[sp]) and np.array_equal(r1[sp], single[sp]), sp
    assert r0["failed"].tolist() == [failed1[s] for s in SPECIES] == r1["failed"].tolist()
    # rank 0 wrote the reference-format cache
    d = np.load(out / "cache" / f"Z_train_chunks_norm_{SPECIES[0]}.npz")
    assert np.array_equal(d["Z"], single[SPECIES[0]])
    assert not [p for p in (out / "cache").iterdir() if p.name.endswith(".tmp")]        # written under a temporary name, renamed
    # the second call of each worker was served from that cache on both ranks
    for r in (0, 1):
        again = np.load(out / f"rank{r}_again.npz")
        for sp in SPECIES:
            assert np.array_equal(again[sp], single[sp]), (r, sp)
    # a partly filled cache: the plan (which species are cached, which files are drawn - the RNG skips cached species,
    # 08_fit_radial_detector.py:468-484) is rank 0's, so both ranks build the same list; equal to one process on the same cache
    import shutil
    (out / "cache" / f"Z_train_chunks_norm_{SPECIES[2]}.npz").unlink()
    shutil.copytree(out / "cache", tmp_path / "cache_copy")
    mixed, failed_m = pfolder.encode_folder(None, root, SPECIES, CFG, max_per_class=3, seed=123, cache_dir=tmp_path / "cache_copy",
                                            encode_fn=_oracle_encode_fn(), verbose=False)
    mp.spawn(_worker, args=(2, _free_port(), str(root), str(out), "_mixed"), nprocs=2, join=True)
    for r in (0, 1):
        got = np.load(out / f"rank{r}_mixed.npz")
        for sp in SPECIES:
            assert np.array_equal(got[sp], mixed[sp]), (r, sp)
        assert got["failed"].tolist() == [failed_m[s] for s in SPECIES]


def _worker8(rank, world, port, root, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), LOCAL_WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        paths = sorted(Path(root).rglob("*.wav"))
        shards, per = pfolder.plan_shards(len(paths), world)
        Z, st = pfolder.encode_file_list(None, paths, CFG, encode_fn=_oracle_encode_fn())
        np.savez(Path(out_dir) / f"w8_rank{rank}.npz", Z=Z, st=st, lo=shards[rank][0], hi=shards[rank][1], per=per,
                 threads=pfolder.default_loader_threads(world))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_eight_ranks_on_thirteen_files_empty_tail_ranks_and_padded_blocks(tmp_path):
    """world_size 8 (the node BASELINE.json's north star names) on 13 files: blocks of ceil(13 / 8) = 2 rows, so ranks 0-5 hold 2
    files, rank 6 one file + one padding row, rank 7 NOTHING (a whole block of padding) - the all-gather still runs on every rank
    and the trimmed result equals one process bit for bit.  One unreadable file: flagged in the gathered status on every rank.
    Also: the loader's thread default splits the affinity mask by LOCAL_WORLD_SIZE."""
    root = tmp_path / "chunks"
    synth.write_synth_folder(root, 13, CFG.n_samples, CFG.sr)
    paths = sorted(root.rglob("*.wav"))
    assert len(paths) == 13
    paths[4].write_bytes(b"broken")
    fn = _oracle_encode_fn()
    Z1, st1 = fn(paths)
    shards, per = pfolder.plan_shards(13, 8)
    assert per == 2 and shards[6] == (12, 13) and shards[7][0] == shards[7][1]          # the last rank owns no file
    out = tmp_path / "out"
    out.mkdir()
    mp.spawn(_worker8, args=(8, _free_port(), str(root), str(out)), nprocs=8, join=True)
    n_aff = len(os.sched_getaffinity(0))
    for r in range(8):
        d = np.load(out / f"w8_rank{r}.npz")
        assert np.array_equal(d["Z"], Z1.numpy()) and np.array_equal(d["st"], st1), r
        assert (int(d["lo"]), int(d["hi"])) == shards[r] and int(d["per"]) == 2
        quota = pfolder._cgroup_cpus()
        cpus = n_aff if quota is None else max(1, min(n_aff, int(quota)))      # (the affinity mask cut to the cgroup's CPU quota)
        assert int(d["threads"]) == max(1, min(16, cpus // 8))
    assert st1[4] != 0 and not st1[np.arange(13) != 4].any()


def test_gather_rows_pads_and_trims_single_process():
    x = torch.arange(12, dtype=torch.float32).reshape(4, 3)
    assert torch.equal(pfolder.gather_rows(x, 4, 3), x[:3])


@pytest.mark.timeout(300)
def test_launcher_starts_eight_ranks_and_forwards_rank_zero(tmp_path):
    """`python bench.py --gpus N` starts its own ranks (launcher.launch_ranks): 8 children over gloo run a stub step; the parent's
    stdout carries rank 0's lines only, the return code is 0; a rank that dies makes the job fail with its code and ends the others."""
    import io
    import json
    from amphibian_vae_latent_detector_amd import launcher
    stub = str(Path(__file__).with_name("_rank_stub.py"))
    out, err = io.StringIO(), io.StringIO()
    rc = launcher.launch_ranks([sys.executable, stub], 8, out=out, err=err)
    assert rc == 0, err.getvalue()
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec == {"world_size": 8, "ranks": list(range(8)), "local_rank": 0, "local_world_size": 8}
    assert "rank 0 chatter" in out.getvalue() and "rank 3 chatter" not in out.getvalue() and "rank 3 chatter" in err.getvalue()
    out2, err2 = io.StringIO(), io.StringIO()
    rc2 = launcher.launch_ranks([sys.executable, stub, "2"], 4, out=out2, err=err2)         # rank 2 exits 7 before the rendezvous
    assert rc2 == 7 and "rank 2 exited with 7" in err2.getvalue()
    assert not launcher.under_launcher({}) and launcher.under_launcher({"RANK": "0", "WORLD_SIZE": "1"})
