#!/usr/bin/env python3
"""A/B sweep of libamph tuning knobs in ONE process with interleaved rounds
(guide §5.4 rule 24).  usage: sweep.py key=v1,v2,... [key=...] [--batch 256] [--chunks 10000]"""
import itertools
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS, SPECS  # noqa: E402
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state  # noqa: E402
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if "=" in a]
    opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
    args = [a for a in args if not a.startswith("--")]
    batch = int(opts.get("batch", 256))
    chunks = int(opts.get("chunks", 10000))
    rounds = int(opts.get("rounds", 3))
    steps = int(opts.get("steps", 3))
    rms = bool(int(opts.get("rms", 1)))            # --rms=0: rows used as they are (already normalised files)
    cfg = PRESETS[opts.get("preset", "bench-1s")]
    spec = SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
    if "lib" in opts:                                # a diagnostic build (e.g. build_ab/libamph_skip.so) instead of the shipped library
        import ctypes as C
        from amphibian_vae_latent_detector_amd import _native as N
        lib = C.CDLL(str(Path(opts["lib"]).resolve()))
        for name in N.EXPORTED_SYMBOLS:
            if hasattr(lib, name):
                fn, ref_ = getattr(lib, name), getattr(N.lib, name)
                fn.restype, fn.argtypes = ref_.restype, ref_.argtypes
        N.lib = lib
    enc = HipEncoder(spec, random_init_state(spec, 0), "cuda:0", max_batch=batch)
    sess = enc.session(cfg)
    pool = synth_batch_pcm16(list(range(1000)), cfg.n_samples, cfg.sr)
    pcm = torch.from_numpy(np.tile(pool, ((chunks + 999) // 1000, 1))[:chunks]).to("cuda:0")
    mu = torch.empty((chunks, spec.latent_dim), dtype=torch.float32, device="cuda:0")
    keys = [a.split("=")[0] for a in args]
    grids = [[int(v) for v in a.split("=")[1].split(",")] for a in args]
    combos = list(itertools.product(*grids)) or [()]
    res = {c: [] for c in combos}
    ref = None
    for r in range(rounds):
        for c in combos:
            for k, v in zip(keys, c):
                if k == "abl_skip":
                    N.lib.amph_abl_skip(int(v))
                else:
                    sess.configure(k, v)
            sess.encode(pcm, apply_rms=rms, out_mu=mu)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(steps):
                sess.encode(pcm, apply_rms=rms, out_mu=mu)
            torch.cuda.synchronize()
            res[c].append(chunks * steps / (time.perf_counter() - t0))
            if ref is None:
                ref = mu.clone()
            elif not torch.equal(ref, mu):
                print("!! result changed with", dict(zip(keys, c)))
    if "stages" in opts:
        for c in combos:
            for k, v in zip(keys, c):
                sess.configure(k, v)
            sess.configure("lanes", 1)
            sess.profile_reset(); sess.profile_enable(True)
            sess.encode(pcm, apply_rms=rms, out_mu=mu)
            torch.cuda.synchronize()
            print("  isolated us/launch", dict(zip(keys, c)), {st["stage"]: round(1e3 * st["total_ms"] / st["launches"], 1) for st in sess.profile_read() if st["launches"]})
            sess.profile_enable(False); sess.profile_reset()
    print(f"batch={batch} chunks={chunks} steps={steps} rounds={rounds}")
    for c in combos:
        v = res[c]
        print("  " + " ".join(f"{k}={x}" for k, x in zip(keys, c)) + f"   median {np.median(v)/1e3:8.1f} k chunks/s   max {max(v)/1e3:8.1f}   all {[round(x/1e3) for x in v]}")


if __name__ == "__main__":
    main()
