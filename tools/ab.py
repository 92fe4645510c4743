#!/usr/bin/env python3
"""A/B two builds of libamph.so in ONE process on ONE device with interleaved rounds
(guide §5.4 rule 24: never compare timings taken on different devices/processes).

    tools/ab.py build_ab/libamph_A.so amphibian_vae_latent_detector_amd/libamph.so [--lanes=1] [--batch=256]

Prints per-stage isolated launch time (lanes = 1, every stage bracketed by HIP events)
and end-to-end throughput for each library, and checks that both give identical latents.
"""
import ctypes as C
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS, SPECS  # noqa: E402
from amphibian_vae_latent_detector_amd import _native as N  # noqa: E402
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state  # noqa: E402
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16  # noqa: E402


def load(path):
    lib = C.CDLL(str(path))
    for name in N.EXPORTED_SYMBOLS:
        if hasattr(lib, name):
            fn, ref = getattr(lib, name), getattr(N.lib, name)
            fn.restype, fn.argtypes = ref.restype, ref.argtypes
    return lib


def main():
    paths = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if a.startswith("--"))
    batch, chunks = int(opts.get("batch", 256)), int(opts.get("chunks", 10000))
    lanes, rounds = int(opts.get("lanes", 4)), int(opts.get("rounds", 4))
    cfg = PRESETS[opts.get("preset", "bench-1s")]
    spec = SPECS[opts.get("spec", "thesis-shape")](cfg.target_frames, cfg.n_mels)
    state = random_init_state(spec, 0)
    pool = synth_batch_pcm16(list(range(1000)), cfg.n_samples, cfg.sr)
    pcm = torch.from_numpy(np.tile(pool, ((chunks + 999) // 1000, 1))[:chunks]).to("cuda:0")
    libs = [load(p) for p in paths]
    sess, mus = [], []
    for lib in libs:
        N.lib = lib
        s = HipEncoder(spec, state, "cuda:0", max_batch=batch).session(cfg)
        sess.append(s)
        mus.append(torch.empty((chunks, spec.latent_dim), dtype=torch.float32, device="cuda:0"))
    stage_ms = [dict() for _ in libs]
    thr = [[] for _ in libs]
    for r in range(rounds):
        for i, lib in enumerate(libs):
            N.lib = lib
            s = sess[i]
            s.configure("lanes", 1)
            s.encode(pcm, apply_rms=True, out_mu=mus[i])
            s.profile_reset()
            s.profile_enable(True)
            s.encode(pcm, apply_rms=True, out_mu=mus[i])
            torch.cuda.synchronize()
            for st in s.profile_read():
                if st["launches"]:
                    stage_ms[i].setdefault(st["stage"], []).append(st["total_ms"] / st["launches"])
            s.profile_enable(False)
            s.profile_reset()
            s.configure("lanes", lanes)
            s.encode(pcm, apply_rms=True, out_mu=mus[i])
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                s.encode(pcm, apply_rms=True, out_mu=mus[i])
            torch.cuda.synchronize()
            thr[i].append(3 * chunks / (time.perf_counter() - t0))
    print(f"batch={batch} chunks={chunks} lanes={lanes} rounds={rounds}   (stage = isolated us/launch, median)")
    names = list(stage_ms[0])
    print("%-14s" % "stage" + "".join("%14s" % Path(p).name[:13] for p in paths))
    for nm in names:
        print("%-14s" % nm + "".join("%14.1f" % (1e3 * np.median(stage_ms[i].get(nm, [np.nan]))) for i in range(len(libs))))
    print("%-14s" % "sum" + "".join("%14.1f" % (1e3 * sum(np.median(v) for v in stage_ms[i].values())) for i in range(len(libs))))
    print("%-14s" % "k chunks/s" + "".join("%14.1f" % (np.median(thr[i]) / 1e3) for i in range(len(libs))))
    for i in range(1, len(libs)):
        same = torch.equal(mus[0], mus[i])
        print(f"latents of {Path(paths[i]).name} vs {Path(paths[0]).name}: " +
              ("bit-identical" if same else "max |d| = %.3e" % float((mus[0] - mus[i]).abs().max())))
    sys.stdout.flush()
    import os
    os._exit(0)      # two copies of the HIP module in one process: skip their unload order at exit


if __name__ == "__main__":
    main()
