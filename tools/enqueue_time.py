#!/usr/bin/env python3
"""How long the host takes to ENQUEUE one encode call (all batches, all lanes) against how long the GPU takes to run it:
if the two are close the call is launch-bound and a larger batch / a captured graph would pay.  GPU box.
usage: enqueue_time.py [--batch=256] [--chunks=10000] [--reps=8]"""
import sys, time
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS, SPECS
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
opts = dict(a.lstrip("-").split("=") for a in sys.argv[1:] if a.startswith("--") and "=" in a)
batch, chunks, reps = int(opts.get("batch", 256)), int(opts.get("chunks", 10000)), int(opts.get("reps", 8))
cfg = PRESETS["bench-1s"]
spec = SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
enc = HipEncoder(spec, random_init_state(spec, 0), "cuda:0", max_batch=batch)
sess = enc.session(cfg)
pool = synth_batch_pcm16(list(range(1000)), cfg.n_samples, cfg.sr)
pcm = torch.from_numpy(np.tile(pool, ((chunks + 999) // 1000, 1))[:chunks]).to("cuda:0")
mu = torch.empty((chunks, spec.latent_dim), dtype=torch.float32, device="cuda:0")
for _ in range(3):
    sess.encode(pcm, apply_rms=True, out_mu=mu)
torch.cuda.synchronize()
enq, tot = [], []
for _ in range(reps):
    t0 = time.perf_counter()
    sess.encode(pcm, apply_rms=True, out_mu=mu)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    enq.append((t1 - t0) * 1e3); tot.append((t2 - t0) * 1e3)
n_b = (chunks + batch - 1) // batch
print(f"batch {batch}, {chunks} chunks = {n_b} batches: host enqueue {np.median(enq):.3f} ms (min {min(enq):.3f}), call + sync {np.median(tot):.3f} ms "
      f"(min {min(tot):.3f}) -> enqueue / total = {np.median(enq) / np.median(tot):.2f}; {np.median(enq) / n_b * 1e3:.1f} us of host time per batch")
