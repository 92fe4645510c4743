#!/usr/bin/env python3
"""Compare the two STFT kernels (stft_indep 0 / 1) on the same rows: where and by how much the mel power differs."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "bench-1s"]
s = AmphSession(cfg, None, None, "cuda:0", max_batch=16)
pcm = synth_batch_pcm16(list(range(16)), cfg.n_samples, cfg.sr)
y = torch.from_numpy(pcm.astype(np.float32) / 32768.0).to("cuda:0")
lens = torch.tensor([cfg.n_samples - 37 * r for r in range(16)], dtype=torch.int32, device="cuda:0")   # ragged rows too
outs = []
for v in (0, 1):
    s.configure("stft_indep", v)
    outs.append((s.mel_power(y).cpu().numpy(), s.mel_power(y, lengths=lens).cpu().numpy()))
for name, a, b in (("f32 full rows", outs[0][0], outs[1][0]), ("f32 ragged", outs[0][1], outs[1][1])):
    d = np.abs(a - b)
    print(name, "shape", a.shape, "max |d|", d.max(), "rel", d.max() / max(a.max(), 1e-30), "frames differing:",
          sorted(set(np.nonzero(d.max(axis=2))[1].tolist()))[:70], "rows", sorted(set(np.nonzero(d.max(axis=2))[0].tolist())))
