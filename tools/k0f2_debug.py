#!/usr/bin/env python3
"""k_stft_pair2's K0F form against K0's own launches (bench-1s-128): where do the images differ?"""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS, SPECS
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS["bench-1s-128"]
spec = SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
s = HipEncoder(spec, random_init_state(spec, 0), "cuda:0", max_batch=64).session(cfg)
pcm = torch.from_numpy(synth_batch_pcm16(list(range(700, 732)), cfg.n_samples, cfg.sr)).to("cuda:0")
r = {}
for rms in (True, False):
    for k in (1, 0):
        s.configure("stft_k0_fuse", k)
        r[k] = s.mel_pcm16(pcm, apply_rms=rms, layout=1).clone().cpu().numpy()
    d = np.abs(r[1] - r[0])
    print("apply_rms", rms, "image", r[1].shape, "max abs diff", d.max(), "elements differing", int((d > 0).sum()), "of", d.size)
    rows = np.nonzero(d.reshape(d.shape[0], -1).max(1) > 0)[0]
    print("  rows differing", rows[:10], "of", d.shape[0])
    if len(rows):
        dd = d[rows[0]]
        fr = np.nonzero(dd.max(1) > 0)[0]
        print("  row", rows[0], "frames (image rows, layout 1 = [frame][band]) differing:", fr[:40], "max per frame", dd.max(1)[fr][:10])
