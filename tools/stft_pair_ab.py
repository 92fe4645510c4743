#!/usr/bin/env python3
"""k_stft_pair against k_stft_mel_w on the same PCM16 rows: mel power and standardised image (fused and separate dB stage),
with and without the RMS stage, full and ragged rows; then isolated stage times of both.  GPU box."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "bench-1s"]
B = 40
s = AmphSession(cfg, None, None, "cuda:0", max_batch=B)
pcm = torch.from_numpy(synth_batch_pcm16(list(range(B)), cfg.n_samples, cfg.sr)).to("cuda:0")
lens = torch.tensor([cfg.n_samples - 371 * r for r in range(B)], dtype=torch.int32, device="cuda:0")
worst = 0.0
for nw in (8, 10, 12):
    for rms in (False, True):
        for ln in (None, lens):
            outs = []
            for v in (0, 1):
                s.configure("stft_pair", v)
                s.configure("stft_pair_waves", nw)
                img, pw = s.mel_pcm16(pcm, lengths=ln, apply_rms=rms, layout=1, want_power=True)
                fused = s.mel_pcm16(pcm, lengths=ln, apply_rms=rms, layout=1)
                outs.append((pw.cpu().numpy(), img.cpu().numpy(), fused.cpu().numpy()))
            (p0, i0, f0), (p1, i1, f1) = outs
            dp = np.abs(p0 - p1).max(axis=(1, 2)) / np.maximum(p0.max(axis=(1, 2)), 1e-30)
            print(f"waves {nw} rms {int(rms)} ragged {int(ln is not None)}: mel power max |d| / row max = {dp.max():.2e}; image (separate dB) max |d| = "
                  f"{np.abs(i0 - i1).max():.2e}; image (fused dB) max |d| = {np.abs(f0 - f1).max():.2e}; fused vs separate (pair) {np.abs(f1 - i1).max():.2e}",
                  flush=True)
            assert np.isfinite(dp).all() and np.isfinite(f1).all() and np.isfinite(i1).all(), "non-finite output"
            worst = max(worst, dp.max())
# several rows per workgroup (the pair counter starts over per row; the dB stage of a row runs between two rows' pairs)
s.configure("stft_pair_waves", 12)
for grid in (7, 3):
    outs = []
    for v in (0, 1):
        s.configure("stft_pair", v)
        s.configure("stft_grid", grid if v else 0)
        img, pw = s.mel_pcm16(pcm, lengths=lens, apply_rms=True, layout=1, want_power=True)
        fused = s.mel_pcm16(pcm, lengths=lens, apply_rms=True, layout=1)
        outs.append((pw.cpu().numpy(), img.cpu().numpy(), fused.cpu().numpy()))
    s.configure("stft_grid", 0)
    (p0, i0, f0), (p1, i1, f1) = outs
    dp = np.abs(p0 - p1).max(axis=(1, 2)) / np.maximum(p0.max(axis=(1, 2)), 1e-30)
    print(f"{grid} workgroups for {B} rows: mel power max |d| / row max = {dp.max():.2e}; image (fused dB) max |d| = {np.abs(f0 - f1).max():.2e}; "
          f"fused vs separate (pair) {np.abs(f1 - i1).max():.2e}", flush=True)
    worst = max(worst, dp.max())
print("worst mel power difference relative to the row's peak:", worst)
assert np.isfinite(worst) and 0.0 < worst < 1e-6, worst
