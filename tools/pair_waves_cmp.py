import sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS["bench-1s"]; B = 256
s = AmphSession(cfg, None, None, "cuda:0", max_batch=B)
pcm = torch.from_numpy(synth_batch_pcm16(list(range(1000, 1000 + B)), cfg.n_samples, cfg.sr)).to("cuda:0")
out = {}
for nw in (12, 11, 10, 8):
    s.configure("stft_pair_waves", nw)
    img, pw = s.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)
    fused = s.mel_pcm16(pcm, apply_rms=True, layout=1)
    out[nw] = (pw.cpu().numpy(), img.cpu().numpy(), fused.cpu().numpy())
for nw in (11, 10, 8):
    dp = np.abs(out[nw][0] - out[12][0]).max(); di = np.abs(out[nw][1] - out[12][1]).max(); df = np.abs(out[nw][2] - out[12][2]).max()
    rows = np.nonzero(np.abs(out[nw][2] - out[12][2]).reshape(B, -1).max(axis=1))[0]
    print(f"waves {nw} vs 12: mel power max |d| {dp:.3e}, image (separate dB) {di:.3e}, image (fused dB) {df:.3e}, rows differing (fused) {rows[:10]} n={len(rows)}")
s.configure("stft_pair_waves", 12)
s.configure("stft_pair_tw_lds", 0)
img, pw = s.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)
print("12 waves, stage-1 twiddles from global memory vs from LDS: mel power max |d|", np.abs(pw.cpu().numpy() - out[12][0]).max())
s.configure("stft_pair_tw_lds", 1)
