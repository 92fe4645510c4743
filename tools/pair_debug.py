import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
from oracle import frontend as ofe
cfg = PRESETS["bench-1s"]
B = 40
s = AmphSession(cfg, None, None, "cuda:0", max_batch=B)
pcm_h = synth_batch_pcm16(list(range(B)), cfg.n_samples, cfg.sr)
pcm = torch.from_numpy(pcm_h).to("cuda:0")
lens_h = np.array([cfg.n_samples - 371 * r for r in range(B)], dtype=np.int32)
lens = torch.from_numpy(lens_h).to("cuda:0")
res = {}
for v in (0, 1, 0, 1):
    s.configure("stft_pair", v)
    img, pw = s.mel_pcm16(pcm, lengths=lens, apply_rms=True, layout=1, want_power=True)
    res.setdefault(v, []).append(pw.cpu().numpy())
for v in (0, 1):
    a, b = res[v]
    print("pair", v, "run-to-run max diff", np.abs(a - b).max())
a, b = res[0][0], res[1][0]
for r in range(B):
    d = np.abs(a[r] - b[r])
    if d.max() > 1e-6 * a[r].max():
        fr = np.nonzero(d.max(axis=1) > 1e-6 * a[r].max())[0]
        yr, good = ofe.normalize_pcm16_chunk(pcm_h[r, :lens_h[r]], True)
        want = ofe.melspectrogram(ofe.fix_length(yr, cfg.n_samples) if hasattr(ofe, "fix_length") else np.pad(yr, (0, cfg.n_samples - yr.size)), cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T
        print("row", r, "len", lens_h[r], "frames differing", fr.tolist(), "rowmax", a[r].max(), "max d", d.max(),
              "| pair0 vs oracle", np.abs(a[r] - want).max() / want.max(), "pair1 vs oracle", np.abs(b[r] - want).max() / want.max())
