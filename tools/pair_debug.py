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
for v in (0, 1):
    s.configure("stft_pair", v)
    outs = []
    for rep in range(3):
        img, pw = s.mel_pcm16(pcm, lengths=lens, apply_rms=True, layout=1, want_power=True)
        outs.append(pw.cpu().numpy())
        nrm, ok = s.rms_normalize(pcm[:, :])
    for a in outs[1:]:
        d = np.abs(a - outs[0]).max(axis=(1, 2))
        print("pair", v, "rows differing run to run:", np.nonzero(d > 0)[0].tolist(), d.max())
    res = outs[-1]
    bad = []
    for r in range(B):
        yr, good = ofe.normalize_pcm16_chunk(pcm_h[r, :lens_h[r]], True)
        yy = np.zeros(cfg.n_samples, np.float32); yy[:yr.size] = yr
        want = ofe.melspectrogram(yy, cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T
        e = np.abs(res[r] - want).max() / want.max()
        if e > 1e-5: bad.append((r, float(e)))
    print("pair", v, "rows off the oracle:", bad)
