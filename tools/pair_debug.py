import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
from oracle import frontend as ofe
cfg = PRESETS[sys.argv[1] if len(sys.argv) > 1 else "bench-1s-128"]
idx = [0, 1, 2, 3, 4, 5, 6, 7, 8, 49, 10, 11]
B = len(idx)
s = AmphSession(cfg, None, None, "cuda:0", max_batch=B)
pcm_h = synth_batch_pcm16(idx, cfg.n_samples, cfg.sr)
pcm = torch.from_numpy(pcm_h).to("cuda:0")
for rms in (False, True):
    want = np.stack([ofe.wave_to_mel(ofe.normalize_pcm16_chunk(q, rms)[0], **cfg.mel_kwargs()).T for q in pcm_h])
    s.configure("stft_pair", 0)
    i0, p0 = (t.cpu().numpy() for t in s.mel_pcm16(pcm, apply_rms=rms, layout=1, want_power=True))
    s.configure("stft_pair", 1)
    i1, p1 = (t.cpu().numpy() for t in s.mel_pcm16(pcm, apply_rms=rms, layout=1, want_power=True))
    for r in range(B):
        print(f"rms {int(rms)} row {r}: old vs oracle {np.abs(i0[r]-want[r]).max():.2e}  pair vs oracle {np.abs(i1[r]-want[r]).max():.2e}  pair vs old {np.abs(i1[r]-i0[r]).max():.2e}"
              f"  power pair vs old / peak {np.abs(p1[r]-p0[r]).max()/max(p0[r].max(),1e-30):.2e}")
