#!/usr/bin/env python3
"""Where does a frame's time go in k_stft_mel?  Runs the diagnostic build (make stamps) and prints
the mean s_memtime deltas per phase.  Shares, not absolute times (the stamps serialise the wave)."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd import _native as N
lib = C.CDLL(str(ROOT / "build_ab" / "libamph_stamps.so"))
for name in N.EXPORTED_SYMBOLS:
    fn, ref = getattr(lib, name), getattr(N.lib, name)
    fn.restype, fn.argtypes = ref.restype, ref.argtypes
N.lib = lib
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS["bench-1s"]
s = AmphSession(cfg, None, None, "cuda:0", max_batch=256)
y = torch.from_numpy(synth_batch_pcm16(list(range(256)), cfg.n_samples, cfg.sr).astype(np.float32) / 32768).to("cuda:0")
for _ in range(3):
    s.mel_power(y)
torch.cuda.synchronize()
s.configure("dump_stamps", 512)
