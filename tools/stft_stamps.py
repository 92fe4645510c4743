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
import os
lib = C.CDLL(str(ROOT / "build_ab" / os.environ.get("AMPH_STAMPS_LIB", "libamph_stamps.so")))   # (or an ablation built with -DAMPH_STAMPS)
for name in N.EXPORTED_SYMBOLS:
    fn, ref = getattr(lib, name), getattr(N.lib, name)
    fn.restype, fn.argtypes = ref.restype, ref.argtypes
N.lib = lib
from amphibian_vae_latent_detector_amd import SPECS
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS["bench-1s"]
spec = SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
s = HipEncoder(spec, random_init_state(spec, 0), "cuda:0", max_batch=256).session(cfg)
s.configure("lanes", 1)
pcm = torch.from_numpy(synth_batch_pcm16(list(range(256)), cfg.n_samples, cfg.sr)).to("cuda:0")
for key in [a for a in sys.argv[2:] if "=" in a]:      # e.g. stft_s16=0
    k, v = key.split("=")
    s.configure(k, int(v))
for _ in range(3):
    s.encode(pcm, apply_rms=True)                       # the PCM16 + RMS path of the benchmark
torch.cuda.synchronize()
s.configure("dump_stamps", int(sys.argv[1]) if len(sys.argv) > 1 else 256)   # workgroups of the launch (12-wave kernel: 256)
try:
    s.configure("dump_timeline", int(sys.argv[1]) if len(sys.argv) > 1 else 256)   # k_stft_pair: tables | pair rounds | dB stage
except Exception as e:                                   # (older diagnostic builds)
    print("no timeline:", e)
