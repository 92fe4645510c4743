#!/usr/bin/env python3
"""Per-role time split of k_conv_gemm (diagnostic build `make stamps`): cycles a consumer / loader wave
spends working vs waiting at the slice barrier.  Shares, not absolute times."""
import ctypes as C, sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS, SPECS
from amphibian_vae_latent_detector_amd import _native as N
lib = C.CDLL(str(ROOT / "build_ab" / "libamph_stamps.so"))
for name in N.EXPORTED_SYMBOLS:
    fn, ref = getattr(lib, name), getattr(N.lib, name)
    fn.restype, fn.argtypes = ref.restype, ref.argtypes
N.lib = lib
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
args = dict(a.split("=") for a in sys.argv[1:])          # key=value knobs, e.g. patch_fused=0 patch_wgs=128; preset=bench-1s-128 bf16_encoder=1
cfg = PRESETS[args.pop("preset", "bench-1s")]
spec = SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
s = HipEncoder(spec, random_init_state(spec, 0), "cuda:0", max_batch=256).session(cfg)
s.configure("lanes", 1)
for k, v in args.items():
    s.configure(k, int(v))
x = torch.randn(256, 1, cfg.target_frames, cfg.n_mels, device="cuda:0")
for _ in range(3):
    s.encoder_forward(x)
torch.cuda.synchronize()
s.configure("dump_gemm_stamps", 0)
