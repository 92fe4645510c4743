#!/usr/bin/env python3
"""Which summation order does K0 implement?  Recovers the GPU's float32 scale per row and
compares it with numpy and with the parallel-tree model (development aid)."""
import sys
from pathlib import Path
import numpy as np
import torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tools"))
from amphibian_vae_latent_detector_amd import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from pairwise_model import numpy_sum, tree_sum, f

fe = AmphSession(PRESETS["bench-1s"], None, None, "cuda:0", max_batch=8)
rng = np.random.default_rng(0)
def np_scale(y):
    rms = np.sqrt(np.mean(y ** 2)); return np.float32(0.05) / (rms + np.float32(1e-8))
def model_scale(y, summer):
    rms = np.sqrt(f(summer(y * y) / f(len(y)))); return np.float32(0.05) / (rms + np.float32(1e-8))
for n in [100, 128, 129, 1000, 4096, 8192, 8193, 16384, 22050, 24576, 30000]:
    ys = []
    for t in range(6):
        y = (rng.standard_normal(n) * 0.01).astype(np.float32)
        if t % 2: y[::97] = 0.5
        ys.append(y)
    Y = np.stack(ys)
    # a probe sample of exactly 1.0/64 lets the scale be read back exactly: out = x * scale
    Y[:, 0] = np.float32(1.0 / 64)
    out, ok = fe.rms_normalize(Y)
    out = out.cpu().numpy()
    res = []
    for r in range(6):
        sc_gpu = np.float32(out[r, 0] * 64)
        sc_np = np_scale(Y[r]); sc_m = model_scale(Y[r], numpy_sum)
        res.append((int(sc_gpu.view(np.int32)) - int(sc_np.view(np.int32)), int(sc_m.view(np.int32)) - int(sc_np.view(np.int32))))
    print(f"n={n:6d} ulp(gpu-numpy, model-numpy) per row: {res}")
