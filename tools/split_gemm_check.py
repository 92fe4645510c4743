#!/usr/bin/env python3
"""Accuracy of the three-term bf16 split GEMM path ("split_gemm" 1) against the torch-CPU fp32 oracle and against the
f32-MFMA path, encoder alone and end to end.  GPU box."""
import sys
from pathlib import Path
import numpy as np, torch
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from amphibian_vae_latent_detector_amd import PRESETS, SPECS
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
from oracle import encoder as oenc

def rel(a, b):
    return float(np.max(np.linalg.norm(a - b, axis=1) / np.maximum(np.linalg.norm(b, axis=1), 1e-30)))

for preset, name in (("bench-1s", "thesis-shape"), ("bench-1s-128", "thesis-shape"), ("ref-default", "thesis-shape")):
    cfg = PRESETS[preset]
    spec = SPECS[name](cfg.target_frames, cfg.n_mels)
    state = random_init_state(spec, 0)
    enc = HipEncoder(spec, state, "cuda:0", max_batch=64)
    sess = enc.session(cfg)
    tenc = oenc.build_torch_encoder(spec.to_dict(), state)
    x = torch.from_numpy(np.random.default_rng(1).standard_normal((70, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32))
    with torch.no_grad():
        mu_o, lv_o = tenc(x)
    out = {}
    for v in (0, 1, 2, 3, 4):                                    # 3: patch-resident fused pair (k_conv_gemm_f4s); 4: with 7 workgroups
        sess.configure("split_gemm", 1 if v else 0)
        sess.configure("split_fused", 1 if v >= 2 else 0)
        sess.configure("patch_fused", 1 if v >= 3 else 0)
        sess.configure("patch_wgs", 7 if v == 4 else 0)
        mu, lv = sess.encoder_forward(x)
        out[v] = (mu.cpu().numpy(), lv.cpu().numpy())
        mu2, _ = sess.encoder_forward(x[:13])
        assert torch.equal(mu2.cpu(), mu[:13].cpu()), "batch dependence"
    assert np.array_equal(out[3][0], out[4][0]), "workgroup-count dependence"
    for pg in (0, 1, 2):
        sess.configure("patch_gemm", pg)
        mu_g, lv_g = sess.encoder_forward(x)
        assert torch.equal(sess.encoder_forward(x[:13])[0], mu_g[:13]), "batch dependence"
        print(f"   + patch_gemm={pg}: mu {rel(mu_g.cpu().numpy(), mu_o.numpy()):.2e} lv {rel(lv_g.cpu().numpy(), lv_o.numpy()):.2e} vs patch_gemm=3 {rel(mu_g.cpu().numpy(), out[4][0]):.2e}")
    sess.configure("patch_gemm", 3)
    print(f"   + patch-resident fused pair: mu {rel(out[3][0], mu_o.numpy()):.2e} lv {rel(out[3][1], lv_o.numpy()):.2e} vs f3s {rel(out[3][0], out[2][0]):.2e}")
    print(f"   + fused pair on bf16x3: mu {rel(out[2][0], mu_o.numpy()):.2e} lv {rel(out[2][1], lv_o.numpy()):.2e} vs f32 MFMA {rel(out[2][0], out[0][0]):.2e}")
    print(f"[{preset}] encoder vs torch-CPU fp32: f32 MFMA mu {rel(out[0][0], mu_o.numpy()):.2e} lv {rel(out[0][1], lv_o.numpy()):.2e} | "
          f"bf16x3 mu {rel(out[1][0], mu_o.numpy()):.2e} lv {rel(out[1][1], lv_o.numpy()):.2e} | bf16x3 vs f32 MFMA {rel(out[1][0], out[0][0]):.2e}")
    sess.configure("split_gemm", 1)
    sess.configure("split_fused", 1)
    sess.configure("patch_fused", 1)
    sess.configure("patch_wgs", 0)
print("ok")
