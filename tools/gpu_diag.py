#!/usr/bin/env python3
"""Stage-by-stage parity diagnostic: every HIP stage against the CPU oracle, with
error metrics printed (development aid; the pass/fail versions live in tests/)."""
import sys
import time
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from amphibian_vae_latent_detector_amd import PRESETS, SPECS  # noqa: E402
from amphibian_vae_latent_detector_amd.encoder import AmphSession, HipEncoder, random_init_state  # noqa: E402
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16  # noqa: E402
from oracle import encoder as oenc  # noqa: E402
from oracle import frontend as ofe  # noqa: E402


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30))


def main():
    preset = sys.argv[1] if len(sys.argv) > 1 else "bench-1s"
    cfg = PRESETS[preset]
    dev = "cuda:0"
    print("device:", torch.cuda.get_device_name(0), "| preset", preset, cfg)
    idx = list(range(12)) + [49, 99]
    pcm = synth_batch_pcm16(idx, cfg.n_samples, cfg.sr)
    B = len(idx)
    fe = AmphSession(cfg, None, None, dev, max_batch=8)     # < B on purpose: exercises batching

    basis_o = ofe.mel_filterbank(cfg.sr, cfg.n_fft, cfg.n_mels, cfg.fmin, cfg.fmax)
    basis_g = fe.mel_basis()
    print("mel basis   max|d| = %.3e (max %.3e) nnz mismatch %d" % (np.abs(basis_g - basis_o).max(), basis_o.max(),
          int(((basis_g != 0) != (basis_o != 0)).sum())))

    # ---- rms_normalize
    yf = ofe.pcm16_to_float(pcm)
    out, ok = fe.rms_normalize(yf)
    out = out.cpu().numpy(); ok = ok.cpu().numpy()
    worst = 0.0
    for r in range(B):
        o, good = ofe.rms_normalize(yf[r])
        assert bool(ok[r]) == bool(good), (r, ok[r], good)
        worst = max(worst, float(np.abs(out[r] - o).max()))
    print("rms f32     max|d| = %.3e  ok=%s" % (worst, ok.tolist()))
    outq, okq = fe.rms_normalize(pcm)
    outq = outq.cpu().numpy()
    nd = 0
    for r in range(B):
        o, good = ofe.rms_normalize(yf[r])
        nd += int((outq[r] != ofe.float_to_pcm16(o)).sum())
    print("rms pcm16   mismatching samples = %d of %d" % (nd, pcm.size))

    # ---- mel power / mel
    ynorm = np.stack([ofe.normalize_pcm16_chunk(pcm[r], True)[0] for r in range(B)])
    mp_g = fe.mel_power(ynorm).cpu().numpy()                     # [B, T_raw, M]
    mp_o = np.stack([ofe.melspectrogram(ofe.fix_length(ynorm[r], cfg.sr, cfg.duration), cfg.sr, cfg.n_fft, cfg.hop_length,
                                        cfg.n_mels, cfg.fmin, cfg.fmax).T for r in range(B)])
    for r in (0, 1, B - 2):
        d = np.abs(mp_g[r] - mp_o[r])
        print("mel power   row %2d rel L2 %.3e  max|d|/max %.3e  (max %.3e)" % (r, rel(mp_g[r], mp_o[r]), d.max() / max(mp_o[r].max(), 1e-30), mp_o[r].max()))
    relbin = np.abs(mp_g - mp_o) / np.maximum(mp_o, 1e-12 * mp_o.max(axis=(1, 2), keepdims=True))
    print("mel power   worst per-bin relative error over all rows: %.3e" % relbin.max())
    mel_g = fe.mel(ynorm, layout=0).cpu().numpy()
    mel_o = np.stack([ofe.wave_to_mel(ynorm[r], **cfg.mel_kwargs()) for r in range(B)])
    print("mel image   max|d| = %.3e  rel L2 = %.3e  finite=%s silent-row max|g| = %.3e" % (
        np.abs(mel_g - mel_o).max(), rel(mel_g, mel_o), np.isfinite(mel_g).all(), np.abs(mel_g[B - 2]).max()))
    mel_g1 = fe.mel(ynorm, layout=1).cpu().numpy()
    print("mel layout1 equals layout0 transposed:", np.array_equal(mel_g1, mel_g.transpose(0, 2, 1)))

    # ---- encoders
    for name in ("tiny", "thesis-shape"):
        spec = SPECS[name](cfg.target_frames, cfg.n_mels)
        state = random_init_state(spec, seed=0)
        enc = HipEncoder(spec, state, dev, max_batch=8)
        sess = enc.session(cfg)
        tenc = oenc.build_torch_encoder(spec.to_dict(), state)
        mu_o, lv_o = oenc.encode_mels_batched(tenc, mel_o)
        x = torch.from_numpy(mel_o).transpose(1, 2).unsqueeze(1).contiguous()
        mu_g, lv_g = sess.encoder_forward(x)
        mu_g, lv_g = mu_g.cpu().numpy(), lv_g.cpu().numpy()
        print("%-12s encoder-only  mu rel %.3e  logvar rel %.3e  |mu| %.3f" % (name, rel(mu_g, mu_o), rel(lv_g, lv_o), np.abs(mu_o).mean()))
        mu_f, _, okf = sess.encode(pcm, apply_rms=True)
        mu_f = mu_f.cpu().numpy()
        per = [rel(mu_f[r], mu_o[r]) for r in range(B)]
        print("%-12s fused encode  worst per-chunk rel L2 %.3e  (rows %s)  max elem |d| %.3e" % (
            name, max(per), np.argsort(per)[-3:].tolist(), np.abs(mu_f - mu_o).max()))
        mu_h, okh = sess.encode_host(pcm, apply_rms=True)
        print("%-12s host path == device path: %s" % (name, np.array_equal(mu_h, mu_f)))
        if name == "thesis-shape":
            # timing, batch 256 x 4
            big = torch.from_numpy(synth_batch_pcm16(list(range(256)), cfg.n_samples, cfg.sr)).to(dev)
            big = big.repeat(4, 1).contiguous()
            enc2 = HipEncoder(spec, state, dev, max_batch=256)
            s2 = enc2.session(cfg)
            s2.encode(big, apply_rms=True)
            torch.cuda.synchronize()
            t0 = time.time()
            for _ in range(5):
                s2.encode(big, apply_rms=True)
            torch.cuda.synchronize()
            dt = (time.time() - t0) / 5
            print("thesis-shape 1024 chunks: %.3f ms -> %.0f chunks/s" % (dt * 1e3, 1024 / dt))
            s2.profile_enable(True)
            for _ in range(3):
                s2.encode(big, apply_rms=True)
            torch.cuda.synchronize()
            for st in s2.profile_read():
                if st["launches"]:
                    ms = st["total_ms"] / st["launches"]
                    print("   %-12s %7.3f ms/launch  %8.2f TFLOP/s  %8.1f GB/s(alg)" % (
                        st["stage"], ms, st["flops_per_chunk"] * 256 / ms / 1e9, st["bytes_per_chunk"] * 256 / ms / 1e6))
            s2.profile_enable(False)


if __name__ == "__main__":
    main()
