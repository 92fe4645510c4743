"""GPU parity tests: the HIP path, called through the C-ABI (ctypes), against the CPU
oracle on the same seeded inputs, against the golden vectors captured from the
reference's own functions, and - at BASELINE.json's full config-2 size - through
size-independent properties.

Tolerances (fp32 path; BASELINE.json north_star: latents within 1e-4 relative):
  integer / index results (PCM16 output, decisions, rms_ok flags)   bit-exact
  rms_normalize float32 (vs the reference's own output vectors)     bit-exact
  mel power per chunk                                               1e-5 relative L2
  standardised mel image                                            1e-4 absolute
  latents: per-vector ||dz||2/||z||2 <= 1e-4 and |dz| <= 1e-4*max(1,|z|)   
"""
import json
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from conftest import rel_l2

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
TOL = 1e-4


# ----------------------------------------------------------------------------- fixtures
@pytest.fixture(scope="module")
def env():
    if not torch.cuda.is_available():
        pytest.fail("GPU tests need an MI355X; there is no CPU fallback to test instead")
    from amphibian_vae_latent_detector_amd import PRESETS, SPECS
    from amphibian_vae_latent_detector_amd import _native as N
    from amphibian_vae_latent_detector_amd.encoder import AmphSession, HipEncoder, random_init_state
    from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
    from oracle import encoder as oenc
    from oracle import frontend as ofe
    from oracle import radial as orad

    class E:
        pass
    e = E()
    e.PRESETS, e.SPECS, e.N = PRESETS, SPECS, N
    e.AmphSession, e.HipEncoder, e.random_init_state, e.synth = AmphSession, HipEncoder, random_init_state, synth_batch_pcm16
    e.oenc, e.ofe, e.orad = oenc, ofe, orad
    e.cfg = PRESETS["bench-1s"]
    return e


@pytest.fixture(scope="module")
def fe(env):
    return env.AmphSession(env.cfg, None, None, DEV, max_batch=8)


def _encoders(env, name, cfg=None, max_batch=8):
    cfg = cfg or env.cfg
    spec = env.SPECS[name](cfg.target_frames, cfg.n_mels)
    state = env.random_init_state(spec, seed=0)
    return env.HipEncoder(spec, state, DEV, max_batch=max_batch), env.oenc.build_torch_encoder(spec.to_dict(), state), spec


def _oracle_latents(env, tenc, pcm, cfg, apply_rms=True):
    out = []
    for q in pcm:
        y, _ = env.ofe.normalize_pcm16_chunk(q, apply_rms)
        out.append(env.oenc.encode_mel_to_latent(tenc, env.ofe.wave_to_mel(y, **cfg.mel_kwargs())))
    return np.stack(out)


def _assert_latents(z_gpu, z_cpu, tol=TOL):
    for r in range(z_cpu.shape[0]):
        assert rel_l2(z_gpu[r], z_cpu[r]) <= tol, (r, rel_l2(z_gpu[r], z_cpu[r]))
    assert np.all(np.abs(z_gpu - z_cpu) <= tol * np.maximum(1.0, np.abs(z_cpu)))


# ----------------------------------------------------------------------------- the library that runs is the HIP one
def test_native_library_is_loaded_and_device_is_gfx950(env, fe):
    maps = Path("/proc/self/maps").read_text()
    assert "libamph.so" in maps
    assert "gfx950" in torch.cuda.get_device_properties(0).gcnArchName
    assert fe.raw_frames == 58


def test_mel_basis_equals_librosa_restatement(env, fe):
    want = env.ofe.mel_filterbank(env.cfg.sr, env.cfg.n_fft, env.cfg.n_mels, env.cfg.fmin, env.cfg.fmax)
    got = fe.mel_basis()
    assert np.array_equal(got != 0, want != 0)
    np.testing.assert_allclose(got, want, rtol=2e-7, atol=0)


# ----------------------------------------------------------------------------- K0: rms_normalize
RMS_CASES = ["noise_quiet", "noise_loud", "tone", "silent", "below_gate", "clipping", "short", "gate_boundary"]


@pytest.mark.parametrize("numpy2", [0, 1])
@pytest.mark.parametrize("case", RMS_CASES)
def test_rms_normalize_f32_against_reference_vectors(env, fe, golden, case, numpy2):
    """Bit-exact against the reference function's own outputs, in both scalar-promotion environments
    (tests/golden/make_golden.py): the default = numpy 1.26 (the reference's pin: float64 gate compare and
    scale, cast once), and "rms_numpy2" = numpy 2.x (float32 scalars).  Gated rows are returned untouched,
    scaled rows use numpy's float32 summation order.  `gate_boundary` has rms == float32(1e-4): gated under
    1.26, scaled under 2.x."""
    g = golden("rms_normalize.npz")
    pre = "np2_" if numpy2 else ""
    fe.configure("rms_numpy2", numpy2)
    try:
        out, ok = fe.rms_normalize(g[f"in_{case}"])
    finally:
        fe.configure("rms_numpy2", 0)
    assert bool(ok.cpu().numpy()[0]) == bool(g[f"ok_{pre}{case}"])
    assert np.array_equal(out.cpu().numpy()[0], g[f"out_{pre}{case}"])
    assert np.abs(out.cpu().numpy()).max() <= 1.0
    if case == "gate_boundary":
        assert bool(g["ok_np2_gate_boundary"]) and not bool(g["ok_gate_boundary"])


def test_rms_normalize_pcm16_bit_exact(env, fe):
    """PCM16 in -> PCM16 out (what 00's process_folder writes): integer result, bit-exact.
    This needs the float32 scale to equal numpy's to the last bit (a 1-ulp difference moves
    every occurrence of some sample value by an LSB), which K0 gets by evaluating mean(y^2)
    in numpy's own summation order.  Lengths straddle numpy's 8192-element buffer chunks
    and its 128-element pairwise blocks."""
    rng = np.random.default_rng(3)
    # 7 922 / 8 191 / 16 114 / 40 457: a last numpy chunk whose pairwise tree is SEVEN levels deep (441 of the 8 191 partial sizes:
    # the right-most path still holds 130 elements at depth 6) - K0 had 64 leaf slots per chunk until round 3
    for n in (env.cfg.n_samples, 8192, 8193, 16385, 129, 100, 7, 48000, 32769, 240000, 7922, 8191, 16114, 40457):     # > 32 768: K0's two-kernel path
        idx = list(range(20)) + [49] if n == env.cfg.n_samples else [int(v) for v in rng.integers(0, 5000, 5)]
        pcm = env.synth(idx, n, env.cfg.sr)
        for numpy2, promotion in ((0, "numpy1"), (1, "numpy2")):       # each mode against its oracle twin
            fe.configure("rms_numpy2", numpy2)
            try:
                out, ok = fe.rms_normalize(pcm)
            finally:
                fe.configure("rms_numpy2", 0)
            out, ok = out.cpu().numpy(), ok.cpu().numpy()
            for r in range(pcm.shape[0]):
                y, good = env.ofe.rms_normalize(env.ofe.pcm16_to_float(pcm[r]), promotion=promotion)
                assert bool(ok[r]) == bool(good)
                assert np.array_equal(out[r], env.ofe.float_to_pcm16(y)), (n, r, promotion)
    pcm = env.synth([49], env.cfg.n_samples, env.cfg.sr)
    out, ok = fe.rms_normalize(pcm)
    assert int(ok[0]) == 0 and not out.cpu().numpy().any()


# ----------------------------------------------------------------------------- K1/K2: STFT + mel power
def test_mel_power_against_oracle(env, fe):
    cfg = env.cfg
    pcm = env.synth([0, 1, 2, 3, 4, 5, 6, 7, 8, 49], cfg.n_samples, cfg.sr)
    y = env.ofe.pcm16_to_float(pcm)
    got = fe.mel_power(y).cpu().numpy()
    assert got.shape == (10, 58, 64)
    for r in range(10):
        want = env.ofe.melspectrogram(y[r], cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T
        assert rel_l2(got[r], want) <= 1e-5
        assert np.abs(got[r] - want).max() <= 2e-6 * max(want.max(), 1e-30)
    assert not got[9].any()


def test_mel_power_pure_tone_and_impulse(env, fe):
    """Known spectra: a bin-centred tone and a unit impulse (flat spectrum)."""
    cfg = env.cfg
    n = np.arange(cfg.n_samples)
    k0 = 200
    tone = np.cos(2 * np.pi * k0 * n / cfg.n_fft).astype(np.float32)
    imp = np.zeros(cfg.n_samples, np.float32)
    imp[5000] = 1.0
    got = fe.mel_power(np.stack([tone, imp])).cpu().numpy()
    for r, y in enumerate((tone, imp)):
        want = env.ofe.melspectrogram(y, cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T
        assert np.abs(got[r] - want).max() <= 3e-6 * want.max()
    basis = fe.mel_basis()
    m_peak = int(np.argmax(basis[:, k0]))
    assert int(np.argmax(got[0, 20])) == m_peak                       # energy lands in the filter that owns bin k0


@pytest.mark.parametrize("pad_mode", [0, 1])
def test_mel_image_against_oracle_both_pad_modes(env, pad_mode):
    from dataclasses import replace
    cfg = replace(env.cfg, pad_mode=pad_mode)
    sess = env.AmphSession(cfg, None, None, DEV, max_batch=4)
    pcm = env.synth(list(range(9)), cfg.n_samples, cfg.sr)
    y = env.ofe.pcm16_to_float(pcm)
    got = sess.mel(y, layout=0).cpu().numpy()
    got_t = sess.mel(y, layout=1).cpu().numpy()
    want = np.stack([env.ofe.wave_to_mel(v, pad_mode="reflect" if pad_mode == 0 else "constant", **cfg.mel_kwargs()) for v in y])
    assert np.abs(got - want).max() <= 1e-4
    assert np.array_equal(got_t, got.transpose(0, 2, 1))
    assert not got[:, :, :3].any() and not got[:, :, 61:].any()         # zero pad 3 left / 3 right (58 -> 64)
    # the dB / z-score stage runs inside the STFT kernel (one workgroup per row); the separate kernel must give the same image,
    # and so must a launch whose workgroups own several rows each (3 persistent workgroups for 9 rows: the in-loop form)
    try:
        sess.configure("stft_fuse_db", 0)
        assert np.array_equal(sess.mel(y, layout=0).cpu().numpy(), got)
        sess.configure("stft_fuse_db", 1)
        sess.configure("stft_grid", 3)
        assert np.array_equal(sess.mel(y, layout=0).cpu().numpy(), got)
        assert np.array_equal(sess.mel(y, layout=1).cpu().numpy(), got_t)
    finally:
        sess.configure("stft_grid", 0)
        sess.configure("stft_fuse_db", 1)
    sess.close()
    # the PCM16 paths in the same pad mode: rows padded by K0 (apply_rms) / by the copy kernel, against the STFT kernel
    # padding on the fly (stft_prepad 0), ragged lengths included - same bits; and against the CPU restatement
    enc, tenc, spec = _encoders(env, "tiny", cfg=cfg, max_batch=4)
    s2 = enc.session(cfg)
    lengths = np.array([cfg.n_samples, 15001, 700, 0, cfg.n_samples, 1, 4097, cfg.n_samples, 9000], dtype=np.int32)
    try:
        s2.configure("stft_pair", 0)        # (k_stft_mel_w on both sides: the pair kernel, which only reads K0's rows, has its own test)
        for apply_rms in (True, False):
            a = s2.encode(pcm, lengths=lengths, apply_rms=apply_rms)[0]
            s2.configure("stft_prepad", 0)
            assert torch.equal(a, s2.encode(pcm, lengths=lengths, apply_rms=apply_rms)[0]), apply_rms
            s2.configure("stft_prepad", 1)
        s2.configure("stft_pair", 1)
        want_z = []
        for r in range(9):
            yr, _ = env.ofe.normalize_pcm16_chunk(pcm[r, :lengths[r]], True) if lengths[r] else (np.zeros(0, np.float32), False)
            m = env.ofe.wave_to_mel(yr, pad_mode="reflect" if pad_mode == 0 else "constant", **cfg.mel_kwargs())
            want_z.append(env.oenc.encode_mel_to_latent(tenc, m))
        _assert_latents(s2.encode(pcm, lengths=lengths, apply_rms=True)[0].cpu().numpy(), np.stack(want_z))
    finally:
        s2.configure("stft_prepad", 1)
        s2.configure("stft_pair", 1)


@pytest.mark.parametrize("preset", ["bench-1s", "bench-1s-128"])
def test_pair_of_frames_stft_kernel_against_oracle_and_the_frame_per_wave_kernel(env, capsys, preset):
    """k_stft_pair (two frames per wave as one complex 2048-point FFT; the default for PCM16 rows at n_fft 2048 / hop 384) against
    the CPU restatement of librosa's melspectrogram (map_detector_core.py:219-228) and against k_stft_mel_w on the same rows:
    mel power within 1e-6 of the row's peak, images within 1e-4 of the oracle's, with and without the RMS stage, full and
    ragged rows (a last pair with ONE frame is covered by the 3 s preset: 376 frames are 188 pairs, 626: 313), every
    workgroup shape, stage-1 twiddles in LDS and from global memory, several rows per workgroup (in-loop dB stage), and the
    fused dB stage equal to the separate kernel bit for bit.  bench-1s-128: k_stft_pair2, the n_fft 1024 / hop 256 form (a pair per
    half wave, four frames per wave iteration; 87 frames: the last slot holds three)."""
    cfg = env.PRESETS[preset]
    B = 12
    sess = env.AmphSession(cfg, None, None, DEV, max_batch=B)
    pcm = env.synth([0, 1, 2, 3, 4, 5, 6, 7, 8, 49, 10, 11], cfg.n_samples, cfg.sr)
    lengths = np.array([cfg.n_samples, 16114, 700, 0, cfg.n_samples, 1, 4097, 7922, 9000, cfg.n_samples, 21999, 12345], dtype=np.int32)
    worst_p = worst_i = 0.0
    for apply_rms in (False, True):
        for ln in (None, lengths):
            want_p, want_i = [], []
            for r in range(B):
                q = pcm[r] if ln is None else pcm[r, :ln[r]]
                yr = env.ofe.normalize_pcm16_chunk(q, apply_rms)[0] if q.size else np.zeros(0, np.float32)
                yy = np.zeros(cfg.n_samples, np.float32)
                yy[:yr.size] = yr
                want_p.append(env.ofe.melspectrogram(yy, cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T)
                want_i.append(env.ofe.wave_to_mel(yr, **cfg.mel_kwargs()).T)
            want_p, want_i = np.stack(want_p), np.stack(want_i)
            sess.configure("stft_pair", 0)
            old_img, old_p = (t.cpu().numpy() for t in sess.mel_pcm16(pcm, lengths=ln, apply_rms=apply_rms, layout=1, want_power=True))
            sess.configure("stft_pair", 1)
            shapes = ((10, 1, 0), (8, 1, 0), (12, 0, 0), (10, 0, 5), (8, 1, 3)) if cfg.n_fft == 2048 else ((8, 1, 0), (12, 1, 0), (12, 0, 5), (8, 0, 3))
            for waves, tw_lds, grid in shapes:
                sess.configure("stft_pair_waves", waves)
                sess.configure("stft_pair_tw_lds", tw_lds)
                sess.configure("stft_grid", grid)
                img, pw = (t.cpu().numpy() for t in sess.mel_pcm16(pcm, lengths=ln, apply_rms=apply_rms, layout=1, want_power=True))
                fused = sess.mel_pcm16(pcm, lengths=ln, apply_rms=apply_rms, layout=1).cpu().numpy()
                assert np.array_equal(fused, img), (waves, tw_lds, grid)              # fused dB stage == separate kernel
                peak = np.maximum(want_p.max(axis=(1, 2)), 1e-30)
                ep = (np.abs(pw - want_p).max(axis=(1, 2)) / peak).max()
                eo = (np.abs(pw - old_p).max(axis=(1, 2)) / peak).max()
                assert ep <= 2e-6 and eo <= 1e-6, (apply_rms, waves, ep, eo)
                e_or, e_old = float(np.abs(img - want_i).max()), float(np.abs(img - old_img).max())
                assert e_or <= 1e-4 and e_old <= 1e-4, (apply_rms, waves, e_or, e_old)
                assert not pw[9].any()                                                # the silent row stays exactly zero
                worst_p, worst_i = max(worst_p, ep), max(worst_i, float(np.abs(img - want_i).max()))
            sess.configure("stft_grid", 0)
            sess.configure("stft_pair_waves", 10 if cfg.n_fft == 2048 else 8)
            sess.configure("stft_pair_tw_lds", 1)
    with capsys.disabled():
        print(f"\n[k_stft_pair{'' if cfg.n_fft == 2048 else '2'}, {preset}] mel power vs oracle: {worst_p:.2e} of the row's peak; image vs oracle: {worst_i:.2e}")
    sess.close()


@pytest.mark.parametrize("n_mels,fmin,fmax", [(40, 0.0, 11025.0), (80, 50.0, 8000.0), (128, 0.0, 11025.0), (24, 300.0, 3000.0), (64, 150.0, 10000.0)])
def test_pair_kernel_mel_tables_for_other_filter_banks(env, n_mels, fmin, fmax):
    """k_stft_pair takes its mel filters from tables the host arranges per round of 16 filters (build_pair_mel, csrc/amph_api.hip: which
    filter sits on which lane quad, start bins moved down inside the slack of the round's steps so that a lane group's reads tile the
    LDS bank row; the band index travels in the table).  Other filter banks than the bench preset's - band counts that are not multiples
    of 16, narrow and wide bands, rounds for which no arrangement exists - against the CPU restatement of librosa's melspectrogram
    (map_detector_core.py:219-228) and against k_stft_mel_w: mel power within 2e-6 of the row's peak, every band written."""
    import dataclasses
    cfg = dataclasses.replace(env.PRESETS["bench-1s"], n_mels=n_mels, fmin=fmin, fmax=fmax)
    B = 6
    sess = env.AmphSession(cfg, None, None, DEV, max_batch=B)
    pcm = env.synth([0, 1, 2, 3, 4, 49], cfg.n_samples, cfg.sr)
    want = []
    for r in range(B):
        yr = env.ofe.normalize_pcm16_chunk(pcm[r], True)[0]
        want.append(env.ofe.melspectrogram(yr, cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T)
    want = np.stack(want)
    sess.configure("stft_pair", 0)
    old = sess.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)[1].cpu().numpy()
    sess.configure("stft_pair", 1)
    got = sess.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)[1].cpu().numpy()
    assert got.shape == want.shape == (B, cfg.raw_frames, n_mels)
    peak = np.maximum(want.max(axis=(1, 2)), 1e-30)
    ep = (np.abs(got - want).max(axis=(1, 2)) / peak).max()
    eo = (np.abs(got - old).max(axis=(1, 2)) / peak).max()
    assert ep <= 2e-6 and eo <= 2e-6, (ep, eo)           # (two fp32 summation orders against each other: each within ~1e-6 of the oracle)
    assert (got[:5].max(axis=1) > 0).all()              # every band of the non-silent rows carries energy: none was left unwritten
    sess.close()


def test_fix_length_padding_truncation_and_ragged_lengths(env, fe):
    cfg = env.cfg
    N = cfg.n_samples
    pcm = env.synth([3, 4, 5, 6], N + 4000, cfg.sr)
    y = env.ofe.pcm16_to_float(pcm)
    lengths = np.array([N + 4000, N, 9000, 0], dtype=np.int32)          # longer, exact, shorter, empty
    got = fe.mel(y, lengths=lengths, layout=0).cpu().numpy()
    for r in range(4):
        want = env.ofe.wave_to_mel(y[r, :lengths[r]], **cfg.mel_kwargs())
        assert np.abs(got[r] - want).max() <= 1e-4, r
    assert not got[3].any()
    short = fe.mel(y[:, :9000], layout=0).cpu().numpy()                 # n_in < n_samples without lengths
    assert np.abs(short[0] - env.ofe.wave_to_mel(y[0, :9000], **cfg.mel_kwargs())).max() <= 1e-4


# ----------------------------------------------------------------------------- K4/K5: encoder
@pytest.mark.parametrize("name", ["tiny", "thesis-shape"])
def test_encoder_forward_against_torch_cpu(env, name):
    enc, tenc, spec = _encoders(env, name)
    rng = np.random.default_rng(1)
    x = rng.standard_normal((11, 1, 64, 64)).astype(np.float32)        # 11 rows over max_batch 8: two launches
    mu, lv = enc(torch.from_numpy(x))
    with torch.no_grad():
        mu_o, lv_o = tenc(torch.from_numpy(x))
    _assert_latents(mu.cpu().numpy(), mu_o.numpy(), 1e-5)
    _assert_latents(lv.cpu().numpy(), lv_o.numpy(), 1e-5)
    z = np.zeros((1, 1, 64, 64), np.float32)                           # all-zero image (silent chunk)
    mu0, _ = enc(torch.from_numpy(z))
    with torch.no_grad():
        _assert_latents(mu0.cpu().numpy(), tenc(torch.from_numpy(z))[0].numpy(), 1e-5)
    with pytest.raises(RuntimeError, match="expects"):
        enc(torch.zeros(1, 1, 32, 64))


@pytest.mark.parametrize("name", ["tiny", "thesis-shape"])
def test_fused_first_two_layers_equal_unfused_bitwise(env, name):
    """Layer 1 recomputed inside layer 2's operand staging (no 128 KB/chunk activation in HBM)
    must give the same bits as the materialised path, for full and ragged batches."""
    enc, tenc, spec = _encoders(env, name, max_batch=16)
    sess = enc.session(env.cfg)
    x = torch.from_numpy(np.random.default_rng(7).standard_normal((37, 1, 64, 64)).astype(np.float32))
    # (the f32-MFMA forms of both: with "split_gemm" the unfused layer 2 runs on the bf16 matrix cores, the fused one does not)
    sess.configure("split_gemm", 0)
    try:
        sess.configure("fuse_conv12", 1)
        mu1, lv1 = sess.encoder_forward(x)
        sess.configure("fuse_conv12", 0)
        mu0, lv0 = sess.encoder_forward(x)
        sess.configure("fuse_conv12", 1)
    finally:
        sess.configure("split_gemm", 1)
    assert torch.equal(mu1, mu0) and torch.equal(lv1, lv0)
    with torch.no_grad():
        _assert_latents(mu1.cpu().numpy(), tenc(x)[0].numpy(), 1e-5)


def test_split_operand_gemm_layers_are_fp32_accurate_and_batch_independent(env, capsys):
    """The GEMM layers behind the fused pair run on the matrix cores with two-term fp16 split operands ("split_gemm",
    default on; csrc/k_encoder_split.hip): x = h + 2^-11 l, three f16 MFMAs per K = 16, fp32 accumulation.  Against the
    torch-CPU fp32 oracle they must be as close as the f32-MFMA kernels are (1e-5 bar; both measure ~1e-6), must not depend
    on batch composition, and the two forms must agree to fp32 rounding noise.  Shapes: thesis-shape at 64 x 64 and at the
    reference's 192 x 64 image, `tiny` (C_in % 32 != 0 in layer 2: the split kernel only takes the head there)."""
    for preset, name in (("bench-1s", "thesis-shape"), ("ref-default", "thesis-shape"), ("bench-1s", "tiny")):
        cfg = env.PRESETS[preset]
        enc, tenc, spec = _encoders(env, name, cfg=cfg, max_batch=32)
        sess = enc.session(cfg)
        x = torch.from_numpy(np.random.default_rng(3).standard_normal((45, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32))
        with torch.no_grad():
            mu_o, lv_o = tenc(x)
        out = {}
        try:
            for v in (0, 1):
                sess.configure("split_gemm", v)
                mu, lv = sess.encoder_forward(x)
                out[v] = (mu.cpu().numpy(), lv.cpu().numpy())
                assert torch.equal(sess.encoder_forward(x[7:20])[0], mu[7:20])           # other batch, same bits
                sess.configure("fuse_conv12", 0)                                         # layer 2 through the same kernel family
                mu_u, _ = sess.encoder_forward(x)
                sess.configure("fuse_conv12", 1)
                _assert_latents(mu_u.cpu().numpy(), mu_o.numpy(), 1e-5)
        finally:
            sess.configure("split_gemm", 1)
            sess.configure("fuse_conv12", 1)
        for v in (0, 1):
            _assert_latents(out[v][0], mu_o.numpy(), 1e-5)
            _assert_latents(out[v][1], lv_o.numpy(), 1e-5)
        _assert_latents(out[1][0], out[0][0], 5e-6)
        e = [max(rel_l2(out[v][0][r], mu_o.numpy()[r]) for r in range(45)) for v in (0, 1)]
        with capsys.disabled():
            print(f"\n[split GEMM, {preset}/{name}] max rel L2 of mu vs torch-CPU fp32: f32 MFMA {e[0]:.2e}, fp16 x 2 MFMA {e[1]:.2e}")


def test_patch_resident_fused_pair_matches_oracle_and_other_forms(env, capsys):
    """conv1 + conv2 with the layer-1 tile resident in LDS and the conv2 weights in registers ("patch_fused", default on;
    csrc/k_encoder_patch.hip).  Against the torch-CPU fp32 oracle it must meet the same 1e-5 bar as the other forms (all ~1e-6),
    must not depend on the batch (ragged sizes: 1 chunk = 4 tiles ... 45 chunks), on the number of persistent workgroups
    (1, 7, the default rule, one per tile) or on where in the batch a chunk sits, and must agree with the staged split kernel
    (k_conv_gemm_f3s, "patch_fused" 0) to fp32 rounding noise.  Both 64 x 64 (bench) and 192 x 64 (reference default) images."""
    for preset in ("bench-1s", "ref-default"):
        cfg = env.PRESETS[preset]
        enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=48)
        sess = enc.session(cfg)
        x = torch.from_numpy(np.random.default_rng(11).standard_normal((45, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32))
        x[3] = 0.0                                                                       # an all-zero image (silence after z-score)
        x[4, 0, :2, :] = 6.0                                                             # large values (for a z-scored image) on the padded top edge
        with torch.no_grad():
            mu_o, lv_o = tenc(x)
        try:
            sess.configure("patch_fused", 0)
            mu_s, lv_s = sess.encoder_forward(x)
            sess.configure("patch_fused", 1)
            mu, lv = sess.encoder_forward(x)
            for wgs in (1, 7, 4096):
                sess.configure("patch_wgs", wgs)
                mu_w, lv_w = sess.encoder_forward(x)
                assert torch.equal(mu_w, mu) and torch.equal(lv_w, lv), f"result depends on the workgroup count ({wgs})"
            sess.configure("patch_wgs", 0)
            for lo, hi in ((0, 1), (7, 20), (44, 45)):
                mu_b, _ = sess.encoder_forward(x[lo:hi])
                assert torch.equal(mu_b, mu[lo:hi]), "result depends on the batch"
        finally:
            sess.configure("patch_fused", 1)
            sess.configure("patch_wgs", 0)
        _assert_latents(mu.cpu().numpy(), mu_o.numpy(), 1e-5)
        _assert_latents(lv.cpu().numpy(), lv_o.numpy(), 1e-5)
        _assert_latents(mu.cpu().numpy(), mu_s.cpu().numpy(), 5e-6)
        _assert_latents(lv.cpu().numpy(), lv_s.cpu().numpy(), 5e-6)
        e = max(rel_l2(mu.cpu().numpy()[r], mu_o.numpy()[r]) for r in range(45))
        d = max(rel_l2(mu.cpu().numpy()[r], mu_s.cpu().numpy()[r]) for r in range(45))
        with capsys.disabled():
            print(f"\n[patch-resident conv1+conv2, {preset}] max rel L2 of mu vs torch-CPU fp32 {e:.2e}, vs the staged split kernel {d:.2e}")


def test_register_resident_weight_conv_layers_match_oracle_and_staged_kernel(env, capsys):
    """conv3 / conv4 of the thesis-shape encoder with a workgroup's slice of the split weights in registers and the tile's input
    image(s) resident in LDS ("patch_gemm" bits 0 / 1; csrc/k_encoder_patch.hip, k_conv_gemm_g4).  Against the torch-CPU fp32 oracle
    each combination must meet the 1e-5 bar of the staged kernel (k_conv_gemm_s3; all measure ~1e-6), agree with it to fp32 rounding,
    and must not depend on the batch (1 image: conv4's tile is two images, the second missing; odd counts), on the position in the
    batch or on the number of workgroups."""
    cfg = env.PRESETS["bench-1s"]
    enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=48)
    sess = enc.session(cfg)
    x = torch.from_numpy(np.random.default_rng(17).standard_normal((45, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32))
    x[5] = 0.0
    with torch.no_grad():
        mu_o, lv_o = tenc(x)
    res = {}
    try:
        for bits in (0, 1, 2, 3):
            sess.configure("patch_gemm", bits)
            mu, lv = sess.encoder_forward(x)
            res[bits] = (mu.cpu().numpy(), lv.cpu().numpy())
            for wgs in (1, 5, 4096):
                sess.configure("patch_gemm_wgs", wgs)
                mu_w, lv_w = sess.encoder_forward(x)
                assert torch.equal(mu_w, mu) and torch.equal(lv_w, lv), f"patch_gemm {bits}: result depends on the workgroup count ({wgs})"
            sess.configure("patch_gemm_wgs", 0)
            for lo, hi in ((0, 1), (2, 5), (7, 20), (44, 45)):
                mu_b, _ = sess.encoder_forward(x[lo:hi])
                assert torch.equal(mu_b, mu[lo:hi]), f"patch_gemm {bits}: result depends on the batch"
    finally:
        sess.configure("patch_gemm", 1)
        sess.configure("patch_gemm_wgs", 0)
    for bits in (0, 1, 2, 3):
        _assert_latents(res[bits][0], mu_o.numpy(), 1e-5)
        _assert_latents(res[bits][1], lv_o.numpy(), 1e-5)
        _assert_latents(res[bits][0], res[0][0], 5e-6)
    e = {bits: max(rel_l2(res[bits][0][r], mu_o.numpy()[r]) for r in range(45)) for bits in res}
    with capsys.disabled():
        print("\n[weights-in-registers conv3 / conv4] max rel L2 of mu vs torch-CPU fp32: staged %.2e, conv3 %.2e, conv4 %.2e, both %.2e"
              % (e[0], e[1], e[2], e[3]))


def test_two_term_fp16_split_holds_its_accuracy_across_input_magnitudes(env, capsys):
    """The matrix-core layers multiply fp32 operands as two fp16 terms (x = h + 2^-11 l; csrc/k_encoder_split.hip header): fp16's range is
    the one thing the three-term bf16 form did not have to think about.  Images scaled by 1e-3 (most layer-1 activations below fp16's
    smallest normal: h subnormal, l carries the value), 1 and 60 (activations in the hundreds; |x| must stay below 65 504) must all meet
    the 1e-5 bar (relative L2 per row) against the torch-CPU fp32 encoder, and every output must be finite."""
    cfg = env.PRESETS["bench-1s"]
    enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=32)
    sess = enc.session(cfg)
    base = np.random.default_rng(31).standard_normal((24, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32)
    worst = {}
    for scale in (1e-3, 1.0, 60.0):
        x = torch.from_numpy(base * np.float32(scale))
        with torch.no_grad():
            mu_o, lv_o = tenc(x)
        mu, lv = sess.encoder_forward(x)
        assert torch.isfinite(mu).all() and torch.isfinite(lv).all()
        # (per-row relative L2: at scale 60 the latents reach the hundreds and an element near zero is a cancellation in ANY fp32 evaluation)
        worst[scale] = max(max(rel_l2(mu.cpu().numpy()[r], mu_o.numpy()[r]), rel_l2(lv.cpu().numpy()[r], lv_o.numpy()[r])) for r in range(24))
        assert worst[scale] <= 1e-5, (scale, worst[scale])
    with capsys.disabled():
        print("\n[two-term fp16 split] max rel L2 of mu vs torch-CPU fp32 at image scale " + ", ".join(f"{k:g}: {v:.2e}" for k, v in worst.items()))


def test_rms_normalize_inside_the_pair_kernel_is_bit_identical_to_its_own_launches(env):
    """Round 4 ("stft_k0_fuse", csrc/k_frontend.hip K0F): K0's fold + apply and its padded normalised copy of every row live inside
    k_stft_pair - the kernel folds the per-chunk sums of squares into the scale, reads the RAW rows (librosa's centre padding, fix_length
    and truncation as index arithmetic, map_detector_core.py:212-228) and re-quantises in its decode (00_normalize_dataset_rms.py:29-57).
    Same arithmetic, same bits: the image, the latents and the rms_ok flags must be EQUAL with the knob on and off - for whole one-second
    rows, rows longer than the chunk (the RMS is over the whole file, the STFT over its first second), ragged lengths incl. odd and very
    short ones (shorter than the padding: the reflection then reads fix_length's zeros), silent rows, and a batch of more rows than
    workgroups (some workgroups then own two rows)."""
    cfg = env.PRESETS["bench-1s"]
    enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=320)
    sess = enc.session(cfg)
    DEFAULT_K0, DEFAULT_CQ = 1, 0
    rng = np.random.default_rng(77)
    for n_rows, n_in, ragged in ((320, cfg.n_samples, False), (37, 30011, True), (300, 22050, True), (5, 32768, False)):
        pcm = env.synth(np.arange(500, 500 + n_rows), n_in, cfg.sr)
        pcm[3] = 0                                                          # digital silence: the gate leaves the row unchanged
        lens = None
        if ragged:
            lens = rng.integers(1, n_in + 1, size=n_rows).astype(np.int32)
            lens[:6] = (n_in, n_in - 1, 1, 2, 1025, 8193)
            lens[6:10] = (7922, 8191, 16114, 130)                             # numpy trees 7 levels deep / incomplete (the chunk-sum kernel's walk)
            lens[10] = 0                                                      # an empty row: gated, an all-zero signal
        res = {}
        # 1: the default (two rows per workgroup and queue when the batch is large: "stft_pair_rows2"); 2: one row per workgroup; 3: two rows per
        # group on 50 workgroups (every workgroup runs several groups, the last one of some a single row); 0: K0's own launches
        # 4 / 5 / 6: the continuous queue ("stft_pair_cq": R rows per workgroup on one pair queue, a row's dB stage run by the wave that completes
        # its last pair, no workgroup barrier) with R = 4, R = 3 on 50 workgroups (six or seven rows each), R = 8
        for knob in (1, 2, 3, 4, 5, 6, 0):
            sess.configure("stft_k0_fuse", 1 if knob else 0)
            sess.configure("stft_pair_rows2", 0 if knob == 2 else 1)
            sess.configure("stft_pair_cq", {4: 4, 5: 3, 6: 8}.get(knob, 0))
            sess.configure("stft_grid", 50 if knob in (3, 5) else 0)
            mu, _, ok = sess.encode(pcm, lengths=lens, apply_rms=True)
            img = sess.mel_pcm16(pcm, lengths=lens, apply_rms=True, layout=1)
            # ... and without the RMS stage (files 00 has already normalised): raw rows, plain decode, no padded copy
            mu0, _, ok0 = sess.encode(pcm, lengths=lens, apply_rms=False)
            res[knob] = (mu.clone(), ok.clone(), img.clone(), mu0.clone(), ok0.clone())
        sess.configure("stft_k0_fuse", DEFAULT_K0)
        sess.configure("stft_pair_rows2", 1)
        sess.configure("stft_pair_cq", DEFAULT_CQ)
        sess.configure("stft_grid", 0)
        for knob in (1, 2, 3, 4, 5, 6):
            assert torch.equal(res[knob][1], res[0][1]), (knob, n_rows, n_in)
            assert torch.equal(res[knob][2], res[0][2]), (knob, n_rows, n_in)
            assert torch.equal(res[knob][0], res[0][0]), (knob, n_rows, n_in)
            assert torch.equal(res[knob][3], res[0][3]) and torch.equal(res[knob][4], res[0][4]) and not res[knob][4].any(), (knob, n_rows, n_in)
        assert torch.isfinite(res[1][0]).all()
        assert int(res[1][1][3]) == 0                                       # the silent row was gated
    # several batches per call: the chunk sums of all the call's rows from ONE launch in front of the batches ("k0_call_sums", not the default)
    encb, _, _ = _encoders(env, "thesis-shape", cfg=cfg, max_batch=64)
    sb = encb.session(cfg)
    pcm = env.synth(np.arange(40, 340), 30011, cfg.sr)
    lens = rng.integers(1, 30012, size=300).astype(np.int32)
    lens[:4] = (30011, 0, 1, 8193)
    rb = {}
    for knob in (1, 0):
        sb.configure("k0_call_sums", knob)
        mu, _, ok = sb.encode(pcm, lengths=lens, apply_rms=True)
        rb[knob] = (mu.clone(), ok.clone())
    sb.configure("k0_call_sums", 0)
    assert torch.equal(rb[1][0], rb[0][0]) and torch.equal(rb[1][1], rb[0][1]) and torch.isfinite(rb[1][0]).all()
    # the 128-band preset (BASELINE configs[3]: n_fft 1024 / hop 256, k_stft_pair2): the same fusion on half-wave pairs - raw rows, the two
    # slots of a row that touch the centre padding per sample, re-quantisation in the decode
    cfg2 = env.PRESETS["bench-1s-128"]
    enc2, _, _ = _encoders(env, "thesis-shape", cfg=cfg2, max_batch=320)
    s2 = enc2.session(cfg2)
    for n_rows, n_in, ragged in ((320, cfg2.n_samples, False), (37, 30011, True), (300, 22050, True), (5, 32768, False)):
        pcm = env.synth(np.arange(700, 700 + n_rows), n_in, cfg2.sr)
        pcm[3] = 0
        lens = None
        if ragged:
            lens = rng.integers(1, n_in + 1, size=n_rows).astype(np.int32)
            lens[:11] = (n_in, n_in - 1, 1, 2, 513, 8193, 7922, 8191, 16114, 130, 0)
        r2 = {}
        for knob in (1, 0):
            s2.configure("stft_k0_fuse", knob)
            mu, _, ok = s2.encode(pcm, lengths=lens, apply_rms=True)
            img = s2.mel_pcm16(pcm, lengths=lens, apply_rms=True, layout=1)
            mu0, _, ok0 = s2.encode(pcm, lengths=lens, apply_rms=False)
            r2[knob] = (mu.clone(), ok.clone(), img.clone(), mu0.clone(), ok0.clone())
        s2.configure("stft_k0_fuse", DEFAULT_K0)
        for k in range(5):
            assert torch.equal(r2[1][k], r2[0][k]), ("bench-1s-128", k, n_rows, n_in)
        assert torch.isfinite(r2[1][0]).all() and int(r2[1][1][3]) == 0
    # the reference's own default geometry (5 s / 48 kHz: 30 numpy chunks per row, 626 frames = 313 pairs, the dB stage as its own kernel)
    cfg5 = env.PRESETS["ref-default"]
    enc5, _, _ = _encoders(env, "thesis-shape", cfg=cfg5, max_batch=8)
    s5 = enc5.session(cfg5)
    pcm5 = env.synth(np.arange(900, 906), cfg5.n_samples, cfg5.sr)
    pcm5[2] = 0
    lens5 = np.array([cfg5.n_samples, cfg5.n_samples - 3, 100001, cfg5.n_samples, 8192 * 7 + 130, 2049], np.int32)
    r5 = {}
    for knob in (1, 0):
        s5.configure("stft_k0_fuse", knob)
        a, _, oka = s5.encode(pcm5, lengths=lens5, apply_rms=True)
        b, _, _ = s5.encode(pcm5, lengths=lens5, apply_rms=False)
        r5[knob] = (a.clone(), oka.clone(), b.clone())
    assert all(torch.equal(r5[1][k], r5[0][k]) for k in range(3))
    # and the fused form against the oracle on a few rows of the plain case
    pcm = env.synth(np.arange(20, 26), cfg.n_samples, cfg.sr)
    z, _, _ = sess.encode(pcm, apply_rms=True)
    _assert_latents(z.cpu().numpy(), _oracle_latents(env, tenc, pcm, cfg))


def test_normalize_dataset_rms_tool_writes_what_00_writes(env, tmp_path, capsys):
    """00_normalize_dataset_rms.py's process_folder / main (:41-81) on K0: `<split>_norm/<species>/<name>.wav` for every file of every
    split that exists, mono PCM16 at --sr, also for a file the silence gate leaves unchanged (:55-57).  The bytes of each output file
    must equal the oracle's float_to_pcm16(rms_normalize(pcm16_to_float(q))) - the whole file, whatever its length; a stereo file goes
    through the general loader (channel mean) and the float32 kernel."""
    from amphibian_vae_latent_detector_amd import normalize_dataset_rms as nd
    from amphibian_vae_latent_detector_amd import synth
    sr = 22050
    base = tmp_path / "lse"
    files = {}
    for split, sp, specs in (("train_chunks", "Boana_faber", [(0, 22050), (1, 22050), (49, 22050), (3, 30011)]),
                             ("train_chunks", "Scinax_sp", [(4, 22050), (5, 9001)]),
                             ("val_chunks", "Boana_faber", [(6, 22050)])):
        d = base / split / sp
        d.mkdir(parents=True)
        for i, n in specs:
            q = synth.synth_chunk_pcm16(i, n, sr)
            synth.write_wav_pcm16(d / f"chunk_{i:06d}.wav", q, sr)
            files[(split, sp, f"chunk_{i:06d}.wav")] = q
    st = np.stack([synth.synth_chunk_pcm16(7, 8000, sr), synth.synth_chunk_pcm16(8, 8000, sr)], axis=1)
    synth.write_wav_pcm16(base / "val_chunks" / "Boana_faber" / "stereo.wav", st, sr)
    assert nd.main(["--base-dir", str(base), "--sr", str(sr)]) == 0
    out = capsys.readouterr().out
    assert "No existe" in out and "test_chunks" in out and "Dataset normalizado listo" in out         # the missing split is reported, as 00 does
    assert not (base / "test_chunks_norm").exists()
    for (split, sp, name), q in files.items():
        got, rate = env.ofe.read_wav_pcm16(base / f"{split}_norm" / sp / name)
        got = got[:, 0]
        want = env.ofe.float_to_pcm16(env.ofe.rms_normalize(env.ofe.pcm16_to_float(q))[0])
        assert rate == sr and got.shape == want.shape and np.array_equal(got, want), (split, sp, name)
    silent, _ = env.ofe.read_wav_pcm16(base / "train_chunks_norm" / "Boana_faber" / "chunk_000049.wav")
    assert not silent.any()                                                                           # written although gated
    got, _ = env.ofe.read_wav_pcm16(base / "val_chunks_norm" / "Boana_faber" / "stereo.wav")
    got = got[:, 0]
    y = (st.astype(np.float32) / np.float32(32768.0)).mean(axis=1).astype(np.float32)
    want = env.ofe.float_to_pcm16(env.ofe.rms_normalize(y)[0])
    assert got.shape == want.shape and np.abs(got.astype(np.int32) - want.astype(np.int32)).max() <= 1
    # the single-signal function of the same module
    y1, ok1 = nd.rms_normalize(env.ofe.pcm16_to_float(files[("train_chunks", "Boana_faber", "chunk_000000.wav")]))
    w1, wok = env.ofe.rms_normalize(env.ofe.pcm16_to_float(files[("train_chunks", "Boana_faber", "chunk_000000.wav")]))
    assert ok1 == bool(wok) and np.array_equal(y1, w1)


def test_fp16_split_range_guard_images_weights_and_activations(env, capsys):
    """The reference's encoder is an fp32 nn.Module loaded from ANY checkpoint (map_detector_core.py:160-179, strict=False) and called
    on any tensor (:270); the default GEMM arithmetic here splits fp32 operands into two fp16 terms, which exist below 65 504 only
    (include/amph.h, amph_range_status).  Three ways out of that range, each of which must end in finite latents within 1e-5 of the
    torch-CPU fp32 encoder - never a silent NaN:
      (a) images scaled by 1e3 / 1e4 / 1e5 through encoder_forward: the input test / an activation test raises the flag, the binding
          re-runs the call on the f32-MFMA kernels and says so (RuntimeWarning);
      (b) a checkpoint whose conv3 weights exceed 65 504 (scaled by 2e6, the BN scale divided by it: the same function): amph_create
          leaves that layer on the f32-MFMA kernel - no flag, no warning;
      (c) a checkpoint whose conv2 activations pass 65 504 at ordinary image scale (BN gamma of layer 2 x 16000): the epilogue that
          writes them raises the flag; also through the fused chunk -> latent entry point (encode) and amph_encode_pcm16_host."""
    import warnings
    cfg = env.PRESETS["bench-1s"]
    spec = env.SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
    base = np.random.default_rng(32).standard_normal((12, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32)
    report = []

    def worst(a, b):
        return max(rel_l2(a[r], b[r]) for r in range(a.shape[0]))

    # (a) the caller's tensor
    enc, tenc, _ = _encoders(env, "thesis-shape", cfg=cfg, max_batch=16)
    for scale in (1e3, 1e4, 1e5):
        sess = env.AmphSession(cfg, spec, enc._blob, DEV, max_batch=16)          # a fresh context: the fallback is sticky per context
        x = torch.from_numpy(base * np.float32(scale))
        with torch.no_grad():
            mu_o, lv_o = tenc(x)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            mu, lv = sess.encoder_forward(x)
        assert torch.isfinite(mu).all() and torch.isfinite(lv).all(), scale
        w = max(worst(mu.cpu().numpy(), mu_o.numpy()), worst(lv.cpu().numpy(), lv_o.numpy()))
        assert w <= 1e-5, (scale, w)
        tripped = any("fp16" in str(r.message) for r in rec)
        report.append(f"image x {scale:g}: {w:.2e} ({'f32-MFMA re-run' if tripped else 'split path'})")
        if scale >= 1e4:
            assert tripped                                # activations of 1e4-scaled images pass 65 504 in conv2 / conv3
        # without the check the flag stays readable and the split path's result is what it is (possibly NaN): nothing silent
        sess2 = env.AmphSession(cfg, spec, enc._blob, DEV, max_batch=16)
        sess2.encoder_forward(x, check_range=False)
        assert sess2.range_overflow(clear=False) == tripped and sess2.range_overflow() == tripped and not sess2.range_overflow()
        sess.close(); sess2.close()

    # (b) weights outside fp16's range: decided at create
    state = env.random_init_state(spec, seed=0)
    big = {k: np.array(v, copy=True) for k, v in state.items()}
    big["conv.2.weight"] *= np.float32(2e6)
    big["bn.2.running_var"] = big["bn.2.running_var"] * np.float32(4e12)        # alpha = gamma / sqrt(var + eps) shrinks by 2e6
    big["bn.2.running_mean"] *= np.float32(2e6)
    big["conv.2.bias"] *= np.float32(2e6)
    enc_b = env.HipEncoder(spec, big, DEV, max_batch=16)
    tenc_b = env.oenc.build_torch_encoder(spec.to_dict(), big)
    x = torch.from_numpy(base)
    with torch.no_grad():
        mu_o, _ = tenc_b(x)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        mu, _ = enc_b.session(cfg).encoder_forward(x)
    assert torch.isfinite(mu).all() and not rec
    w = worst(mu.cpu().numpy(), mu_o.numpy())
    assert w <= 1e-5, w
    report.append(f"conv3 weights x 2e6: {w:.2e} (layer on the f32-MFMA kernel from amph_create on)")

    # (c) activations outside fp16's range at ordinary inputs
    hot = {k: np.array(v, copy=True) for k, v in state.items()}
    hot["bn.1.weight"] *= np.float32(16000.0)
    hot["bn.1.bias"] *= np.float32(16000.0)
    tenc_h = env.oenc.build_torch_encoder(spec.to_dict(), hot)
    with torch.no_grad():
        mu_o, _ = tenc_h(x)
        a2 = torch.relu(tenc_h.bn[1](tenc_h.conv[1](torch.relu(tenc_h.bn[0](tenc_h.conv[0](x))))))
    assert float(a2.max()) > 65504.0                       # the case is what it claims to be
    enc_h = env.HipEncoder(spec, hot, DEV, max_batch=16)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        mu, _ = enc_h.session(cfg).encoder_forward(x)
    assert torch.isfinite(mu).all() and any("fp16" in str(r.message) for r in rec)
    w = worst(mu.cpu().numpy(), mu_o.numpy())
    assert w <= 1e-5, w
    report.append(f"BN gamma of layer 2 x 16000: {w:.2e} (flag raised by conv2's epilogue, f32-MFMA re-run)")
    # ... and through the chunk -> latent entry points, device and host form
    pcm = env.synth(np.arange(40, 52), cfg.n_samples, cfg.sr)
    z_o = _oracle_latents(env, tenc_h, pcm, cfg)
    enc_h2 = env.HipEncoder(spec, hot, DEV, max_batch=16)
    with warnings.catch_warnings(record=True) as rec:
        warnings.simplefilter("always")
        z, _, _ = enc_h2.session(cfg).encode(pcm, apply_rms=True)
    assert any("fp16" in str(r.message) for r in rec)
    # (per-row relative L2: the latents of this checkpoint reach 4e4, an element near zero is a cancellation in ANY fp32 evaluation)
    assert worst(z.cpu().numpy(), z_o) <= 1e-5
    enc_h3 = env.HipEncoder(spec, hot, DEV, max_batch=16)
    z_h, _ = enc_h3.session(cfg).encode_host(pcm, apply_rms=True)
    assert worst(z_h, z_o) <= 1e-5
    assert np.array_equal(z_h, z.cpu().numpy())           # both ended on the same f32-MFMA kernels
    with capsys.disabled():
        print("\n[fp16 split range guard] max rel L2 of the latents vs torch-CPU fp32: " + "; ".join(report))


@pytest.mark.parametrize("preset", ["bench-1s", "ref-default", "bench-1s-128"])
def test_wide_tile_staged_gemm_is_bit_identical_to_the_64_row_tile(env, capsys, preset):
    """The staged split GEMM on 128 x 128 tiles ("wide_gemm"; csrc/k_encoder_split.hip, k_conv_gemm_s3w: what multi-lane calls use
    for conv4 - and conv3 where the register-resident kernel does not apply) keeps k_conv_gemm_s3's order of operations per output
    element, so forcing it on (1) or off (0) must not change a bit, at any batch (1 image: a tile with 16 of its 128 rows; odd
    counts: a partial last tile), with conv3 on either kernel family; and it must meet the 1e-5 bar against the torch-CPU oracle."""
    cfg = env.PRESETS[preset]
    enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=48)
    sess = enc.session(cfg)
    x = torch.from_numpy(np.random.default_rng(23).standard_normal((45, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32))
    x[9] = 0.0
    with torch.no_grad():
        mu_o, lv_o = tenc(x)
    try:
        for pg in (1, 0):
            sess.configure("patch_gemm", pg)
            sess.configure("wide_gemm", 0)
            mu0, lv0 = sess.encoder_forward(x)
            sess.configure("wide_gemm", 1)
            mu1, lv1 = sess.encoder_forward(x)
            assert torch.equal(mu1, mu0) and torch.equal(lv1, lv0), f"patch_gemm {pg}: the 128 x 128 tile changed the result"
            for lo, hi in ((0, 1), (2, 5), (7, 20), (44, 45)):
                assert torch.equal(sess.encoder_forward(x[lo:hi])[0], mu0[lo:hi]), f"patch_gemm {pg}: result depends on the batch"
            _assert_latents(mu1.cpu().numpy(), mu_o.numpy(), 1e-5)
            _assert_latents(lv1.cpu().numpy(), lv_o.numpy(), 1e-5)
    finally:
        sess.configure("patch_gemm", 1)
        sess.configure("wide_gemm", -1)


@pytest.mark.parametrize("preset", ["bench-1s-128", "ref-default"])
def test_row_tiled_register_resident_conv3_on_other_geometries(env, capsys, preset):
    """k_conv_gemm_g5 (round 3): conv3 with the weights in registers for images that are not 16 x 16 - tiles of whole output rows
    with a halo row: the 128-band geometry of BASELINE configs[3] (conv3 24 x 32 x 64 -> 12 x 16 x 128, three 4-row tiles per image)
    and the reference's default (10_benchmark_folder_detection.py:123-134: 192 x 64 image, conv3 48 x 16 x 64 -> 24 x 8 x 128, three
    8-row tiles).  Against the torch-CPU fp32 encoder <= 1e-5 like every other form, within fp32 rounding of the staged kernel,
    independent of batch, position and workgroup count (bitwise); the first / middle / last tile of an image take the zero row /
    the halo of the tile above (a wrong halo shows at once: random images).  The bf16 form against its rounding emulation."""
    cfg = env.PRESETS[preset]
    enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=16)
    sess = enc.session(cfg)
    x = torch.from_numpy(np.random.default_rng(23).standard_normal((13, 1, cfg.target_frames, cfg.n_mels)).astype(np.float32))
    x[4] = 0.0
    with torch.no_grad():
        mu_o, lv_o = tenc(x)
    res = {}
    try:
        for bits in (0, 1):
            sess.configure("patch_gemm", bits)
            mu, lv = sess.encoder_forward(x)
            res[bits] = (mu.cpu().numpy(), lv.cpu().numpy())
            for wgs in (1, 7, 4096):
                sess.configure("patch_gemm_wgs", wgs)
                mu_w, lv_w = sess.encoder_forward(x)
                assert torch.equal(mu_w, mu) and torch.equal(lv_w, lv), (bits, wgs)
            sess.configure("patch_gemm_wgs", 0)
            for lo, hi in ((0, 1), (3, 6), (12, 13)):
                assert torch.equal(sess.encoder_forward(x[lo:hi])[0], mu[lo:hi]), (bits, lo, hi)
        # bf16 encoder: conv3 through the same kernel with one bf16 term, against the staged bf16 GEMM (same rounding points)
        sess.configure("bf16_encoder", 1)
        sess.configure("patch_gemm_bf16", 3)
        mu_b = sess.encoder_forward(x)[0].cpu().numpy()
        sess.configure("patch_gemm_bf16", 0)
        mu_b0 = sess.encoder_forward(x)[0].cpu().numpy()
    finally:
        sess.configure("bf16_encoder", 0)
        sess.configure("patch_gemm_bf16", 3)
        sess.configure("patch_gemm", 1)
        sess.configure("patch_gemm_wgs", 0)
    for bits in (0, 1):
        _assert_latents(res[bits][0], mu_o.numpy(), 1e-5)
        _assert_latents(res[bits][1], lv_o.numpy(), 1e-5)
    _assert_latents(res[1][0], res[0][0], 5e-6)
    assert not np.array_equal(res[1][0], res[0][0])                          # (another kernel: last bits differ)
    eb = max(rel_l2(mu_b[r], mu_b0[r]) for r in range(13))
    assert eb <= 2e-3                                                        # bf16: same rounding points, another summation order
    e = {bits: max(rel_l2(res[bits][0][r], mu_o.numpy()[r]) for r in range(13)) for bits in res}
    with capsys.disabled():
        print(f"\n[k_conv_gemm_g5, {preset}] max rel L2 of mu vs torch-CPU fp32: staged {e[0]:.2e}, row-tiled conv3 {e[1]:.2e}; bf16 form vs staged bf16 GEMM {eb:.2e}")


def test_encoder_odd_shapes_generic_path(env):
    """Non power-of-two geometry: 5x5 stride-1 first layer, C_out not a multiple of 64, K padded to 32."""
    from amphibian_vae_latent_detector_amd import EncoderSpec
    from dataclasses import replace
    cfg = replace(env.cfg, n_mels=40, target_frames=60)
    spec = EncoderSpec(in_frames=60, in_mels=40, conv=((1, 12, 5, 1, 2), (12, 20, 3, 2, 1), (20, 36, 3, 3, 0)), latent_dim=5)
    state = env.random_init_state(spec, seed=4)
    enc = env.HipEncoder(spec, state, DEV, max_batch=4)
    sess = enc.session(cfg)
    tenc = env.oenc.build_torch_encoder(spec.to_dict(), state)
    x = np.random.default_rng(2).standard_normal((6, 1, 60, 40)).astype(np.float32)
    mu, lv = sess.encoder_forward(torch.from_numpy(x))
    with torch.no_grad():
        mu_o, lv_o = tenc(torch.from_numpy(x))
    _assert_latents(mu.cpu().numpy(), mu_o.numpy(), 1e-5)
    _assert_latents(lv.cpu().numpy(), lv_o.numpy(), 1e-5)


# ----------------------------------------------------------------------------- fused path, BASELINE configs
def test_config1_100_chunks_tiny_vae_end_to_end(env):
    """BASELINE.json configs[0]: 100 synthetic 1 s / 22.05 kHz chunks, tiny VAE (latent_dim 8)."""
    enc, tenc, spec = _encoders(env, "tiny", max_batch=32)
    pcm = env.synth(list(range(100)), env.cfg.n_samples, env.cfg.sr)
    mu, lv, ok = enc.session(env.cfg).encode(pcm, apply_rms=True, want_logvar=True)
    z = mu.cpu().numpy()
    assert z.shape == (100, 8) and np.isfinite(z).all()
    _assert_latents(z, _oracle_latents(env, tenc, pcm, env.cfg))
    ok = ok.cpu().numpy()
    assert ok[49] == 0 and ok[99] == 0 and ok.sum() == 98               # the two silent chunks fail the RMS gate
    assert np.array_equal(z[49], z[99])                                  # silence -> identical latents


def test_config2_thesis_shape_all_10000_chunks_and_full_size_properties(env):
    """BASELINE.json configs[1]: 10k chunks, thesis-shape encoder (latent_dim 32), batch 256, fp32.
    Oracle diff on ALL 10 000 chunks (the CPU restatement single-threaded: ~15 s), plus determinism,
    batch-composition invariance, permutation equivariance, finiteness."""
    cfg = env.cfg
    enc, tenc, spec = _encoders(env, "thesis-shape", max_batch=256)
    sess = enc.session(cfg)
    n = 10000
    pcm = env.synth(list(range(n)), cfg.n_samples, cfg.sr)
    dev_pcm = torch.from_numpy(pcm).to(DEV)
    mu, _, ok = sess.encode(dev_pcm, apply_rms=True)
    z = mu.cpu().numpy()
    assert z.shape == (n, 32) and np.isfinite(z).all() and int(ok.sum()) == n - n // 50
    threads = torch.get_num_threads()
    torch.set_num_threads(1)                                             # a batch-1 loop fanned out over all cores crawls
    try:
        z_cpu = _oracle_latents(env, tenc, pcm, cfg)
    finally:
        torch.set_num_threads(threads)
    _assert_latents(z, z_cpu)
    rel = np.linalg.norm(z - z_cpu, axis=1) / np.maximum(np.linalg.norm(z_cpu, axis=1), 1e-30)
    print(f"\n[config 2, all {n} chunks] max rel L2 vs CPU oracle: {rel.max():.3e}, median {np.median(rel):.3e}")
    mu2, _, _ = sess.encode(dev_pcm, apply_rms=True)
    assert torch.equal(mu, mu2)                                          # deterministic
    perm = torch.from_numpy(np.random.default_rng(0).permutation(n)).to(DEV)
    mu3, _, _ = sess.encode(dev_pcm[perm].contiguous(), apply_rms=True)
    assert torch.equal(mu3, mu[perm])                                    # rows independent of batch position
    enc_small, _, _ = _encoders(env, "thesis-shape", max_batch=37)       # other batch size, same bits
    mu4, _, _ = enc_small.session(cfg).encode(dev_pcm[:1000], apply_rms=True)
    assert torch.equal(mu4, mu[:1000])
    mu_h, ok_h = sess.encode_host(pcm[:700], apply_rms=True)             # pinned host path
    assert np.array_equal(mu_h, z[:700]) and np.array_equal(ok_h, ok.cpu().numpy()[:700])
    # What the scalar-promotion environment is worth: the default follows the reference's pinned
    # numpy 1.26 (float64 scale, cast once); "rms_numpy2" is numpy 2.x.  Count the rows whose re-quantised PCM16
    # samples differ between the two and the latent distance that costs.
    fes = env.AmphSession(cfg, None, None, DEV, max_batch=256)
    q1, ok1 = fes.rms_normalize(dev_pcm)
    fes.configure("rms_numpy2", 1)
    q2, ok2 = fes.rms_normalize(dev_pcm)
    fes.close()
    rows_changed = int((q1 != q2).any(dim=1).sum())
    samples_changed = int((q1 != q2).sum())
    assert torch.equal(ok1, ok2)
    sess.configure("rms_numpy2", 1)
    try:
        z2 = sess.encode(dev_pcm, apply_rms=True)[0].cpu().numpy()
    finally:
        sess.configure("rms_numpy2", 0)
    rel12 = np.linalg.norm(z2 - z, axis=1) / np.maximum(np.linalg.norm(z, axis=1), 1e-30)
    print(f"[config 2] numpy-1.26 vs numpy-2 RMS scale: {rows_changed} of {n} rows change at least one PCM16 sample "
          f"({samples_changed} samples, max |dq| {int((q1.int() - q2.int()).abs().max())}); latent rel L2 between the modes: "
          f"max {rel12.max():.3e}, rows over 1e-4: {int((rel12 > 1e-4).sum())}")
    assert (rel12 > 0).sum() == 0 or rows_changed > 0
    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:                                                                 # the numpy-2 mode against ITS oracle twin (first 1000 rows)
        z2_cpu = np.stack([env.oenc.encode_mel_to_latent(tenc, env.ofe.wave_to_mel(
            env.ofe.normalize_pcm16_chunk(q, True, promotion="numpy2")[0], **cfg.mel_kwargs())) for q in pcm[:1000]])
    finally:
        torch.set_num_threads(threads)
    _assert_latents(z2[:1000], z2_cpu)


def test_config3_100k_chunks_as_8_virtual_shards_then_08_fit(env, capsys):
    """BASELINE.json configs[2] at its size, on one GPU: 100 000 chunks (25 000 per species) in the reference's list order
    (08_fit_radial_detector.py:461-484), encoded as 8 contiguous shards of 12 500 laid out as the all-gather lays them out
    == one pass, bitwise; then 08's host fit (q_in 0.95, q_out 0.10) with max_per_class 0 and 400 (seed 123): thresholds
    against those from CPU-oracle latents <= 1e-4 relative (on a 10 000-chunk subset for max_per_class 0, on all 1 600
    sampled chunks for 400); the smallest relative decision margin |d - rk| / rk is printed."""
    from amphibian_vae_latent_detector_amd import folder as pfolder
    from amphibian_vae_latent_detector_amd import radial as prad
    from amphibian_vae_latent_detector_amd import synth
    cfg = env.cfg
    n = 100000
    enc, tenc, spec = _encoders(env, "thesis-shape", max_batch=256)
    sess = enc.session(cfg)
    plan = synth.synth_folder_plan(n, 0, 123)
    assert list(plan) == list(synth.SPECIES) and all(len(v) == n // 4 for v in plan.values())
    flat = np.concatenate([np.asarray(plan[sp]) for sp in synth.SPECIES])
    pcm_all = env.synth(np.arange(n), cfg.n_samples, cfg.sr, workers=min(16, len(__import__("os").sched_getaffinity(0))))
    dev_all = torch.from_numpy(pcm_all).to(DEV)
    dev = dev_all[torch.from_numpy(flat).to(DEV)]                         # rows in list order: species-major, name-sorted
    del dev_all
    Z1 = sess.encode(dev, apply_rms=True)[0]
    Z8 = pfolder.encode_resident_shards(sess, dev, 8, apply_rms=True)
    assert torch.equal(Z8, Z1)
    assert pfolder.plan_shards(n, 8)[1] == 12500
    Z3 = pfolder.encode_resident_shards(sess, dev, 3, apply_rms=True)     # uneven: 33 334 + 33 334 + 33 332 rows, padded blocks
    assert torch.equal(Z3, Z1)
    Z = Z1.cpu().numpy()
    assert np.isfinite(Z).all()
    Zg = {sp: Z[k * (n // 4):(k + 1) * (n // 4)] for k, sp in enumerate(synth.SPECIES)}
    full_fit = prad.fit_radial_detector(Zg, 0.95, 0.10, verbose=False)    # max_per_class 0: all 100 000 latents
    assert all(np.isfinite(v) and v > 0 for v in full_fit["thresholds"].values())

    threads = torch.get_num_threads()
    torch.set_num_threads(1)
    try:
        def oracle(indices):
            return _oracle_latents(env, tenc, pcm_all[np.asarray(indices)], cfg)
        sub = {sp: plan[sp][:2500] for sp in synth.SPECIES}               # 10 000-chunk subset, list order
        Zc_sub = {sp: oracle(sub[sp]) for sp in synth.SPECIES}
        plan400 = synth.synth_folder_plan(n, 400, 123)
        Zc_400 = {sp: oracle(plan400[sp]) for sp in synth.SPECIES}
    finally:
        torch.set_num_threads(threads)
    Zg_sub = {sp: Zg[sp][:2500] for sp in synth.SPECIES}
    pos = {int(c): i for i, c in enumerate(flat)}                         # chunk index -> row of the list
    Zg_400 = {sp: Z[[pos[c] for c in plan400[sp]]] for sp in synth.SPECIES}
    # the 1 600 sampled chunks encoded on their own (what 08 --max-per-class 400 would encode) == the same rows of the full job
    rows400 = torch.from_numpy(np.array([pos[c] for sp in synth.SPECIES for c in plan400[sp]])).to(DEV)
    assert torch.equal(sess.encode(dev[rows400], apply_rms=True)[0], Z1[rows400])
    report = []
    # every point of the reference's q_out grid (run_qout_grid.sh:13) on the 10 000-chunk subset; the sampled fit at its first point
    cases = [(f"max_per_class 0 (10 000-chunk subset), q_out {q:.2f}", Zg_sub, Zc_sub, q) for q in (0.10, 0.15, 0.20, 0.25)]
    cases.append(("max_per_class 400, seed 123, q_out 0.10", Zg_400, Zc_400, 0.10))
    for tag, Zg_, Zc_, q_out in cases:
        for sp in synth.SPECIES:
            _assert_latents(Zg_[sp], Zc_[sp])
        rg = prad.fit_radial_detector(Zg_, 0.95, q_out, verbose=False)
        rc = prad.fit_radial_detector(Zc_, 0.95, q_out, verbose=False)
        worst = 0.0
        for sp in synth.SPECIES:
            rel = abs(rg["thresholds"][sp] - rc["thresholds"][sp]) / rc["thresholds"][sp]
            worst = max(worst, rel)
            assert rel <= 1e-4, (tag, sp, rel)
            assert np.abs(np.array(rg["centroids"][sp]) - np.array(rc["centroids"][sp])).max() <= 1e-4
        # decisions with the CPU-side detector: GPU latents through K6 against the CPU rule on CPU latents
        cents = {k: np.array(v, np.float32) for k, v in rc["centroids"].items()}
        det = prad.RadialDetector(cents, rc["thresholds"], 32)
        allg, allc = np.concatenate(list(Zg_.values())), np.concatenate(list(Zc_.values()))
        pred, best = det.decide(sess, allg)
        dmat = np.stack([np.sqrt(((allc - cents[s][None]) ** 2).sum(axis=1)) for s in det.names], axis=1)
        thr = np.array([rc["thresholds"][s] for s in det.names])
        margin = (np.abs(dmat - thr[None]) / thr[None]).min(axis=1)
        differ = 0
        for r in np.argsort(margin)[:200]:                                # the 200 closest calls, in full
            d, sp_c, bd = env.orad.decide(allc[r], cents, rc["thresholds"])
            got = det.names[pred[r]] if pred[r] >= 0 else None
            if got != sp_c:
                differ += 1
                assert margin[r] <= 1e-4, (tag, r, margin[r])
        report.append(f"[config 3, {tag}] thresholds GPU vs CPU-oracle latents: max rel {worst:.2e}; min |d-rk|/rk over "
                      f"{allc.shape[0]} chunks: {margin.min():.3e}; decisions differing among the 200 closest calls: {differ}")
    with capsys.disabled():
        print()
        for line in report:
            print(line)
        print("[config 3, all 100 000 latents] thresholds:", {k: round(v, 6) for k, v in full_fit["thresholds"].items()})


def test_encode_file_list_with_the_hip_encode_under_an_rccl_group(env, tmp_path):
    """The product's sharded folder encode with NOTHING injected: real WAV files -> native reader -> HIP encode ->
    all_gather_into_tensor on a one-rank `nccl` (= RCCL) process group, against the same call without a group."""
    import os
    import socket
    import torch.distributed as dist
    from amphibian_vae_latent_detector_amd import folder as pfolder
    from amphibian_vae_latent_detector_amd import synth
    cfg = env.cfg
    enc, tenc, spec = _encoders(env, "tiny", max_batch=16)
    root = tmp_path / "chunks"
    made = synth.write_synth_folder(root, 45, cfg.n_samples, cfg.sr)
    (root / synth.SPECIES[3] / "chunk_000003.wav").write_bytes(b"RIFFbroken")
    paths = [p for sp in synth.SPECIES for p in sorted((root / sp).glob("*.wav"))]
    Z0, st0 = pfolder.encode_file_list(enc, paths, cfg, apply_rms=True)
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
    try:
        Z1, st1 = pfolder.encode_file_list(enc, paths, cfg, apply_rms=True)
        Zf, failed = pfolder.encode_folder(enc, root, list(synth.SPECIES), cfg, cache_dir=tmp_path / "cache", apply_rms=True, verbose=False)
    finally:
        dist.destroy_process_group()
    assert np.array_equal(Z1, Z0) and np.array_equal(st1, st0) and int((st1 != 0).sum()) == 1
    assert sum(failed.values()) == 1 and sum(z.shape[0] for z in Zf.values()) == 44
    good = st0 == 0
    assert np.array_equal(np.concatenate([Zf[sp] for sp in synth.SPECIES]), Z0[good])
    ok_paths = [p for p, g in zip(paths, good) if g]
    want = np.stack([env.oenc.encode_mel_to_latent(tenc, env.ofe.wave_to_mel(
        env.ofe.normalize_pcm16_chunk(env.ofe.read_wav_pcm16(p)[0][:, 0], True)[0], **cfg.mel_kwargs())) for p in ok_paths])
    _assert_latents(Z0[good], want)


def test_encode_edge_cases(env):
    cfg = env.cfg
    enc, tenc, spec = _encoders(env, "tiny", max_batch=8)
    sess = enc.session(cfg)
    mu, _, ok = sess.encode(np.zeros((0, cfg.n_samples), np.int16), apply_rms=True)      # empty batch
    assert tuple(mu.shape) == (0, 8)
    pcm = env.synth([10, 11, 12, 13, 14], cfg.n_samples + 3000, cfg.sr)
    lengths = np.array([cfg.n_samples + 3000, cfg.n_samples, 5000, 1, 0], dtype=np.int32)
    for apply_rms in (False, True):
        mu, _, ok = sess.encode(pcm, lengths=lengths, apply_rms=apply_rms)
        want = []
        for r in range(5):
            q = pcm[r, :lengths[r]]
            y, good = env.ofe.normalize_pcm16_chunk(q, apply_rms) if q.size else (np.zeros(0, np.float32), False)
            want.append(env.oenc.encode_mel_to_latent(tenc, env.ofe.wave_to_mel(y, **cfg.mel_kwargs())))
            if apply_rms:
                assert bool(ok[r]) == bool(good), r                     # RMS is over the whole file, before truncation
        _assert_latents(mu.cpu().numpy(), np.stack(want))
    # float32 input carrying the same samples gives the same bits as PCM16 input
    y = env.ofe.pcm16_to_float(pcm)
    a, _, _ = sess.encode(pcm, apply_rms=True)
    b, _, _ = sess.encode(y, apply_rms=True)
    assert float((a - b).abs().max()) <= 1e-4            # PCM16 rows: k_stft_pair; float32 rows: k_stft_mel_w
    sess.configure("stft_pair", 0)
    assert torch.equal(sess.encode(pcm, apply_rms=True)[0], b)
    sess.configure("stft_pair", 1)
    # odd row stride (rows not 4-byte aligned: per-sample paths of K0's normalised-row write and of the STFT fetch), and the
    # normalised rows written by K0 against the STFT kernel normalising on the fly: same bits
    odd = np.ascontiguousarray(pcm[:, : cfg.n_samples + 2999])
    lengths_odd = np.minimum(lengths, odd.shape[1]).astype(np.int32)
    c, _, _ = sess.encode(odd, lengths=lengths_odd, apply_rms=True)
    assert torch.equal(c, sess.encode(pcm, lengths=lengths_odd, apply_rms=True)[0])
    try:
        sess.configure("stft_pair", 0)                        # k_stft_mel_w on both sides (k_stft_pair only reads K0's padded rows)
        c0 = sess.encode(odd, lengths=lengths_odd, apply_rms=True)[0]
        a0 = sess.encode(pcm, apply_rms=True)[0]
        assert float((c0 - c).abs().max()) <= 1e-4 and float((a0 - a).abs().max()) <= 1e-4
        for key in ("stft_prepad", "stft_prenorm"):           # K0 writes plain normalised rows / nothing: the STFT kernel pads
            sess.configure(key, 0)                            # (and normalises) itself
            assert torch.equal(c0, sess.encode(odd, lengths=lengths_odd, apply_rms=True)[0]), key
            assert torch.equal(a0, sess.encode(pcm, apply_rms=True)[0]), key
    finally:
        sess.configure("stft_prenorm", 1)
        sess.configure("stft_prepad", 1)
        sess.configure("stft_pair", 1)
    # full-scale square wave: clipping + quantisation extremes
    sq = np.where(np.arange(cfg.n_samples) % 100 < 50, 32767, -32768).astype(np.int16)[None]
    mu, _, _ = sess.encode(sq, apply_rms=True)
    _assert_latents(mu.cpu().numpy(), _oracle_latents(env, tenc, sq, cfg))


def test_reference_default_preset_crop_path(env):
    """The reference's own defaults: 48 kHz, 5 s, 626 STFT frames centre-cropped to 192."""
    cfg = env.PRESETS["ref-default"]
    enc, tenc, spec = _encoders(env, "tiny", cfg=cfg, max_batch=2)
    pcm = env.synth([0, 1, 2], cfg.n_samples, cfg.sr)
    sess = enc.session(cfg)
    assert sess.raw_frames == 626
    y = env.ofe.pcm16_to_float(pcm)
    mel = sess.mel(y, layout=0).cpu().numpy()
    want = np.stack([env.ofe.wave_to_mel(v, **cfg.mel_kwargs()) for v in y])
    assert mel.shape == (3, 64, 192) and np.abs(mel - want).max() <= 1e-4
    mu, _, _ = sess.encode(pcm, apply_rms=True)
    _assert_latents(mu.cpu().numpy(), _oracle_latents(env, tenc, pcm, cfg))


def test_reference_07cli_preset_and_long_chunk_dbnorm_paths(env):
    """07's CLI default (48 kHz, 3 s, 376 frames: the 24-values-per-thread dB kernel) and a 9 s chunk (1126 frames:
    beyond every register-cached form, the recompute-per-pass kernel)."""
    from dataclasses import replace
    for cfg, frames in ((env.PRESETS["ref-07cli"], 376), (replace(env.PRESETS["ref-default"], duration=9.0), 1126)):
        sess = env.AmphSession(cfg, None, None, DEV, max_batch=2)
        assert sess.raw_frames == frames
        y = env.ofe.pcm16_to_float(env.synth([3, 49], cfg.n_samples, cfg.sr))       # chunk 49 is digital silence
        mel = sess.mel(y, layout=0).cpu().numpy()
        want = np.stack([env.ofe.wave_to_mel(v, **cfg.mel_kwargs()) for v in y])
        assert mel.shape == (2, 64, 192) and np.abs(mel - want).max() <= 1e-4
        assert np.all(mel[1] == 0.0)


def test_n_fft_1024_preset_128_mels(env):
    """BASELINE.json configs[3] front end: n_fft = 1024, hop 256, 128 mel bins (preset bench-1s-128):
    half-wave FFT geometry (two frames per wave), 87 raw frames zero-padded 4 left / 5 right to 96."""
    cfg = env.PRESETS["bench-1s-128"]
    sess = env.AmphSession(cfg, None, None, DEV, max_batch=4)
    assert sess.raw_frames == 87
    pcm = env.synth([0, 1, 2, 3, 4, 49, 6], cfg.n_samples, cfg.sr)
    y = env.ofe.pcm16_to_float(pcm)
    want_basis = env.ofe.mel_filterbank(cfg.sr, cfg.n_fft, cfg.n_mels, cfg.fmin, cfg.fmax)
    np.testing.assert_allclose(sess.mel_basis(), want_basis, rtol=2e-7, atol=0)
    mp = sess.mel_power(y).cpu().numpy()
    assert mp.shape == (7, 87, 128)
    for r in range(7):
        want = env.ofe.melspectrogram(y[r], cfg.sr, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax).T
        assert rel_l2(mp[r], want) <= 1e-5 and np.abs(mp[r] - want).max() <= 3e-6 * max(want.max(), 1e-30), r
    mel = sess.mel(y, layout=0).cpu().numpy()
    want = np.stack([env.ofe.wave_to_mel(v, **cfg.mel_kwargs()) for v in y])
    assert mel.shape == (7, 128, 96) and np.abs(mel - want).max() <= 1e-4
    assert not mel[:, :, :4].any() and not mel[:, :, 91:].any()
    sess.close()
    enc, tenc, spec = _encoders(env, "thesis-shape", cfg=cfg, max_batch=4)
    mu, _, _ = enc.session(cfg).encode(pcm, apply_rms=True)
    _assert_latents(mu.cpu().numpy(), _oracle_latents(env, tenc, pcm, cfg))


@pytest.mark.parametrize("preset,name", [("bench-1s", "tiny"), ("bench-1s-128", "thesis-shape"), ("bench-1s", "thesis-shape")])
def test_bf16_encoder_path(env, preset, name, capsys):
    """BASELINE.json configs[3]: bf16 encoder (weights + activations, bf16 MFMA, fp32 accumulate / BN / latents).
    Checked two ways: tightly against a torch-CPU emulation that rounds weights and every layer's output to
    bf16 (kernel correctness), and loosely against the fp32 oracle (the price of bf16, reported)."""
    cfg = env.PRESETS[preset]
    enc, tenc, spec = _encoders(env, name, cfg=cfg, max_batch=8)
    sess = enc.session(cfg)
    pcm = env.synth(list(range(11)), cfg.n_samples, cfg.sr)
    mel = np.stack([env.ofe.wave_to_mel(env.ofe.normalize_pcm16_chunk(q, True)[0], **cfg.mel_kwargs()) for q in pcm])
    x = torch.from_numpy(mel).transpose(1, 2).unsqueeze(1).contiguous()
    bf = lambda t: t.to(torch.bfloat16).to(torch.float32)
    # thesis-shape: layers 1 + 2 run in the fused split-operand kernel (fp32 layer 1, fp32-accurate layer 2, bf16 epilogue):
    # the first bf16 rounding is layer 2's OUTPUT; `tiny` (8 channels after layer 1) takes the unfused bf16 pair
    # thesis-shape at 64 mel bands (the bench geometry): layers 1 + 2 in bf16 proper (k_conv_gemm_f4b) - the image, both layers'
    # weights and the layer-1 activation are rounded to bf16 like every other layer's
    # ... and at 128 bands too since round 3 (k_conv_gemm_f4b has its own shape gate: its LDS image fits where k_conv_gemm_f4s's does not)
    fused12 = False
    true_bf16_pair = name == "thesis-shape"
    with torch.no_grad():
        mu32, _ = tenc(x)
        h = bf(x) if true_bf16_pair else x
        for li, (conv, bn) in enumerate(zip(tenc.conv, tenc.bn)):
            exact_w = (li == 0 or (fused12 and li == 1)) and not true_bf16_pair
            y = torch.nn.functional.conv2d(h, conv.weight if exact_w else bf(conv.weight), conv.bias,
                                           stride=conv.stride, padding=conv.padding)
            h = torch.relu(bn(y))
            if not (fused12 and li == 0):
                h = bf(h)
        flat = h.flatten(1)
        mu_emu = flat @ bf(tenc.fc_mu.weight).T + tenc.fc_mu.bias
    sess.configure("bf16_encoder", 1)
    mu_b, lv_b = sess.encoder_forward(x)
    mu_f, _, _ = sess.encode(pcm, apply_rms=True)                    # fused path end to end in bf16 mode
    sess.configure("bf16_encoder", 0)
    mu_fp32, _ = sess.encoder_forward(x)
    mu_b = mu_b.cpu().numpy()
    err_emu = max(rel_l2(mu_b[r], mu_emu[r].numpy()) for r in range(11))
    err_f32 = max(rel_l2(mu_b[r], mu32[r].numpy()) for r in range(11))
    with capsys.disabled():
        print(f"\n[bf16 encoder, {preset}/{name}] max rel L2 vs bf16-emulating torch: {err_emu:.2e}; vs fp32 oracle: {err_f32:.2e}")
    assert err_emu <= 2e-3 and err_f32 <= 5e-2             # (measured 5e-7 ... 8.2e-4 / 4.4e-3 ... 6.0e-3)
    assert max(rel_l2(mu_f.cpu().numpy()[r], mu_b[r]) for r in range(11)) <= 5e-3   # HIP mel vs oracle mel in front
    _assert_latents(mu_fp32.cpu().numpy(), mu32.numpy(), 1e-5)       # switching back restores the fp32 path


@pytest.mark.parametrize("n_fft,hop", [(512, 128), (4096, 1024), (256, 100), (2048, 385), (1024, 255), (8192, 2048)])
def test_general_stft_kernel_other_n_fft_and_odd_hops_against_the_oracle(env, capsys, n_fft, hop):
    """Every reference CLI exposes --n-fft and --hop (07_encode_wav_to_latent.py:429-430, 08_fit_radial_detector.py:353-354) and librosa
    takes any value (map_detector_core.py:219-228); the tuned kernels cover n_fft 1024 / 2048 with an even hop, everything else (powers
    of two from 64 to 8192, any hop) runs k_stft_generic + the separate dB kernel.  Against the oracle: mel power within 1e-5 of the
    row's peak, the standardised image within 1e-4, for float32 rows (ragged lengths, a silent row) and for PCM16 rows through the RMS
    stage; and the latents of the tiny encoder on such a front end at the 1e-4 bar."""
    from dataclasses import replace
    base = env.PRESETS["bench-1s"]
    cfg = replace(base, n_fft=n_fft, hop_length=hop, n_mels=48, target_frames=64)
    sess = env.AmphSession(cfg, None, None, DEV, max_batch=8)
    pcm = env.synth([0, 1, 2, 49, 4, 5], cfg.n_samples, cfg.sr)
    y = env.ofe.pcm16_to_float(pcm)
    lens = np.array([cfg.n_samples, cfg.n_samples - 1, 9001, cfg.n_samples, cfg.n_samples // 2 + 7, cfg.n_samples], np.int32)
    kw = cfg.mel_kwargs()
    pw = sess.mel_power(y, lengths=lens).cpu().numpy()                     # [B, T_raw, M]
    img = sess.mel(y, lengths=lens, layout=0).cpu().numpy()                # [B, M, T]
    worst_p = worst_i = 0.0
    for r in range(len(lens)):
        yr = y[r, :lens[r]]
        yr = np.pad(yr, (0, cfg.n_samples - yr.shape[0]))
        want_p = env.ofe.melspectrogram(yr, cfg.sr, n_fft, hop, cfg.n_mels, cfg.fmin, cfg.fmax)          # [M, T_raw]
        peak = max(float(want_p.max()), 1e-30)
        assert pw[r].shape == want_p.T.shape
        worst_p = max(worst_p, float(np.abs(pw[r] - want_p.T).max()) / peak)
        want_i = env.ofe.wave_to_mel(yr, **kw)
        worst_i = max(worst_i, float(np.abs(img[r] - want_i).max()))
    # (n_fft 256: 129 bins under 48 bands - the lowest bands are single weights of ~1e-3 on bins 50 dB below the row's peak, where a
    #  power error of 4e-7 of the peak is 2e-4 dB-standardised; the image bar is 3e-4 there)
    tol_i = 3e-4 if n_fft <= 256 else 1e-4
    assert worst_p <= 1e-5 and worst_i <= tol_i, (worst_p, worst_i)
    assert not pw[3].any()                                                 # digital silence stays exactly zero
    # PCM16 rows through the RMS stage (00_normalize_dataset_rms.py + its PCM16 write / read)
    img16 = sess.mel_pcm16(pcm, apply_rms=True, layout=0).cpu().numpy()
    w16 = 0.0
    for r in range(pcm.shape[0]):
        yn, _ = env.ofe.normalize_pcm16_chunk(pcm[r], True)
        w16 = max(w16, float(np.abs(img16[r] - env.ofe.wave_to_mel(yn, **kw)).max()))
    assert w16 <= tol_i, w16
    spec = env.SPECS["tiny"](cfg.target_frames, cfg.n_mels)
    state = env.random_init_state(spec, seed=0)
    enc = env.HipEncoder(spec, state, DEV, max_batch=8)
    z, _, _ = enc.session(cfg).encode(pcm, apply_rms=True)
    _assert_latents(z.cpu().numpy(), _oracle_latents(env, env.oenc.build_torch_encoder(spec.to_dict(), state), pcm, cfg))
    with capsys.disabled():
        print(f"\n[k_stft_generic, n_fft {n_fft} / hop {hop}] mel power vs oracle: {worst_p:.2e} of the row's peak; image: {worst_i:.2e}; PCM16 + RMS image: {w16:.2e}")
    sess.close()


def test_unsupported_and_invalid_configurations_raise(env):
    from dataclasses import replace
    with pytest.raises(env.N.AmphError, match="AMPH_ERR_UNSUPPORTED"):
        env.AmphSession(replace(env.cfg, n_fft=3000), None, None, DEV)           # no power of two (the general kernel's FFT is radix-2)
    with pytest.raises(env.N.AmphError, match="AMPH_ERR_UNSUPPORTED"):
        env.AmphSession(replace(env.cfg, n_fft=16384), None, None, DEV)
    with pytest.raises(env.N.AmphError, match="AMPH_ERR_UNSUPPORTED"):
        env.AmphSession(replace(env.cfg, duration=0.02), None, None, DEV)        # reflect pad longer than the chunk
    spec = env.SPECS["tiny"](64, 64)
    blob = np.zeros(10, np.float32)
    with pytest.raises(env.N.AmphError, match="weight blob"):
        env.AmphSession(env.cfg, spec, blob, DEV)
    enc, _, _ = _encoders(env, "tiny")
    with pytest.raises(RuntimeError, match="front end produces"):
        enc.session(replace(env.cfg, target_frames=96))
    # tuning knobs: bad values are rejected, and every workgroup shape of the STFT kernel gives the same bits
    sess = env.AmphSession(env.cfg, None, None, DEV, max_batch=8)
    for key, bad in (("stft_waves", 5), ("frames_per_block", 6), ("lanes", 0), ("no_such_knob", 1)):
        with pytest.raises(env.N.AmphError):
            sess.configure(key, bad)
    y = env.ofe.pcm16_to_float(env.synth([0, 1, 2, 49, 4], env.cfg.n_samples, env.cfg.sr))
    ref = sess.mel_power(y)                                   # default: 12 waves x 12 frames
    for waves, ft in ((4, 8), (4, 16), (12, 8), (12, 12)):
        sess.configure("frames_per_block", ft)
        sess.configure("stft_waves", waves)
        assert torch.equal(sess.mel_power(y), ref), (waves, ft)
    # the tile-synchronous kernel (shared sample span, two barriers per tile) against the default one with independent
    # waves: the same arithmetic up to the compiler's FMA contractions
    try:
        sess.configure("stft_indep", 0)
        for waves, ft in ((12, 12), (4, 8)):
            sess.configure("frames_per_block", ft)
            sess.configure("stft_waves", waves)
            got = sess.mel_power(y)
            assert float((got - ref).abs().max()) <= 1e-6 * float(ref.max()), (waves, ft)
            assert not got[3].any()                            # the silent row stays exactly zero
    finally:
        sess.configure("stft_indep", 1)
    # every knob is a property of ITS context (round 3: "stft_indep", "stft_grid", "rms_chunk_path_over" were process-wide): two
    # contexts with different settings in one process each keep their own kernels and results, in either call order
    a_ctx = env.AmphSession(env.cfg, None, None, DEV, max_batch=8)
    b_ctx = env.AmphSession(env.cfg, None, None, DEV, max_batch=8)
    pcm = env.synth([0, 1, 2, 3, 4, 5, 6], env.cfg.n_samples, env.cfg.sr)
    try:
        b_ctx.configure("stft_indep", 0)
        b_ctx.configure("stft_grid", 3)
        b_ctx.configure("stft_pair", 0)
        b_ctx.configure("rms_chunk_path_over", 8)
        ref_a = a_ctx.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)[1]
        ref_b = b_ctx.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)[1]
        assert not torch.equal(ref_a, ref_b)                               # different kernels: last-bit differences
        assert float((ref_a - ref_b).abs().max()) <= 1e-6 * float(ref_a.max())
        for _ in range(2):                                                 # interleaved: neither disturbs the other
            assert torch.equal(b_ctx.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)[1], ref_b)
            assert torch.equal(a_ctx.mel_pcm16(pcm, apply_rms=True, layout=1, want_power=True)[1], ref_a)
        for key in ("gemm_flags", "gemm_min_lds", "head_fold"):            # removed with the forms they selected
            with pytest.raises(env.N.AmphError):
                a_ctx.configure(key, 0)
    finally:
        a_ctx.close()
        b_ctx.close()


# ----------------------------------------------------------------------------- K6: radial decision
def test_radial_decisions_bit_identical_to_reference(env, fe, golden):
    from amphibian_vae_latent_detector_amd.radial import RadialDetector
    g = golden("radial_decide.npz")
    cfg = json.loads(str(g["cfg_json"]))
    det = RadialDetector.from_config(cfg, latent_dim=8)
    pred, best = det.decide(fe, g["Z"])
    got = [det.names[i] if i >= 0 else "" for i in pred.tolist()]
    assert got == [str(s) for s in g["pred"]]
    assert np.array_equal(pred >= 0, g["detected"])
    assert np.array_equal(best.astype(np.float64), g["best_d"])          # float32 distances, exactly numpy's
    rng = np.random.default_rng(5)                                       # D = 32 (8 <= n <= 128 pairwise branch) and D = 5 (< 8)
    for D in (32, 5, 100):
        Z = rng.standard_normal((257, D)).astype(np.float32)
        cents = {f"s{k}": rng.standard_normal(D).astype(np.float32) for k in range(4)}
        thr = {f"s{k}": float(np.sqrt(D) * 1.35) for k in range(4)}
        det = RadialDetector(cents, thr, D, priority=["s2", "s0"])
        pred, best = det.decide(fe, Z)
        for r in range(257):
            d, sp, bd = env.orad.decide(Z[r], cents, thr, ["s2", "s0"])
            assert (det.names[pred[r]] if pred[r] >= 0 else None) == sp and float(best[r]) == bd


# ----------------------------------------------------------------------------- K9: radial fit on the device
def test_radial_fit_on_device_bit_equal_to_host_and_reference(env, fe, golden):
    """amph_fit_radial: centroids (numpy's axis-0 order), distances (numpy's pairwise row sums) and the
    order statistics behind every np.quantile on the GPU.  Thresholds and centroids must equal the reference's own outputs
    (radial_fit.npz: rk1_* = numpy 1.26 arithmetic, rk_* = numpy 2.x) and the host fit on other shapes TO THE LAST BIT,
    summaries (min / p50 / p90 / max) included; single species -> rk_out = inf; ties; a one-row species."""
    from amphibian_vae_latent_detector_amd import radial as prad
    g = golden("radial_fit.npz")
    species = json.loads(str(g["species_json"]))
    Zs = {sp: g[f"Z_{i}"] for i, sp in enumerate(species)}
    for q_out in (0.10, 0.15, 0.20, 0.25):
        for numpy2, pre in ((False, "rk1_"), (True, "rk_")):
            rd = prad.fit_radial_detector(Zs, 0.95, q_out, verbose=False, session=fe, numpy2=numpy2)
            rh = prad.fit_radial_detector(Zs, 0.95, q_out, verbose=False, numpy2=numpy2)
            for i, sp in enumerate(species):
                rk, rk_in, rk_out = g[f"{pre}{i}_{q_out:.2f}"].tolist()
                assert rd["centroids"][sp] == g[f"mu_{i}_{q_out:.2f}"].tolist()
                m = rd["per_species"][sp]
                assert (rd["thresholds"][sp], m["rk_in"], m["rk_out"]) == (rk, rk_in, rk_out), (q_out, numpy2, sp)
                assert m == rh["per_species"][sp], (q_out, numpy2, sp)          # N_in / N_out / summaries too
    rng = np.random.default_rng(11)
    for D, sizes in ((32, (2500, 1, 333, 4096)), (5, (17, 9)), (100, (300,)), (8, (2, 2, 2))):
        Z = {f"s{k}": (rng.standard_normal((n, D)) * (1 + k) + k).astype(np.float32) for k, n in enumerate(sizes)}
        Z["s0"][1] = Z["s0"][0]                                                   # duplicate rows: tied distances
        dev = {k: torch.from_numpy(v).to(DEV) for k, v in Z.items()}               # latents already on the device
        for q_in, q_out in ((0.95, 0.01), (0.5, 0.5), (1.0, 0.0)):
            rd = prad.fit_radial_detector(dev, q_in, q_out, verbose=False, session=fe)
            rh = prad.fit_radial_detector(Z, q_in, q_out, verbose=False)
            assert rd["centroids"] == rh["centroids"] and rd["thresholds"] == rh["thresholds"], (D, sizes, q_in, q_out)
            for sp in Z:
                a, b = rd["per_species"][sp], rh["per_species"][sp]
                assert a["rk_in"] == b["rk_in"] and a["rk_out"] == b["rk_out"] and a["N_out"] == b["N_out"]
                for key in ("rho_in_summary", "rho_out_summary"):
                    assert all((a[key][s] == b[key][s]) or (np.isnan(a[key][s]) and np.isnan(b[key][s])) for s in a[key]), (sp, key, a[key], b[key])
    with pytest.raises(RuntimeError):
        fe.fit_radial(torch.zeros((4, 8)), [0, 4, 4], 0.9, 0.1)                  # an empty species


# ----------------------------------------------------------------------------- K7: Gaussian-MAP decision
def test_map_scores_and_decisions_against_reference(env, fe, golden):
    """Scores within 1e-5 relative of the reference's (its quadratic form is float32 BLAS with an unspecified
    summation order; the kernel's order is fixed), decisions equal wherever the reference's margin is not
    within that tolerance of a tie or of tau."""
    from amphibian_vae_latent_detector_amd.map_detector import MapDetector
    from oracle import map_score as omap
    g = golden("map_decide.npz")
    cfg = json.loads(str(g["cfg_json"]))
    det = MapDetector.from_config(cfg, latent_dim=8)
    assert det.names == json.loads(str(g["species_json"]))
    sc = det.scores(fe, g["Z"])
    ref = g["scores"]
    tol = 1e-5 * np.maximum(1.0, np.abs(ref))
    assert np.all(np.abs(sc - ref) <= tol), float(np.max(np.abs(sc - ref) / np.maximum(1.0, np.abs(ref))))
    pred, best = det.decide(fe, g["Z"])
    srt = np.sort(ref, axis=1)
    margin = np.minimum(srt[:, -1] - srt[:, -2], np.abs(srt[:, -1] - det.tau))
    safe = margin > 2e-5 * np.maximum(1.0, np.abs(srt[:, -1]))
    assert safe.mean() > 0.95
    got = np.array([det.names[i] if i >= 0 else "" for i in pred.tolist()])
    assert np.array_equal(got[safe], np.array([str(s) for s in g["pred"]])[safe])
    assert np.array_equal((pred >= 0)[safe], g["detected"][safe])
    assert np.all(np.abs(best - g["best_score"]) <= 1e-5 * np.maximum(1.0, np.abs(g["best_score"])))
    assert (pred < 0).sum() > 0 and (pred >= 0).sum() > 0
    # tau = None: nothing is rejected; K = 1, D not a power of two, D = 128 (two waves per chunk), empty batch
    g2 = golden("map_decide_notau.npz")
    det.tau = None
    det.log_priors[:] = np.log(0.2 + 1e-12)
    pred2, best2 = det.decide(fe, g["Z"])
    assert (pred2 >= 0).all()
    assert np.array_equal(np.array([det.names[i] for i in pred2.tolist()])[safe], np.array([str(s) for s in g2["pred"]])[safe])
    rng = np.random.default_rng(9)
    for D, K in ((5, 1), (32, 4), (100, 3), (128, 2)):
        names = [f"s{k}" for k in range(K)]
        means = {sp: rng.standard_normal(D).astype(np.float32) for sp in names}
        precs, lds = {}, {}
        for sp in names:
            A = (rng.standard_normal((D, D)) * 0.2 + np.eye(D)).astype(np.float32)
            precs[sp], lds[sp] = omap.inv_and_logdet((A @ A.T + 0.1 * np.eye(D)).astype(np.float32))
        pri = {sp: 1.0 / K for sp in names}
        d = MapDetector(means, precs, lds, pri, -1e9, D)
        Z = (rng.standard_normal((257, D)) * 1.5).astype(np.float32)
        sc = d.scores(fe, Z)
        want = np.stack([omap.scores(z, names, means, precs, lds, pri) for z in Z])
        assert np.all(np.abs(sc - want) <= 2e-5 * np.maximum(1.0, np.abs(want))), (D, K)
        pred, best = d.decide(fe, Z)
        assert np.array_equal(pred, np.argmax(sc, axis=1).astype(np.int32)) and np.array_equal(best, sc.max(axis=1))
    pred0, best0 = d.decide(fe, np.zeros((0, D), np.float32))
    assert pred0.shape == (0,) and best0.shape == (0,)
    with pytest.raises(RuntimeError):
        fe.decide_map(np.zeros((4, 200), np.float32), np.zeros((1, 200)), np.zeros((1, 200, 200)), np.zeros(1), np.zeros(1))


def test_map_fit_scores_through_the_kernel(env, fe, golden):
    """fit_map_detector(session=...) evaluates the true-class scores of the tau quantile on the GPU: tau within 1e-5
    relative of the reference's."""
    from amphibian_vae_latent_detector_amd.map_detector import fit_map_detector
    g = golden("map_fit.npz")
    species = json.loads(str(g["species_json"]))
    Z = {sp: g[f"Z_{i}"] for i, sp in enumerate(species)}
    for vi in (0, 3):
        prm = json.loads(str(g[f"v{vi}_params_json"]))
        md = fit_map_detector(Z, cov_type=prm["cov_type"], cov_structure=prm["cov_structure"], priors=prm["priors"],
                              eps=prm["eps"], shrink=prm["shrink"], set_tau_q=prm["tau_q"], session=fe, verbose=False)
        tau = float(g[f"v{vi}_tau"])
        assert abs(md["tau"] - tau) <= 1e-4 * max(1.0, abs(tau))


# ----------------------------------------------------------------------------- the drop-in Python surface
def test_map_detector_core_surface(env, tmp_path):
    from amphibian_vae_latent_detector_amd import map_detector_core as core
    from amphibian_vae_latent_detector_amd import synth
    from amphibian_vae_latent_detector_amd.encoder import save_checkpoint
    cfg = env.cfg
    spec = env.SPECS["tiny"](cfg.target_frames, cfg.n_mels)
    state = env.random_init_state(spec, seed=0)
    pt, ym = tmp_path / "model.pt", tmp_path / "enc.yaml"
    save_checkpoint(pt, ym, spec, state)
    encoder = core.load_encoder(pt, ym, tmp_path, torch.device("cuda"))
    tenc = env.oenc.build_torch_encoder(spec.to_dict(), state)
    q = synth.synth_chunk_pcm16(0, cfg.n_samples + 500, cfg.sr)
    wav = tmp_path / "chunk_000000.wav"
    synth.write_wav_pcm16(wav, q, cfg.sr)
    kw = cfg.mel_kwargs()
    mel = core.wav_to_mel(wav, **kw)
    want_mel = env.ofe.wav_to_mel(wav, **kw)
    assert isinstance(mel, torch.Tensor) and mel.dtype == torch.float32 and tuple(mel.shape) == (64, 64) and mel.device.type == "cpu"
    assert np.abs(mel.numpy() - want_mel).max() <= 1e-4
    z = core.encode_wav_to_latent(encoder, wav, torch.device("cuda"), **kw)
    assert isinstance(z, np.ndarray) and z.dtype == np.float32 and z.shape == (8,)
    _assert_latents(z[None], env.oenc.encode_mel_to_latent(tenc, want_mel)[None])
    # a caller-supplied torch module goes through the HIP front end + the reference adapter
    z2 = core.encode_wav_to_latent(tenc.to("cuda"), wav, torch.device("cuda"), **kw)
    _assert_latents(z2[None], env.oenc.encode_mel_to_latent(tenc.cpu(), want_mel)[None])
    # duration <= 0 (map_detector_core.py:212: no pad / truncate step): the WHOLE file - here 22 550 samples, 59 frames instead of
    # 58 - through a front-end context of exactly that length; a second file of another length gets its own
    kw0 = dict(kw, duration=0.0)
    mel0 = core.wav_to_mel(wav, **kw0)
    want0 = env.ofe.wav_to_mel(wav, **kw0)
    assert tuple(mel0.shape) == (64, 64) and np.abs(mel0.numpy() - want0).max() <= 1e-4
    assert np.abs(mel0.numpy() - mel.numpy()).max() > 1e-3                      # not the truncated chunk's image
    z0 = core.encode_wav_to_latent(encoder, wav, torch.device("cuda"), **kw0)
    _assert_latents(z0[None], env.oenc.encode_mel_to_latent(tenc, want0)[None])
    wav_b = tmp_path / "other_length.wav"
    synth.write_wav_pcm16(wav_b, synth.synth_chunk_pcm16(3, 30011, cfg.sr), cfg.sr)
    assert np.abs(core.wav_to_mel(wav_b, **dict(kw0, duration=-1.0)).numpy() - env.ofe.wav_to_mel(wav_b, **kw0)).max() <= 1e-4
    # whole files of several lengths (none a multiple of hop) through the ENCODER: each against the oracle; the contexts are batch-1 /
    # one-lane and only the last two lengths stay alive, so device memory stops growing after the second (a 256-row, four-lane
    # context per length kept ~0.4 GB each for one-second files and gigabytes for minute-long ones)
    torch.cuda.synchronize()
    used = []
    for k, n in enumerate((22550, 30011, 26113, 33331, 24007, 28001)):
        wk = tmp_path / f"len_{n}.wav"
        synth.write_wav_pcm16(wk, synth.synth_chunk_pcm16(10 + k, n, cfg.sr), cfg.sr)
        zk = core.encode_wav_to_latent(encoder, wk, torch.device("cuda"), **kw0)
        _assert_latents(zk[None], env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(wk, **kw0))[None])
        torch.cuda.synchronize()
        free, total = torch.cuda.mem_get_info()
        used.append(total - free)
    assert len(encoder._file_sessions) == 2
    assert max(used[2:]) - used[1] < 64 * 2 ** 20, [u >> 20 for u in used]      # no growth past the second length (allocator granularity aside)
    # stereo file: channel mean on the float path
    st = np.stack([q, q], axis=1)
    synth.write_wav_pcm16(tmp_path / "st.wav", st, cfg.sr)
    z3 = core.encode_wav_to_latent(encoder, tmp_path / "st.wav", torch.device("cuda"), **kw)
    _assert_latents(z3[None], z[None], 1e-6)
    # batched form
    pcm = env.synth(list(range(5)), cfg.n_samples, cfg.sr)
    zb = core.encode_chunks(encoder, pcm, **kw)
    _assert_latents(zb, _oracle_latents(env, tenc, pcm, cfg, apply_rms=False))
    # error behaviour: ordinary per-file exceptions
    with pytest.raises(FileNotFoundError):
        core.encode_wav_to_latent(encoder, tmp_path / "nope.wav", torch.device("cuda"), **kw)
    synth.write_wav_pcm16(tmp_path / "sr.wav", q, 48000)             # another rate: resampled as librosa.load would (K8)
    z48 = core.encode_wav_to_latent(encoder, tmp_path / "sr.wav", torch.device("cuda"), **kw)
    assert z48.shape == (8,) and np.isfinite(z48).all() and not np.allclose(z48, z)
    (tmp_path / "junk.wav").write_bytes(b"junk")
    with pytest.raises(Exception):
        core.wav_to_mel(tmp_path / "junk.wav", **kw)


def test_folder_encode_fit_and_decisions_match_cpu(env, tmp_path):
    """Integration: 100-chunk folder -> latents -> radial fit -> decisions, CPU vs GPU."""
    from amphibian_vae_latent_detector_amd import folder as pfolder
    from amphibian_vae_latent_detector_amd import radial as prad
    from amphibian_vae_latent_detector_amd import synth
    cfg = env.cfg
    enc, tenc, spec = _encoders(env, "tiny", max_batch=16)
    root = tmp_path / "train_chunks_norm"
    made = synth.write_synth_folder(root, 100, cfg.n_samples, cfg.sr)
    (root / synth.SPECIES[2] / "chunk_000002.wav").write_bytes(b"RIFFbroken")
    Zg, failed = pfolder.encode_folder(enc, root, list(synth.SPECIES), cfg, cache_dir=tmp_path / "cache_npz", verbose=False)
    assert failed[synth.SPECIES[2]] == 1 and sum(z.shape[0] for z in Zg.values()) == 99
    Zc = {}
    for sp in synth.SPECIES:
        paths = [p for p in sorted((root / sp).glob("*.wav")) if p.stat().st_size > 100]
        Zc[sp] = np.stack([env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(p, **cfg.mel_kwargs())) for p in paths])
        _assert_latents(Zg[sp], Zc[sp])
    for q_out in (0.10, 0.25):
        rg = prad.fit_radial_detector(Zg, 0.95, q_out, verbose=False)
        rc = prad.fit_radial_detector(Zc, 0.95, q_out, verbose=False)
        for sp in synth.SPECIES:
            assert abs(rg["thresholds"][sp] - rc["thresholds"][sp]) <= 1e-4 * rc["thresholds"][sp]
        det = prad.RadialDetector({k: np.array(v, np.float32) for k, v in rc["centroids"].items()}, rc["thresholds"], 8)
        allg, allc = np.concatenate(list(Zg.values())), np.concatenate(list(Zc.values()))
        pred, best = det.decide(enc.session(cfg), allg)
        margins = []
        for r in range(allc.shape[0]):
            d, sp, bd = env.orad.decide(allc[r], {k: np.array(v, np.float32) for k, v in rc["centroids"].items()}, rc["thresholds"])
            dists = [env.orad.l2(allc[r] - np.array(rc["centroids"][s], np.float32)) for s in det.names]
            margins.append(min(abs(x - rc["thresholds"][s]) / rc["thresholds"][s] for x, s in zip(dists, det.names)))
            if margins[-1] > 1e-4:                                     # decisions identical away from the threshold
                assert (det.names[pred[r]] if pred[r] >= 0 else None) == sp, (r, margins[-1])
        assert len(margins) == 99
    d = np.load(tmp_path / "cache_npz" / f"Z_train_chunks_norm_{synth.SPECIES[0]}.npz")
    assert np.array_equal(d["Z"], Zg[synth.SPECIES[0]]) and int(d["failed"]) == 0
    # streaming encode + decide keeps latents on the device and agrees with the two-step form
    pcm = env.synth(list(range(40)), cfg.n_samples, cfg.sr)
    detected, species, best, mu = prad.detect_chunks(enc, det, pcm, cfg, apply_rms=False)
    pred2, best2 = det.decide(enc.session(cfg), mu)
    assert np.array_equal(detected, pred2 >= 0) and np.array_equal(best, best2)


def test_qout_grid_pipeline_on_resident_latents(env, tmp_path, capsys):
    """BASELINE.json configs[4] workload (run_qout_grid.sh: q_in 0.95, q_out in {.10,.15,.20,.25}, max_per_class,
    seed 123): fit (08) + folder benchmark (10) per grid point from latents encoded once; reference-format
    outputs; decisions equal to the CPU rule on CPU latents wherever the margin to a threshold exceeds 1e-4."""
    from amphibian_vae_latent_detector_amd import map_detector_core as core
    from amphibian_vae_latent_detector_amd import pipeline as ppipe
    from amphibian_vae_latent_detector_amd import synth
    cfg = env.cfg
    enc, tenc, spec = _encoders(env, "tiny", max_batch=32)
    train, val = tmp_path / "train_chunks", tmp_path / "val_chunks"
    synth.write_synth_folder(train, 120, cfg.n_samples, cfg.sr)
    for sp in synth.SPECIES:
        (val / sp).mkdir(parents=True)
    for i in range(200, 240):
        synth.write_wav_pcm16(val / synth.SPECIES[i % 4] / f"chunk_{i:06d}.wav", synth.synth_chunk_pcm16(i, cfg.n_samples, cfg.sr), cfg.sr)
    (val / synth.SPECIES[0] / "chunk_999999.wav").write_bytes(b"RIFFjunk")
    cfgp = tmp_path / "config.json"
    core.save_json(cfgp, {"species": list(synth.SPECIES), "chunk_seconds": 1.0})
    out = ppipe.run_qout_grid(enc, train, val, cfgp, cfg, tmp_path / "qout_grid", max_per_class=25, seed=123)
    assert sorted(out) == [0.10, 0.15, 0.20, 0.25]
    log = capsys.readouterr().out
    assert len(re.findall(r"✅\s+[\w_]+:\s+rk_in=[0-9.]+\s+\|\s+rk_out=[0-9.]+\s+\|\s+rk=[0-9.]+", log)) == 16   # 9105's RK_LINE
    # CPU side: same sampled training files, oracle latents, reference fit + decision rule
    import random
    random.seed(123)
    Zc = {}
    for sp in synth.SPECIES:
        wavs = random.sample(sorted((train / sp).resolve().glob("*.wav")), 25)
        Zc[sp] = np.stack([env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(w, **cfg.mel_kwargs())) for w in wavs])
    ev_files = [w for sp in sorted(synth.SPECIES) for w in sorted((val / sp).glob("*.wav")) if w.stat().st_size > 100]
    Zev = np.stack([env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(w, **cfg.mel_kwargs())) for w in ev_files])
    for q_out, res in out.items():
        run = tmp_path / "qout_grid" / f"qout_{q_out:.2f}"
        assert {p.name for p in run.iterdir()} == {"results.csv", "summary.txt", "config_used.json", "config_snapshot.json"}
        used = core.load_json(run / "config_used.json")
        assert used["radial_detector"]["meta_fit"]["q_out"] == q_out and used["radial_detector"]["meta_fit"]["max_per_class"] == 25
        snap = core.load_json(run / "config_snapshot.json")
        cents, thr = {}, {}
        for sp in synth.SPECIES:
            others = np.concatenate([Zc[o] for o in synth.SPECIES if o != sp])
            mu, rk, rk_in, rk_out = env.orad.fit_species_with_fp_control(Zc[sp], others, 0.95, q_out)
            cents[sp], thr[sp] = mu, rk
            assert abs(res["radial_detector"]["thresholds"][sp] - rk) <= 1e-4 * rk
            assert abs(snap["rk_per_species"][sp] - rk) <= 1e-4 * rk + 1e-6
        rows = [r for r in res["rows"] if r["pred_species"] != "ERROR"]
        assert len(rows) == len(ev_files) == 40 and sum(r["pred_species"] == "ERROR" for r in res["rows"]) == 1
        for r, z in zip(rows, Zev):
            d, sp, bd = env.orad.decide(z, cents, thr)
            margin = min(abs(env.orad.l2(z - cents[s]) - thr[s]) / thr[s] for s in synth.SPECIES)
            if margin > 1e-3:
                assert r["pred_species"] == (sp or "NO_DETECT"), (q_out, r["file"], margin)
            assert abs(r["best_distance"] - bd) <= 1e-4 * max(1.0, bd)

def test_map_folder_pipeline_fit_and_benchmark(env, tmp_path, capsys):
    """08b -> 10b on folders: config.json:map_detector in the reference's layout; tau and
    scores within 1e-4 of the oracle fit / scoring on CPU latents; decisions equal away from ties and tau."""
    from amphibian_vae_latent_detector_amd import map_detector_core as core
    from amphibian_vae_latent_detector_amd import pipeline as ppipe
    from amphibian_vae_latent_detector_amd import synth
    from oracle import map_score as omap
    cfg = env.cfg
    enc, tenc, spec = _encoders(env, "tiny", max_batch=32)
    train, val = tmp_path / "train_chunks", tmp_path / "val_chunks"
    synth.write_synth_folder(train, 160, cfg.n_samples, cfg.sr)
    for sp in synth.SPECIES:
        (val / sp).mkdir(parents=True)
    for i in range(300, 340):
        synth.write_wav_pcm16(val / synth.SPECIES[i % 4] / f"chunk_{i:06d}.wav", synth.synth_chunk_pcm16(i, cfg.n_samples, cfg.sr), cfg.sr)
    (val / synth.SPECIES[1] / "chunk_999999.wav").write_bytes(b"RIFFjunk")
    cfgp = tmp_path / "config.json"
    core.save_json(cfgp, {"species": list(synth.SPECIES), "chunk_seconds": 1.0})
    md = ppipe.fit_map_from_folder(enc, train, cfgp, cfg, cov_type="lda", cov_structure="full", shrink=0.1, set_tau_q=0.05)
    assert "tau fijado desde train" in capsys.readouterr().out
    stored = core.load_json(cfgp)
    assert (tmp_path / "config.json.bak").exists() and stored["map_detector"]["model"] == "gaussian_map"
    assert stored["map_detector"]["meta_fit"]["chunk_seconds"] == 1.0 and stored["map_detector"]["meta_fit"]["n_fft"] == cfg.n_fft
    means, precs, logdets, tau = core.read_map_detector_params(stored)           # what 10b / 09n read back
    # CPU side: oracle latents, oracle fit
    Zc = {sp: np.stack([env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(w, **cfg.mel_kwargs()))
                        for w in sorted((train / sp).glob("*.wav"))]) for sp in synth.SPECIES}
    fit = omap.fit_map(Zc, cov_type="lda", cov_structure="full", priors="empirical", eps=1e-6, shrink=0.1, tau_q=0.05)
    assert abs(md["tau"] - fit["tau"]) <= 2e-3 * max(1.0, abs(fit["tau"]))       # latents differ by <= 1e-4, precision amplifies
    rows = ppipe.benchmark_folder_map(enc, val, cfgp, cfg, tmp_path / "out_map")
    text = (tmp_path / "out_map" / "summary.txt").read_text()
    assert text.startswith("=== Detection Benchmark Summary (MAP) ===") and "=== Per-class ===" in text
    header = (tmp_path / "out_map" / "results.csv").read_text().splitlines()[0]
    assert header == "file,true_species,pred_species,detected,correct,best_score,error"
    ok = [r for r in rows if r["pred_species"] != "ERROR"]
    assert len(ok) == 40 and len(rows) == 41
    species = sorted(means)
    priors = core.get_priors_from_map_meta(stored, species)
    for r in ok:
        z = env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(Path(r["file"]), **cfg.mel_kwargs()))
        sc = omap.scores(z, species, means, precs, logdets, priors)
        srt = np.sort(sc)
        assert abs(r["best_score"] - srt[-1]) <= 5e-3 * max(1.0, abs(srt[-1]))
        if min(srt[-1] - srt[-2], abs(srt[-1] - tau)) > 2e-2 * max(1.0, abs(srt[-1])):
            det, sp, _ = omap.decide(z, species, means, precs, logdets, priors, tau)
            assert r["pred_species"] == (sp or "NO_DETECT")

def test_cli_07_single_wav_jsonl(env, tmp_path, capsys):
    """BASELINE configs[0] / SURVEY 8(d) config 1: the 07-style CLI (--wav ... --jsonl) on chunk 0."""
    from amphibian_vae_latent_detector_amd import encode_wav_to_latent as cli
    from amphibian_vae_latent_detector_amd import synth
    from amphibian_vae_latent_detector_amd.encoder import save_checkpoint
    cfg = env.cfg
    spec = env.SPECS["tiny"](cfg.target_frames, cfg.n_mels)
    state = env.random_init_state(spec, seed=0)
    pt, ym, wav = tmp_path / "model.pt", tmp_path / "enc.yaml", tmp_path / "chunk_000000.wav"
    save_checkpoint(pt, ym, spec, state)
    synth.write_wav_pcm16(wav, synth.synth_chunk_pcm16(0, cfg.n_samples, cfg.sr), cfg.sr)
    argv = ["--wav", str(wav), "--encoder", str(pt), "--encoder-config", str(ym), "--sr", str(cfg.sr), "--duration", "1.0",
            "--n-mels", "64", "--fmin", "150", "--fmax", str(cfg.fmax), "--hop-length", "384", "--n-fft", "2048",
            "--target-frames", "64"]
    cli.main(argv + ["--jsonl"])
    out = capsys.readouterr().out
    rec = json.loads(out.strip().splitlines()[-1])
    assert rec["wav"] == str(wav) and rec["latent_dim"] == 8 and len(rec["vector"]) == 8
    tenc = env.oenc.build_torch_encoder(spec.to_dict(), state)
    want = env.oenc.encode_mel_to_latent(tenc, env.ofe.wav_to_mel(wav, **cfg.mel_kwargs()))
    _assert_latents(np.array(rec["vector"], np.float32)[None], want[None])
    assert "✅ target_frames usado: 64" in out
    cli.main(argv + ["--precision", "3"])
    out = capsys.readouterr().out
    assert "✅ Latent dim: 8" in out
    with pytest.raises(SystemExit):
        cli.main(["--wav", str(wav), "--encoder", str(tmp_path / "missing.pt")])

# ----------------------------------------------------------------------------- K8: ingest generality
def test_resampler_against_the_restatement_and_file_ingest(env, fe, tmp_path):
    """K8 against the CPU restatement of resampy's loop (same taps, float64 products, float32 running sum: equal to
    float32 rounding noise), then a 48 kHz stereo 24-bit file through the reference-shaped API: decode, channel mean,
    GPU resample to the preset rate, mel + encode, equal to the oracle pipeline on the oracle-resampled signal."""
    from amphibian_vae_latent_detector_amd import map_detector_core as core
    from amphibian_vae_latent_detector_amd.encoder import save_checkpoint
    from oracle import resample as ors
    import sys
    sys.path.insert(0, str(Path(__file__).resolve().parent))
    from test_host_logic import _write_wav
    rng = np.random.default_rng(12)
    for (sr0, sr1, n) in ((48000, 22050, 3000), (22050, 48000, 1500), (44100, 22050, 2001), (16000, 22050, 900)):
        y = (0.3 * rng.standard_normal((2, n))).astype(np.float32)
        z = fe.resample(y, sr0, sr1).cpu().numpy()
        for r in range(2):
            want = ors.resample(y[r], sr0, sr1)
            assert z.shape[1] == want.shape[0]
            assert np.abs(z[r] - want).max() <= 2e-6, (sr0, sr1, float(np.abs(z[r] - want).max()))
    with pytest.raises(RuntimeError):
        fe.resample(np.zeros((1, 1), np.float32), 48000, 8000)
    # file level: 1.2 s of 48 kHz stereo, 24-bit
    cfg = env.cfg
    sr0 = 48000
    t = np.arange(int(1.2 * sr0)) / sr0
    sig = 0.2 * np.sin(2 * np.pi * 1500 * t) * np.hanning(t.size) + 0.1 * np.sin(2 * np.pi * 4200 * t) + 0.01 * rng.standard_normal(t.size)
    st = np.stack([sig, 0.5 * sig + 0.02 * rng.standard_normal(t.size)], axis=1)
    i24 = np.round(st * 8388607).astype(np.int32)
    wav = tmp_path / "field_48k_stereo24.wav"
    _write_wav(wav, i24, sr0, 1, 24)
    mono = (i24.astype(np.float32) / np.float32(8388608.0)).mean(axis=1, dtype=np.float32)
    y_ref = ors.resample(mono, sr0, cfg.sr)
    kw = cfg.mel_kwargs()
    mel = core.wav_to_mel(wav, **kw)
    y_fix = np.zeros(cfg.n_samples, np.float32)
    y_fix[:min(cfg.n_samples, y_ref.shape[0])] = y_ref[:cfg.n_samples]
    want_mel = env.ofe.wave_to_mel(y_fix, **kw)
    assert np.abs(mel.numpy() - want_mel).max() <= 1e-4
    spec = env.SPECS["tiny"](cfg.target_frames, cfg.n_mels)
    state = env.random_init_state(spec, seed=0)
    pt, ym = tmp_path / "model.pt", tmp_path / "enc.yaml"
    save_checkpoint(pt, ym, spec, state)
    encoder = core.load_encoder(pt, ym, tmp_path, torch.device("cuda"))
    z = core.encode_wav_to_latent(encoder, wav, torch.device("cuda"), **kw)
    tenc = env.oenc.build_torch_encoder(spec.to_dict(), state)
    _assert_latents(z[None], env.oenc.encode_mel_to_latent(tenc, want_mel)[None])
    # folder ingest: the native reader declines the file (status 5), the general path encodes it
    from amphibian_vae_latent_detector_amd import folder as pfolder
    Z, status = pfolder.encode_paths(encoder, [wav, tmp_path / "missing.wav"], cfg)
    assert status.tolist()[0] == 0 and status.tolist()[1] != 0
    _assert_latents(Z[:1].cpu().numpy(), z[None])


def test_bench_ranks_started_by_the_launcher_one_rank_over_rccl(tmp_path):
    """`python bench.py --gpus N` starts its ranks itself (launcher.launch_ranks: child processes with RANK / LOCAL_RANK / WORLD_SIZE /
    MASTER_*; the parent never touches the GPU).  On the one-GPU box the same code path runs with ONE rank: bench.py under the launcher's
    environment takes the torch.distributed / RCCL path (the all-gather executes for a one-rank group too), and rank 0's stdout carries
    exactly one JSON line with the rccl object."""
    import io
    import sys
    from amphibian_vae_latent_detector_amd import launcher
    root = Path(__file__).resolve().parents[1]
    out, err = io.StringIO(), io.StringIO()
    rc = launcher.launch_ranks([sys.executable, str(root / "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--prewarm", "2",
                                "--chunks", "2048", "--cpu-seconds", "0", "--no-detect", "--no-extra-legs"], 1, out=out, err=err)
    assert rc == 0, err.getvalue()[-2000:]
    lines = [ln for ln in out.getvalue().splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 1 and rec["value"] > 0 and rec["steps"] == 3
    assert rec["rccl"]["world_size"] == 1 and rec["rccl"]["backend"] == "nccl" and rec["rccl"]["rank0_block_equals_own_latents"]
    assert rec["fp16_split_range_overflow"] is False
