"""Host-side logic of the product package; no GPU needed (the library loads and its
host-only entry points - WAV ingest - run on the CPU; no compute kernels are called)."""
import ctypes
import json
import random
import re
from pathlib import Path

import numpy as np
import pytest
import torch

from amphibian_vae_latent_detector_amd import PRESETS, SPECS, EncoderSpec
from amphibian_vae_latent_detector_amd import _native as N
from amphibian_vae_latent_detector_amd import encoder as penc
from amphibian_vae_latent_detector_amd import folder as pfolder
from amphibian_vae_latent_detector_amd import map_detector_core as core
from amphibian_vae_latent_detector_amd import pipeline as ppipe
from amphibian_vae_latent_detector_amd import radial as prad
from amphibian_vae_latent_detector_amd import synth, wavio

ROOT = Path(__file__).resolve().parents[1]


# ------------------------------------------------------------------ C-ABI surface
def test_library_exports_every_declared_symbol():
    header = (ROOT / "include" / "amph.h").read_text()
    declared = set(re.findall(r"\b(amph_[a-z0-9_]+)\s*\(", header))
    assert declared == set(N.EXPORTED_SYMBOLS)
    lib = ctypes.CDLL(str(N.LIB_PATH))
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert N.lib.amph_abi_version() == 1


def test_no_cpu_device_and_no_fallback():
    with pytest.raises(RuntimeError, match="no CPU path"):
        penc._device_index("cpu")
    src = "".join((ROOT / "amphibian_vae_latent_detector_amd" / f).read_text()
                  for f in ("map_detector_core.py", "encoder.py", "folder.py", "radial.py", "wavio.py", "_native.py",
                            "map_detector.py", "resample.py", "pipeline.py", "encode_wav_to_latent.py", "config.py", "synth.py"))
    assert "oracle" not in src                       # the product never touches the checker


def test_create_rejects_bad_arguments_without_a_gpu():
    ctx = ctypes.c_void_p()
    st = N.lib.amph_create(None, None, None, 0, 0, 8, ctypes.byref(ctx))
    assert st == N.AMPH_ERR_INVALID_ARG and not ctx.value and "amph_create" in N.last_error(None)


# ------------------------------------------------------------------ presets / spec
def test_presets_geometry():
    assert (PRESETS["ref-default"].n_samples, PRESETS["ref-default"].raw_frames) == (240000, 626)
    assert (PRESETS["ref-07cli"].n_samples, PRESETS["ref-07cli"].raw_frames) == (144000, 376)
    b = PRESETS["bench-1s"]
    assert (b.n_samples, b.raw_frames, b.n_bins, b.target_frames) == (22050, 58, 1025, 64)
    assert PRESETS["bench-1s-128"].raw_frames == 87
    t = SPECS["thesis-shape"](64, 64)
    assert t.out_hw() == [(32, 32), (16, 16), (8, 8), (4, 4)] and t.flat == 4096 and t.latent_dim == 32
    assert SPECS["tiny"](64, 64).flat == 16 * 16 * 16
    assert EncoderSpec.from_dict(t.to_dict()) == t


def test_pack_weights_layout_and_checks():
    spec = SPECS["tiny"](64, 64)
    st = penc.random_init_state(spec, seed=3)
    blob = penc.pack_weights(spec, st)
    n = sum(int(np.prod(s)) for s in penc.state_shapes(spec).values())
    assert blob.dtype == np.float32 and blob.size == n
    assert np.array_equal(blob[:72], st["conv.0.weight"].ravel())
    assert np.array_equal(blob[-8:], st["fc_logvar.bias"])
    bad = dict(st)
    bad["fc_mu.weight"] = bad["fc_mu.weight"][:, :-1]
    with pytest.raises(RuntimeError, match="shape"):
        penc.pack_weights(spec, bad)
    del bad["fc_mu.weight"]
    with pytest.raises(RuntimeError, match="missing"):
        penc.pack_weights(spec, bad)
    st2 = penc.random_init_state(spec, seed=3)
    assert all(np.array_equal(st[k], st2[k]) for k in st)


def test_checkpoint_files_roundtrip(tmp_path):
    spec = SPECS["tiny"](64, 64)
    st = penc.random_init_state(spec, seed=1)
    pt, ym = tmp_path / "model.pt", tmp_path / "enc.yaml"
    penc.save_checkpoint(pt, ym, spec, st)
    assert core._load_spec_yaml(ym) == spec
    ck = torch.load(str(pt), map_location="cpu", weights_only=True)
    assert set(ck["state_dict"]) == set(penc.state_keys(spec))
    with pytest.raises(RuntimeError, match="no CPU path"):          # loads fine, then refuses the cpu device
        core.load_encoder(pt, ym, tmp_path, torch.device("cpu"))
    (tmp_path / "bad.yaml").write_text("encoder:\n  foo: 1\n")
    with pytest.raises(RuntimeError, match="_target_"):
        core._load_spec_yaml(tmp_path / "bad.yaml")
    with pytest.raises(FileNotFoundError):
        core.load_encoder(tmp_path / "nope.pt", ym, tmp_path, torch.device("cuda"))


def test_path_helpers(tmp_path):
    (tmp_path / "downloaded_models").mkdir()
    (tmp_path / "latent_space_exploration" / "x" / "y").mkdir(parents=True)
    assert core.find_project_root(tmp_path / "latent_space_exploration" / "x" / "y") == tmp_path.resolve()
    with pytest.raises(FileNotFoundError):
        core.resolve_default_config(tmp_path)
    core.save_json(tmp_path / "config.json", {"species": ["a"], "ñ": 1})
    assert core.load_json(core.resolve_default_config(tmp_path)) == {"species": ["a"], "ñ": 1}
    (tmp_path / "list.json").write_text("[1]")
    with pytest.raises(ValueError):
        core.load_json(tmp_path / "list.json")
    s = core.summarize_1d(np.arange(101, dtype=np.float64))
    assert s == {"min": 0.0, "p05": 5.0, "p50": 50.0, "p95": 95.0, "max": 100.0}
    assert all(np.isnan(v) for v in core.summarize_1d(np.array([])).values())


# ------------------------------------------------------------------ golden vectors through the product's host functions
def test_crop_or_pad_time_matches_reference(golden):
    g = golden("crop_or_pad.npz")
    for k in [k[3:] for k in g.files if k.startswith("in_")]:
        assert np.array_equal(core.crop_or_pad_time(g[f"in_{k}"], int(k.split("_")[1])), g[f"out_{k}"])


def test_output_adapter_matches_reference(golden):
    g = golden("adapter.npz")
    W = torch.tensor(g["W"])
    x = torch.tensor(g["mel"]).T.unsqueeze(0).unsqueeze(0)
    base = x.reshape(1, -1) @ W.T
    outs = {"tensor": base, "tuple": (base, base * 2), "list_nontensor_first": ["skip", base + 1],
            "dict_mu": {"logvar": base * 3, "mu": base}, "dict_z_over_mu": {"mu": base * 3, "z": base - 1},
            "dict_other": {"foo": base + 5}, "ndim3": torch.stack([base, base * 3], dim=1),
            "ndim4": base.reshape(1, 2, 2, 1)}
    for name, out in outs.items():
        assert np.array_equal(core._adapt_output(out), g[f"z_{name}"]), name
    for bad in (torch.zeros(2, 4), "nope", (), {"a": 1}):
        with pytest.raises(RuntimeError):
            core._adapt_output(bad)


def test_host_radial_fit_matches_reference(golden, capsys):
    g = golden("radial_fit.npz")
    species = json.loads(str(g["species_json"]))
    Zs = {sp: g[f"Z_{i}"] for i, sp in enumerate(species)}
    for q_out in (0.10, 0.25):
        rd = prad.fit_radial_detector(Zs, 0.95, q_out)              # default: the reference's pinned numpy 1.26 quantile arithmetic
        for i, sp in enumerate(species):
            assert rd["centroids"][sp] == g[f"mu_{i}_{q_out:.2f}"].tolist()
            rk, rk_in, rk_out = g[f"rk1_{i}_{q_out:.2f}"].tolist()
            assert rd["thresholds"][sp] == rk
            m = rd["per_species"][sp]
            assert (m["rk_in"], m["rk_out"], m["rk_final"]) == (rk_in, rk_out, rk)
    out = capsys.readouterr().out
    pat = re.compile(r"✅ (\S+): rk_in=([0-9.]+) \| rk_out=([0-9.]+|nan) \| rk=([0-9.]+)")   # 9105's regex shape
    assert len(pat.findall(out)) == 8


def test_priority_rank_rule(golden):
    g = golden("radial_decide.npz")
    cfg = json.loads(str(g["cfg_json"]))
    det = prad.RadialDetector.from_config(cfg, latent_dim=8)
    assert det.names == list(cfg["radial_detector"]["centroids"])           # config order kept
    # emulate the kernel's selection rule on the host with numpy distances
    for z, want in zip(g["Z"], g["pred"]):
        d = np.array([np.sqrt(np.sum((z - c) * (z - c))) for c in det.centroids], dtype=np.float32)
        acc = [k for k in range(len(det.names)) if float(d[k]) <= det.thresholds[k]]
        got = det.names[min(acc, key=lambda k: det.rank[k])] if acc else ""
        assert got == str(want)
    # species without threshold / with another dimension are skipped as in the reference loop
    det2 = prad.RadialDetector({"a": np.zeros(8, np.float32), "b": np.zeros(4, np.float32), "c": np.zeros(8, np.float32)},
                               {"a": 1.0, "b": 1.0}, latent_dim=8)
    assert det2.names == ["a"]
    with pytest.raises(ValueError):
        prad.get_detector_from_config({})



# ------------------------------------------------------------------ Gaussian-MAP detector: host fit + config readers
def test_host_map_fit_matches_reference(golden, capsys):
    from amphibian_vae_latent_detector_amd import map_detector as pmap
    g = golden("map_fit.npz")
    species = json.loads(str(g["species_json"]))
    Z = {sp: g[f"Z_{i}"] for i, sp in enumerate(species)}
    same = str(g["numpy_version"]) == np.__version__
    for vi in range(4):
        prm = json.loads(str(g[f"v{vi}_params_json"]))
        md = pmap.fit_map_detector(Z, cov_type=prm["cov_type"], cov_structure=prm["cov_structure"], priors=prm["priors"],
                                   eps=prm["eps"], shrink=prm["shrink"], set_tau_q=prm["tau_q"], meta={"chunk_seconds": 1.0})
        assert md["model"] == "gaussian_map" and sorted(md["means"]) == sorted(species)
        assert set(md) == {"model", "cov_type", "cov_structure", "priors", "means", "cov", "precision", "logdet_cov", "tau",
                           "meta_fit"}                                               # the keys 08b writes (:322-351)
        for i, sp in enumerate(species):
            assert md["means"][sp] == g[f"v{vi}_mean_{i}"].astype(float).tolist()
            assert md["meta_fit"]["per_species"][sp]["prior"] == float(g[f"v{vi}_prior_{i}"])
            if same:
                assert md["precision"][sp] == g[f"v{vi}_prec_{i}"].astype(float).tolist()
                assert md["logdet_cov"][sp] == float(g[f"v{vi}_logdet_{i}"])
            else:
                np.testing.assert_allclose(md["logdet_cov"][sp], float(g[f"v{vi}_logdet_{i}"]), rtol=1e-5)
        if prm["tau_q"] is None:
            assert md["tau"] is None
        else:
            np.testing.assert_allclose(md["tau"], float(g[f"v{vi}_tau"]), rtol=0 if same else 1e-4)
    assert "tau fijado desde train" in capsys.readouterr().out
    with pytest.raises(ValueError):
        pmap.fit_map_detector(Z, shrink=1.5)


def test_map_config_readers_match_reference(golden):
    from amphibian_vae_latent_detector_amd import map_detector as pmap
    from amphibian_vae_latent_detector_amd import map_detector_core as core
    g = golden("map_decide.npz")
    cfg = json.loads(str(g["cfg_json"]))
    means, precs, logdets, tau = core.read_map_detector_params(cfg)
    species = sorted(set(means) & set(precs) & set(logdets))
    assert species == json.loads(str(g["species_json"])) and tau == -14.0
    assert core.get_priors_from_map_meta(cfg, species) == json.loads(str(g["priors_json"]))
    assert core.get_chunk_seconds_for_map(cfg) == float(g["duration"]) == 1.0     # meta_fit wins over chunk_seconds
    det = pmap.MapDetector.from_config(cfg, latent_dim=8)
    assert det.names == species and det.precisions.shape == (5, 8, 8) and det.tau == -14.0
    # host scores with the mirrored core function reproduce the reference's
    z = g["Z"][3]
    sc = [core.gaussian_logpdf_from_precision(z, means[sp], precs[sp], logdets[sp]) + det.log_priors[k]
          for k, sp in enumerate(species)]
    np.testing.assert_allclose(sc, g["scores"][3], rtol=1e-6, atol=1e-6)
    for bad in ({}, {"map_detector": {"model": "other"}}, {"map_detector": {"model": "gaussian_map", "means": []}}):
        with pytest.raises(ValueError):
            core.read_map_detector_params(bad)
    det2 = pmap.MapDetector({"a": np.zeros(8), "b": np.zeros(4)}, {"a": np.eye(8), "b": np.eye(4)}, {"a": 0.0, "b": 0.0},
                            {}, None, latent_dim=8)
    assert det2.names == ["a"]                                     # another dimension: skipped (10b:153-154)


def test_cli_flags_are_the_reference_07_flags():
    """07_encode_wav_to_latent.py:416-441: same flag names and defaults (device excepted: no CPU path)."""
    from amphibian_vae_latent_detector_amd.encode_wav_to_latent import parse_args
    a = parse_args(["--wav", "x.wav"])
    assert (a.sr, a.duration, a.n_mels, a.fmin, a.fmax, a.hop_length, a.n_fft, a.target_frames) == \
        (48000, 3.0, 64, 150.0, 15000.0, 384, 2048, 192)
    assert (a.jsonl, a.precision, a.auto_frames, a.auto_max_frames, a.auto_step, a.encoder, a.encoder_config) == \
        (False, 6, False, 512, 8, None, None)
    assert a.device == "cuda"
    with pytest.raises(SystemExit):
        parse_args([])                                             # --wav is required


def test_07_output_adapter_variant_and_whole_file_configuration():
    """07's own extract_vector (07_encode_wav_to_latent.py:264-297): the extra dict key "enc", ValueError, the empty time axis
    guard, [B, D] tensor out; and the duration <= 0 configuration (map_detector_core.py:212: no pad / truncate step - the
    whole file): n_samples equals the file length for every length."""
    import torch
    from amphibian_vae_latent_detector_amd.encode_wav_to_latent import extract_vector
    from amphibian_vae_latent_detector_amd import map_detector_core as core
    t = torch.arange(12, dtype=torch.float32).reshape(1, 3, 4)
    assert torch.equal(extract_vector(t), t.mean(dim=1))                      # [B, T, C] -> mean over T
    assert extract_vector(torch.zeros(2, 0, 5)).shape == (2, 0, 5)            # empty T axis: returned as it is, t[:, :0] (core's adapter would average it to NaN)
    assert torch.equal(extract_vector({"enc": t[:, 0]}), t[:, 0])              # 07's extra key
    assert torch.equal(extract_vector({"other": 1, "enc": t[:, 1], "mu": t[:, 2]}), t[:, 2])   # ... after the core's keys, in order
    assert torch.equal(extract_vector({"x": "no", "y": t[:, 1]}), t[:, 1])     # else the first tensor value
    assert torch.equal(extract_vector((None, t[:, 0], t[:, 1])), t[:, 0])
    assert extract_vector(torch.zeros(2, 3, 4, 5)).shape == (2, 60)           # more than three axes: flattened, no mean
    for bad in ([], {"a": 1}, 3.0):
        with pytest.raises(ValueError):
            extract_vector(bad)
    kw = dict(n_mels=64, fmin=150.0, fmax=11025.0, hop_length=384, n_fft=2048, target_frames=64)
    for sr in (22050, 48000, 44100):
        for n in (1025, 22050, 22051, 65537, 240000, 999983):
            assert core._config(sr, 0.0, whole_file_samples=n, **kw).n_samples == n
            assert core._config(sr, -1.0, whole_file_samples=n, **kw).n_samples == n
    with pytest.raises(RuntimeError):
        core._config(22050, 0.0, **kw)                                        # whole file without its length
    assert core._config(22050, 1.0, **kw).n_samples == 22050


def _write_wav(path, data, rate, tag, bits, extensible=False):
    """Minimal RIFF writer for the loader tests (data: [n, ch] array already in the sample dtype)."""
    import struct
    ch = data.shape[1]
    if bits == 24:
        v = data.astype(np.int32) & 0xFFFFFF
        body = np.stack([v & 0xFF, (v >> 8) & 0xFF, (v >> 16) & 0xFF], axis=-1).astype(np.uint8).tobytes()
    else:
        body = data.tobytes()
    block = ch * bits // 8
    if extensible:
        guid = struct.pack("<H", tag) + bytes.fromhex("000000001000800000aa00389b71")
        fmt = struct.pack("<HHIIHHHHI", 0xFFFE, ch, rate, rate * block, block, bits, 22, bits, 0) + guid
    else:
        fmt = struct.pack("<HHIIHH", tag, ch, rate, rate * block, block, bits)
    chunks = b"fmt " + struct.pack("<I", len(fmt)) + fmt + b"LIST" + struct.pack("<I", 4) + b"abcd" + \
        b"data" + struct.pack("<I", len(body)) + body
    path.write_bytes(b"RIFF" + struct.pack("<I", 4 + len(chunks)) + b"WAVE" + chunks)


def test_general_wav_loader_subtypes_and_channel_mean(tmp_path):
    """librosa.load's decode for everything the native PCM16-mono reader declines: libsndfile's float scaling per
    subtype, channel mean, native sample rate reported (resampling itself needs the GPU)."""
    rng = np.random.default_rng(3)
    x = rng.uniform(-0.9, 0.9, size=(500, 2))
    i16 = np.round(x * 32767).astype("<i2")
    _write_wav(tmp_path / "s16.wav", i16, 22050, 1, 16)
    y, rate = wavio.load_audio(tmp_path / "s16.wav")
    assert rate == 22050 and y.dtype == np.float32 and y.shape == (500,)
    want = (i16.astype(np.float32) / np.float32(32768.0)).mean(axis=1, dtype=np.float32)
    assert np.array_equal(y, want)
    i24 = np.round(x * 8388607).astype(np.int32)
    _write_wav(tmp_path / "s24.wav", i24, 48000, 1, 24)
    y, rate = wavio.load_audio(tmp_path / "s24.wav")
    assert rate == 48000 and np.allclose(y, (i24 / 8388608.0).mean(axis=1), atol=1e-7)
    i32 = np.round(x[:, :1] * 2147483647).astype("<i4")
    _write_wav(tmp_path / "s32.wav", i32, 44100, 1, 32, extensible=True)
    y, rate = wavio.load_audio(tmp_path / "s32.wav")
    assert rate == 44100 and np.allclose(y, i32[:, 0] / 2147483648.0, atol=1e-7)
    u8 = np.round(x[:, :1] * 127 + 128).astype(np.uint8)
    _write_wav(tmp_path / "u8.wav", u8, 8000, 1, 8)
    y, _ = wavio.load_audio(tmp_path / "u8.wav")
    assert np.array_equal(y, (u8[:, 0].astype(np.float32) - 128) / 128)
    f32 = x.astype("<f4")
    _write_wav(tmp_path / "f32.wav", f32, 22050, 3, 32)
    y, _ = wavio.load_audio(tmp_path / "f32.wav")
    assert np.array_equal(y, f32.mean(axis=1, dtype=np.float32))
    _write_wav(tmp_path / "f64.wav", x.astype("<f8"), 22050, 3, 64)
    y, _ = wavio.load_audio(tmp_path / "f64.wav")
    assert np.allclose(y, x.astype(np.float32).mean(axis=1), atol=1e-7)
    # same rate: no session needed; another rate without a session: a Python exception, not an abort
    assert np.array_equal(wavio.load_float_mono(tmp_path / "s16.wav", 22050), want)
    with pytest.raises(RuntimeError, match="sample rate"):
        wavio.load_float_mono(tmp_path / "s24.wav", 22050)
    (tmp_path / "bad.wav").write_bytes(b"RIFFxxxxWAVEjunk")
    with pytest.raises(RuntimeError):
        wavio.load_audio(tmp_path / "bad.wav")
    _write_wav(tmp_path / "adpcm.wav", i16, 22050, 2, 16)
    with pytest.raises(RuntimeError, match="unsupported"):
        wavio.load_audio(tmp_path / "adpcm.wav")


def test_kaiser_best_table_is_the_published_construction():
    from amphibian_vae_latent_detector_amd.resample import KAISER_BEST, kaiser_best_window
    win, num_table = kaiser_best_window()
    assert (KAISER_BEST["num_zeros"], KAISER_BEST["precision"]) == (64, 9) and num_table == 512
    assert win.dtype == np.float64 and win.shape == (32769,) and win[0] == KAISER_BEST["rolloff"]
    assert N.lib.amph_resample_length(22050, 48000, 22050) == int(np.ceil(22050 * 22050 / 48000))
    assert N.lib.amph_resample_length(1000, 22050, 48000) == int(np.ceil(1000 * 48000 / 22050))

# ------------------------------------------------------------------ synthetic data + WAV ingest
def test_synth_is_deterministic_and_shaped():
    a = synth.synth_chunk_pcm16(7, 22050, 22050)
    assert a.dtype == np.int16 and a.shape == (22050,) and np.array_equal(a, synth.synth_chunk_pcm16(7, 22050, 22050))
    assert not synth.synth_chunk_pcm16(49, 22050, 22050).any() and not synth.synth_chunk_pcm16(99, 22050, 22050).any()
    assert not np.array_equal(a, synth.synth_chunk_pcm16(8, 22050, 22050))


def test_native_wav_reader(tmp_path):
    q = synth.synth_chunk_pcm16(3, 22050, 22050)
    p = tmp_path / "a.wav"
    synth.write_wav_pcm16(p, q, 22050)
    out, n = wavio.read_pcm16_mono(p, 22050, 22050)
    assert n == 22050 and np.array_equal(out, q)
    out, n = wavio.read_pcm16_mono(p, 22050, 30000)                        # longer buffer: zero filled
    assert n == 22050 and np.array_equal(out[:22050], q) and not out[22050:].any()
    out, n = wavio.read_pcm16_mono(p, 22050, 1000)                         # truncation
    assert n == 1000 and np.array_equal(out, q[:1000])
    with pytest.raises(NotImplementedError):
        wavio.read_pcm16_mono(p, 48000, 22050)                             # wrong rate: no resampler
    st = np.stack([q, -q], axis=1)
    synth.write_wav_pcm16(tmp_path / "st.wav", st, 22050)
    with pytest.raises(NotImplementedError):
        wavio.read_pcm16_mono(tmp_path / "st.wav", 22050, 22050)
    y = wavio.load_float_mono(tmp_path / "st.wav", 22050)                  # stereo -> channel mean, float32
    assert y.dtype == np.float32 and np.abs(y).max() <= 1.0 / 32768 + 1e-9
    (tmp_path / "junk.wav").write_bytes(b"not a wav at all")
    with pytest.raises(OSError):
        wavio.read_pcm16_mono(tmp_path / "junk.wav", 22050, 100)
    with pytest.raises(OSError):
        wavio.read_pcm16_mono(tmp_path / "missing.wav", 22050, 100)


def test_native_reader_skips_extra_chunks(tmp_path):
    q = np.arange(-50, 50, dtype="<i2")
    body = (b"fmt " + (16).to_bytes(4, "little") + (1).to_bytes(2, "little") + (1).to_bytes(2, "little") +
            (22050).to_bytes(4, "little") + (44100).to_bytes(4, "little") + (2).to_bytes(2, "little") + (16).to_bytes(2, "little") +
            b"LIST" + (5).to_bytes(4, "little") + b"abcde\x00" +            # odd-sized chunk + pad byte
            b"data" + (200).to_bytes(4, "little") + q.tobytes())
    (tmp_path / "x.wav").write_bytes(b"RIFF" + (4 + len(body)).to_bytes(4, "little") + b"WAVE" + body)
    out, n = wavio.read_pcm16_mono(tmp_path / "x.wav", 22050, 100)
    assert n == 100 and np.array_equal(out, q)


def test_read_many_flags_bad_files(tmp_path):
    paths = []
    for i in range(5):
        p = tmp_path / f"c{i}.wav"
        synth.write_wav_pcm16(p, synth.synth_chunk_pcm16(i, 2000 + 100 * i, 22050), 22050)
        paths.append(p)
    (tmp_path / "bad.wav").write_bytes(b"RIFFxxxxWAVEjunk")
    paths.insert(2, tmp_path / "bad.wav")
    pcm, lengths, status, ptr = wavio.read_many_pcm16(paths, 22050, 2200, n_threads=3)
    assert ptr is None and pcm.shape == (6, 2200)
    assert status.tolist() == [0, 0, N.AMPH_ERR_IO, 0, 0, 0]
    assert lengths.tolist() == [2000, 2100, 0, 2200, 2200, 2200] and not pcm[2].any()
    assert np.array_equal(pcm[0, :2000], synth.synth_chunk_pcm16(0, 2000, 22050)) and not pcm[0, 2000:].any()


def test_read_many_across_directories_trailing_chunks_relative_names_and_sizes(tmp_path, monkeypatch):
    """Round 4's reader: openat on per-thread directory descriptors + one preadv per canonical file.  Files of SIX directories in one
    call (the descriptor cache holds four), a canonical file with a LIST chunk BEHIND its samples (those bytes are not samples: the data
    chunk's size rules), a data chunk longer than the file (truncated: keep what is there), a bare file name relative to the working
    directory, a missing directory; amph_wav_sizes over the same list; a PathArray window in the middle of the list."""
    import struct
    paths, want = [], []
    for d in range(6):
        sub = tmp_path / f"sp{d}"
        sub.mkdir()
        for i in range(3):
            q = synth.synth_chunk_pcm16(10 * d + i, 500 + 10 * i, 22050)
            synth.write_wav_pcm16(sub / f"c{i}.wav", q, 22050)
            paths.append(sub / f"c{i}.wav")
            want.append(q)
    q = synth.synth_chunk_pcm16(77, 300, 22050)
    head = b"WAVEfmt " + struct.pack("<IHHIIHH", 16, 1, 1, 22050, 44100, 2, 16)
    body = b"data" + struct.pack("<I", 600) + q.tobytes() + b"LIST" + struct.pack("<I", 8) + b"\x7f" * 8
    (tmp_path / "sp0" / "tail.wav").write_bytes(b"RIFF" + struct.pack("<I", len(head) + len(body)) + head + body)
    paths.append(tmp_path / "sp0" / "tail.wav"); want.append(q)
    body = b"data" + struct.pack("<I", 2000) + q[:100].tobytes()                          # claims 1000 samples, holds 100
    (tmp_path / "sp1" / "short.wav").write_bytes(b"RIFF" + struct.pack("<I", len(head) + len(body)) + head + body)
    paths.append(tmp_path / "sp1" / "short.wav"); want.append(q[:100])
    synth.write_wav_pcm16(tmp_path / "bare.wav", q, 22050)
    monkeypatch.chdir(tmp_path)
    paths.append("bare.wav"); want.append(q)
    paths.append(tmp_path / "no_such_dir" / "c.wav"); want.append(None)
    rows = np.full((len(paths), 640), 7, np.int16)
    lengths, status = wavio.read_many_into(rows, paths, 22050, n_threads=3)
    for i, w in enumerate(want):
        if w is None:
            assert status[i] == N.AMPH_ERR_IO and lengths[i] == 0 and not rows[i].any()
            continue
        assert status[i] == 0 and lengths[i] == len(w), (i, status[i], lengths[i])
        assert np.array_equal(rows[i, :len(w)], w) and not rows[i, len(w):].any(), i
    sizes = wavio.file_sizes(paths, n_threads=2)
    assert sizes[-1] == -1 and sizes[0] == 44 + 2 * 500 and sizes[-2] == 44 + 600
    pa = wavio.PathArray(paths)
    rows2 = np.zeros((4, 640), np.int16)
    l2, s2 = wavio.read_many_into(rows2, pa, 22050, n_threads=2, offset=5)
    assert np.array_equal(rows2, rows[5:9]) and l2.tolist() == lengths[5:9].tolist() and not s2.any()
    with pytest.raises(IndexError):
        pa.window(len(paths) - 1, 2)


# ------------------------------------------------------------------ folder driver (host logic, fake encode)
def _fake_encode(paths):
    """Deterministic stand-in for the HIP encode: a vector derived from the file name."""
    Z = torch.tensor([[float(int(re.findall(r"(\d+)", p.stem)[-1]) * 10 + j) for j in range(4)] for p in paths],
                     dtype=torch.float32).reshape(len(paths), 4)
    st = np.array([N.AMPH_ERR_IO if "bad" in p.stem else 0 for p in paths], dtype=np.int32)
    return Z, st


def _make_tree(root, counts):
    for sp, n in counts.items():
        (root / sp).mkdir(parents=True)
        for i in range(n):
            (root / sp / f"chunk_{i:04d}.wav").write_bytes(b"")
    return root


def test_plan_shards():
    assert pfolder.plan_shards(10, 4) == ([(0, 3), (3, 6), (6, 9), (9, 10)], 3)
    assert pfolder.plan_shards(3, 8)[0][3:] == [(3, 3)] * 5
    assert pfolder.plan_shards(0, 2) == ([(0, 0), (0, 0)], 1)
    assert pfolder.plan_shards(100000, 8)[1] == 12500


def test_list_species_wavs_sampling_is_the_reference_rule(tmp_path):
    root = _make_tree(tmp_path / "train_chunks_norm", {"b_sp": 30, "a_sp": 5, "empty_sp": 0})
    random.seed(123)
    got = pfolder.list_species_wavs(root, ["b_sp", "missing", "a_sp", "empty_sp"], max_per_class=10)
    random.seed(123)
    want_b = random.sample(sorted((root / "b_sp").resolve().glob("*.wav")), 10)
    assert list(got) == ["b_sp", "a_sp"] and got["b_sp"] == want_b and len(got["a_sp"]) == 5
    random.seed(123)
    got2 = pfolder.list_species_wavs(root, ["a_sp", "b_sp"], max_per_class=10, skip=["a_sp"])
    assert list(got2) == ["b_sp"] and got2["b_sp"] == want_b               # skipped species draw nothing


def test_encode_folder_cache_and_failures(tmp_path, capsys):
    cfg = PRESETS["bench-1s"]
    root = _make_tree(tmp_path / "val_chunks_norm", {"s1": 6, "s2": 3, "s3": 2})
    (root / "s2" / "bad_0009.wav").write_bytes(b"")
    for p in (root / "s3").glob("*.wav"):
        p.rename(p.with_name("bad_" + p.name))
    cache = tmp_path / "cache_npz"
    Z, failed = pfolder.encode_folder(None, root, ["s1", "s2", "s3", "s4"], cfg, cache_dir=cache, encode_fn=_fake_encode)
    assert list(Z) == ["s1", "s2"] and Z["s1"].shape == (6, 4) and Z["s2"].shape == (3, 4) and failed == {"s1": 0, "s2": 1}
    assert Z["s1"][:, 0].tolist() == [0.0, 10.0, 20.0, 30.0, 40.0, 50.0]     # name-sorted order
    out = capsys.readouterr().out
    assert "❌ s3" in out and "fallo bad_0009.wav" in out and "s4" in out
    d = np.load(cache / "Z_val_chunks_norm_s2.npz")
    assert set(d.files) == {"Z", "failed", "root"} and int(d["failed"]) == 1 and str(d["root"]) == str(root.resolve())
    assert np.array_equal(d["Z"], Z["s2"])
    # second call is served from the cache (encode_fn would now fail loudly)
    Z2, failed2 = pfolder.encode_folder(None, root, ["s1", "s2"], cfg, cache_dir=cache,
                                        encode_fn=lambda p: (_ for _ in ()).throw(AssertionError("cache miss")))
    assert np.array_equal(Z2["s1"], Z["s1"]) and failed2 == failed


# ------------------------------------------------------------------ reference-format outputs (08 / 10 / 9100 / 9105)
def test_write_radial_config_keys_and_backup(tmp_path, golden):
    g = golden("radial_fit.npz")
    species = json.loads(str(g["species_json"]))
    Zs = {sp: g[f"Z_{i}"] for i, sp in enumerate(species)}
    cfgp = tmp_path / "config.json"
    core.save_json(cfgp, {"species": species, "chunk_seconds": 1.0, "other": {"keep": True}})
    rd = prad.fit_radial_detector(Zs, 0.95, 0.15, failed={species[0]: 2}, verbose=False)
    fe = PRESETS["bench-1s"]
    bak = ppipe.write_radial_config(cfgp, rd, chunks_dir=tmp_path / "train_chunks", q_in=0.95, q_out=0.15,
                                    chunk_seconds=1.0, fe=fe, max_per_class=400, seed=123)
    assert bak.name == "config.json.bak" and "radial_detector" not in core.load_json(bak)   # backup = previous file
    out = core.load_json(cfgp)
    assert out["other"] == {"keep": True} and set(out["radial_detector"]) == {"centroids", "thresholds", "meta_fit"}
    mf = out["radial_detector"]["meta_fit"]                              # 08_fit_radial_detector.py:568-583
    assert list(mf) == ["chunks_dir", "chunks_name", "q_in", "q_out", "chunk_seconds", "sr", "n_mels", "target_frames",
                        "fmin", "fmax", "hop_length", "n_fft", "max_per_class", "seed", "per_species"]
    assert mf["chunks_name"] == "train_chunks" and mf["max_per_class"] == 400 and mf["n_fft"] == 2048
    ps = mf["per_species"][species[0]]
    assert set(ps) == {"N_in", "N_out", "rk_in", "rk_out", "rk_final", "failed", "used", "rho_in_summary", "rho_out_summary"}
    assert ps["failed"] == 2 and set(ps["rho_in_summary"]) == {"min", "p50", "p90", "max"}
    # the written detector is what 09's get_detector_from_config reads back
    cents, thr, dur = prad.get_detector_from_config(out)
    assert list(cents) == species and thr == out["radial_detector"]["thresholds"] and dur == 1.0
    assert cents[species[1]].dtype == np.float32


def test_results_csv_and_summary_are_parseable_by_the_reference_scripts(tmp_path):
    files = [("a_sp", Path("/x/a_sp/1.wav")), ("a_sp", Path("/x/a_sp/2.wav")), ("a_sp", Path("/x/a_sp/3.wav")),
             ("b_sp", Path("/x/b_sp/1.wav")), ("b_sp", Path("/x/b_sp/2.wav"))]
    status = np.array([0, 0, 4, 0, 0])
    rows = ppipe.rows_from_decisions(files, status, ["a_sp", None, None, "b_sp", "a_sp"], np.array([1.5, 9.0, 0.0, 0.25, 2.0], np.float32))
    assert [r["pred_species"] for r in rows] == ["a_sp", "NO_DETECT", "ERROR", "b_sp", "a_sp"]
    assert [r["correct"] for r in rows] == [True, False, False, True, False] and "error" in rows[2]
    csv_path, summary_path = ppipe.write_results(rows, tmp_path / "detection_benchmark")
    import pandas as pd
    df = pd.read_csv(csv_path)
    assert list(df.columns) == ["file", "true_species", "pred_species", "detected", "correct", "best_distance", "error"]
    assert df["detected"].tolist() == [True, False, False, True, True] and np.isnan(df["best_distance"][2])
    assert df["best_distance"][3] == 0.25
    txt = summary_path.read_text(encoding="utf-8")
    # the three patterns 9100_spearman_rk_analysis.py:53-58 applies to summary.txt
    g = re.search(r"Correct:\s*\d+\s*\|\s*Accuracy:\s*([0-9.]+)%", txt)
    n = re.search(r"NO_DETECT:\s*\d+\s*\|\s*Rate:\s*([0-9.]+)%", txt)
    cls = re.compile(r"^\-\s*(?P<sp>.+?)\s+n=\s*(?P<n>\d+)\s+acc=\s*(?P<acc>[0-9.]+)%\s+no_detect=\s*(?P<nd>[0-9.]+)%", re.I | re.M)
    assert float(g.group(1)) == 50.0 and float(n.group(1)) == 25.0          # ERROR rows are left out of the summary
    per = {m.group("sp"): (int(m.group("n")), float(m.group("acc")), float(m.group("nd"))) for m in cls.finditer(txt)}
    assert per == {"a_sp": (2, 50.0, 50.0), "b_sp": (2, 50.0, 0.0)}
    assert txt.splitlines()[0] == "=== Detection Benchmark Summary ===" and "Total files: 4" in txt


def test_list_audio_files_is_recursive_and_sorted(tmp_path):
    for rel in ("sp/b.wav", "sp/sub/a.WAV", "sp/c.txt", "sp/a.wav"):
        (tmp_path / rel).parent.mkdir(parents=True, exist_ok=True)
        (tmp_path / rel).write_bytes(b"")
    assert [p.name for p in ppipe.list_audio_files(tmp_path / "sp")] == ["a.wav", "b.wav", "a.WAV"]

# ------------------------------------------------------------------ properties (hypothesis)
def test_properties_shards_loader_and_crop():
    from hypothesis import given, settings
    from hypothesis import strategies as st
    from amphibian_vae_latent_detector_amd import map_detector_core as core

    @settings(max_examples=200, deadline=None)
    @given(st.integers(0, 100000), st.integers(1, 16))
    def shards(n, world):
        plan, per = pfolder.plan_shards(n, world)
        assert len(plan) == world and per == max(1, (n + world - 1) // world)
        covered = [i for a, b in plan for i in range(a, b)] if n <= 2000 else None
        assert all(0 <= a <= b <= n and b - a <= per for a, b in plan)
        assert sum(b - a for a, b in plan) == n and (covered is None or covered == list(range(n)))
    shards()

    @settings(max_examples=100, deadline=None)
    @given(st.integers(1, 300), st.integers(1, 300), st.integers(1, 5))
    def crop(T, target, M):
        mel = np.arange(M * T, dtype=np.float32).reshape(M, T)
        out = core.crop_or_pad_time(mel, target)
        assert out.shape == (M, target) and out.dtype == np.float32
        if T >= target:
            s0 = (T - target) // 2
            assert np.array_equal(out, mel[:, s0:s0 + target])
        else:
            left = (target - T) // 2
            assert np.array_equal(out[:, left:left + T], mel) and out[:, :left].sum() == 0 and out[:, left + T:].sum() == 0
    crop()

    @settings(max_examples=60, deadline=None)
    @given(st.integers(1, 400), st.integers(1, 3), st.sampled_from([8000, 22050, 48000]), st.integers(0, 2 ** 31 - 1))
    def pcm16_roundtrip(n, ch, rate, seed):
        import tempfile
        q = np.random.default_rng(seed).integers(-32768, 32768, size=(n, ch)).astype("<i2")
        with tempfile.TemporaryDirectory() as td:
            pth = Path(td) / "x.wav"
            _write_wav(pth, q, rate, 1, 16)
            y, r = wavio.load_audio(pth)
        assert r == rate and y.shape == (n,) and y.dtype == np.float32
        want = (q.astype(np.float32) / np.float32(32768.0))
        assert np.array_equal(y, want.mean(axis=1, dtype=np.float32) if ch > 1 else want[:, 0])
    pcm16_roundtrip()


def test_synth_folder_plan_equals_the_reference_listing_of_a_written_folder(tmp_path):
    """bench.py --config3 / the config-3 GPU test order HBM-resident chunks with synth_folder_plan instead of writing
    100 000 files: it must be the list 08_fit_radial_detector.py:374,461-484 builds over write_synth_folder's layout
    (species order, name-sorted, ONE generator seeded once, random.sample per species)."""
    import random
    from amphibian_vae_latent_detector_amd import folder as pfolder
    from amphibian_vae_latent_detector_amd import synth
    synth.write_synth_folder(tmp_path / "r", 41, 64, 22050)
    for mpc in (0, 7):
        random.seed(123)
        listed = pfolder.list_species_wavs(tmp_path / "r", list(synth.SPECIES), mpc)
        plan = synth.synth_folder_plan(41, mpc, 123)
        assert list(plan) == list(synth.SPECIES)
        for sp in synth.SPECIES:
            assert [int(p.stem.split("_")[1]) for p in listed[sp]] == plan[sp], (mpc, sp)
    # the parallel generator gives the same chunks as the serial one
    a = synth.synth_batch_pcm16(np.arange(100, 612), 2000, 22050, workers=4)
    assert np.array_equal(a, synth.synth_batch_pcm16(np.arange(100, 612), 2000, 22050))


def test_pad_block_layout_of_the_virtual_shards():
    """encode_resident_shards lays its blocks out as all_gather_into_tensor does (rank-major, each padded to ceil(n / R) rows)."""
    import torch
    from amphibian_vae_latent_detector_amd import folder as pfolder

    class FakeSession:
        latent_dim = 3
        device = torch.device("cpu")

        def encode(self, pcm, lengths=None, apply_rms=True):
            return pcm[:, :3].float() * 2.0, None, None
    x = torch.arange(10 * 5).reshape(10, 5)
    for shards in (1, 3, 4, 8, 16):
        got = pfolder.encode_resident_shards(FakeSession(), x, shards)
        assert torch.equal(got, x[:, :3].float() * 2.0), shards


def test_lds_swizzle_model_of_the_staged_gemm_is_conflict_free():
    """tools/lds_swizzle_model.py: the slot swizzle h_sw of k_conv_gemm_s3 / _s3w (csrc/k_encoder_split.hip) costs 16 array cycles per
    fragment read and 8 per staging store under the gfx950 banking rules - no conflicts; the two-bit swizzle it replaced 32 / 16 / 8."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("lds_swizzle_model", Path(__file__).resolve().parents[1] / "tools" / "lds_swizzle_model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.slice_cycles(mod.new) == (16, 8, 8)
    assert mod.slice_cycles(mod.old) == (32, 16, 8)


def test_mel_read_bank_model_places_the_filters_conflict_free():
    """tools/mel_read_bank_model.py: k_stft_pair's power-spectrum reads of the mel phase under the gfx950 banking rules - 302 array cycles per
    pair of frames with the old lane -> bin map, 104 (conflict-free) with contiguous quads and the placement search that build_pair_mel
    (csrc/amph_api.hip) runs per round of 16 filters; the weight reads stay conflict-free."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("mel_read_bank_model", Path(__file__).resolve().parents[1] / "tools" / "mel_read_bank_model.py")
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    res = mod.main()
    assert res["placed"] == (52, 104, 13)
    assert [v[1] for k, v in res.items() if k != "placed"] == [302, 228]
