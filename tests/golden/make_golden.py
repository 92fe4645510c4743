#!/usr/bin/env python3
"""Capture golden vectors from the REFERENCE's own functions (build container only).

Run once, here, as `python tests/golden/make_golden.py`; the .npz/.json files it
writes next to itself are committed.  /root/reference does not exist on the GPU
box and nothing in tests/, smoke() or bench.py reads it: they read the fixtures.

The reference modules import librosa/soundfile at top level and neither is
installed (nor installable) in this image.  They are registered as EMPTY
placeholder modules (no attributes at all) so that the module bodies execute;
every function captured below is pure numpy/torch and never touches them.  The
functions that do need librosa (`wav_to_mel`) cannot run and are NOT captured:
the mel front end stays "parity unpinned" (oracle/__init__.py).

Captured (reference file:line -> fixture):
  rms_normalize            00_normalize_dataset_rms.py:29-38        rms_normalize.npz
  crop_or_pad_time         map_detector_core.py:185-195             crop_or_pad.npz
  encode_wav_to_latent's post-mel half (orientation, adapter, batch check)
                           map_detector_core.py:267-300             adapter.npz
  extract_vector (07 variant)  07_encode_wav_to_latent.py:264-297   adapter.npz
  fit_species_with_fp_control  08_fit_radial_detector.py:310-333    radial_fit.npz
  l2 / get_detector_from_config / DetectorSession.predict_one decision block
                           09_evaluate_wav_detection.py:113-149,354-355;
                           10_benchmark_folder_detection.py:152-199  radial_decide.npz
"""
import importlib.util
import json
import sys
import types
from pathlib import Path

import numpy as np
import torch

HERE = Path(__file__).resolve().parent
REF = Path("/root/reference/latent_space_exploration")


def _load(fname, modname):
    for name in ("librosa", "soundfile"):
        sys.modules.setdefault(name, types.ModuleType(name))  # inert, attribute-free
    spec = importlib.util.spec_from_file_location(modname, REF / fname)
    m = importlib.util.module_from_spec(spec)
    sys.modules[modname] = m
    spec.loader.exec_module(m)
    return m


def main():
    m00 = _load("00_normalize_dataset_rms.py", "ref00")
    core = _load("map_detector_core.py", "refcore")
    m07 = _load("07_encode_wav_to_latent.py", "ref07")
    m08 = _load("08_fit_radial_detector.py", "ref08")
    m09 = _load("09_evaluate_wav_detection.py", "ref09")
    m10 = _load("10_benchmark_folder_detection.py", "ref10")

    # ---------------------------------------------------------------- rms
    rng = np.random.default_rng(7)
    cases = {}
    n = 22050
    t = np.arange(n, dtype=np.float32) / 22050.0
    cases["noise_quiet"] = (0.004 * rng.standard_normal(n)).astype(np.float32)
    cases["noise_loud"] = (0.4 * rng.standard_normal(n)).astype(np.float32)
    cases["tone"] = (0.3 * np.sin(2 * np.pi * 440.0 * t)).astype(np.float32)
    cases["silent"] = np.zeros(n, dtype=np.float32)
    cases["below_gate"] = (5e-5 * rng.standard_normal(n)).astype(np.float32)
    spiky = (0.001 * rng.standard_normal(n)).astype(np.float32)
    spiky[::500] = 0.9                       # rms small, peaks clip after scaling
    cases["clipping"] = spiky
    cases["short"] = (0.1 * rng.standard_normal(37)).astype(np.float32)
    # rms == float32(1e-4) exactly: 9.99999975e-05 < 1e-4 in float64 (numpy 1.26: gated), not < float32(1e-4) (numpy 2: scaled)
    # (its own generator, so the stream the later fixtures draw from does not move)
    _b = np.random.default_rng(0).standard_normal(64).astype(np.float32)
    cases["gate_boundary"] = (_b * np.float32(0.00010935918)).astype(np.float32)
    assert np.sqrt(np.mean(cases["gate_boundary"] ** 2)) == np.float32(1e-4)
    out = {}
    # The reference pins numpy 1.26.4 (requirements-thesis-baseline-macos-arm64.txt:46); this image has numpy 2.x, whose
    # scalar promotion differs (NEP 50): `rms < rms_min` and `target_rms / (rms + eps)` with rms an np.float32 SCALAR are
    # float64 operations under 1.26 and float32 ones under 2.x.  The reference's own function is run both ways:
    #   out_np2_*  as it is, on this numpy (float32 scalars);
    #   out_*      with ONE substitution in its module namespace: `np.sqrt` of a scalar returns the same value as a Python
    #              float.  A Python float is what NEP 50 calls a weak scalar: `rms < rms_min`, `rms + eps` and the division are
    #              then double arithmetic, and `y * scale` converts the double to float32 once and multiplies in float32 -
    #              operation for operation what value-based casting did under numpy 1.26.  Every other line of the
    #              reference function executes unchanged.
    class _LegacyScalarNumpy:
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def sqrt(x):
            r = np.sqrt(x)
            return float(r) if np.ndim(r) == 0 else r

    for name, y in cases.items():
        y_norm, ok = m00.rms_normalize(y.copy())
        out[f"in_{name}"] = y
        out[f"out_np2_{name}"] = np.asarray(y_norm)
        out[f"ok_np2_{name}"] = np.array(bool(ok))
        m00.np = _LegacyScalarNumpy()
        try:
            y_norm, ok = m00.rms_normalize(y.copy())
        finally:
            m00.np = np
        assert np.asarray(y_norm).dtype == np.float32
        out[f"out_{name}"] = np.asarray(y_norm)
        out[f"ok_{name}"] = np.array(bool(ok))
    out["numpy_version"] = np.array(np.__version__)
    out["promotion"] = np.array("out_* = numpy 1.26 scalar promotion (emulated, see make_golden.py); out_np2_* = numpy 2.x as run")
    np.savez_compressed(HERE / "rms_normalize.npz", **out)

    # ------------------------------------------------------- crop_or_pad_time
    out = {}
    for (T_raw, target) in [(58, 64), (87, 96), (376, 192), (626, 192), (192, 192), (63, 64), (1, 4)]:
        mel = rng.standard_normal((5, T_raw)).astype(np.float32)
        out[f"in_{T_raw}_{target}"] = mel
        a = core.crop_or_pad_time(mel, target)
        b = m07.crop_or_pad_time(mel, target)
        assert np.array_equal(a, b)
        out[f"out_{T_raw}_{target}"] = np.ascontiguousarray(a)
    np.savez_compressed(HERE / "crop_or_pad.npz", **out)

    # ----------------------------------------------------------------- adapter
    # encode_wav_to_latent with wav_to_mel substituted by a function returning a
    # prepared [M, T] tensor; the "encoder" is a stub whose output form varies.
    M, T, D = 6, 5, 4
    mel = torch.tensor(rng.standard_normal((M, T)).astype(np.float32))
    core.wav_to_mel = lambda **kw: mel
    kw = dict(sr=1, duration=1.0, n_mels=M, fmin=0.0, fmax=1.0, hop_length=1, n_fft=2, target_frames=T)
    W = torch.tensor(rng.standard_normal((D, T * M)).astype(np.float32))

    def base(x):  # x is [1,1,T,M]; depends on orientation
        assert tuple(x.shape) == (1, 1, T, M), x.shape
        return x.reshape(1, -1) @ W.T

    forms = {
        "tensor": lambda x: base(x),
        "tuple": lambda x: (base(x), base(x) * 2),
        "list_nontensor_first": lambda x: ["skip", base(x) + 1],
        "dict_mu": lambda x: {"logvar": base(x) * 3, "mu": base(x)},
        "dict_z_over_mu": lambda x: {"mu": base(x) * 3, "z": base(x) - 1},
        "dict_other": lambda x: {"foo": base(x) + 5},
        "ndim3": lambda x: torch.stack([base(x), base(x) * 3], dim=1),       # [1,2,D] -> mean dim1
        "ndim4": lambda x: (base(x).reshape(1, 2, 2, 1)),                    # flatten
    }
    out = {"mel": mel.numpy(), "W": W.numpy()}
    for name, fn in forms.items():
        class Enc(torch.nn.Module):
            def forward(self, x, _fn=fn):
                return _fn(x)
        z = core.encode_wav_to_latent(Enc(), Path("unused.wav"), torch.device("cpu"), **kw)
        out[f"z_{name}"] = z
        v07 = m07.extract_vector(fn(mel.T.unsqueeze(0).unsqueeze(0)))
        out[f"z07_{name}"] = v07.numpy()[0]
    # error behaviour
    errs = {}
    for name, fn in {"batch2": lambda x: torch.zeros(2, D), "str": lambda x: "nope",
                     "empty_tuple": lambda x: (), "dict_no_tensor": lambda x: {"a": 1}}.items():
        class Enc(torch.nn.Module):
            def forward(self, x, _fn=fn):
                return _fn(x)
        try:
            core.encode_wav_to_latent(Enc(), Path("unused.wav"), torch.device("cpu"), **kw)
            errs[name] = "no error"
        except Exception as e:  # noqa: BLE001
            errs[name] = type(e).__name__
    out["errors_json"] = np.array(json.dumps(errs))
    np.savez_compressed(HERE / "adapter.npz", **out)

    # -------------------------------------------------------------- radial fit
    out = {}
    D = 32
    species = list(m09.PRIORITY_ORDER)
    centers = rng.standard_normal((4, D)).astype(np.float32) * 1.5
    Z = {sp: (centers[i] + rng.standard_normal((60 + 17 * i, D))).astype(np.float32) for i, sp in enumerate(species)}
    for i, sp in enumerate(species):
        out[f"Z_{i}"] = Z[sp]
    for q_out in (0.10, 0.15, 0.20, 0.25):
        for i, sp in enumerate(species):
            Z_out = np.concatenate([Z[o] for o in species if o != sp], axis=0)
            mu, rk, rk_in, rk_out, extra = m08.fit_species_with_fp_control(Z[sp], Z_out, 0.95, q_out)
            out[f"mu_{i}_{q_out:.2f}"] = mu
            out[f"rk_{i}_{q_out:.2f}"] = np.array([rk, rk_in, rk_out], dtype=np.float64)
    mu, rk, rk_in, rk_out, extra = m08.fit_species_with_fp_control(Z[species[0]], None, 0.95, 0.1)
    out["rk_noout"] = np.array([rk, rk_in, rk_out], dtype=np.float64)
    # np.quantile(x_float32, q) under the reference's pinned numpy 1.26.4 keeps q, the virtual index and the interpolation weight
    # in float64 and evaluates a + (b - a) * t in float64 (0-d float32 array x 0-d float64 array -> float64); numpy 2.x casts a
    # Python-float q to the array's dtype first and does all of it in float32 (numpy/lib/_function_base_impl.py, quantile():
    # "Use dtype of array if possible").  rk1_* = the reference function with np.quantile routed PAST that one cast, into
    # numpy's own unchanged implementation (q as a float64 array): operation for operation the 1.26 arithmetic.
    from numpy.lib import _function_base_impl as _fb

    class _LegacyQuantileNumpy:
        def __getattr__(self, name):
            return getattr(np, name)

        @staticmethod
        def quantile(a, q):
            return _fb._quantile_unchecked(np.asanyarray(a), np.asanyarray(q, dtype=np.float64))

    m08.np = _LegacyQuantileNumpy()
    try:
        for q_out in (0.10, 0.15, 0.20, 0.25):
            for i, sp in enumerate(species):
                Z_out = np.concatenate([Z[o] for o in species if o != sp], axis=0)
                mu, rk, rk_in, rk_out, extra = m08.fit_species_with_fp_control(Z[sp], Z_out, 0.95, q_out)
                assert isinstance(rk_in, float)
                out[f"rk1_{i}_{q_out:.2f}"] = np.array([rk, rk_in, rk_out], dtype=np.float64)
    finally:
        m08.np = np
    out["promotion"] = np.array("rk_* = numpy 2.x as run; rk1_* = numpy 1.26 quantile arithmetic (see make_golden.py)")
    out["species_json"] = np.array(json.dumps(species))
    np.savez_compressed(HERE / "radial_fit.npz", **out)

    # ----------------------------------------------------------- radial decide
    # DetectorSession.predict_one with the module's encode_wav_to_latent replaced
    # by a lookup of prepared latents; config goes through get_detector_from_config.
    names = ["Pleurodema_thaul", "Batrachyla_taeniata", "Zeta_extra", "Alpha_extra", "Batrachyla_leptopus"]
    D = 8
    cents = {sp: rng.standard_normal(D).astype(np.float32) for sp in names}
    thr = {"Pleurodema_thaul": 2.4, "Batrachyla_taeniata": 2.6, "Zeta_extra": 2.2, "Alpha_extra": 2.3,
           "Batrachyla_leptopus": 2.5}
    cfg = {"chunk_seconds": 1.0,
           "radial_detector": {"centroids": {sp: [float(v) for v in cents[sp]] for sp in names},
                               "thresholds": thr}}
    centroids, thresholds, dur = m09.get_detector_from_config(cfg)
    zs = []
    for _ in range(300):
        zs.append((rng.standard_normal(D) * rng.uniform(0.2, 1.6)).astype(np.float32))
    for sp in names:                                   # exact-on-threshold and near misses
        c = cents[sp]
        u = rng.standard_normal(D).astype(np.float32)
        u /= np.linalg.norm(u)
        for s in (0.999, 1.0, 1.001):
            zs.append((c + u * np.float32(thr[sp] * s)).astype(np.float32))
    zs.append(np.full(D, 50.0, dtype=np.float32))      # NO_DETECT
    # only non-priority species accepted -> sorted(accepted)[0]
    zs.append((cents["Zeta_extra"] + 0.01).astype(np.float32))
    zs = np.stack(zs)

    class FakeMod:
        PRIORITY_ORDER = m09.PRIORITY_ORDER
        l2 = staticmethod(m09.l2)
        cur = None

        @staticmethod
        def encode_wav_to_latent(**kw):
            return FakeMod.cur

    sess = m10.DetectorSession(module=FakeMod, project_root=Path("."), config_path=Path("."),
                               encoder_pt=Path("."), encoder_yaml=Path("."), device="cpu")
    sess.centroids, sess.thresholds, sess.duration, sess.encoder = centroids, thresholds, dur, None
    det, pred, best = [], [], []
    for z in zs:
        FakeMod.cur = z
        d, sp, bd = sess.predict_one(Path("unused.wav"))
        det.append(bool(d))
        pred.append(sp if sp is not None else "")
        best.append(bd)
    np.savez_compressed(HERE / "radial_decide.npz", Z=zs, detected=np.array(det), pred=np.array(pred),
                        best_d=np.array(best, dtype=np.float64),
                        cfg_json=np.array(json.dumps(cfg)), priority_json=np.array(json.dumps(m09.PRIORITY_ORDER)))
    print("golden fixtures written to", HERE)


if __name__ == "__main__":
    main()
