"""Encoder objects on top of libamph: weight packing, checkpoints, the HIP session.

`HipEncoder` is what `load_encoder` returns in place of the reference's
`torch.nn.Module` (latent_space_exploration/map_detector_core.py:150-179): a
callable taking x = [B,1,T,M] (map_detector_core.py:268) and returning the
tuple (mu, logvar), so the reference's output adapter ("first tensor of a
tuple", map_detector_core.py:275-276) yields mu as the latent.
All compute runs in hand-written HIP kernels; torch only owns device memory
and streams.
"""
from __future__ import annotations

import ctypes as C
import warnings
from collections import OrderedDict
from pathlib import Path
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _native as N
from .config import EncoderSpec, FrontendConfig

TARGET = "amphibian_vae_latent_detector_amd.encoder.ConvVAEEncoder"


# ----------------------------------------------------------------------------
# weights
# ----------------------------------------------------------------------------
def state_keys(spec: EncoderSpec):
    keys = []
    for i in range(len(spec.conv)):
        keys += [f"conv.{i}.weight", f"conv.{i}.bias", f"bn.{i}.weight", f"bn.{i}.bias",
                 f"bn.{i}.running_mean", f"bn.{i}.running_var"]
    keys += ["fc_mu.weight", "fc_mu.bias", "fc_logvar.weight", "fc_logvar.bias"]
    return keys


def state_shapes(spec: EncoderSpec) -> Dict[str, Tuple[int, ...]]:
    shapes = {}
    for i, (ci, co, k, _, _) in enumerate(spec.conv):
        shapes[f"conv.{i}.weight"] = (co, ci, k, k)
        for nm in ("conv.%d.bias", "bn.%d.weight", "bn.%d.bias", "bn.%d.running_mean", "bn.%d.running_var"):
            shapes[nm % i] = (co,)
    d, flat = spec.latent_dim, spec.flat
    shapes.update({"fc_mu.weight": (d, flat), "fc_mu.bias": (d,), "fc_logvar.weight": (d, flat), "fc_logvar.bias": (d,)})
    return shapes


def random_init_state(spec: EncoderSpec, seed: int = 0) -> Dict[str, np.ndarray]:
    """Random-init weights of the given architecture (there are no checkpoints:
    the reference's model.pt is an external asset, README.md:78-81).  He-style
    conv weights and non-trivial BatchNorm running statistics so that every
    term of the eval-mode BN formula is exercised."""
    rng = np.random.default_rng(seed)
    st: Dict[str, np.ndarray] = {}
    for i, (ci, co, k, _, _) in enumerate(spec.conv):
        fan_in = ci * k * k
        st[f"conv.{i}.weight"] = (rng.standard_normal((co, ci, k, k)) * np.sqrt(2.0 / fan_in)).astype(np.float32)
        st[f"conv.{i}.bias"] = (0.05 * rng.standard_normal(co)).astype(np.float32)
        st[f"bn.{i}.weight"] = rng.uniform(0.5, 1.5, co).astype(np.float32)
        st[f"bn.{i}.bias"] = (0.1 * rng.standard_normal(co)).astype(np.float32)
        st[f"bn.{i}.running_mean"] = (0.1 * rng.standard_normal(co)).astype(np.float32)
        st[f"bn.{i}.running_var"] = rng.uniform(0.5, 1.5, co).astype(np.float32)
    d, flat = spec.latent_dim, spec.flat
    for head in ("fc_mu", "fc_logvar"):
        st[f"{head}.weight"] = (rng.standard_normal((d, flat)) / np.sqrt(flat)).astype(np.float32)
        st[f"{head}.bias"] = (0.05 * rng.standard_normal(d)).astype(np.float32)
    return st


def pack_weights(spec: EncoderSpec, state: Dict[str, np.ndarray]) -> np.ndarray:
    """Flat fp32 blob in the order include/amph.h documents for amph_create."""
    shapes = state_shapes(spec)
    parts = []
    for k in state_keys(spec):
        if k not in state:
            raise RuntimeError(f"state_dict is missing '{k}'")
        a = state[k]
        a = a.detach().cpu().numpy() if torch.is_tensor(a) else np.asarray(a)
        if tuple(a.shape) != shapes[k]:
            raise RuntimeError(f"'{k}' has shape {tuple(a.shape)}, spec needs {shapes[k]}")
        parts.append(np.ascontiguousarray(a, dtype=np.float32).ravel())
    return np.concatenate(parts)


def save_checkpoint(encoder_pt: Path, encoder_yaml: Path, spec: EncoderSpec, state: Dict[str, np.ndarray]) -> None:
    """Writes the two assets `load_encoder` takes: a `{"state_dict": ...}` torch
    checkpoint (one of the forms the reference accepts, map_detector_core.py:124-132)
    and a YAML with an `encoder:` block carrying `_target_` (:117-121)."""
    import yaml
    sd = {k: torch.as_tensor(np.asarray(state[k])) for k in state_keys(spec)}
    torch.save({"state_dict": sd}, str(encoder_pt))
    block = {"_target_": TARGET, **spec.to_dict()}
    with open(encoder_yaml, "w", encoding="utf-8") as f:
        yaml.safe_dump({"encoder": block}, f, sort_keys=False)


# ----------------------------------------------------------------------------
# HIP session
# ----------------------------------------------------------------------------
def _device_index(device) -> int:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise RuntimeError(
            f"device '{device}': this build runs only on an MI355X (spell it 'cuda' / 'cuda:N' as PyTorch-ROCm does); "
            "there is no CPU path")
    return dev.index if dev.index is not None else torch.cuda.current_device()


def _ptr(t: Optional[torch.Tensor]):
    return C.c_void_p(t.data_ptr()) if t is not None else None


class AmphSession:
    """One amph_ctx (one device, one front-end configuration, optional encoder)."""

    def __init__(self, cfg: FrontendConfig, spec: Optional[EncoderSpec], blob: Optional[np.ndarray], device,
                 max_batch: int = 256):
        self.cfg, self.spec = cfg, spec
        self.device_index = _device_index(device)
        self.device = torch.device("cuda", self.device_index)
        torch.cuda.init()
        ccfg = N.CFrontendConfig.from_config(cfg)
        cspec = N.CEncoderSpec.from_spec(spec) if spec is not None else None
        ctx = C.c_void_p()
        if blob is not None:
            blob = np.ascontiguousarray(blob, dtype=np.float32)
            wptr, wn = blob.ctypes.data_as(C.c_void_p), blob.size
        else:
            wptr, wn = None, 0
        st = N.lib.amph_create(C.byref(ccfg), C.byref(cspec) if cspec is not None else None, wptr, wn,
                               self.device_index, int(max_batch), C.byref(ctx))
        if st != N.AMPH_OK:
            raise N.AmphError(st, N.last_error(None))
        self._ctx = ctx
        self.max_batch = int(max_batch)
        self.raw_frames = N.lib.amph_raw_frames(ctx)
        self.latent_dim = N.lib.amph_latent_dim(ctx) if spec is not None else 0

    def close(self):
        if getattr(self, "_ctx", None):
            N.lib.amph_destroy(self._ctx)
            self._ctx = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass

    # -- helpers
    def _stream(self):
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _dev(self, t, dtype) -> torch.Tensor:
        if not torch.is_tensor(t):
            t = torch.from_numpy(np.ascontiguousarray(t))
        return t.to(device=self.device, dtype=dtype).contiguous()

    def _chk(self, st):
        N.check(st, self._ctx)

    # -- entry points (tensors in, tensors out, all on self.device)
    def mel_basis(self) -> np.ndarray:
        out = np.empty((self.cfg.n_mels, self.cfg.n_bins), dtype=np.float32)
        self._chk(N.lib.amph_get_mel_basis(self._ctx, out.ctypes.data_as(C.c_void_p), out.size))
        return out

    def rms_normalize(self, y):
        """00_normalize_dataset_rms.py:29-38 on rows of float32 or int16 -> (y_norm, ok)."""
        y = torch.as_tensor(y) if not torch.is_tensor(y) else y
        is_pcm = y.dtype == torch.int16
        y = self._dev(y, torch.int16 if is_pcm else torch.float32)
        if y.ndim == 1:
            y = y[None]
        out = torch.empty_like(y)
        ok = torch.empty(y.shape[0], dtype=torch.int32, device=self.device)
        fn = N.lib.amph_rms_normalize_pcm16 if is_pcm else N.lib.amph_rms_normalize_f32
        with torch.cuda.device(self.device):
            self._chk(fn(self._ctx, _ptr(y), y.shape[0], y.shape[1], _ptr(out), _ptr(ok), self._stream()))
        return out, ok

    def mel(self, wave, lengths=None, layout: int = 0) -> torch.Tensor:
        """wav_to_mel for rows of float32 samples -> [B, M, T] (layout 0) or [B, T, M] (layout 1)."""
        wave = self._dev(wave, torch.float32)
        if wave.ndim == 1:
            wave = wave[None]
        B = wave.shape[0]
        lens = self._dev(lengths, torch.int32) if lengths is not None else None
        shape = (B, self.cfg.n_mels, self.cfg.target_frames) if layout == 0 else (B, self.cfg.target_frames, self.cfg.n_mels)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_mel_f32(self._ctx, _ptr(wave), B, wave.shape[1], _ptr(lens), layout, _ptr(out), self._stream()))
        return out

    def mel_power(self, wave, lengths=None) -> torch.Tensor:
        wave = self._dev(wave, torch.float32)
        if wave.ndim == 1:
            wave = wave[None]
        B = wave.shape[0]
        lens = self._dev(lengths, torch.int32) if lengths is not None else None
        out = torch.empty((B, self.raw_frames, self.cfg.n_mels), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_mel_power_f32(self._ctx, _ptr(wave), B, wave.shape[1], _ptr(lens), _ptr(out), self._stream()))
        return out

    def mel_pcm16(self, pcm, lengths=None, apply_rms: bool = False, layout: int = 0, want_power: bool = False):
        """wav_to_mel for rows of decoded PCM16 samples (optionally through 00_normalize_dataset_rms.py's rms_normalize +
        PCM16 write / read first) -> image [B, M, T] / [B, T, M]; want_power: also the raw mel power [B, T_raw, M]."""
        pcm = self._dev(pcm, torch.int16)
        if pcm.ndim == 1:
            pcm = pcm[None]
        B = pcm.shape[0]
        lens = self._dev(lengths, torch.int32) if lengths is not None else None
        shape = (B, self.cfg.n_mels, self.cfg.target_frames) if layout == 0 else (B, self.cfg.target_frames, self.cfg.n_mels)
        out = torch.empty(shape, dtype=torch.float32, device=self.device)
        pw = torch.empty((B, self.raw_frames, self.cfg.n_mels), dtype=torch.float32, device=self.device) if want_power else None
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_mel_pcm16(self._ctx, _ptr(pcm), B, pcm.shape[1], _ptr(lens), 1 if apply_rms else 0, layout,
                                           _ptr(out), _ptr(pw), self._stream()))
        return (out, pw) if want_power else out

    # -- the fp16 range of the default GEMM arithmetic (include/amph.h, amph_range_status)
    def range_overflow(self, clear: bool = True) -> bool:
        """True when a kernel of this context met a value the two-term fp16 split cannot hold (>= 65 504 or NaN) since the last
        clear: the latents of those calls are not valid.  Synchronises the current stream first."""
        torch.cuda.current_stream(self.device).synchronize()
        return bool(N.lib.amph_range_status(self._ctx, 1 if clear else 0))

    def _rerun_in_f32_if_out_of_range(self, run):
        """The reference's module is fp32 with fp32's range (map_detector_core.py:175-179, :270).  After `run()`: if the range flag is
        up, this context moves to the f32-MFMA kernels for good ("split_gemm" 0: its checkpoint / data are outside the split's range)
        and the call runs again."""
        if not self.range_overflow():
            return
        warnings.warn("an activation or input of this encoder is outside fp16's range (>= 65 504 or NaN): the two-term fp16 GEMM path "
                      "cannot represent it; this context now runs the f32-MFMA kernels (amph_configure split_gemm = 0)", RuntimeWarning,
                      stacklevel=3)
        self.configure("split_gemm", 0)
        self.configure("split_fused", 0)
        run()
        self.range_overflow()

    def encoder_forward(self, x: torch.Tensor, check_range: bool = True) -> Tuple[torch.Tensor, torch.Tensor]:
        x = self._dev(x, torch.float32)
        if x.ndim != 4 or x.shape[1] != 1 or x.shape[2] != self.cfg.target_frames or x.shape[3] != self.cfg.n_mels:
            raise RuntimeError(f"encoder expects [B,1,{self.cfg.target_frames},{self.cfg.n_mels}], got {tuple(x.shape)}")
        B, D = x.shape[0], self.latent_dim
        mu = torch.empty((B, D), dtype=torch.float32, device=self.device)
        lv = torch.empty((B, D), dtype=torch.float32, device=self.device)

        def run():
            with torch.cuda.device(self.device):
                self._chk(N.lib.amph_encoder_forward(self._ctx, _ptr(x), B, _ptr(mu), _ptr(lv), self._stream()))
        run()
        if check_range:
            self._rerun_in_f32_if_out_of_range(run)
        return mu, lv

    def encode(self, wave, lengths=None, apply_rms: bool = False, want_logvar: bool = False, out_mu=None, check_range: bool = True):
        """Fused chunk -> latent.  wave: [B, n_in] int16 (PCM16) or float32, device or host.
        Returns (mu [B,D], logvar or None, rms_ok [B] int32) as device tensors.
        check_range (default): synchronise the stream and, should an activation have left the range of the two-term fp16 GEMM
        arithmetic, run the call again on the f32-MFMA kernels (include/amph.h, amph_range_status).  A caller that keeps calls in
        flight passes False and asks `range_overflow()` at its own synchronisation point."""
        wave = torch.as_tensor(wave) if not torch.is_tensor(wave) else wave
        is_pcm = wave.dtype == torch.int16
        wave = self._dev(wave, torch.int16 if is_pcm else torch.float32)
        if wave.ndim == 1:
            wave = wave[None]
        B, D = wave.shape[0], self.latent_dim
        lens = self._dev(lengths, torch.int32) if lengths is not None else None
        mu = out_mu if out_mu is not None else torch.empty((B, D), dtype=torch.float32, device=self.device)
        lv = torch.empty((B, D), dtype=torch.float32, device=self.device) if want_logvar else None
        ok = torch.empty(B, dtype=torch.int32, device=self.device)
        fn = N.lib.amph_encode_pcm16 if is_pcm else N.lib.amph_encode_f32

        def run():
            with torch.cuda.device(self.device):
                self._chk(fn(self._ctx, _ptr(wave), B, wave.shape[1], _ptr(lens), 1 if apply_rms else 0, _ptr(mu), _ptr(lv),
                             _ptr(ok), self._stream()))
        run()
        if check_range:
            self._rerun_in_f32_if_out_of_range(run)
        return mu, lv, ok

    def encode_host(self, pcm: np.ndarray, lengths=None, apply_rms: bool = False):
        """Host int16 [B, n_in] -> host mu [B, D] through the pinned, copy-overlapped path."""
        pcm = np.ascontiguousarray(pcm, dtype=np.int16)
        B, D = pcm.shape[0], self.latent_dim
        mu = np.empty((B, D), dtype=np.float32)
        ok = np.empty(B, dtype=np.int32)
        lens = np.ascontiguousarray(lengths, dtype=np.int32) if lengths is not None else None
        self._chk(N.lib.amph_encode_pcm16_host(
            self._ctx, pcm.ctypes.data_as(C.c_void_p), B, pcm.shape[1],
            lens.ctypes.data_as(C.c_void_p) if lens is not None else None, 1 if apply_rms else 0,
            mu.ctypes.data_as(C.c_void_p), None, ok.ctypes.data_as(C.c_void_p)))
        return mu, ok

    def decide_radial(self, z: torch.Tensor, centroids: np.ndarray, thresholds: np.ndarray, rank: np.ndarray):
        z = self._dev(z, torch.float32)
        B, D = z.shape
        cent = np.ascontiguousarray(centroids, dtype=np.float32)
        thr = np.ascontiguousarray(thresholds, dtype=np.float64)
        rk = np.ascontiguousarray(rank, dtype=np.int32)
        pred = torch.empty(B, dtype=torch.int32, device=self.device)
        best = torch.empty(B, dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_decide_radial(self._ctx, _ptr(z), B, D, cent.ctypes.data_as(C.c_void_p),
                                               thr.ctypes.data_as(C.c_void_p), rk.ctypes.data_as(C.c_void_p),
                                               cent.shape[0], _ptr(pred), _ptr(best), self._stream()))
            torch.cuda.current_stream(self.device).synchronize()   # host arrays above must outlive the async copies
        return pred, best

    def fit_radial(self, z: torch.Tensor, offsets, q_in: float, q_out: float, numpy2: bool = False):
        """fit_species_with_fp_control (08_fit_radial_detector.py:310-333) for every species at once on resident latents
        (K9, amph_fit_radial).  z: [n, D] device tensor, rows of species k at offsets[k]:offsets[k + 1].
        -> (centroids float32 [K, D], rk_in, rk_out, rk float64 [K], summaries float64 [K, 2, 4])."""
        z = self._dev(z, torch.float32)
        off = np.ascontiguousarray(offsets, dtype=np.int64)
        K, D = off.shape[0] - 1, z.shape[1]
        if int(off[-1]) != z.shape[0]:
            raise RuntimeError(f"offsets end at {int(off[-1])}, z has {z.shape[0]} rows")
        cent = np.empty((K, D), dtype=np.float32)
        rk_in, rk_out, rk = (np.empty(K, dtype=np.float64) for _ in range(3))
        summ = np.empty((K, 2, 4), dtype=np.float64)
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_fit_radial(self._ctx, _ptr(z), off.ctypes.data_as(C.c_void_p), K, D, float(q_in), float(q_out),
                                            1 if numpy2 else 0, cent.ctypes.data_as(C.c_void_p), rk_in.ctypes.data_as(C.c_void_p),
                                            rk_out.ctypes.data_as(C.c_void_p), rk.ctypes.data_as(C.c_void_p),
                                            summ.ctypes.data_as(C.c_void_p), self._stream()))
        return cent, rk_in, rk_out, rk, summ

    def resample(self, y, orig_sr: int, target_sr: int) -> torch.Tensor:
        """librosa.load's rate conversion (resampy kaiser_best + fix_length) for rows of float32 samples, on the GPU
        (K8, amph_resample_f32).  y: [n] or [B, n] -> device tensor [B, ceil(n * target_sr / orig_sr)]."""
        from .resample import kaiser_best_window
        y = self._dev(y, torch.float32)
        if y.ndim == 1:
            y = y[None]
        B, n_in = y.shape
        n_out = N.lib.amph_resample_length(int(n_in), int(orig_sr), int(target_sr))
        if n_out < 1:
            raise RuntimeError("Input signal length is too small to resample")
        win, num_table = kaiser_best_window()
        out = torch.empty((B, n_out), dtype=torch.float32, device=self.device)
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_resample_f32(self._ctx, _ptr(y), B, n_in, int(orig_sr), int(target_sr),
                                              win.ctypes.data_as(C.c_void_p), win.shape[0], num_table, _ptr(out), n_out,
                                              self._stream()))
        return out

    def decide_map(self, z: torch.Tensor, means: np.ndarray, precisions: np.ndarray, logdets: np.ndarray,
                   log_priors: np.ndarray, tau: Optional[float] = None, want_scores: bool = False):
        """Gaussian-MAP scoring + decision (K7, include/amph.h amph_decide_map).
        -> (pred int32 [B] (-1 = NO_DETECT), best_score float64 [B], scores float64 [B, K] or None)."""
        z = self._dev(z, torch.float32)
        B, D = z.shape
        mu = np.ascontiguousarray(means, dtype=np.float32).reshape(-1, D)
        K = mu.shape[0]
        pr = np.ascontiguousarray(precisions, dtype=np.float32).reshape(K, D, D)
        ld = np.ascontiguousarray(logdets, dtype=np.float64).reshape(K)
        lp = np.ascontiguousarray(log_priors, dtype=np.float64).reshape(K)
        pred = torch.empty(B, dtype=torch.int32, device=self.device)
        best = torch.empty(B, dtype=torch.float64, device=self.device)
        scores = torch.empty((B, K), dtype=torch.float64, device=self.device) if want_scores else None
        with torch.cuda.device(self.device):
            self._chk(N.lib.amph_decide_map(self._ctx, _ptr(z), B, D, mu.ctypes.data_as(C.c_void_p),
                                            pr.ctypes.data_as(C.c_void_p), ld.ctypes.data_as(C.c_void_p),
                                            lp.ctypes.data_as(C.c_void_p), K, 0 if tau is None else 1,
                                            0.0 if tau is None else float(tau), _ptr(pred), _ptr(best), _ptr(scores),
                                            self._stream()))
            torch.cuda.current_stream(self.device).synchronize()   # host arrays above must outlive the async copies
        return pred, best, scores

    def configure(self, key: str, value: int):
        """Tuning knobs of include/amph.h's amph_configure ("lanes", "frames_per_block")."""
        self._chk(N.lib.amph_configure(self._ctx, key.encode(), int(value)))

    # -- profiling
    def profile_enable(self, on, only_stage: Optional[int] = None):
        """on=False: off; on=True: every stage; only_stage=i: bracket stage i alone."""
        mode = 0 if not on else (1 if only_stage is None else 2 + int(only_stage))
        self._chk(N.lib.amph_profile_enable(self._ctx, mode))

    def profile_reset(self):
        self._chk(N.lib.amph_profile_reset(self._ctx))

    def profile_read(self):
        out = []
        for i in range(N.lib.amph_profile_stage_count(self._ctx)):
            n, ms, fl, by = C.c_int64(), C.c_double(), C.c_double(), C.c_double()
            self._chk(N.lib.amph_profile_read(self._ctx, i, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)))
            out.append({"stage": N.lib.amph_profile_stage_name(self._ctx, i).decode(), "launches": n.value,
                        "total_ms": ms.value, "flops_per_chunk": fl.value, "bytes_per_chunk": by.value})
        return out


class HipEncoder:
    """Stand-in for the reference's encoder `nn.Module` (see module docstring)."""

    def __init__(self, spec: EncoderSpec, state: Dict[str, np.ndarray], device, max_batch: int = 256):
        self.spec = spec
        self.device = torch.device("cuda", _device_index(device))
        self.max_batch = max_batch
        self._blob = pack_weights(spec, state)
        self._sessions: Dict[FrontendConfig, AmphSession] = {}
        self._file_sessions: "OrderedDict[FrontendConfig, AmphSession]" = OrderedDict()

    # nn.Module-flavoured no-ops the reference's callers use (map_detector_core.py:179)
    def eval(self):
        return self

    def to(self, device):
        if torch.device(device) != self.device and _device_index(device) != self.device.index:
            raise RuntimeError("HipEncoder is bound to its device; call load_encoder with the target device")
        return self

    def session(self, cfg: FrontendConfig) -> AmphSession:
        if (cfg.target_frames, cfg.n_mels) != (self.spec.in_frames, self.spec.in_mels):
            raise RuntimeError(f"front end produces [{cfg.target_frames},{cfg.n_mels}] but the encoder was built for "
                               f"[{self.spec.in_frames},{self.spec.in_mels}]")
        s = self._sessions.get(cfg)
        if s is None:
            s = AmphSession(cfg, self.spec, self._blob, self.device, self.max_batch)
            self._sessions[cfg] = s
        return s

    WHOLE_FILE_SESSIONS = 2      # contexts kept for the duration <= 0 path (one per distinct file length; least recently used goes)

    def whole_file_session(self, cfg: FrontendConfig) -> AmphSession:
        """A context for ONE whole file of cfg.n_samples samples (map_detector_core.py:212, duration <= 0: no pad / truncate): batch 1 on
        one lane - its workspaces scale with the file's frame count, so a 256-row, four-lane context per distinct length would keep
        gigabytes per file - and only the last WHOLE_FILE_SESSIONS lengths stay alive (the evicted context is destroyed: its device
        memory returns to the allocator)."""
        if (cfg.target_frames, cfg.n_mels) != (self.spec.in_frames, self.spec.in_mels):
            raise RuntimeError(f"front end produces [{cfg.target_frames},{cfg.n_mels}] but the encoder was built for "
                               f"[{self.spec.in_frames},{self.spec.in_mels}]")
        s = self._file_sessions.pop(cfg, None)
        if s is None:
            s = AmphSession(cfg, self.spec, self._blob, self.device, 1)
            s.configure("lanes", 1)
        self._file_sessions[cfg] = s
        while len(self._file_sessions) > self.WHOLE_FILE_SESSIONS:
            _, old = self._file_sessions.popitem(last=False)
            old.close()
        return s

    def _any_session(self) -> AmphSession:
        if self._sessions:
            return next(iter(self._sessions.values()))
        from .config import PRESETS
        base = PRESETS["bench-1s"]
        cfg = FrontendConfig(sr=base.sr, duration=base.duration, n_fft=2048, hop_length=base.hop_length,
                             n_mels=self.spec.in_mels, fmin=base.fmin, fmax=base.fmax, target_frames=self.spec.in_frames)
        return self.session(cfg)

    @torch.no_grad()
    def __call__(self, x: torch.Tensor) -> Tuple[torch.Tensor, torch.Tensor]:
        return self._any_session().encoder_forward(x)

    forward = __call__
