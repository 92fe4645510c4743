"""ctypes binding of libamph.so (include/amph.h).  No fallback: if the HIP
library is missing or fails to load, importing this module raises."""
from __future__ import annotations

import ctypes as C
import os
import sys
import warnings
from pathlib import Path

from .config import EncoderSpec, FrontendConfig

AMPH_MAX_CONV_LAYERS = 8
LIB_PATH = Path(__file__).resolve().parent / "libamph.so"

STATUS_NAMES = {0: "AMPH_OK", 1: "AMPH_ERR_INVALID_ARG", 2: "AMPH_ERR_NO_DEVICE", 3: "AMPH_ERR_HIP",
                4: "AMPH_ERR_IO", 5: "AMPH_ERR_UNSUPPORTED"}
AMPH_OK, AMPH_ERR_INVALID_ARG, AMPH_ERR_NO_DEVICE, AMPH_ERR_HIP, AMPH_ERR_IO, AMPH_ERR_UNSUPPORTED = range(6)


class AmphError(RuntimeError):
    def __init__(self, status: int, message: str):
        super().__init__(f"{STATUS_NAMES.get(status, status)}: {message}")
        self.status = status


class CFrontendConfig(C.Structure):
    _fields_ = [("sr", C.c_int32), ("n_samples", C.c_int32), ("n_fft", C.c_int32), ("hop_length", C.c_int32),
                ("n_mels", C.c_int32), ("fmin", C.c_float), ("fmax", C.c_float), ("target_frames", C.c_int32),
                ("pad_mode", C.c_int32)]

    @staticmethod
    def from_config(cfg: FrontendConfig) -> "CFrontendConfig":
        return CFrontendConfig(cfg.sr, cfg.n_samples, cfg.n_fft, cfg.hop_length, cfg.n_mels, cfg.fmin, cfg.fmax,
                               cfg.target_frames, cfg.pad_mode)


class CConvLayer(C.Structure):
    _fields_ = [("c_in", C.c_int32), ("c_out", C.c_int32), ("kernel", C.c_int32), ("stride", C.c_int32),
                ("pad", C.c_int32)]


class CEncoderSpec(C.Structure):
    _fields_ = [("in_frames", C.c_int32), ("in_mels", C.c_int32), ("n_conv", C.c_int32),
                ("conv", CConvLayer * AMPH_MAX_CONV_LAYERS), ("bn_eps", C.c_float), ("latent_dim", C.c_int32)]

    @staticmethod
    def from_spec(spec: EncoderSpec) -> "CEncoderSpec":
        if len(spec.conv) > AMPH_MAX_CONV_LAYERS:
            raise ValueError(f"at most {AMPH_MAX_CONV_LAYERS} conv layers")
        s = CEncoderSpec()
        s.in_frames, s.in_mels, s.n_conv = spec.in_frames, spec.in_mels, len(spec.conv)
        for i, (ci, co, k, st, p) in enumerate(spec.conv):
            s.conv[i] = CConvLayer(ci, co, k, st, p)
        s.bn_eps, s.latent_dim = spec.bn_eps, spec.latent_dim
        return s


def _hardware_queues() -> None:
    """The encode entry points spread their batches over 4 HIP streams ("lanes", include/amph.h) and a process gets 4
    hardware queues by default: with RCCL's streams, or the copy stream of the host path, beside them the lanes end up
    sharing queues and lose their overlap (measured 7.3 vs 6.1 ms per 10 000-chunk step under torch.distributed).  The
    ROCm runtime reads GPU_MAX_HW_QUEUES once, when HIP initialises, so it is defaulted here - import of this package
    normally precedes the first HIP call - and a late import is reported instead of silently running slower."""
    if "GPU_MAX_HW_QUEUES" in os.environ:
        return
    os.environ["GPU_MAX_HW_QUEUES"] = "8"
    torch = sys.modules.get("torch")
    if torch is not None and torch.cuda.is_initialized():
        warnings.warn("amphibian_vae_latent_detector_amd was imported after HIP had been initialised without "
                      "GPU_MAX_HW_QUEUES: the 4 encode lanes share the default 4 hardware queues with every other "
                      "stream of the process. Export GPU_MAX_HW_QUEUES=8 before the first torch.cuda call.",
                      RuntimeWarning, stacklevel=3)


def _load() -> C.CDLL:
    _hardware_queues()
    if not LIB_PATH.exists():
        raise ImportError(
            f"{LIB_PATH} not found: the HIP library has not been built. Run "
            f"`python -c 'import __graft_entry__ as g; g.build()'` (or `make -C {LIB_PATH.parent / 'csrc'}`). "
            "There is no CPU fallback.")
    lib = C.CDLL(str(LIB_PATH))
    vp, i32p, f32p, i16p, f64p = C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p
    sigs = {
        "amph_abi_version": (C.c_int, []),
        "amph_create": (C.c_int, [C.POINTER(CFrontendConfig), C.POINTER(CEncoderSpec), f32p, C.c_size_t, C.c_int,
                                  C.c_int, C.POINTER(vp)]),
        "amph_destroy": (None, [vp]),
        "amph_last_error": (C.c_char_p, [vp]),
        "amph_configure": (C.c_int, [vp, C.c_char_p, C.c_int]),
        "amph_raw_frames": (C.c_int, [vp]),
        "amph_latent_dim": (C.c_int, [vp]),
        "amph_max_batch": (C.c_int, [vp]),
        "amph_get_mel_basis": (C.c_int, [vp, f32p, C.c_size_t]),
        "amph_rms_normalize_f32": (C.c_int, [vp, f32p, C.c_int, C.c_int, f32p, i32p, vp]),
        "amph_rms_normalize_pcm16": (C.c_int, [vp, i16p, C.c_int, C.c_int, i16p, i32p, vp]),
        "amph_mel_f32": (C.c_int, [vp, f32p, C.c_int, C.c_int, i32p, C.c_int, f32p, vp]),
        "amph_mel_power_f32": (C.c_int, [vp, f32p, C.c_int, C.c_int, i32p, f32p, vp]),
        "amph_mel_pcm16": (C.c_int, [vp, i16p, C.c_int, C.c_int, i32p, C.c_int, C.c_int, f32p, f32p, vp]),
        "amph_encoder_forward": (C.c_int, [vp, f32p, C.c_int, f32p, f32p, vp]),
        "amph_range_status": (C.c_int, [vp, C.c_int]),
        "amph_encode_pcm16": (C.c_int, [vp, i16p, C.c_int64, C.c_int, i32p, C.c_int, f32p, f32p, i32p, vp]),
        "amph_encode_f32": (C.c_int, [vp, f32p, C.c_int64, C.c_int, i32p, C.c_int, f32p, f32p, i32p, vp]),
        "amph_encode_pcm16_host": (C.c_int, [vp, i16p, C.c_int64, C.c_int, i32p, C.c_int, f32p, f32p, i32p]),
        "amph_decide_radial": (C.c_int, [vp, f32p, C.c_int64, C.c_int, f32p, f64p, i32p, C.c_int, i32p, f32p, vp]),
        "amph_fit_radial": (C.c_int, [vp, f32p, vp, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int, f32p, f64p, f64p, f64p, f64p, vp]),
        "amph_decide_map": (C.c_int, [vp, f32p, C.c_int64, C.c_int, f32p, f32p, f64p, f64p, C.c_int, C.c_int, C.c_double,
                                      i32p, f64p, f64p, vp]),
        "amph_resample_length": (C.c_int, [C.c_int, C.c_int, C.c_int]),
        "amph_resample_f32": (C.c_int, [vp, f32p, C.c_int, C.c_int, C.c_int, C.c_int, f64p, C.c_int, C.c_int, f32p, C.c_int, vp]),
        "amph_wav_read_pcm16": (C.c_int, [C.c_char_p, C.c_int, i16p, C.c_int, C.POINTER(C.c_int32)]),
        "amph_wav_read_many": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, C.c_int, i16p, C.c_int, i32p, i32p, C.c_int]),
        "amph_wav_sizes": (C.c_int, [C.POINTER(C.c_char_p), C.c_int, vp, C.c_int]),
        "amph_host_alloc": (vp, [C.c_size_t]),
        "amph_host_free": (None, [vp]),
        "amph_profile_enable": (C.c_int, [vp, C.c_int]),
        "amph_profile_stage_count": (C.c_int, [vp]),
        "amph_profile_stage_name": (C.c_char_p, [vp, C.c_int]),
        "amph_profile_read": (C.c_int, [vp, C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                        C.POINTER(C.c_double)]),
        "amph_profile_reset": (C.c_int, [vp]),
    }
    for name, (res, args) in sigs.items():
        fn = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        fn.restype = res
        fn.argtypes = args
    if lib.amph_abi_version() != 1:
        raise ImportError(f"libamph ABI version {lib.amph_abi_version()} != 1")
    return lib


EXPORTED_SYMBOLS = (
    "amph_abi_version", "amph_create", "amph_destroy", "amph_last_error", "amph_configure", "amph_raw_frames", "amph_latent_dim",
    "amph_max_batch", "amph_get_mel_basis", "amph_rms_normalize_f32", "amph_rms_normalize_pcm16", "amph_mel_f32",
    "amph_mel_power_f32", "amph_mel_pcm16", "amph_encoder_forward", "amph_range_status", "amph_encode_pcm16", "amph_encode_f32", "amph_encode_pcm16_host",
    "amph_decide_radial", "amph_fit_radial", "amph_decide_map", "amph_resample_length", "amph_resample_f32", "amph_wav_read_pcm16", "amph_wav_read_many", "amph_wav_sizes", "amph_host_alloc", "amph_host_free",
    "amph_profile_enable", "amph_profile_stage_count", "amph_profile_stage_name", "amph_profile_read",
    "amph_profile_reset",
)

lib = _load()


def last_error(ctx) -> str:
    msg = lib.amph_last_error(ctx)
    return msg.decode("utf-8", "replace") if msg else ""


def check(status: int, ctx=None) -> None:
    if status != AMPH_OK:
        raise AmphError(status, last_error(ctx))
