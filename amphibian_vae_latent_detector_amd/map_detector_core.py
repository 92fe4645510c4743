"""Drop-in mirror of the reference's importable hot-path surface,
`latent_space_exploration/map_detector_core.py`, on MI355X HIP kernels.

Same names, keyword sets, return types and error behaviour:

  load_encoder(encoder_pt, encoder_yaml, project_root, device)      ref :150-179
  wav_to_mel(wav_path, *, sr, duration, n_mels, fmin, fmax,
             hop_length, n_fft, target_frames) -> torch f32 [M, T]   ref :198-237
  encode_wav_to_latent(encoder, wav_path, device, *, ...) -> np f32 [D]   ref :240-300
  crop_or_pad_time(mel, target_frames)                               ref :185-195
  find_project_root / resolve_default_* / load_json / save_json / summarize_1d   ref :48-101
  inv_and_logdet / gaussian_logpdf_from_precision / get_priors_from_map_meta /
  get_chunk_seconds_for_map / read_map_detector_params (host numpy, as in the reference)   ref :306-420

plus the batched form the reference lacks (it loops batch-1,
08_fit_radial_detector.py:488-503): `encode_chunks`.

Per-file failures raise ordinary Python exceptions (callers catch `Exception`
per file and continue: 08_fit_radial_detector.py:504-506); nothing aborts the
process.  There is no CPU path: `device` must be a `cuda` device.
"""
from __future__ import annotations

import json
from collections import OrderedDict
from pathlib import Path
from typing import Any, Dict, Optional

import numpy as np
import torch

from . import wavio
from .config import PAD_REFLECT, EncoderSpec, FrontendConfig
from .encoder import TARGET, AmphSession, HipEncoder


# ---------------------------------------------------------------------------
# Paths / IO (host-side helpers the reference's callers import alongside)
# ---------------------------------------------------------------------------
def find_project_root(start: Path) -> Path:
    here = Path(start).resolve()
    probe = here
    for _ in range(15):
        if (probe / "downloaded_models").exists() and (probe / "latent_space_exploration").exists():
            return probe
        probe = probe.parent
    return here


def _must_exist(p: Path, what: str) -> Path:
    if not p.exists():
        raise FileNotFoundError(f"No encontré {what} en: {p}")
    return p


def resolve_default_config(project_root: Path) -> Path:
    return _must_exist(Path(project_root) / "config.json", "config.json")


_MODEL_DIR = "bird_net_vae_audio_splitted_encoder_v0"


def resolve_default_encoder_pt(project_root: Path) -> Path:
    return _must_exist(Path(project_root) / "models" / _MODEL_DIR / "model.pt", "encoder .pt")


def resolve_default_encoder_yaml(project_root: Path) -> Path:
    return _must_exist(Path(project_root) / "models" / _MODEL_DIR / "bird_net_vae_audio_splitted.yaml", "encoder YAML")


def load_json(path: Path) -> Dict[str, Any]:
    with open(path, "r", encoding="utf-8") as fh:
        data = json.load(fh)
    if not isinstance(data, dict):
        raise ValueError("JSON no es un objeto dict.")
    return data


def save_json(path: Path, obj: Dict[str, Any]) -> None:
    with open(path, "w", encoding="utf-8") as fh:
        json.dump(obj, fh, indent=2, ensure_ascii=False)


def summarize_1d(x: np.ndarray) -> Dict[str, float]:
    keys = (("min", None), ("p05", 0.05), ("p50", 0.50), ("p95", 0.95), ("max", None))
    if x.size == 0:
        return {k: float("nan") for k, _ in keys}
    out = {}
    for k, q in keys:
        out[k] = float(np.min(x) if k == "min" else np.max(x) if k == "max" else np.quantile(x, q))
    return out


# ---------------------------------------------------------------------------
# Encoder loading
# ---------------------------------------------------------------------------
def _load_spec_yaml(encoder_yaml: Path) -> EncoderSpec:
    import yaml
    with open(encoder_yaml, "r", encoding="utf-8") as fh:
        cfg = yaml.safe_load(fh)
    enc = cfg.get("encoder") if isinstance(cfg, dict) else None
    if not (isinstance(enc, dict) and "_target_" in enc):
        raise RuntimeError("El YAML no contiene un bloque 'encoder:' con _target_.")
    if enc["_target_"] != TARGET:
        raise RuntimeError(f"encoder._target_ = {enc['_target_']!r}: this build only instantiates {TARGET!r} "
                           "(the reference's soundscape_vae factory is not available)")
    return EncoderSpec.from_dict(enc)


def load_encoder(encoder_pt: Path, encoder_yaml: Path, project_root: Path, device: torch.device,
                 max_batch: int = 256) -> HipEncoder:
    """Checkpoint -> encoder on `device`, eval semantics.  Accepts the reference's
    `{"state_dict": ...}` and flat-tensor-dict checkpoint forms; a pickled
    nn.Module cannot be loaded safely (weights_only) and is rejected."""
    del project_root  # the reference only uses it to extend sys.path for Hydra
    try:
        ckpt = torch.load(str(encoder_pt), map_location="cpu", weights_only=True)
    except FileNotFoundError:
        raise
    except Exception as e:  # noqa: BLE001
        raise RuntimeError(f"No pude leer el checkpoint de forma segura (weights_only): {e}") from e
    state = None
    if isinstance(ckpt, dict):
        if isinstance(ckpt.get("state_dict"), dict):
            state = ckpt["state_dict"]
        elif ckpt and all(torch.is_tensor(v) for v in ckpt.values()):
            state = ckpt
    if state is None:
        raise RuntimeError("No encontré state_dict en el checkpoint.")
    spec = _load_spec_yaml(Path(encoder_yaml))
    return HipEncoder(spec, state, device, max_batch=max_batch).eval()


# ---------------------------------------------------------------------------
# Audio -> mel -> latent
# ---------------------------------------------------------------------------
def crop_or_pad_time(mel: np.ndarray, target_frames: int) -> np.ndarray:
    frames = mel.shape[1]
    extra = frames - target_frames
    if extra == 0:
        return mel
    if extra > 0:
        lo = extra // 2
        return mel[:, lo:lo + target_frames]
    left = (-extra) // 2
    return np.pad(mel, ((0, 0), (left, -extra - left)), mode="constant")


def _config(sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames, pad_mode=PAD_REFLECT,
            whole_file_samples: Optional[int] = None) -> FrontendConfig:
    """duration > 0: the fixed-length chunk of map_detector_core.py:212-217.  duration <= 0 (the reference then skips the
    pad / truncate step and transforms the WHOLE file, :212): the caller passes the file's length and gets a configuration
    whose n_samples is exactly that length - one front-end context per distinct length (cached like any other)."""
    if not duration > 0:
        if whole_file_samples is None:
            raise RuntimeError("duration <= 0 means the whole file: its length is needed (whole_file_samples)")
        n = int(whole_file_samples)
        duration = (n + 0.5) / float(sr)                  # int(sr * duration) == n for every n (FrontendConfig.n_samples)
    cfg = FrontendConfig(sr=int(sr), duration=float(duration), n_fft=int(n_fft), hop_length=int(hop_length),
                         n_mels=int(n_mels), fmin=float(fmin), fmax=float(fmax), target_frames=int(target_frames),
                         pad_mode=pad_mode)
    if whole_file_samples is not None and cfg.n_samples != int(whole_file_samples):
        raise RuntimeError(f"internal: n_samples {cfg.n_samples} != file length {whole_file_samples}")
    return cfg


def _whole_file(wav_path: Path, sr: int, device=None):
    """The whole signal as librosa.load(path, sr=sr, mono=True) returns it (the duration <= 0 path): float32 [n]."""
    wav_path = Path(wav_path)
    if not wav_path.exists():
        raise FileNotFoundError(f"No existe WAV: {wav_path}")
    y, rate = wavio.load_audio(wav_path)
    if rate != int(sr):
        y = wavio.load_float_mono(wav_path, int(sr), session=_frontend_session(PRESETS_FOR_RESAMPLE(int(sr)), device))
    return np.ascontiguousarray(y, dtype=np.float32)


def PRESETS_FOR_RESAMPLE(sr: int) -> FrontendConfig:
    """any valid configuration at this rate: the resampler (K8) only needs a context on the device"""
    return FrontendConfig(sr=sr, duration=1.0, n_fft=2048, hop_length=384, n_mels=64, fmin=0.0, fmax=sr / 2.0, target_frames=64)


_frontend_sessions: Dict[tuple, AmphSession] = {}
_file_frontend_sessions: "OrderedDict[tuple, AmphSession]" = OrderedDict()
_FILE_SESSIONS = 2               # whole-file front-end contexts kept (one per distinct file length; least recently used goes)


def _frontend_session(cfg: FrontendConfig, device, whole_file: bool = False) -> AmphSession:
    """whole_file: a context for ONE file of cfg.n_samples samples (duration <= 0): batch 1, one lane, and only the last
    _FILE_SESSIONS lengths stay alive - its workspaces scale with the file's length (HipEncoder.whole_file_session)."""
    dev = torch.device(device if device is not None else "cuda")
    key = (cfg, dev.index if dev.index is not None else torch.cuda.current_device())
    if whole_file:
        s = _file_frontend_sessions.pop(key, None)
        if s is None:
            s = AmphSession(cfg, None, None, dev, max_batch=1)
            s.configure("lanes", 1)
        _file_frontend_sessions[key] = s
        while len(_file_frontend_sessions) > _FILE_SESSIONS:
            _, old = _file_frontend_sessions.popitem(last=False)
            old.close()
        return s
    s = _frontend_sessions.get(key)
    if s is None:
        s = AmphSession(cfg, None, None, dev, max_batch=64)
        _frontend_sessions[key] = s
    return s


def _load_wave(wav_path: Path, cfg: FrontendConfig, device=None):
    """-> (array [n], n_valid): int16 through the native reader for mono PCM16,
    float32 otherwise.  Only the first n_samples are needed (fix_length truncates)."""
    wav_path = Path(wav_path)
    if not wav_path.exists():
        raise FileNotFoundError(f"No existe WAV: {wav_path}")
    try:
        return wavio.read_pcm16_mono(wav_path, cfg.sr, cfg.n_samples)
    except NotImplementedError:                          # stereo, another subtype or another rate: general path
        y = wavio.load_float_mono(wav_path, cfg.sr, session=_frontend_session(cfg, device))[:cfg.n_samples]
        return y, y.shape[0]


def wav_to_mel(wav_path: Path, *, sr: int, duration: float, n_mels: int, fmin: float, fmax: float, hop_length: int,
               n_fft: int, target_frames: int, device: Optional[torch.device] = None) -> torch.Tensor:
    """float32 [n_mels, target_frames] on the host, computed by the HIP front end."""
    if not duration > 0:                                  # whole file (ref :212: no pad / truncate step)
        y = _whole_file(Path(wav_path), sr, device)
        cfg = _config(sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames, whole_file_samples=y.shape[0])
        return _frontend_session(cfg, device, whole_file=True).mel(y[None], layout=0)[0].cpu()
    cfg = _config(sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames)
    sess = _frontend_session(cfg, device)
    y, n_valid = _load_wave(Path(wav_path), cfg, device)
    if y.dtype == np.int16:
        y = y.astype(np.float32) / np.float32(32768.0)
    mel = sess.mel(y[None], lengths=np.array([n_valid], dtype=np.int32), layout=0)
    return mel[0].cpu()


def _adapt_output(out) -> np.ndarray:
    """The reference's output adapter (ref :272-300), same rules and messages."""
    if isinstance(out, torch.Tensor):
        t = out
    elif isinstance(out, (list, tuple)):
        t = next((o for o in out if isinstance(o, torch.Tensor)), None)
        if t is None:
            raise RuntimeError("Salida del encoder es tuple/list sin tensor.")
    elif isinstance(out, dict):
        t = next((out[k] for k in ("z", "latent", "mu", "mean", "embedding")
                  if isinstance(out.get(k), torch.Tensor)), None)
        if t is None:
            t = next((v for v in out.values() if isinstance(v, torch.Tensor)), None)
        if t is None:
            raise RuntimeError("Salida del encoder es dict sin tensor.")
    else:
        raise RuntimeError(f"No sé interpretar salida del encoder: {type(out)}")
    if t.ndim == 3:
        t = t.mean(dim=1)
    if t.ndim > 2:
        t = t.reshape(t.shape[0], -1)
    z = t.detach().cpu().numpy()
    if z.shape[0] != 1:
        raise RuntimeError(f"Esperaba batch=1, obtuve: {z.shape}")
    return z[0].astype(np.float32)


@torch.no_grad()
def encode_wav_to_latent(encoder, wav_path: Path, device: torch.device, *, sr: int, duration: float, n_mels: int,
                         fmin: float, fmax: float, hop_length: int, n_fft: int, target_frames: int) -> np.ndarray:
    if not duration > 0:                                  # whole file (ref :212): a context per file length
        y = _whole_file(Path(wav_path), sr, device)
        cfg = _config(sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames, whole_file_samples=y.shape[0])
        n_valid = y.shape[0]
    else:
        cfg = _config(sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames)
        y, n_valid = _load_wave(Path(wav_path), cfg, device)
    whole = not duration > 0
    if isinstance(encoder, HipEncoder):
        sess = encoder.whole_file_session(cfg) if whole else encoder.session(cfg)
        mu, _, _ = sess.encode(y[None], lengths=np.array([n_valid], dtype=np.int32), apply_rms=False)
        return _adapt_output((mu,))
    # a caller-supplied module (any callable on x = [1,1,T,M]): HIP front end + the reference adapter
    sess = _frontend_session(cfg, device, whole_file=whole)
    if y.dtype == np.int16:
        y = y.astype(np.float32) / np.float32(32768.0)
    x = sess.mel(y[None], lengths=np.array([n_valid], dtype=np.int32), layout=1).unsqueeze(1)
    return _adapt_output(encoder(x.to(device)))


@torch.no_grad()
def encode_chunks(encoder: HipEncoder, chunks, device: Optional[torch.device] = None, *, sr: int, duration: float,
                  n_mels: int, fmin: float, fmax: float, hop_length: int, n_fft: int, target_frames: int,
                  lengths=None, apply_rms: bool = False, return_tensor: bool = False):
    """Batched encode: chunks [B, n] int16 (raw PCM16) or float32, host or device -> latents [B, D].
    apply_rms=True fuses 00_normalize_dataset_rms.py (incl. its PCM16 write/read) in front."""
    del device
    cfg = _config(sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames)
    mu, _, _ = encoder.session(cfg).encode(chunks, lengths=lengths, apply_rms=apply_rms)
    return mu if return_tensor else mu.cpu().numpy()


# ---------------------------------------------------------------------------
# MAP helpers (host numpy, ref :306-420).  The batched on-GPU form of the scoring
# is map_detector.MapDetector.decide -> HIP kernel K7 (amph_decide_map).
# ---------------------------------------------------------------------------
def inv_and_logdet(cov: np.ndarray):
    sign, ld = np.linalg.slogdet(cov)
    if sign <= 0:
        d = cov.shape[0]
        cov2 = cov + (1e-3 * np.eye(d, dtype=cov.dtype))
        sign, ld = np.linalg.slogdet(cov2)
        if sign <= 0:
            raise RuntimeError("Covarianza no PD incluso tras regularización.")
        cov = cov2
    prec = np.linalg.inv(cov).astype(np.float32)
    return prec, float(ld)


def gaussian_logpdf_from_precision(z: np.ndarray, mu: np.ndarray, prec: np.ndarray, logdet_cov: float) -> float:
    d = int(z.shape[0])
    diff = (z - mu).astype(np.float32)
    quad = float(diff.T @ prec @ diff)
    return -0.5 * (quad + float(logdet_cov) + d * float(np.log(2.0 * np.pi)))


def _map_block(cfg: Dict[str, Any]) -> Dict[str, Any]:
    md = cfg.get("map_detector", {})
    return md if isinstance(md, dict) else {}


def _meta_fit(cfg: Dict[str, Any]) -> Dict[str, Any]:
    meta = _map_block(cfg).get("meta_fit", {})
    return meta if isinstance(meta, dict) else {}


def get_priors_from_map_meta(cfg: Dict[str, Any], species) -> Dict[str, float]:
    """Priors stored by the fit (`map_detector.meta_fit.per_species[sp].prior`), clipped at 0 and renormalised when
    every species has one; otherwise uniform (ref :326-356)."""
    per = _meta_fit(cfg).get("per_species", {})
    found: Dict[str, float] = {}
    for sp in species:
        try:
            found[sp] = float(per.get(sp, {}).get("prior"))
        except Exception:  # noqa: BLE001  (missing / None / non-numeric: fall back for ALL species)
            found = {}
            break
    total = sum(max(0.0, v) for v in found.values())
    if found and total > 0:
        return {sp: max(0.0, v) / total for sp, v in found.items()}
    if found:
        return found
    return {sp: 1.0 / len(species) for sp in species} if len(species) else {}


def get_chunk_seconds_for_map(cfg: Dict[str, Any]) -> float:
    """meta_fit.chunk_seconds of the MAP fit if present, else config.json:chunk_seconds, else 5.0 (ref :359-371)."""
    for source in (_meta_fit(cfg), cfg):
        if "chunk_seconds" in source:
            try:
                return float(source["chunk_seconds"])
            except Exception:  # noqa: BLE001
                if source is cfg:
                    return 5.0
    return 5.0


def read_map_detector_params(cfg: Dict[str, Any]):
    """-> (means {sp: f32[D]}, precisions {sp: f32[D,D]}, logdets {sp: float}, tau float|None) as 08b stores them
    (ref :374-420; same checks and messages)."""
    md = cfg.get("map_detector", None)
    if not isinstance(md, dict):
        raise ValueError("config.json no contiene map_detector (dict). Ejecuta antes 08b_fit_map_detector.py")
    if md.get("model", "") != "gaussian_map":
        raise ValueError(f"map_detector.model inesperado: {md.get('model')}")
    raw = [md.get(key, None) for key in ("means", "precision", "logdet_cov")]
    if not all(isinstance(r, dict) for r in raw):
        raise ValueError("map_detector debe contener 'means', 'precision' y 'logdet_cov' como dicts.")
    usable = lambda sp, v: isinstance(sp, str) and isinstance(v, list) and len(v) > 0   # noqa: E731
    means = {sp: np.asarray(v, dtype=np.float32) for sp, v in raw[0].items() if usable(sp, v)}
    precisions: Dict[str, np.ndarray] = {}
    for sp, mat in raw[1].items():
        if not usable(sp, mat):
            continue
        P = np.asarray(mat, dtype=np.float32)
        if P.ndim != 2 or P.shape[0] != P.shape[1]:
            raise ValueError(f"precision[{sp}] debe ser matriz cuadrada, obtuve shape={P.shape}")
        precisions[sp] = P
    logdets = {sp: float(v) for sp, v in raw[2].items() if isinstance(sp, str)}
    if not (means and precisions and logdets):
        raise ValueError("means/precision/logdet_cov vacíos o mal formateados en config.json.")
    tau = md.get("tau", None)
    return means, precisions, logdets, (None if tau is None else float(tau))
