#!/usr/bin/env python3
"""Single-WAV encode with the CLI of the reference's north-star script,
latent_space_exploration/07_encode_wav_to_latent.py (:416-441 flags, :516-526 output):

    python -m amphibian_vae_latent_detector_amd.encode_wav_to_latent --wav chunk.wav \
        --encoder model.pt --encoder-config enc.yaml --device cuda [--jsonl] [--precision 6]

Same flag names, defaults (sr 48000, duration 3.0, 64 mels, 150-15000 Hz, hop 384, n_fft 2048,
target_frames 192), printed lines and JSONL record {"wav", "latent_dim", "vector"}.
Differences: `--device` defaults to `cuda` (there is no CPU path); `--auto-frames` (07:355-409, a
shape probe for an encoder whose architecture is unknown) exits with a message: the encoder spec
states its input size, there is nothing to probe.

`extract_vector` and the positional `wav_to_mel` are 07's own variants of the core functions
(07:223-233, 264-297): the adapter also accepts the dict key "enc", raises ValueError, returns
a tensor [B, D] and keeps an empty time axis empty instead of averaging over it.
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import numpy as np
import torch


def wav_to_mel(wav_path, sr, duration, n_mels, fmin, fmax, hop_length, n_fft, target_frames, device=None) -> torch.Tensor:
    """07's positional form of wav_to_mel (07:223-233): float32 [n_mels, target_frames], HIP front end."""
    from . import map_detector_core as core
    return core.wav_to_mel(Path(wav_path), sr=sr, duration=duration, n_mels=n_mels, fmin=fmin, fmax=fmax, hop_length=hop_length,
                           n_fft=n_fft, target_frames=target_frames, device=device)


def extract_vector(out) -> torch.Tensor:
    """07's output adapter (07:264-297) -> tensor [B, D]: Tensor -> itself; list / tuple -> first tensor; dict -> first of the keys
    z, latent, mu, mean, embedding, enc, else the first tensor value; [B, T, C] -> mean over T (an EMPTY T axis gives [B, 0]);
    more than two axes -> flattened."""
    if isinstance(out, torch.Tensor):
        t = out
    elif isinstance(out, (list, tuple)):
        t = next((x for x in out if isinstance(x, torch.Tensor)), None)
        if t is None:
            raise ValueError("Salida list/tuple sin tensors.")
    elif isinstance(out, dict):
        t = next((out[k] for k in ("z", "latent", "mu", "mean", "embedding", "enc") if isinstance(out.get(k), torch.Tensor)), None)
        if t is None:
            t = next((v for v in out.values() if isinstance(v, torch.Tensor)), None)
        if t is None:
            raise ValueError("Salida dict sin tensors.")
    else:
        raise ValueError(f"No sé interpretar salida: {type(out)}")
    if t.ndim == 3:
        if t.shape[1] == 0:
            return t[:, :0]
        t = t.mean(dim=1)
    if t.ndim > 2:
        t = t.reshape(t.shape[0], -1)
    return t


def parse_args(argv=None) -> argparse.Namespace:
    p = argparse.ArgumentParser()
    p.add_argument("--wav", required=True, type=str)
    p.add_argument("--encoder", type=str, default=None)
    p.add_argument("--encoder-config", type=str, default=None)
    p.add_argument("--device", type=str, default="cuda")
    p.add_argument("--sr", type=int, default=48000)
    p.add_argument("--duration", type=float, default=3.0)
    p.add_argument("--n-mels", type=int, default=64)
    p.add_argument("--fmin", type=float, default=150.0)
    p.add_argument("--fmax", type=float, default=15000.0)
    p.add_argument("--hop-length", type=int, default=384)
    p.add_argument("--n-fft", type=int, default=2048)
    p.add_argument("--target-frames", type=int, default=192)
    p.add_argument("--auto-frames", action="store_true")
    p.add_argument("--auto-max-frames", type=int, default=512)
    p.add_argument("--auto-step", type=int, default=8)
    p.add_argument("--jsonl", action="store_true")
    p.add_argument("--precision", type=int, default=6)
    return p.parse_args(argv)


def main(argv=None) -> None:
    from . import map_detector_core as core
    args = parse_args(argv)
    project_root = core.find_project_root(Path(__file__).resolve().parent)
    wav_path = Path(args.wav)
    if not wav_path.is_absolute():
        wav_path = (Path.cwd() / wav_path).resolve()
    device = torch.device(args.device)
    try:
        encoder_path = Path(args.encoder) if args.encoder else core.resolve_default_encoder_pt(project_root)
    except FileNotFoundError:
        encoder_path = None
    if encoder_path is None or not encoder_path.exists():
        raise SystemExit("❌ No encontré encoder por defecto. Pasa la ruta con --encoder ...")
    try:
        encoder_config_path = Path(args.encoder_config) if args.encoder_config else core.resolve_default_encoder_yaml(project_root)
    except FileNotFoundError:
        raise SystemExit("❌ No encontré el YAML del encoder. Pasa la ruta con --encoder-config ...")

    print(f"📌 Project root: {project_root}")
    print(f"🎧 WAV: {wav_path}")
    print(f"🧠 Encoder: {encoder_path}")
    print(f"🧾 Encoder config: {encoder_config_path}")
    print(f"🖥️ Device: {device}\n")

    if args.auto_frames:
        raise SystemExit("❌ --auto-frames no aplica aquí: el spec del encoder (YAML) fija el tamaño de entrada; "
                         "usa --target-frames con el valor del spec (in_frames).")
    encoder = core.load_encoder(encoder_path, encoder_config_path, project_root, device)
    print(f"✅ target_frames usado: {args.target_frames}")
    # 07's own forward (07:469-512): mel -> x = mel.T[None, None] -> encoder -> extract_vector -> batch check
    mel = wav_to_mel(wav_path, args.sr, args.duration, args.n_mels, args.fmin, args.fmax, args.hop_length, args.n_fft,
                     args.target_frames, device=device)
    x = mel.T.unsqueeze(0).unsqueeze(0).to(device)
    with torch.no_grad():
        vec = extract_vector(encoder(x))
    if vec.shape[0] != 1:
        raise RuntimeError(f"Esperaba batch=1, obtuve: {tuple(vec.shape)}")
    vec_np = vec[0].detach().cpu().numpy().astype(np.float32)
    np.set_printoptions(precision=int(args.precision), suppress=True, linewidth=180)
    if args.jsonl:
        print(json.dumps({"wav": str(wav_path), "latent_dim": int(vec_np.size), "vector": vec_np.astype(float).tolist()},
                         ensure_ascii=False))
    else:
        print(f"✅ Latent dim: {vec_np.size}")
        print(vec_np)


if __name__ == "__main__":
    main()
