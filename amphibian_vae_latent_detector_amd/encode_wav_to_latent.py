#!/usr/bin/env python3
"""Single-WAV encode with the CLI of the reference's north-star script,
latent_space_exploration/07_encode_wav_to_latent.py (:416-441 flags, :516-526 output):

    python -m amphibian_vae_latent_detector_amd.encode_wav_to_latent --wav chunk.wav \
        --encoder model.pt --encoder-config enc.yaml --device cuda [--jsonl] [--precision 6]

Same flag names, defaults (sr 48000, duration 3.0, 64 mels, 150-15000 Hz, hop 384, n_fft 2048,
target_frames 192), printed lines and JSONL record {"wav", "latent_dim", "vector"}.
Differences: `--device` defaults to `cuda` (there is no CPU path); `--auto-frames` (07:355-409, a
shape probe for an encoder whose architecture is unknown) is accepted and ignored with a note,
because the encoder spec states its input size.
"""
from __future__ import annotations

import argparse
import json
from pathlib import Path

import numpy as np
import torch


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

    encoder = core.load_encoder(encoder_path, encoder_config_path, project_root, device)
    if args.auto_frames:
        print("ℹ️ --auto-frames ignorado: el spec del encoder fija target_frames.")
    print(f"✅ target_frames usado: {args.target_frames}")
    vec_np = core.encode_wav_to_latent(encoder, wav_path, device, sr=args.sr, duration=args.duration, n_mels=args.n_mels,
                                       fmin=args.fmin, fmax=args.fmax, hop_length=args.hop_length, n_fft=args.n_fft,
                                       target_frames=args.target_frames)
    np.set_printoptions(precision=int(args.precision), suppress=True, linewidth=180)
    if args.jsonl:
        print(json.dumps({"wav": str(wav_path), "latent_dim": int(vec_np.size), "vector": vec_np.astype(float).tolist()},
                         ensure_ascii=False))
    else:
        print(f"✅ Latent dim: {vec_np.size}")
        print(vec_np)


if __name__ == "__main__":
    main()
