#!/usr/bin/env python3
"""Mirror of latent_space_exploration/00_normalize_dataset_rms.py on K0 (the HIP rms_normalize kernels).

Same tool, same flags: `--base-dir` holds `train_chunks/`, `val_chunks/`, `test_chunks/` (each with one
directory per species); `<split>_norm/<species>/<name>.wav` is written next to them - every file, also
the ones the silence gate leaves unchanged (00_normalize_dataset_rms.py:55-57), as mono PCM16 at `--sr`
(soundfile's default WAV subtype).  The original dataset is not touched.

    python -m amphibian_vae_latent_detector_amd.normalize_dataset_rms --base-dir latent_space_exploration --sr 48000

What runs where: the files of a species directory are read by the native RIFF reader (mono PCM16 at `sr`) in
groups of equal length and normalised whole - the RMS is over the whole file, 00:51-53 - by
`amph_rms_normalize_pcm16` (K0: mean(y^2) in numpy's float32 summation order, gate and scale in the pinned
numpy's scalar arithmetic, clip, and libsndfile's float -> PCM16 conversion rint(x * 32767)); other subtypes,
channel counts and rates take `wavio.load_float_mono` (channel mean, K8 resampler) and the float32 kernel.
"""
from __future__ import annotations

import argparse
from collections import defaultdict
from pathlib import Path
from typing import Dict, List, Optional, Tuple

import numpy as np
import torch

from . import wavio
from .map_detector_core import PRESETS_FOR_RESAMPLE, _frontend_session
from .synth import write_wav_pcm16

SETS = ("train_chunks", "val_chunks", "test_chunks")          # 00_normalize_dataset_rms.py:68
GROUP_ROWS = 64                                               # rows per K0 launch (the front-end context's max_batch)


def rms_normalize(y, target_rms: float = 0.05, rms_min: float = 1e-4, eps: float = 1e-8, device=None):
    """00_normalize_dataset_rms.py:29-38 on the GPU: float32 [n] -> (y_norm float32 [n], ok).  The constants are the
    reference's defaults (K0 has them built in); other values are refused rather than silently ignored."""
    if (target_rms, rms_min, eps) != (0.05, 1e-4, 1e-8):
        raise RuntimeError("the HIP rms_normalize implements target_rms=0.05, rms_min=1e-4, eps=1e-8 (the reference's defaults)")
    y = np.ascontiguousarray(y, dtype=np.float32)
    sess = _frontend_session(PRESETS_FOR_RESAMPLE(22050), device)
    out, ok = sess.rms_normalize(y[None])
    return out[0].cpu().numpy(), bool(ok[0].item())


def _pcm16_file_length(path: Path) -> Optional[int]:
    """Samples of a canonical mono PCM16 file (44-byte header), by size alone; None = let the general loader look at it."""
    try:
        size = path.stat().st_size
    except OSError:
        return None
    return (size - 44) // 2 if size >= 44 else None


def process_folder(src_root: Path, dst_root: Path, sr: int = 48000, device=None, verbose: bool = True) -> Tuple[int, int]:
    """00_normalize_dataset_rms.py:41-57.  -> (files written, files that could not be read)."""
    src_root, dst_root = Path(src_root), Path(dst_root)
    sess = _frontend_session(PRESETS_FOR_RESAMPLE(int(sr)), device)
    written = failed = 0
    for sp_dir in [d for d in src_root.iterdir() if d.is_dir()]:
        dst_sp = dst_root / sp_dir.name
        dst_sp.mkdir(parents=True, exist_ok=True)
        wavs = list(sp_dir.glob("*.wav"))
        # groups of equal length: one K0 launch per group of up to GROUP_ROWS files
        by_len: Dict[int, List[Path]] = defaultdict(list)
        general: List[Path] = []
        for w in wavs:
            n = _pcm16_file_length(w)
            (by_len[n] if n and n > 0 else general).append(w)
        for n, group in by_len.items():
            for off in range(0, len(group), GROUP_ROWS):
                part = group[off:off + GROUP_ROWS]
                rows = np.zeros((len(part), n), dtype=np.int16)
                lengths, status = wavio.read_many_into(rows, part, int(sr), n_threads=min(8, len(part)))
                good = [i for i in range(len(part)) if status[i] == 0 and lengths[i] == n]
                general += [part[i] for i in range(len(part)) if i not in good]      # another subtype / rate / a longer header: general loader
                if not good:
                    continue
                out, _ = sess.rms_normalize(torch.from_numpy(rows[good]))
                out = out.cpu().numpy()
                for k, i in enumerate(good):
                    write_wav_pcm16(dst_sp / part[i].name, out[k], int(sr))
                    written += 1
        for w in general:
            try:
                y = wavio.load_float_mono(w, int(sr), session=sess)              # librosa.load(wav, sr=sr, mono=True), 00:51
                out, _ = sess.rms_normalize(np.ascontiguousarray(y, dtype=np.float32)[None])
                q = np.rint(np.clip(out[0].cpu().numpy(), -1.0, 1.0) * np.float32(32767.0)).astype(np.int16)   # sf.write: float -> PCM_16
                write_wav_pcm16(dst_sp / w.name, q, int(sr))
                written += 1
            except Exception as e:  # noqa: BLE001   (the reference would raise here; a broken file should not cost the rest of the split)
                failed += 1
                if verbose:
                    print(f"⚠ {w}: {e}")
        if verbose:
            print(f"{src_root.name}/{sp_dir.name}: {len(wavs)} wav")
    return written, failed


def main(argv=None) -> int:
    parser = argparse.ArgumentParser()
    parser.add_argument("--base-dir", type=str, default="latent_space_exploration")
    parser.add_argument("--sr", type=int, default=48000)
    parser.add_argument("--device", type=str, default="cuda", help="cuda o cuda:N (no hay ruta de CPU)")
    args = parser.parse_args(argv)
    base = Path(args.base_dir).resolve()
    for s in SETS:
        src, dst = base / s, base / f"{s}_norm"
        if not src.exists():
            print(f"⚠ No existe {src}")
            continue
        print(f"\nProcesando {s} → {s}_norm")
        process_folder(src, dst, sr=args.sr, device=args.device)
    print("\n✅ Dataset normalizado listo.")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())
