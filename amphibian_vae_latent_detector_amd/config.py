"""Front-end configuration + encoder specification (data, not code).

The mel flag set is the one every reference script exposes with identical
names (latent_space_exploration/08_fit_radial_detector.py:348-354,
10_benchmark_folder_detection.py:123-134, 07_encode_wav_to_latent.py:424-432).
The reference's encoder architecture is not in its repository (un-vendored
`soundscape_vae`), so the encoder is described by an explicit `EncoderSpec`.
"""
from __future__ import annotations

from dataclasses import asdict, dataclass, field
from typing import List, Tuple

PAD_REFLECT = 0   # librosa 0.9.2 stft default (the version the reference pins)
PAD_CONSTANT = 1  # librosa >= 0.10 default


@dataclass(frozen=True)
class FrontendConfig:
    sr: int = 48000
    duration: float = 5.0
    n_fft: int = 2048
    hop_length: int = 384
    n_mels: int = 64
    fmin: float = 150.0
    fmax: float = 15000.0
    target_frames: int = 192
    pad_mode: int = PAD_REFLECT

    @property
    def n_samples(self) -> int:
        # map_detector_core.py:213  target_len = int(sr * duration)
        return int(self.sr * self.duration)

    @property
    def n_bins(self) -> int:
        return 1 + self.n_fft // 2

    @property
    def raw_frames(self) -> int:
        return 1 + self.n_samples // self.hop_length

    def mel_kwargs(self) -> dict:
        """kwargs in the reference's `encode_wav_to_latent` keyword form."""
        return dict(sr=self.sr, duration=self.duration, n_mels=self.n_mels, fmin=self.fmin,
                    fmax=self.fmax, hop_length=self.hop_length, n_fft=self.n_fft,
                    target_frames=self.target_frames)


@dataclass(frozen=True)
class EncoderSpec:
    """conv(k,s,p)+BN(eval)+ReLU stack on [B,1,T,M] -> flatten (NCHW) -> mu/logvar heads."""
    in_frames: int = 64
    in_mels: int = 64
    conv: Tuple[Tuple[int, int, int, int, int], ...] = ()   # (c_in, c_out, k, stride, pad)
    bn_eps: float = 1e-5
    latent_dim: int = 32

    def out_hw(self) -> List[Tuple[int, int]]:
        h, w = self.in_frames, self.in_mels
        dims = []
        for (_, _, k, s, p) in self.conv:
            h = (h + 2 * p - k) // s + 1
            w = (w + 2 * p - k) // s + 1
            dims.append((h, w))
        return dims

    @property
    def flat(self) -> int:
        h, w = self.out_hw()[-1]
        return self.conv[-1][1] * h * w

    def to_dict(self) -> dict:
        d = asdict(self)
        d["conv"] = [list(c) for c in self.conv]
        return d

    @staticmethod
    def from_dict(d: dict) -> "EncoderSpec":
        return EncoderSpec(in_frames=int(d["in_frames"]), in_mels=int(d["in_mels"]),
                           conv=tuple(tuple(int(v) for v in c) for c in d["conv"]),
                           bn_eps=float(d.get("bn_eps", 1e-5)), latent_dim=int(d["latent_dim"]))


# ---------------------------------------------------------------------------
# Presets: `ref-default`/`ref-07cli` are the reference's own
# defaults; `bench-*` are the benchmark presets BASELINE.json's configs name.
# ---------------------------------------------------------------------------
PRESETS = {
    "ref-default": FrontendConfig(),                                      # 10_benchmark...:123-134
    "ref-07cli": FrontendConfig(duration=3.0),                            # 07_encode...:424-432
    "bench-1s": FrontendConfig(sr=22050, duration=1.0, n_fft=2048, hop_length=384, n_mels=64,
                               fmin=150.0, fmax=11025.0, target_frames=64),
    "bench-1s-128": FrontendConfig(sr=22050, duration=1.0, n_fft=1024, hop_length=256, n_mels=128,
                                   fmin=150.0, fmax=11025.0, target_frames=96),
}


def tiny_spec(in_frames: int = 64, in_mels: int = 64) -> EncoderSpec:
    """BASELINE.json configs[0]: 2 conv + fc, latent_dim=8."""
    return EncoderSpec(in_frames=in_frames, in_mels=in_mels,
                       conv=((1, 8, 3, 2, 1), (8, 16, 3, 2, 1)), latent_dim=8)


def thesis_shape_spec(in_frames: int = 64, in_mels: int = 64) -> EncoderSpec:
    """BASELINE.json configs[1]: 4x[conv3x3 s2 p1 + BN + ReLU] 1->32->64->128->256, latent_dim=32."""
    return EncoderSpec(in_frames=in_frames, in_mels=in_mels,
                       conv=((1, 32, 3, 2, 1), (32, 64, 3, 2, 1), (64, 128, 3, 2, 1), (128, 256, 3, 2, 1)),
                       latent_dim=32)


SPECS = {"tiny": tiny_spec, "thesis-shape": thesis_shape_spec}
