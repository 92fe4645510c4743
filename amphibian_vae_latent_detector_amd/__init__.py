"""MI355X-native chunk -> latent encoder of the amphibian VAE latent detector.

Hot path only: RMS normalise -> STFT/mel -> VAE encoder -> latent,
as hand-written HIP kernels for gfx950 behind the reference's
`latent_space_exploration/map_detector_core.py` function surface.

Importing `config` / `synth` needs no GPU library; everything that computes
(`map_detector_core`, `encoder`, `folder`, `radial`, `wavio`) imports `_native`,
which raises if libamph.so has not been built.  There is no CPU fallback.
"""
from .config import PRESETS, SPECS, EncoderSpec, FrontendConfig  # noqa: F401

__all__ = ["PRESETS", "SPECS", "EncoderSpec", "FrontendConfig"]
