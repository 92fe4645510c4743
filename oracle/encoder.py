"""Oracle: VAE encoder forward + the reference's output adapter, torch-CPU fp32.

TEST INFRASTRUCTURE (see oracle/__init__.py).  The reference's encoder
architecture is NOT in /root/reference (un-vendored `soundscape_vae`,
requirements-thesis-baseline-macos-arm64.txt:82-83) => the architecture is the
build's explicit `EncoderSpec` (conv3x3+BN(eval)+ReLU stack, flatten, two
linear heads) and is **parity unpinned** against the thesis checkpoint.
What IS pinned (tests/golden/adapter_*.npz, captured from the reference's own
code) is everything around the module call:

  x = mel.T[None, None]        latent_space_exploration/map_detector_core.py:267-268
  out = encoder(x) (no_grad, eval)                       map_detector_core.py:179,240,270
  output adapter               latent_space_exploration/map_detector_core.py:272-300
"""
from __future__ import annotations

import numpy as np
import torch
from torch import nn


class TorchEncoder(nn.Module):
    """conv2d(k,s,p)+BatchNorm2d(eval)+ReLU stack -> flatten (NCHW order) ->
    (fc_mu, fc_logvar).  `spec` is the plain dict form of EncoderSpec:
    {"in_frames", "in_mels", "conv": [[c_in, c_out, k, stride, pad], ...],
     "bn_eps", "latent_dim"}."""

    def __init__(self, spec: dict):
        super().__init__()
        self.spec = spec
        self.conv = nn.ModuleList()
        self.bn = nn.ModuleList()
        h, w = int(spec["in_frames"]), int(spec["in_mels"])
        c_last = 1
        for (c_in, c_out, k, s, p) in spec["conv"]:
            self.conv.append(nn.Conv2d(c_in, c_out, k, stride=s, padding=p, bias=True))
            self.bn.append(nn.BatchNorm2d(c_out, eps=float(spec.get("bn_eps", 1e-5))))
            h = (h + 2 * p - k) // s + 1
            w = (w + 2 * p - k) // s + 1
            c_last = c_out
        self.flat = c_last * h * w
        d = int(spec["latent_dim"])
        self.fc_mu = nn.Linear(self.flat, d)
        self.fc_logvar = nn.Linear(self.flat, d)

    def forward(self, x):
        for conv, bn in zip(self.conv, self.bn):
            x = torch.relu(bn(conv(x)))
        h = x.flatten(1)
        return self.fc_mu(h), self.fc_logvar(h)


def build_torch_encoder(spec: dict, state: dict) -> TorchEncoder:
    """state: {name: np.ndarray | torch.Tensor} with TorchEncoder's key names."""
    m = TorchEncoder(spec)
    sd = {k: torch.as_tensor(np.asarray(v)) if not torch.is_tensor(v) else v for k, v in state.items()}
    missing, unexpected = m.load_state_dict(sd, strict=False)
    missing = [k for k in missing if not k.endswith("num_batches_tracked")]
    if missing or unexpected:
        raise RuntimeError(f"state mismatch: missing={missing} unexpected={unexpected}")
    return m.eval()


def extract_latent(out):
    """map_detector_core.py:272-300 (adapter + batch check), returns float32 [D]."""
    if isinstance(out, torch.Tensor):
        t = out
    elif isinstance(out, (list, tuple)):
        t = next((z for z in out if isinstance(z, torch.Tensor)), None)
        if t is None:
            raise RuntimeError("Salida del encoder es tuple/list sin tensor.")
    elif isinstance(out, dict):
        t = None
        for k in ("z", "latent", "mu", "mean", "embedding"):
            if k in out and isinstance(out[k], torch.Tensor):
                t = out[k]
                break
        if t is None:
            t = next((v for v in out.values() if isinstance(v, torch.Tensor)), None)
        if t is None:
            raise RuntimeError("Salida del encoder es dict sin tensor.")
    else:
        raise RuntimeError(f"No sé interpretar salida del encoder: {type(out)}")
    if t.ndim == 3:
        t = t.mean(dim=1)
    if t.ndim > 2:
        t = t.view(t.shape[0], -1)
    z = t.detach().cpu().numpy()
    if z.shape[0] != 1:
        raise RuntimeError(f"Esperaba batch=1, obtuve: {z.shape}")
    return z[0].astype(np.float32)


@torch.no_grad()
def encode_mel_to_latent(encoder, mel):
    """map_detector_core.py:267-300 given the [M, T] mel array."""
    mel_t = torch.as_tensor(np.asarray(mel), dtype=torch.float32)
    x = mel_t.T.unsqueeze(0).unsqueeze(0)
    return extract_latent(encoder(x))


@torch.no_grad()
def encode_mels_batched(encoder, mels):
    """Batched convenience for tests: mels [B, M, T] -> mu [B, D], logvar [B, D].
    (The reference itself is batch-1; eval-mode BN has no batch coupling, so the
    rows are the batch-1 results up to the CPU library's blocking.)"""
    x = torch.as_tensor(np.asarray(mels), dtype=torch.float32).transpose(1, 2).unsqueeze(1).contiguous()
    mu, logvar = encoder(x)
    return mu.numpy().astype(np.float32), logvar.numpy().astype(np.float32)
