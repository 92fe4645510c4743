"""Diagnostic (ab/libamph_rdbT.so = -DAMPH_ABL_RDB=3 copied over the in-tree library): s_memtime stamps of k_stft_pair2's register dB stage, per wave,
read back from the first 96 values of every row's image.  Stamps: 0 kernel start, 1 stage entered, 2 at barrier 1, 3 behind it, 4 at barrier 3,
5 behind it, 6 stores issued, 7 stores acknowledged."""
import sys, numpy as np, torch
sys.path.insert(0, ".")
from amphibian_vae_latent_detector_amd.config import PRESETS
from amphibian_vae_latent_detector_amd.encoder import AmphSession
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16 as synth
cfg = PRESETS["bench-1s-128"]
B = 256
sess = AmphSession(cfg, None, None, "cuda:0", max_batch=B)
pcm = synth(list(range(B)), cfg.n_samples, cfg.sr)
for it in range(3):
    img = sess.mel_pcm16(pcm, apply_rms=True, layout=1).cpu().numpy()
st = img.reshape(B, -1)[:, :96].copy().view(np.uint32).reshape(B, 12, 8).astype(np.int64)
d = (st - st[:, :1, :1]) & 0xFFFFFFFF          # relative to wave 0's kernel start
np.set_printoptions(linewidth=200)
print("median over rows, per wave (rows) x stamp (columns), cycles since the workgroup's start:")
print(np.median(d, axis=0).astype(int))
