import sys, time
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from amphibian_vae_latent_detector_amd import PRESETS, SPECS
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
cfg = PRESETS["bench-1s"]; spec = SPECS["thesis-shape"](cfg.target_frames, cfg.n_mels)
enc = HipEncoder(spec, random_init_state(spec, 0), "cuda:0", max_batch=256); sess = enc.session(cfg)
pool = synth_batch_pcm16(list(range(1000)), cfg.n_samples, cfg.sr)
pcm = torch.from_numpy(np.tile(pool, (10, 1))[:10000]).to("cuda:0")
mu = torch.empty((10000, spec.latent_dim), dtype=torch.float32, device="cuda:0")
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    for _ in range(3): sess.encode(pcm, apply_rms=True, out_mu=mu)
    torch.cuda.synchronize()
    for rep in range(3):
        t0 = time.perf_counter()
        for _ in range(5): sess.encode(pcm, apply_rms=True, out_mu=mu)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        print(f"5 steps: host submission {1e3*(t1-t0):.2f} ms, until the GPU is done {1e3*(t2-t0):.2f} ms")
