"""Does creating the lane streams AFTER the RCCL communicator cost concurrency?  (development probe)"""
import os, sys, time, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
import numpy as np
from amphibian_vae_latent_detector_amd import PRESETS, SPECS
from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
first = sys.argv[1]
cfg = PRESETS["bench-1s"]; spec = SPECS["thesis-shape"](64, 64)
def mk():
    return HipEncoder(spec, random_init_state(spec, 0), dev, max_batch=256).session(cfg)
if first == "dist":
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    sess = mk()
else:
    sess = mk()
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
pool = synth_batch_pcm16(list(range(500)), cfg.n_samples, cfg.sr)
pcm = torch.from_numpy(np.tile(pool, (20, 1))).to(dev)
mu = torch.empty((10000, 32), device=dev)
def run(n=10):
    for _ in range(3): sess.encode(pcm, apply_rms=True, out_mu=mu)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): sess.encode(pcm, apply_rms=True, out_mu=mu)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
for lanes in (1, 2, 4, 6):
    sess.configure("lanes", lanes)
    print(f"first={first} GPU_MAX_HW_QUEUES={os.environ.get('GPU_MAX_HW_QUEUES')} lanes={lanes}: {run():.3f} ms/step")
dist.destroy_process_group()
