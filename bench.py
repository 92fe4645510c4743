#!/usr/bin/env python3
"""Headline benchmark: 1 s / 22.05 kHz audio chunks encoded to latents per second.

    python bench.py --gpus N --steps K --warmup W

One "step" = one pass of the hot path (RMS normalise + PCM16 hand-off -> STFT/mel ->
dB/standardise -> thesis-shape VAE encoder -> mu) over the BASELINE.json configs[1]
workload: 10 000 synthetic chunks resident in HBM, processed in batches of 256, fp32.
With N > 1 every rank holds its own 10 000-chunk shard (weak scaling, one process per
GPU) and the step ends with the single RCCL all-gather that concatenates the latent
matrix.  `python bench.py --gpus N` starts its N ranks itself (one child process per GPU,
amphibian_vae_latent_detector_amd/launcher.py; the parent never touches the GPU); under
`python -m torch.distributed.run --nproc-per-node N` the ranks already exist and it just runs.
Rank 0 prints ONE JSON line.  Before the W warm-up steps come `--prewarm` (default 40)
untimed passes: the GPU idles through the set-up (the synthetic chunks are made on the host), and
its first ~100 ms of work afterwards run ~5 % slower (profiles/r03_bench_warmup_sweep.txt); the
timed region is exactly the K steps, bracketed by barrier + synchronize.  `value_cold` is the same
K steps measured BEFORE those passes (W warm-up steps only), so the two effects can be told apart.
The CPU baseline leg runs last.

Extra objects on that line:
  roofline      the dominant kernel (largest share of GPU time in a fully instrumented
                single-lane pass of the same step): `achieved` = SURVEY 8(d)'s algorithmic
                bytes per chunk (A = 2N + 4D, PCM16 in, mu out) x the chunks one launch
                processes / that kernel's mean launch duration ALONE on the GPU (HIP events
                on its launch stream, one lane: the figure `rocprofv3 --kernel-trace --stats`
                of `bench.py --lanes 1` reproduces, profiles/*_lanes1.csv), against 8 TB/s.
                `traffic` (= `traffic_from_committed_profile`) = HBM bytes per launch from the
                COMMITTED rocprofv3 PMC profile of this configuration (bench.py cannot profile
                itself), corrected as MI355X_MICROARCH.md prescribes (FETCH_SIZE x 2 + WRITE_SIZE).
                `in_timed_region` = the same kernel bracketed inside the timed region, where
                four lanes share the GPU: that figure includes queueing behind the other
                lanes' kernels and is NOT a kernel duration.
  fp32_roofline the whole path's ALGORITHMIC fp32 FLOP/s against the 157.3 TF fp32 peak.  The
                conv / head GEMMs do not run as fp32 instructions: they are two-term fp16 splits
                (x = h + 2^-11 l, three MFMAs per K = 16) on the matrix cores (fp32-accurate, fp32
                accumulate), so this is a work rate,
                not the utilisation of any one pipe - mixed_roofline is.
  mixed_roofline the bound the kernels as built can reach: matrix-pipe time of the MFMA kernels
                (committed SQ_INSTS_MFMA x 32 cycles / (1024 SIMDs x 2.4 GHz), profiles/mfma_latest.json)
                + fp32 VALU time of the others (their FLOPs at 157.3 TF), per 256-chunk batch, against
                the measured batch period; MFMA-busy share of every MFMA kernel's isolated duration.
  step_ms       per-step wall time inside the timed region (events on the stream): min / median / max.
  rccl          (N > 1 or under torch.distributed.run) what the collective saw: backend, world size as
                the process group reports it, bytes gathered per step.
  hbm_roofline  the whole path's algorithmic bytes/s against 8 TB/s - the figure
                BASELINE.json's metric asks for.
  kernels       every kernel of the step: isolated duration, algorithmic work, corrected
                HBM traffic and its ratio to the kernel's own algorithmic bytes.
  detect        BASELINE configs[4] leg: encode + radial decision on the device
                (10_benchmark_folder_detection.py:387-418 as radial.detect_chunks).
  value_f32_mfma the same steps with every GEMM layer on the true f32 MFMA kernels ("split_gemm" 0: fp32 operands,
                v_mfma_f32_32x32x2_f32) - the reference's own arithmetic beside the headline's two-term fp16 split.
  from_files    end to end from WAV files (parse on the host threads -> pinned -> HBM -> encode), files/s per
                loader thread count; pcie_inclusive: host rows through amph_encode_pcm16_host.  Never the headline.
  config4       BASELINE configs[3]: preset bench-1s-128 (n_fft 1024 / hop 256 / 128 bands), bf16 encoder, 3 steps.
  cpu_baseline  the CPU oracle driven like the reference's batch-1 loop
                (08_fit_radial_detector.py:488-503) on a bounded sample, rank 0, N = 1.

`--config3` runs BASELINE configs[2] on ONE GPU instead: 100 000 resident chunks in the
reference's list order encoded as 8 contiguous virtual shards (the data path of the
8-GPU job with the all-gather replaced by the layout it produces), then 08's host fit.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

# The encode spreads its batches over 4 HIP streams ("lanes").  A process gets 4 hardware queues by
# default and streams are mapped onto them in creation order, so once RCCL has created its own
# streams the lanes would share queues and serialise (measured: 7.3 instead of 6.1 ms/step).  Must be
# set before the HIP runtime initialises.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: what RCCL needs on this driver (already exported on the pool)

import numpy as np  # noqa: E402
import torch  # noqa: E402

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

PEAK_F32_MFMA_TFLOPS = 157.3      # MI355X_MICROARCH.md: f32-input MFMA = fp32 vector peak
IN_REGION_BRACKET_EVERY = 4     # the dominant kernel's HIP-event bracket inside the timed region: every n-th launch
PEAK_HBM_GBS = 8000.0             # HBM3E spec peak


def cpu_baseline(cfg, spec, state, pcm, budget_s):
    """Reference-shaped loop on the host cores: per chunk, batch 1, in order."""
    from oracle import encoder as oenc
    from oracle import frontend as ofe
    tenc = oenc.build_torch_encoder(spec.to_dict(), state)
    kw = cfg.mel_kwargs()

    def run(max_s):
        t0 = time.perf_counter()
        n = 0
        for q in pcm:
            y, _ = ofe.normalize_pcm16_chunk(q, True)
            oenc.encode_mel_to_latent(tenc, ofe.wave_to_mel(y, **kw))
            n += 1
            if time.perf_counter() - t0 > max_s:
                break
        return n / (time.perf_counter() - t0), n

    default_threads = torch.get_num_threads()
    run(1.0)                                         # warm caches / thread pools
    v_def, n_def = run(budget_s * 0.6)
    torch.set_num_threads(1)
    v_one, n_one = run(budget_s * 0.4)
    torch.set_num_threads(default_threads)
    # headline = the single-thread figure (cores = 1: what a scalar port uses); the reference itself never sets a thread
    # count, so the library-default run (every torch op fanned out over all cores by a batch-1 loop) is reported beside it
    return {"value": round(v_one, 2), "unit": "chunks/s", "cores": 1, "kind": "port",
            "sample": f"first {n_one} chunks of the config-2 set, batch-1 loop as 08_fit_radial_detector.py:488-503, numpy/scipy "
                      f"front end + torch-CPU encoder, torch.set_num_threads(1)",
            "value_default_threads": round(v_def, 2), "threads_default": default_threads,
            "sample_default_threads": f"{n_def} chunks at library-default threading ({default_threads} torch threads, "
                                      f"os.cpu_count()={os.cpu_count()})"}

def from_files_leg(enc, cfg, pcm_host, args, dev, world, rank, use_dist, dist):
    """End to end from FILES (SURVEY 8(d) "feed bounds"; 08_fit_radial_detector.py:461-520's per-file loop): PCM16 WAVs on local disk ->
    native RIFF reader on the host thread pool -> pinned rows -> H2D -> encode -> latents on the device.  Page cache warm (just
    written).  EVERY rank runs it on its own --files files at the same time with loader threads = CPUs in the affinity mask / ranks
    (N ranks of one node share the host: the figure a sharded folder job gets), bracketed by barriers; rank 0 reports the sum."""
    import tempfile
    from concurrent.futures import ThreadPoolExecutor
    from amphibian_vae_latent_detector_amd import folder as _folder
    from amphibian_vae_latent_detector_amd.synth import write_wav_pcm16
    n_aff = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    share = max(1, n_aff // max(1, world))

    def sync():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    with tempfile.TemporaryDirectory(prefix=f"amph_bench_r{rank}_") as td:
        paths = []
        for i in range(args.files):
            pth = Path(td) / f"chunk_{i:06d}.wav"
            write_wav_pcm16(pth, pcm_host[i % pcm_host.shape[0]], cfg.sr)
            paths.append(pth)
        sweep = sorted({t for t in (2, 4, 8, 16, 32, 64, min(share, 128)) if t <= share}) or [share]
        _folder.encode_paths(enc, paths[:min(512, len(paths))], cfg, apply_rms=True, loader_threads=share)   # warm-up
        by_threads, per_thread, failed = {}, {}, 0
        for threads in sweep:
            sync()
            t1 = time.perf_counter()
            _, st_f = _folder.encode_paths(enc, paths, cfg, apply_rms=True, loader_threads=threads)
            sync()
            dt = time.perf_counter() - t1
            if use_dist:
                t = torch.tensor([dt], dtype=torch.float64, device=dev)
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
                dt = float(t.item())
            by_threads[threads] = round(world * args.files / dt, 1)
            per_thread[threads] = round(world * args.files / dt / threads, 1)
            failed = int((st_f != 0).sum())

        # the floor under any loader: open + read + close of every file, no parsing, no GPU (same thread pool size)
        def _slurp(pth):
            with open(pth, "rb") as f:
                return len(f.read())
        sync()
        t1 = time.perf_counter()
        with ThreadPoolExecutor(max_workers=min(64, share)) as ex:
            sum(ex.map(_slurp, paths, chunksize=64))
        open_read_floor = round(args.files / (time.perf_counter() - t1), 1)
        sync()
    best_t = max(by_threads, key=by_threads.get)
    return {"value": by_threads[best_t], "unit": "files/s (all ranks)", "files_per_rank": args.files, "ranks": world,
            "loader_threads_per_rank": best_t, "files_per_s_by_loader_threads": by_threads, "files_per_s_per_thread": per_thread,
            "cpus_in_affinity_mask": n_aff,
            "cpu_share_per_rank": share, "python_open_read_floor_files_per_s_rank0": open_read_floor, "failed_rank0": failed,
            "note": "folder.encode_paths on every rank at once: amph_wav_read_many on the host thread pool (affinity / ranks threads) -> "
                    "pinned -> HBM -> encode (file-level RMS); page cache warm; max over ranks of the wall time; never the headline value"}


def detect_leg(enc, sess, cfg, pcm, mu, args, dev):
    """BASELINE configs[4]: streaming encode + radial decision (centroids + thresholds from 08's fit, q_in 0.95 /
    q_out 0.10, on this shard's own latents split round-robin into the four species), latents never leave HBM."""
    from amphibian_vae_latent_detector_amd import radial as prad
    from amphibian_vae_latent_detector_amd.synth import SPECIES
    Z = mu.cpu().numpy()
    rd = prad.fit_radial_detector({sp: Z[k::len(SPECIES)] for k, sp in enumerate(SPECIES)}, 0.95, 0.10, verbose=False)
    det = prad.RadialDetector({k: np.array(v, np.float32) for k, v in rd["centroids"].items()}, rd["thresholds"], Z.shape[1])
    sess_d = enc.session(cfg)

    def once():
        m, _, _ = sess_d.encode(pcm, apply_rms=True)
        return sess_d.decide_radial(m, det.centroids, det.thresholds, det.rank)
    once()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    reps = max(3, args.steps // 2)
    for _ in range(reps):
        pred, best = once()
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t0
    pred = pred.cpu().numpy()
    return {"value": round(reps * args.chunks / dt, 1), "unit": "chunks/s (encode + radial decision)", "reps": reps,
            "detected_frac": round(float((pred >= 0).mean()), 4),
            "note": "radial.detect_chunks' device path: amph_encode_pcm16 + amph_decide_radial (K6) per 10 000-chunk call, "
                    "the detector's centroids / thresholds are uploaded once (cached by content), one stream sync per call; decisions stay on the device"}


def config4_leg(pcm, args, dev):
    """BASELINE configs[3]: preset bench-1s-128 (n_fft 1024 / hop 256 / 128 mel bands -> 96 x 128 image), thesis-shape encoder in bf16
    (weights, activations, bf16 MFMA, fp32 accumulate), the same resident chunks; its parity bar is the relaxed one (tests: 2e-3 vs a
    bf16-emulating torch encoder), so it is reported beside the headline, never as it."""
    from amphibian_vae_latent_detector_amd import PRESETS, SPECS
    from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
    cfg4 = PRESETS["bench-1s-128"]
    spec4 = SPECS["thesis-shape"](cfg4.target_frames, cfg4.n_mels)
    enc4 = HipEncoder(spec4, random_init_state(spec4, seed=0), dev, max_batch=args.batch)
    s4 = enc4.session(cfg4)
    s4.configure("bf16_encoder", 1)
    mu4 = torch.empty((pcm.shape[0], spec4.latent_dim), dtype=torch.float32, device=dev)
    for _ in range(3):
        s4.encode(pcm, apply_rms=True, out_mu=mu4, check_range=False)
    torch.cuda.synchronize(dev)
    t1 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        s4.encode(pcm, apply_rms=True, out_mu=mu4, check_range=False)
    torch.cuda.synchronize(dev)
    dt = time.perf_counter() - t1
    out = {"value": round(reps * pcm.shape[0] / dt, 1), "unit": "chunks/s", "steps": reps, "ms_per_step": round(dt / reps * 1e3, 4),
           "dtype": "bf16 encoder (fp32 front end, fp32 accumulate)", "finite": bool(torch.isfinite(mu4).all().item()),
           "config": {"workload": f"BASELINE.json configs[3]: {pcm.shape[0]} chunks resident in HBM, preset bench-1s-128 (n_fft {cfg4.n_fft}, hop "
                                  f"{cfg4.hop_length}, {cfg4.n_mels} mel bands, image {cfg4.target_frames} x {cfg4.n_mels}), thesis-shape encoder, bf16, "
                                  f"batch {args.batch}"}}
    s4.close()
    return out


def config3_fit(mu, plan, args, sess=None):
    """08's fit on the gathered latent matrix (08_fit_radial_detector.py:310-333,525-587): q_in 0.95, q_out 0.10,
    max_per_class 0 (all chunks) and 400 (seed 123: rows of the same matrix); on the host (numpy) and on the device
    (K9, amph_fit_radial, latents stay in HBM) - the two must agree to the last bit."""
    from amphibian_vae_latent_detector_amd import radial as prad
    from amphibian_vae_latent_detector_amd.synth import SPECIES, synth_folder_plan
    Z = mu.cpu().numpy()
    out, pos, off = {}, {}, 0
    Zby = {}
    for sp in SPECIES:
        k = len(plan[sp])
        Zby[sp] = Z[off:off + k]
        for i, c in enumerate(plan[sp]):
            pos[c] = off + i
        off += k
    for mpc in (0, 400):
        t0 = time.perf_counter()
        if mpc:
            p400 = synth_folder_plan(args.chunks, mpc, 123)
            Zs = {sp: Z[[pos[c] for c in p400[sp]]] for sp in SPECIES}
        else:
            Zs = Zby
        rd = prad.fit_radial_detector(Zs, 0.95, 0.10, verbose=False)
        e = {"thresholds": {k: round(v, 6) for k, v in rd["thresholds"].items()}, "fit_seconds_host": round(time.perf_counter() - t0, 4)}
        if sess is not None:
            dev_parts = {sp: (mu[torch.as_tensor([pos[c] for c in p400[sp]], device=mu.device)] if mpc else
                              mu[pos[plan[sp][0]]:pos[plan[sp][0]] + len(plan[sp])]) for sp in SPECIES}
            prad.fit_radial_detector(dev_parts, 0.95, 0.10, verbose=False, session=sess)          # warm
            t0 = time.perf_counter()
            rd2 = prad.fit_radial_detector(dev_parts, 0.95, 0.10, verbose=False, session=sess)
            e["fit_seconds_device"] = round(time.perf_counter() - t0, 4)
            e["device_equals_host"] = rd2["thresholds"] == rd["thresholds"] and rd2["centroids"] == rd["centroids"]
        out[f"max_per_class_{mpc}"] = e
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--prewarm", type=int, default=40,
                    help="untimed passes BEFORE the W warm-up steps (not counted in W): the GPU has idled through the CPU baseline leg and the "
                         "set-up, and its first ~100 ms of work run 5 %% slower (10 steps after 2 warm-up steps: 3.47 M chunks/s, 200 after "
                         "20: 3.68 M - profiles/r03_bench_warmup_sweep.txt); 0 = none")
    ap.add_argument("--chunks", type=int, default=10000, help="chunks per GPU per step (config 2: 10k)")
    ap.add_argument("--batch", type=int, default=256, help="chunks per kernel launch (config 2: 256)")
    ap.add_argument("--preset", default="bench-1s")
    ap.add_argument("--spec", default="thesis-shape")
    ap.add_argument("--cpu-seconds", type=float, default=15.0, help="0 disables the CPU baseline leg")
    ap.add_argument("--pcie", type=int, default=-1, help="time the host-buffer (PCIe-inclusive) entry point: 1 / 0; default: on for a plain "
                                                         "one-GPU run (no --knob / --config3)")
    ap.add_argument("--files", type=int, default=-1, help="time the end-to-end folder ingest on this many WAV files (written to a temp dir "
                                                          "first; parse + H2D + encode; never the headline value); default 8000 for a plain "
                                                          "one-GPU run, 0 = off")
    ap.add_argument("--no-extra-legs", action="store_true", help="skip value_f32_mfma, config4, from_files and pcie_inclusive (profiling runs)")
    ap.add_argument("--lanes", type=int, default=0, help="HIP streams the batches are spread over (0 = library default)")
    ap.add_argument("--config3", action="store_true", help="BASELINE configs[2] on one GPU: --chunks (default 100000) resident chunks as "
                                                           "--shards contiguous virtual shards + the 08 fit on the host")
    ap.add_argument("--shards", type=int, default=8)
    ap.add_argument("--no-detect", action="store_true", help="skip the encode + radial decision leg")
    ap.add_argument("--bf16", action="store_true", help="BASELINE.json configs[3]: encoder in bf16 (weights, activations, "
                                                        "bf16 MFMA, fp32 accumulate); relaxes the parity bar to ~5e-3, "
                                                        "so it is never the default / headline configuration")
    ap.add_argument("--knob", action="append", default=[], metavar="KEY=VALUE",
                    help="amph_configure(KEY, VALUE) on the session before anything is timed (include/amph.h lists the keys); "
                         "recorded in the JSON line's config - a line with knobs is a tuning experiment, not the headline")
    args = ap.parse_args()
    if args.config3 and args.chunks == 10000:
        args.chunks = 100000

    from amphibian_vae_latent_detector_amd import launcher
    if args.gpus > 1 and not launcher.under_launcher():
        # one process per GPU, started here: nothing has touched the GPU yet (device_count() does not initialise HIP on this
        # image), the ranks are CHILD processes and this parent only forwards rank 0's JSON line and the worst exit code
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus:
            sys.exit(f"--gpus {args.gpus}: only {n_dev} GPU(s) visible to this process (torch.cuda.device_count())")
        sys.exit(launcher.launch_ranks([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit(f"WORLD_SIZE={world} but --gpus {args.gpus}")
    plain = world == 1 and not args.knob and not args.config3 and not args.no_extra_legs and args.preset == "bench-1s" \
        and args.spec == "thesis-shape" and not args.bf16
    if args.files < 0:
        args.files = 8000 if plain else 0
    if args.pcie < 0:
        args.pcie = 1 if plain else 0
    import torch.distributed as dist
    if not torch.cuda.is_available():
        sys.exit("bench.py needs an MI355X; there is no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    use_dist = world > 1 or "RANK" in os.environ          # under torch.distributed.run: always take the RCCL path
    # stdout carries exactly one JSON line: RCCL prints a banner to fd 1 when it initialises, so fd 1 points at stderr until then
    saved_stdout = os.dup(1)
    sys.stdout.flush()
    os.dup2(2, 1)
    if use_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)   # nccl == RCCL on ROCm

    from amphibian_vae_latent_detector_amd import PRESETS, SPECS
    from amphibian_vae_latent_detector_amd.encoder import HipEncoder, random_init_state
    from amphibian_vae_latent_detector_amd.synth import synth_batch_pcm16

    cfg = PRESETS[args.preset]
    spec = SPECS[args.spec](cfg.target_frames, cfg.n_mels)
    state = random_init_state(spec, seed=0)
    enc = HipEncoder(spec, state, dev, max_batch=args.batch)
    sess = enc.session(cfg)
    if args.bf16:
        sess.configure("bf16_encoder", 1)
    for kv in args.knob:
        k, v = kv.split("=")
        sess.configure(k, int(v))
    D = spec.latent_dim

    # synthetic chunks of this rank's shard (distinct per rank), resident in HBM before timing
    idx = np.arange(rank * args.chunks, (rank + 1) * args.chunks)
    n_cpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 8)
    pcm_host = synth_batch_pcm16(idx, cfg.n_samples, cfg.sr, workers=max(1, min(32, n_cpu // max(1, world))))
    pcm = torch.from_numpy(pcm_host).to(dev)
    plan = None
    if args.config3:
        # the reference's list order (08_fit_radial_detector.py:461-484): species in config order, name-sorted inside
        from amphibian_vae_latent_detector_amd import folder as _folder
        from amphibian_vae_latent_detector_amd.synth import SPECIES, synth_folder_plan
        if world != 1:
            sys.exit("--config3 is the one-GPU form of the 8-GPU job; run it with --gpus 1")
        plan = synth_folder_plan(args.chunks, 0, 123)
        flat = np.concatenate([np.asarray(plan[sp]) for sp in SPECIES if sp in plan])
        pcm = pcm[torch.from_numpy(flat).to(dev)].contiguous()
    mu = torch.empty((args.chunks, D), dtype=torch.float32, device=dev)
    gathered = torch.empty((world * args.chunks, D), dtype=torch.float32, device=dev) if use_dist else None

    def step():
        if args.config3:
            mu.copy_(_folder.encode_resident_shards(sess, pcm, args.shards, apply_rms=True))
            return
        sess.encode(pcm, apply_rms=True, out_mu=mu, check_range=False)     # (range flag read after the timed region)
        if use_dist:
            # the one exchange step: concatenate the latent matrix.  (Issued asynchronously, so that it ran under the next step's
            # encode on two alternating buffers, it cost more than it hid: 1.50 M against 1.95 M chunks/s on one GPU under
            # torch.distributed.run - RCCL's kernel beside the four lanes - so it stays ordered behind its own step.)
            dist.all_gather_into_tensor(gathered, mu)

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize(dev)

    # warm-up.  The first warm-up step runs on ONE lane (kernels back to back, nothing overlapped)
    # with every stage instrumented: it names the dominant kernel and gives each kernel's
    # isolated launch duration.  The remaining warm-up steps run as the timed region does.
    sess.configure("lanes", 1)
    sess.encode(pcm[:args.batch * 2], apply_rms=True)             # first-touch of the workspaces
    sess.profile_enable(True)
    step()
    torch.cuda.synchronize(dev)
    stages = sess.profile_read()
    sess.profile_enable(False)
    sess.profile_reset()
    dom = max(range(len(stages)), key=lambda i: stages[i]["total_ms"])
    n_lanes = args.lanes if args.lanes > 0 else 4
    sess.configure("lanes", n_lanes)

    def timed(k):
        """k steps bracketed by barrier + synchronize on both sides; max over ranks; -> seconds"""
        fence()
        t_a = time.perf_counter()
        for _ in range(k):
            step()
        fence()
        dt_ = time.perf_counter() - t_a
        if use_dist:
            t_ = torch.tensor([dt_], dtype=torch.float64, device=dev)
            dist.all_reduce(t_, op=dist.ReduceOp.MAX)
            dt_ = float(t_.item())
        return dt_

    # value_cold: W - 1 warm-up steps + the K steps with NO pre-warm passes - the GPU has idled for seconds while the host made the
    # synthetic chunks (what `--prewarm 0` measures as the headline)
    for _ in range(max(1, args.warmup - 1)):
        step()
    cold_s = timed(args.steps)
    # the dominant stage's event bracket is on from here: the pre-warm and warm-up passes run exactly as the timed steps do, and the timing
    # events the timed region needs (two per batch) exist before it starts - its 1 600 hipEventCreate calls used to sit inside it
    sess.profile_enable(True, only_stage=dom)
    sess.configure("profile_every", IN_REGION_BRACKET_EVERY)
    for _ in range(max(0, args.prewarm)):              # see --prewarm: brings the GPU out of its idle state; W below is untouched
        step()
    for _ in range(max(1, args.warmup - 1)):
        step()
    torch.cuda.synchronize(dev)
    sess.profile_reset()                               # (events back to the pool, counters to zero)
    if args.prewarm + args.warmup - 1 < args.steps:    # (a short pre-warm: grow the pool to what the timed region takes)
        for _ in range(args.steps - max(0, args.prewarm) - max(1, args.warmup - 1)):
            step()
        torch.cuda.synchronize(dev)
        sess.profile_reset()

    # per-step completion events on the caller's stream (no synchronisation inside the timed region: the lanes join that stream
    # at the end of every encode call)
    evs = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    fence()
    t0 = time.perf_counter()
    evs[0].record()
    for i in range(args.steps):
        step()
        evs[i + 1].record()
    fence()
    elapsed = time.perf_counter() - t0
    step_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(args.steps)]
    if use_dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if rank == 0 and not torch.equal(gathered[:args.chunks], mu):
            sys.exit("all-gather did not return this rank's latents in place 0")
    prof = sess.profile_read()[dom]
    sess.profile_enable(False)
    sess.configure("profile_every", 1)
    range_overflow = sess.range_overflow()             # the flag the two-term fp16 kernels raised inside the timed region, if any (include/amph.h)
    files_leg = from_files_leg(enc, cfg, pcm_host, args, dev, world, rank, use_dist, dist) if args.files > 0 else None

    # value_f32_mfma: the same step with every GEMM layer on the true f32-MFMA kernels (fp32 operands, v_mfma_f32_32x32x2_f32: the
    # reference's own arithmetic) - the headline's two-term fp16 split sits beside it in one record
    f32_leg = None
    if plain and not args.bf16:
        mu_split = mu.clone()
        sess.configure("split_gemm", 0)
        sess.configure("split_fused", 0)
        step()
        k32 = max(3, min(5, args.steps))
        dt32 = timed(k32)
        d = (mu - mu_split).norm(dim=1) / mu_split.norm(dim=1).clamp_min(1e-12)
        f32_leg = {"value": round(world * args.chunks * k32 / dt32, 1), "unit": "chunks/s", "steps": k32,
                   "ms_per_step": round(dt32 / k32 * 1e3, 4),
                   "max_rel_l2_vs_headline_latents": float(d.max().item()),
                   "note": "amph_configure split_gemm 0 / split_fused 0: conv2..conv4 and the head as f32-input MFMA kernels "
                           "(k_conv_gemm_f3 / _p3, v_mfma_f32_32x32x2_f32, fp32 operands end to end); same workload, lanes and batch"}
        sess.configure("split_gemm", 1)
        sess.configure("split_fused", 1)
        step()
        torch.cuda.synchronize(dev)

    if rank == 0:
        peak_mfma = 2500.0 if args.bf16 else PEAK_F32_MFMA_TFLOPS      # dense bf16 MFMA peak / f32-input MFMA peak
        total_chunks = world * args.chunks * args.steps
        value = total_chunks / elapsed
        per_gpu = value / world
        alg_bytes = 2 * cfg.n_samples + 4 * D                          # A = 2N + 4D per chunk (PCM16 in, mu out)
        alg_flops = sum(s["flops_per_chunk"] for s in stages)          # 33.4 MFLOP at bench-1s / thesis-shape

        def is_mfma(name):
            return (name.startswith("conv") and name != "conv1") or name == "head_gemm"

        # HBM-side bytes per launch from the committed PMC profile of this same configuration (bench.py cannot run
        # rocprofv3 on itself), corrected as MI355X_MICROARCH.md (HBM) prescribes: FETCH_SIZE x 2 + WRITE_SIZE
        traffic, traffic_src = {}, None
        if args.batch == 256 and args.preset == "bench-1s" and args.spec == "thesis-shape" and not args.bf16:
            try:
                tj = json.load(open(ROOT / "profiles" / "traffic_latest.json"))
                traffic_src = tj.get("_source", "profiles/traffic_latest.json")
                # the profile is keyed by KERNEL; a stage that launches two kernels (K0: chunk sums, then fold + apply) owns both
                alias = {"rms_apply": "rms_scale", "rms_fold_apply": "rms_scale"}
                not_on_path = ("k_decide_radial", "conv4_wide")
                for k, e in tj["kernels"].items():
                    if k in not_on_path:
                        continue
                    k = alias.get(k, k)
                    traffic[k] = traffic.get(k, 0) + int(e["fetch_kb"] * 1024 * 2 + e["write_kb"] * 1024)
            except (OSError, ValueError, KeyError):
                pass
        rocprof_iso, rocprof_grid = {}, {}
        try:                                                            # rocprofv3 --kernel-trace --stats of `bench.py --lanes 1`
            rj = json.load(open(ROOT / "profiles" / "kernel_durations_latest.json"))
            if args.batch == 256 and args.preset == "bench-1s" and not args.bf16:
                rocprof_iso, rocprof_grid = rj.get("avg_us", {}), rj.get("grid", {})
        except (OSError, ValueError):
            pass

        kernels = []
        for s_ in stages:
            if not s_["launches"]:
                continue
            ms_i = s_["total_ms"] / s_["launches"]
            cpl = args.chunks / s_["launches"]                          # chunks one launch processes (isolated pass: one step)
            e = {"kernel": s_["stage"], "ms_per_launch": round(ms_i, 5), "chunks_per_launch": round(cpl, 1),
                 "own_algorithmic_bytes_per_launch": int(s_["bytes_per_chunk"] * cpl),
                 "flops_per_launch": int(s_["flops_per_chunk"] * cpl)}
            tfl = s_["flops_per_chunk"] * cpl / (ms_i * 1e-3) / 1e12
            if is_mfma(s_["stage"]):
                e.update({"bound": "mfma", "achieved_TFLOPs": round(tfl, 2), "frac": round(tfl / peak_mfma, 4)})
            else:
                gbs = s_["bytes_per_chunk"] * cpl / (ms_i * 1e-3) / 1e9
                e.update({"bound": "hbm", "achieved_GBps": round(gbs, 1), "frac": round(gbs / PEAK_HBM_GBS, 5),
                          "valu_TFLOPs": round(tfl, 2), "valu_frac_of_fp32_peak": round(tfl / PEAK_F32_MFMA_TFLOPS, 4)})
            if s_["stage"] in traffic:
                e["traffic_bytes_per_launch"] = traffic[s_["stage"]]
                e["traffic_over_own_algorithmic"] = round(traffic[s_["stage"]] / max(1.0, s_["bytes_per_chunk"] * args.batch), 2)
            if s_["stage"] in rocprof_iso:
                e["rocprofv3_avg_us"] = rocprof_iso[s_["stage"]]
            kernels.append(e)

        # ---- roofline of the dominant kernel: isolated duration (one lane, nothing beside it)
        iso_ms = stages[dom]["total_ms"] / max(1, stages[dom]["launches"])
        iso_cpl = args.chunks / max(1, stages[dom]["launches"])
        name = stages[dom]["stage"]
        ach = alg_bytes * iso_cpl / (iso_ms * 1e-3) / 1e9
        roof = {"kernel": name, "bound": "hbm", "achieved": round(ach, 2), "peak": PEAK_HBM_GBS, "unit": "GB/s",
                "frac": round(ach / PEAK_HBM_GBS, 5),
                "traffic": traffic.get(name), "traffic_from_committed_profile": traffic.get(name),
                "algorithmic_bytes_per_launch": int(alg_bytes * iso_cpl),
                "avg_launch_ms": round(iso_ms, 5), "launches": int(stages[dom]["launches"]), "chunks_per_launch": round(iso_cpl, 1),
                "measured": "HIP events on the launch stream, 1 lane (kernel alone on the GPU), one whole step",
                "share_of_gpu_time": round(stages[dom]["total_ms"] / max(1e-9, sum(s_["total_ms"] for s_ in stages)), 3)}
        if traffic_src:
            roof["traffic_source"] = traffic_src
            if roof["traffic"]:
                roof["traffic_over_algorithmic"] = round(roof["traffic"] / max(1, alg_bytes * args.batch), 2)
        if name in rocprof_iso:
            roof["rocprofv3_avg_launch_ms"] = round(rocprof_iso[name] / 1e3, 5)
        if name in rocprof_grid:
            # the dominant kernel may occupy only part of the chip by design (round 4: k_stft_pair runs two rows per workgroup on 128 of the
            # 256 CUs - one 768-thread, LDS-filling workgroup per CU - and the other lanes' kernels run beside it): `frac` above prices its
            # launch DURATION against the whole chip's peak, as the contract defines it; frac_on_occupied_cus divides by the share of the
            # CUs the launch holds, the figure that is comparable with a one-workgroup-per-CU launch of earlier rounds
            wg_ = rocprof_grid[name]
            share = min(1.0, wg_["workgroups"] / 256.0) if wg_["threads"] >= 512 else 1.0
            roof["workgroups_per_launch"] = wg_["workgroups"]
            roof["cu_share"] = round(share, 4)
            roof["frac_on_occupied_cus"] = round(ach / PEAK_HBM_GBS / share, 5)
        tfl = stages[dom]["flops_per_chunk"] * iso_cpl / (iso_ms * 1e-3) / 1e12
        if is_mfma(name):
            roof["compute"] = {"bound": "mfma", "achieved_TFLOPs": round(tfl, 2), "peak": peak_mfma, "frac": round(tfl / peak_mfma, 4)}
        else:
            roof["compute"] = {"bound": "fp32 valu", "achieved_TFLOPs": round(tfl, 2), "peak": PEAK_F32_MFMA_TFLOPS,
                               "frac": round(tfl / PEAK_F32_MFMA_TFLOPS, 4)}
        launches = max(1, prof["launches"])
        roof["in_timed_region"] = {"avg_bracket_ms": round(prof["total_ms"] / launches, 5), "launches": int(prof["launches"]),
                                   "bracketed": "every %d-th launch of the kernel inside the timed region (two event records each; "
                                                "every launch costs ~2 %% of the headline)" % IN_REGION_BRACKET_EVERY,
                                   "concurrent_lanes": n_lanes,
                                   "note": "event bracket around the same kernel while the other lanes' kernels share the GPU: "
                                           "queue-inclusive, not a kernel duration"}
        # ---- mixed roofline: what the kernels AS BUILT are bounded by (matrix pipe for the two-term fp16 split GEMMs, fp32 VALU for the rest)
        mixed = None
        try:
            mj = json.load(open(ROOT / "profiles" / "mfma_latest.json"))
            if args.batch == 256 and args.preset == "bench-1s" and args.spec == "thesis-shape" and not args.bf16:
                SIMDS, CLK = 1024, 2.4e9
                per_k, pipe_us, valu_us = [], 0.0, 0.0
                for s_ in stages:
                    if not s_["launches"]:
                        continue
                    iso_us = 1e3 * s_["total_ms"] / s_["launches"]
                    cplk = args.chunks / s_["launches"]
                    m = mj["kernels"].get(s_["stage"], {})
                    n_mfma = m.get("mfma_insts", 0.0) * cplk / 256.0           # committed count is per 256-chunk launch
                    if n_mfma > 0:
                        t_us = n_mfma * 32.0 / (SIMDS * CLK) * 1e6
                        pipe_us += t_us * 256.0 / cplk
                        busy_c = (round(m.get("mfma_busy_cycles", 0.0) / SIMDS / max(1.0, m.get("gui_active_sum", 0.0) / 8.0), 3)
                                  if m.get("gui_active_sum") else None)
                        pk = {"kernel": s_["stage"], "mfma_insts_per_launch": int(n_mfma), "matrix_pipe_us": round(t_us, 2),
                              "isolated_us": round(iso_us, 2), "mfma_busy_frac_of_isolated": round(t_us / iso_us, 3),
                              "mfma_busy_frac_counter": busy_c}
                        g_ = rocprof_grid.get(s_["stage"])
                        if g_ and busy_c is not None:
                            # a persistent kernel holds only part of the chip by design (round 4: conv1+2 and conv3 on 64 LDS-filling workgroups =
                            # a quarter of the CUs; the other lanes' kernels run beside them): the counter above averages over all 1024 SIMDs
                            share_ = min(1.0, g_["workgroups"] / (256.0 * (1 if g_["threads"] >= 512 else 2)))
                            pk["cu_share"] = round(share_, 4)
                            pk["mfma_busy_frac_on_occupied_cus"] = round(busy_c / share_, 3)
                        per_k.append(pk)
                    else:
                        t_us = s_["flops_per_chunk"] * cplk / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e6
                        valu_us += t_us * 256.0 / cplk
                        per_k.append({"kernel": s_["stage"], "fp32_valu_us_at_peak": round(t_us, 2), "isolated_us": round(iso_us, 2),
                                      "valu_frac_of_isolated": round(t_us / iso_us, 3)})
                batch_period_us = elapsed / args.steps / (args.chunks / 256.0) * 1e6
                mixed = {"matrix_pipe_us_per_batch": round(pipe_us, 2), "fp32_valu_us_per_batch": round(valu_us, 2),
                         "bound_us_per_batch": round(pipe_us + valu_us, 2), "measured_batch_period_us": round(batch_period_us, 2),
                         "frac": round((pipe_us + valu_us) / batch_period_us, 4),
                         "matrix_pipe_busy_over_timed_region": round(pipe_us / batch_period_us, 4),
                         "kernels": per_k, "source": mj.get("_source", "profiles/mfma_latest.json"),
                         "note": "GEMM kernels: v_mfma_f32_32x32x16_f16 on two-term fp16 splits of fp32 operands (x = h + 2^-11 l: 3 MFMAs per K = 16, fp32 "
                                 "accumulate): 32 matrix-pipe cycles each on one of 1024 SIMDs at 2.4 GHz; other kernels: algorithmic fp32 FLOPs at "
                                 "the 157.3 TF vector peak; batch period = ms_per_step / (chunks / 256)"}
        except (OSError, ValueError, KeyError):
            pass
        srt = sorted(step_ms)
        line = {
            "metric": "1 s audio chunks encoded to latent/sec", "value": round(value, 1), "unit": "chunks/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "value_cold": round(total_chunks / cold_s, 1),
            "fp16_split_range_overflow": bool(range_overflow),
            "dtype": "bf16 encoder (fp32 front end, fp32 accumulate)" if args.bf16 else
                     "f32 (results fp32-accurate; conv / head GEMMs as two-term fp16 splits, x = h + 2^-11 l, on the f16 MFMA with fp32 accumulate; front end fp32 VALU)",
            "data": "synthetic",
            "step_ms": {"min": round(srt[0], 4), "median": round(srt[len(srt) // 2], 4), "max": round(srt[-1], 4), "n": len(srt),
                        "measured": "events on the caller's stream after every step, read after the closing barrier"},
            "config": {"workload": (f"BASELINE.json configs[2] on one GPU: {args.chunks} synthetic {cfg.duration:g} s / {cfg.sr} Hz PCM16 chunks "
                                    f"resident in HBM in the reference's list order, encoded as {args.shards} contiguous virtual shards laid out "
                                    f"as the all-gather lays them out, " if args.config3 else
                                    f"BASELINE.json configs[1]: {args.chunks} synthetic {cfg.duration:g} s / {cfg.sr} Hz PCM16 chunks per GPU "
                                    f"resident in HBM, ") +
                                   f"preset {args.preset}, {args.spec} VAE encoder (latent_dim {D}), "
                                   f"batch {args.batch}, RMS normalise + mel + encoder, " + ("bf16 encoder" if args.bf16 else "fp32"),
                       "chunks_per_gpu": args.chunks, "batch": args.batch, "lanes": n_lanes, "prewarm_steps": max(0, args.prewarm),
                       **({"knobs": list(args.knob)} if args.knob else {}),
                       "parallelism": f"dp{world} (chunk shards, 1 all-gather)" if not args.config3 else f"1 GPU, {args.shards} virtual shards"},
            "roofline": roof,
            "fp32_roofline": {"algorithmic_flops_per_chunk": int(alg_flops), "achieved_TFLOPs_per_gpu": round(per_gpu * alg_flops / 1e12, 2),
                              "peak_TFLOPs": PEAK_F32_MFMA_TFLOPS, "frac": round(per_gpu * alg_flops / 1e12 / PEAK_F32_MFMA_TFLOPS, 4),
                              "note": "ALGORITHMIC fp32 FLOP/s against the f32 peak; the GEMM kernels run two-term fp16 splits on the f16 matrix cores "
                                      "(v_mfma_f32_32x32x16_f16; mixed_roofline prices them): a work rate, not a pipe utilisation.  The path is compute-bound "
                                      "(algorithmic intensity ~750 FLOP/B), not HBM-bound"},
            **({"mixed_roofline": mixed} if mixed else {}),
            "hbm_roofline": {"algorithmic_bytes_per_chunk": alg_bytes, "achieved_GBps_per_gpu": round(per_gpu * alg_bytes / 1e9, 2),
                             "peak_GBps": PEAK_HBM_GBS, "frac": round(per_gpu * alg_bytes / 1e9 / PEAK_HBM_GBS, 6)},
            "kernels": kernels,
        }
        if use_dist:
            line["rccl"] = {"backend": dist.get_backend(), "world_size": dist.get_world_size(),
                            "all_gather_bytes_per_step": int(world * args.chunks * D * 4), "collectives_per_step": 1,
                            "rank0_block_equals_own_latents": True,      # (checked above: a mismatch exits)
                            "note": "dist.all_gather_into_tensor of the [chunks, D] fp32 latent matrix, inside the timed step"}
        if traffic:
            tot = sum(traffic.values())                   # every kernel of the batch (both K0 kernels: the alias above)
            line["path_traffic"] = {"bytes_per_batch": int(tot), "algorithmic_bytes_per_batch": int(alg_bytes * args.batch),
                                    "ratio": round(tot / (alg_bytes * args.batch), 2), "source": traffic_src,
                                    "by_stage": {k: int(v) for k, v in sorted(traffic.items())},
                                    "note": "sum over profiles/traffic_latest.json (FETCH_SIZE x 2 + WRITE_SIZE per kernel of one 256-chunk "
                                            "batch) against 256 x A; from the committed rocprofv3 PMC passes, not measured in this run"}
        if world == 1 and not args.no_detect and not args.config3:
            line["detect"] = detect_leg(enc, sess, cfg, pcm, mu, args, dev)
        if args.config3:
            line["config3_fit"] = config3_fit(mu, plan, args, sess)
        if f32_leg:
            line["value_f32_mfma"] = f32_leg["value"]
            line["f32_mfma"] = f32_leg
        if plain:
            sess.configure("lanes", 1)                  # (frees the main context's three lane streams: the second context's lanes get hardware queues of their own)
            line["config4"] = config4_leg(pcm, args, dev)
            sess.configure("lanes", n_lanes)
        if world == 1 and args.pcie:
            # host-resident PCM16 -> (pinned staging ->) H2D on a copy stream overlapped with compute -> latents back.  Two forms of
            # the caller's buffer: pageable rows (staged through the library's pinned double buffer) and rows the caller already
            # keeps in pinned memory (copied to the device straight from there).  Bound: 63 GB/s PCIe Gen5 x16 = 1.43 M chunks/s.
            from amphibian_vae_latent_detector_amd import wavio as _wavio
            sess.encode_host(pcm_host[:args.batch * 4], apply_rms=True)
            t1 = time.perf_counter()
            for _ in range(3):
                sess.encode_host(pcm_host, apply_rms=True)
            v_pageable = 3 * args.chunks / (time.perf_counter() - t1)
            with torch.cuda.device(dev):
                pinned = _wavio.PINNED.rows(7, args.chunks, cfg.n_samples)
            pinned[:] = pcm_host
            mu_p, _ = sess.encode_host(pinned, apply_rms=True)
            t1 = time.perf_counter()
            for _ in range(3):
                mu_p, _ = sess.encode_host(pinned, apply_rms=True)
            v_pinned = 3 * args.chunks / (time.perf_counter() - t1)
            line["pcie_inclusive"] = {"value": round(max(v_pageable, v_pinned), 1), "unit": "chunks/s",
                                      "pageable_rows": round(v_pageable, 1), "pinned_rows": round(v_pinned, 1),
                                      "pcie_bound_chunks_per_s": round(63e9 / (2 * cfg.n_samples), 1),
                                      "pinned_equals_resident": bool(np.array_equal(mu_p, mu.cpu().numpy())),
                                      "note": "amph_encode_pcm16_host: host rows -> HBM on a copy stream under the encode, latents copied "
                                              "back; pageable rows go through a pinned double buffer (one memcpy pass on a host core), "
                                              "pinned rows are copied straight from the caller's memory; never the headline value"}
        if files_leg is not None:
            line["from_files"] = files_leg
        if world == 1 and args.cpu_seconds > 0:
            line["cpu_baseline"] = cpu_baseline(cfg, spec, state, pcm_host[:4000], args.cpu_seconds)
        sys.stdout.flush()
        os.dup2(saved_stdout, 1)                        # the JSON line is the only thing written to the real stdout
        print(json.dumps(line), flush=True)
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
