/*
 * amph.h — C-ABI of the MI355X (gfx950) chunk -> latent encoder.
 *
 * The reference (vpobleteacustica/amphibian-vae-latent-detector) has no FFI,
 * plugin or operator layer: its seam for this path is the Python function
 * surface of latent_space_exploration/map_detector_core.py.  Each entry point
 * below names the reference interface it replaces (file:line, relative to the
 * reference repository).  INTEGRATION.md shows the ctypes binding a reference
 * maintainer would add.
 *
 * Conventions
 *   - plain C types only; all `d_*` pointers are DEVICE (HBM) pointers, `h_*`
 *     pointers are host pointers; buffers are caller-owned.
 *   - every call returns an amph_status; amph_last_error(ctx) gives the text.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream);
 *     device entry points are stream-ordered and return without synchronising.
 *   - one context per device and per host thread (thread-compatible).
 */
#ifndef AMPH_H_
#define AMPH_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define AMPH_ABI_VERSION 1
#define AMPH_MAX_CONV_LAYERS 8

typedef enum amph_status {
    AMPH_OK = 0,
    AMPH_ERR_INVALID_ARG = 1,   /* bad pointer / size / unsupported configuration */
    AMPH_ERR_NO_DEVICE = 2,     /* no gfx950 device, or HIP runtime failure at create */
    AMPH_ERR_HIP = 3,           /* a HIP call failed; see amph_last_error */
    AMPH_ERR_IO = 4,            /* WAV file unreadable / not PCM16 / wrong sample rate */
    AMPH_ERR_UNSUPPORTED = 5    /* valid request outside what the kernels implement */
} amph_status;

/* Mel front-end parameters: the keyword set of wav_to_mel / encode_wav_to_latent
 * (map_detector_core.py:198-209, 241-254). */
typedef struct amph_frontend_config {
    int32_t sr;
    int32_t n_samples;      /* int(sr * duration), map_detector_core.py:213 */
    int32_t n_fft;
    int32_t hop_length;
    int32_t n_mels;
    float fmin;
    float fmax;
    int32_t target_frames;
    int32_t pad_mode;       /* 0 = reflect (librosa 0.9.2 stft default), 1 = constant */
} amph_frontend_config;

typedef struct amph_conv_layer {
    int32_t c_in, c_out, kernel, stride, pad;
} amph_conv_layer;

/* Encoder = n_conv x [conv2d + BatchNorm2d(eval) + ReLU] on [B,1,T,M], flatten in
 * NCHW order, two linear heads (mu, logvar).  Stands in for the module the
 * reference instantiates at map_detector_core.py:175-179 (not in its repo). */
typedef struct amph_encoder_spec {
    int32_t in_frames;      /* H = target_frames */
    int32_t in_mels;        /* W = n_mels        */
    int32_t n_conv;
    amph_conv_layer conv[AMPH_MAX_CONV_LAYERS];
    float bn_eps;
    int32_t latent_dim;
} amph_encoder_spec;

typedef struct amph_ctx amph_ctx;

/* -------------------------------------------------------------------------
 * Lifetime.  Replaces load_encoder (map_detector_core.py:150-179): weights are
 * handed over as one flat fp32 blob in this order, PyTorch layouts:
 *   per conv layer i: weight[c_out][c_in][k][k], bias[c_out],
 *                     bn_weight[c_out], bn_bias[c_out], bn_running_mean[c_out],
 *                     bn_running_var[c_out]
 *   fc_mu.weight[D][flat], fc_mu.bias[D], fc_logvar.weight[D][flat], fc_logvar.bias[D]
 * `h_weights` may be NULL to build a front-end-only context (spec may be NULL too).
 * `max_batch` sizes the per-batch workspace (chunks processed per kernel launch).
 * ------------------------------------------------------------------------- */
int amph_abi_version(void);
int amph_create(const amph_frontend_config* cfg, const amph_encoder_spec* spec,
                const float* h_weights, size_t n_weight_floats,
                int device, int max_batch, amph_ctx** out_ctx);
void amph_destroy(amph_ctx* ctx);
const char* amph_last_error(const amph_ctx* ctx);   /* ctx may be NULL: create-time error */

/* Tuning knobs (all have working defaults):
 *   "lanes"            1..8  HIP streams + workspaces the encode entry points spread their
 *                            batches over (default 4; lane 0 is the caller's stream, the
 *                            others are forked from / joined to it with events).  A process has 4
 *                            hardware queues unless GPU_MAX_HW_QUEUES says otherwise (read by the ROCm
 *                            runtime when HIP initialises): export GPU_MAX_HW_QUEUES=8 when other streams
 *                            (RCCL, copies) run beside the lanes; the Python binding defaults it on import
 *   "rms_numpy2"       0/1   scalar arithmetic of rms_normalize: 0 (default) = numpy 1.26.4, the reference's
 *                            pin (requirements-thesis-baseline-macos-arm64.txt:46): np.float32 scalar <op>
 *                            Python float in float64 - gate `rms < 1e-4` and scale `0.05 / (rms + 1e-8)`
 *                            evaluated in double, the scale cast to float32 once; 1 = numpy 2.x (NEP 50):
 *                            float32 throughout.  The scales differ by 1 ulp on about a third of all chunks,
 *                            which the PCM16 re-quantisation behind it turns into whole LSBs
 *   "split_gemm"       0/1   GEMM-shaped layers behind the fused pair (and the head) on the f16 matrix cores with two-term
 *                            split operands: x = h + 2^-11 l (h, l fp16: 22 significand bits), three v_mfma_f32_32x32x16_f16
 *                            per K = 16, fp32 accumulation - as close to the torch-CPU fp32 result as the f32-MFMA kernels
 *                            (1e-6), which 0 selects; results differ from those in the last bits (default 1)
 *   "split_fused"      0/1   ... the fused conv1 + conv2 kernel too (default 1)
 *   "rms_chunk_path_over" 1.. PCM16 rows of more than this many numpy chunks (8 192 samples) are normalised by the (row, chunk) + (row, tile)
 *                            kernel pair instead of one workgroup per row (default 2; same bits)
 *   "patch_fused"      0/1   ... in its patch-resident form where the shape allows (3x3 stride-2 pad-1 twice, 32 layer-1 channels,
 *                            conv2 output 16 wide): the layer-1 activation of a 64-pixel tile is built once in LDS (layer 1 itself
 *                            on the matrix cores, split operands) and the conv2 weights stay in registers (default 1; 0 = the
 *                            staged kernel; the two agree to fp32 rounding, ~7e-7)
 *   "patch_gemm"       0..3  bit 0: conv3 (16 x 16 x 64 -> 8 x 8 x 128), bit 1: conv4 (8 x 8 x 128 -> 4 x 4 x 256) of the thesis-shape
 *                            encoder as k_conv_gemm_g4: a workgroup's slice of the split weights stays in registers, the tile's input
 *                            image(s) are split once into an LDS patch, no barrier in the multiply loop.  Default 1 (conv4 in that form
 *                            is faster alone and slower beside the other lanes' kernels); other shapes use k_conv_gemm_s3.  Same 1e-6
 *                            agreement with the fp32 encoder; last bits differ (K is summed in 4 / 8 parts)
 *   "patch_gemm_bf16"  0..3  the same two bits for the bf16 encoder ("bf16_encoder" 1): bf16 activations copied into the patch, bf16 weights
 *                            in 36 registers, one MFMA per k-step (default 3)
 *   "bf16_pair"        0/1   bf16 encoder: conv1 + conv2 in bf16 proper (k_conv_gemm_f4b: image, weights and the layer-1 activation
 *                            rounded to bf16, fp32 accumulate / BN; default 1) instead of the fp32-accurate pair with a bf16 epilogue
 *   "patch_gemm_wgs"   0..   workgroups of that kernel (0 = four tiles each, at most half the CUs; results do not depend on it)
 *   "wide_gemm"        -1/0/1  staged split GEMM layers on 128 x 128 tiles (a quarter of the workgroups, each longer: less CU-time per batch, more
 *                      latency alone; bit-identical results).  -1 (default) = when the call's batches run on several lanes; 0 / 1 = never / always
 *   "patch_wgs"        0..   persistent workgroups of the "patch_fused" kernel (0 = eight tiles each, at most half the CUs; results do
 *                            not depend on it)
 *   "stft_fuse_db"     0/1   power_to_db + z-score + crop / pad inside the STFT kernel (one workgroup per row, the row's mel
 *                            power stays in LDS) when it fits; 0 = the separate kernel (default 1; same image up to the last
 *                            bit of the float32 mean in a few rows)
 *   "stft_waves"       4/12  waves per STFT workgroup: 12 = one 768-thread workgroup per CU (three waves
 *                            per SIMD; the default when its LDS image fits 160 KB), 4 = two 256-thread ones
 *   "frames_per_block" 4..32 STFT frames per workgroup tile (multiple of 4; default 12 with 12 waves
 *                            at n_fft = 2048, 24 at n_fft = 1024, 8 with 4 waves)
 *   "stft_indep"       0/1   frame-per-wave STFT kernel with independent waves (default 1: every wave fetches its own frame, no
 *                            workgroup barrier per tile) or the tile-synchronous one with a shared sample span in LDS
 *                            (0); the two agree to the last bit or two (8e-8 relative)
 *   "stft_pair"        0/1   PCM16 rows padded by K0 / the pad kernel ("stft_prenorm" + "stft_prepad") at n_fft 2048 / hop 384 or
 *                            n_fft 1024 / hop 256 go through the pair kernels (default 1): two consecutive frames as ONE complex FFT -
 *                            k_stft_pair (a pair per wave, 32 points per lane) / k_stft_pair2 (a pair per half wave).  Mel power
 *                            within 5e-7 of the row's peak of the frame-per-wave kernels' (0 selects those)
 *   "stft_pair_waves"  8..12 waves per workgroup of that kernel (8, 10, 11, 12 at n_fft 2048; 8 or 12 at 1024); amph_create picks
 *                            the largest whose LDS image fits; 10, 11 and 12 give identical bits, the 8-wave build differs
 *                            from them in the last bits (3e-7 of a row's peak mel power: another FMA contraction)
 *   "stft_pair_tw_lds" 0/1   its stage-1 twiddles in LDS when the image still fits 160 KB (default 1), else buffer loads
 *   "stft_k0_fuse"     0/1   PCM16 rows that take k_stft_pair at 12 waves (n_fft 2048 / hop 384) or k_stft_pair2's 12-wave form with the dB stage on
 *                            register-resident mel power (n_fft 1024 / hop 256), an even row length; with or without the RMS stage:
 *                            K0's second kernel (fold + apply) and its padded normalised copy of every row move INTO the STFT kernel - it folds
 *                            the per-chunk sums of squares (K0's first kernel: still a launch) into the scale, reads the raw rows (centre padding,
 *                            fix_length and truncation as index arithmetic) and re-quantises in its decode (default 1; 0 = K0's two launches +
 *                            the padded rows; same bits; profiles/r04_k0_fuse.txt)
 *   "stft_pair_rows2"  0/1   ... and, when a batch has more rows than half the CUs, two rows per workgroup: one queue over the pairs of both rows (a
 *                            one-second row's 29 pairs on 12 waves leave the third round 5 / 12 busy; 58 pairs are four full rounds + 10), a mel-power
 *                            image per row in LDS, the tables staged once for both (default 1; same bits)
 *   "stft_pair_cq"     0..64 ... or, > 1: that many rows per workgroup on ONE continuous pair queue - a row's dB stage is run by the wave that
 *                            completes its last pair while the others are on the next row's pairs, no workgroup barrier behind the tables'
 *                            (default 0: 12 % less CU-time but a longer launch - nothing end to end at a 256-row batch, +1 - 3 % from 512 rows;
 *                            same bits; profiles/r04_stft_cq.txt)
 *   "k0_call_sums"     0/1   K0's chunk sums for all rows of a call in one launch in front of its batches (default 0: -1.4 %; same bits)
 *   "stft_prenorm"     0/1   PCM16 input with apply_rms: the RMS kernel also writes the normalised, re-quantised PCM16
 *                            rows (what 00_normalize_dataset_rms.py puts on disk) and the STFT kernel reads those
 *                            (default 1) instead of normalising every sample of every frame itself (0); same bits
 *   "stft_prepad"      0/1   ... and writes them as the independent-wave STFT kernel reads them: truncated / zero padded to
 *                            n_samples and centre padded (n_fft / 2 reflected samples, or zeros, on both sides), so that no
 *                            frame needs index reflection (default 1; same bits)
 *   "stft_grid"        0..   persistent STFT workgroups (0 = fill the chip; results do not depend on it)
 * Every knob is a property of its context (round 3: "stft_indep", "stft_grid" and "rms_chunk_path_over" used to be process-wide;
 * "head_fold", "gemm_flags" and "gemm_min_lds" are gone with the measured dead ends they selected).
 *   "bf16_encoder"     0/1   encoder in bf16 (weights, activations, v_mfma_f32_32x32x16_bf16) with fp32
 *                            accumulation, BN/ReLU and latents (default 0 = fp32, the path the 1e-4 parity
 *                            bar applies to); the front end is always fp32
 *   "fuse_conv12"      0/1   recompute conv layer 1 inside layer 2's operand staging instead of
 *                            materialising it (default 1; results are bit-identical either way) */
int amph_configure(amph_ctx* ctx, const char* key, int value);

/* Geometry derived from the configuration. */
int amph_raw_frames(const amph_ctx* ctx);     /* T_raw = 1 + n_samples / hop_length */
int amph_latent_dim(const amph_ctx* ctx);
int amph_max_batch(const amph_ctx* ctx);
/* Dense Slaney mel filterbank the context uses, float32 [n_mels][1 + n_fft/2]
 * (what librosa.filters.mel returns inside melspectrogram, map_detector_core.py:219). */
int amph_get_mel_basis(const amph_ctx* ctx, float* h_out, size_t n_floats);

/* -------------------------------------------------------------------------
 * rms_normalize — 00_normalize_dataset_rms.py:29-38 (target 0.05, gate 1e-4,
 * eps 1e-8, clip [-1,1]).  d_ok[b] = 1 if scaled, 0 if the silence gate left the
 * row unchanged.  Rows are `n` samples long with a row stride of `n`.
 * The _pcm16 form is what 00's process_folder writes to disk
 * (00_normalize_dataset_rms.py:51-57): decode q/32768, normalise, quantise
 * rint(x*32767).
 * ------------------------------------------------------------------------- */
int amph_rms_normalize_f32(amph_ctx* ctx, const float* d_in, int n_rows, int n,
                           float* d_out, int32_t* d_ok, void* stream);
int amph_rms_normalize_pcm16(amph_ctx* ctx, const int16_t* d_in, int n_rows, int n,
                             int16_t* d_out, int32_t* d_ok, void* stream);

/* -------------------------------------------------------------------------
 * wav_to_mel — map_detector_core.py:198-237, starting from the decoded signal.
 * Rows of `n_in` float32 samples (row stride n_in); d_lengths (nullable) gives
 * the valid sample count per row; each row is zero-padded / truncated to
 * n_samples (:212-217).  layout 0: d_mel_out is [B][n_mels][target_frames] (the
 * tensor wav_to_mel returns); layout 1: [B][target_frames][n_mels] (= the
 * [B,1,T,M] encoder input of map_detector_core.py:268).
 * ------------------------------------------------------------------------- */
int amph_mel_f32(amph_ctx* ctx, const float* d_wave, int n_rows, int n_in,
                 const int32_t* d_lengths, int layout, float* d_mel_out, void* stream);
/* Raw mel power spectrogram before dB (librosa.feature.melspectrogram output,
 * map_detector_core.py:219-228), [B][T_raw][n_mels]; for kernel-level parity tests. */
int amph_mel_power_f32(amph_ctx* ctx, const float* d_wave, int n_rows, int n_in,
                       const int32_t* d_lengths, float* d_melpow_out, void* stream);

/* The same from decoded PCM16 samples (what librosa.load reads from the files 00_normalize_dataset_rms.py writes, and the
 * layout amph_encode_pcm16 takes): rows of n_in int16 samples; apply_rms != 0 runs rms_normalize + the PCM16 write / read of
 * 00_normalize_dataset_rms.py:29-57 first.  d_mel_out (nullable) as amph_mel_f32; d_melpow_out (nullable) receives the raw mel
 * power [B][T_raw][n_mels] (map_detector_core.py:219-228).  At least one of the two must be given. */
int amph_mel_pcm16(amph_ctx* ctx, const int16_t* d_pcm, int n_rows, int n_in, const int32_t* d_lengths, int apply_rms,
                   int layout, float* d_mel_out, float* d_melpow_out, void* stream);

/* -------------------------------------------------------------------------
 * encoder(x) — map_detector_core.py:270 with x = [B,1,T,M] float32 (:268).
 * Writes mu[B][D] and (nullable) logvar[B][D].
 * ------------------------------------------------------------------------- */
int amph_encoder_forward(amph_ctx* ctx, const float* d_x, int n_rows,
                         float* d_mu, float* d_logvar, void* stream);

/* -------------------------------------------------------------------------
 * Range of the default arithmetic.  With "split_gemm" 1 (default) the conv / head GEMMs run fp32 operands as two fp16 terms
 * (x = h + 2^-11 l), which exist for |x| < 65 504 only.  Weights are tested at amph_create (a layer whose weights are outside
 * takes the f32-MFMA kernel); ACTIVATIONS depend on the data: every kernel whose output is the next layer's split operand tests
 * its values and raises a per-context flag, and amph_encoder_forward tests the caller's tensor first.  The reference module
 * (map_detector_core.py:175-179, any checkpoint, fp32) has no such limit, so a raised flag means: the latents of the calls since
 * the last clear are NOT valid (they may be NaN) - amph_configure(ctx, "split_gemm", 0) (+ "split_fused", 0) and run them again.
 * amph_range_status reads the flag (1 = raised) and clears it when `clear` != 0; meaningful once the streams of the calls in
 * question have synchronised.  amph_encode_pcm16_host, which synchronises, does all of this itself (and leaves the context on
 * the f32-MFMA kernels).  The Python binding (AmphSession.encode / encoder_forward) does the same after its calls.
 * ------------------------------------------------------------------------- */
int amph_range_status(amph_ctx* ctx, int clear);

/* -------------------------------------------------------------------------
 * encode — the fused chunk -> latent path: [rms_normalize + PCM16 write/read
 * (00_normalize_dataset_rms.py:51-57)] -> wav_to_mel -> encoder -> mu
 * (encode_wav_to_latent, map_detector_core.py:240-300, for n_rows chunks at once;
 * the reference is batch-1 and loops, 08_fit_radial_detector.py:488-503).
 * Any n_rows >= 0; processed internally in batches of max_batch.
 * d_rms_ok (nullable): 1 if the row was RMS-scaled, 0 if gated / apply_rms == 0.
 * ------------------------------------------------------------------------- */
int amph_encode_pcm16(amph_ctx* ctx, const int16_t* d_pcm, int64_t n_rows, int n_in,
                      const int32_t* d_lengths, int apply_rms,
                      float* d_mu, float* d_logvar, int32_t* d_rms_ok, void* stream);
int amph_encode_f32(amph_ctx* ctx, const float* d_wave, int64_t n_rows, int n_in,
                    const int32_t* d_lengths, int apply_rms,
                    float* d_mu, float* d_logvar, int32_t* d_rms_ok, void* stream);
/* Host-buffer convenience: stages through pinned memory, copies H2D on a side
 * stream overlapped with compute, copies mu back; synchronises before returning. */
int amph_encode_pcm16_host(amph_ctx* ctx, const int16_t* h_pcm, int64_t n_rows, int n_in,
                           const int32_t* h_lengths, int apply_rms,
                           float* h_mu, float* h_logvar, int32_t* h_rms_ok);

/* -------------------------------------------------------------------------
 * Radial decision — 10_benchmark_folder_detection.py:175-199 with l2 from
 * 09_evaluate_wav_detection.py:354-355.  K species in config order; accept k if
 * (double)||z - mu_k||_2 <= thr[k]; among accepted choose the smallest
 * h_rank[k] (rank = index in PRIORITY_ORDER, then sorted-name order for the
 * rest; computed by the host binding).  d_pred[b] = species index or -1
 * (NO_DETECT); d_best_d[b] = min_k distance.  Norms use numpy's float32
 * pairwise summation order so equal z gives an equal decision.
 * ------------------------------------------------------------------------- */
int amph_decide_radial(amph_ctx* ctx, const float* d_z, int64_t n_rows, int dim,
                       const float* h_centroids, const double* h_thresholds,
                       const int32_t* h_rank, int n_species,
                       int32_t* d_pred, float* d_best_d, void* stream);

/* -------------------------------------------------------------------------
 * Radial fit - fit_species_with_fp_control, 08_fit_radial_detector.py:310-333, for every species at once on latents that
 * are resident in HBM: d_z is [n][dim] float32 with the rows of species k at h_offsets[k] .. h_offsets[k + 1] - 1
 * (h_offsets[0] == 0, every species non-empty; n < 2^24).  For species k:
 *   centroid  = np.mean(Z_k, axis=0)                      (numpy's float32 order: row after row, then / N)
 *   rk_in     = np.quantile(||Z_k - centroid||, q_in)     rk_out = np.quantile(||Z_others - centroid||, q_out)  (+inf when
 *   there are no other rows);   rk = min(rk_in, rk_out)
 * numpy2 = 0: np.quantile's scalar arithmetic of numpy 1.26.4, the reference's pin (float64 interpolation); 1: numpy 2.x
 * (float32).  Host outputs h_centroids [K][dim], h_rk_in / h_rk_out / h_rk [K] and (nullable) h_summaries [K][2][4] =
 * summarize_dist (08:115-123: min, p50, p90, max) of the in-class and of the out-class distances (NaN when there are
 * none); synchronises `stream`.
 * ------------------------------------------------------------------------- */
int amph_fit_radial(amph_ctx* ctx, const float* d_z, const int64_t* h_offsets, int n_species, int dim,
                    double q_in, double q_out, int numpy2,
                    float* h_centroids, double* h_rk_in, double* h_rk_out, double* h_rk, double* h_summaries, void* stream);

/* -------------------------------------------------------------------------
 * Gaussian-MAP decision - MapDetectorSession.predict_one,
 * 10b_benchmark_folder_detection_map.py:143-169, with gaussian_logpdf_from_precision
 * (map_detector_core.py:319-323).  K species in the caller's order (the reference
 * iterates sorted(names), 10b:113); for each:
 *   score_k = -0.5 * (diff^T P_k diff + logdet_k + D log 2pi) + h_log_prior[k],
 *   diff = z - mu_k and the quadratic form in float32, the rest in float64;
 * h_log_prior[k] = log(prior_k + 1e-12) (10b:156).  d_pred[b] = first k with the
 * largest score (strict >), or -1 (NO_DETECT) when has_tau and best < tau (10b:166);
 * d_best_score[b] = that score; d_scores (nullable) receives all [n_rows][K] scores
 * (what 08b_fit_map_detector.py:302-306 evaluates for the true class when it sets tau).
 * ------------------------------------------------------------------------- */
int amph_decide_map(amph_ctx* ctx, const float* d_z, int64_t n_rows, int dim,
                    const float* h_means, const float* h_precision,
                    const double* h_logdet, const double* h_log_prior, int n_species,
                    int has_tau, double tau,
                    int32_t* d_pred, double* d_best_score, double* d_scores, void* stream);

/* -------------------------------------------------------------------------
 * Sample-rate conversion - the resample step of librosa.load(path, sr=sr)
 * (map_detector_core.py:210; 00_normalize_dataset_rms.py:51) for files whose rate
 * differs from `sr`: resampy's windowed-sinc interpolation (librosa 0.9.2 default
 * res_type "kaiser_best") followed by librosa's fix_length to ceil(n_in * ratio).
 * Rows of n_in float32 samples (row stride n_in) -> rows of amph_resample_length()
 * samples at row stride out_stride.  h_half_window is the UNSCALED half window of the
 * filter (resampy.filters.sinc_window output, float64 [n_window]), num_table its
 * samples per zero crossing; the library applies resampy's ratio scaling and builds the
 * interpolation deltas (uploaded per call: the call synchronises `stream` twice and is meant
 * for ingest, not for a hot loop).  Parity unpinned (resampy and its table are absent from the
 * build image; oracle/resample.py restates the published algorithm).
 * ------------------------------------------------------------------------- */
int amph_resample_length(int n_in, int orig_sr, int target_sr);
int amph_resample_f32(amph_ctx* ctx, const float* d_in, int n_rows, int n_in, int orig_sr, int target_sr,
                      const double* h_half_window, int n_window, int num_table,
                      float* d_out, int out_stride, void* stream);

/* -------------------------------------------------------------------------
 * WAV ingest — the libsndfile decode inside librosa.load
 * (map_detector_core.py:210) for mono PCM16 RIFF files already at `sr`.
 * Reads up to max_samples int16 samples into h_out (zero-filled beyond the
 * file's length); *out_len = samples in the file (clamped to max_samples).
 * amph_wav_read_many fills rows of a [n_files][max_samples] (pinned or
 * pageable) buffer with a pool of `n_threads` host threads; h_status[i] is an
 * amph_status per file (bad files are flagged, the batch continues:
 * 08_fit_radial_detector.py:504-506).
 * ------------------------------------------------------------------------- */
int amph_wav_read_pcm16(const char* path, int expect_sr, int16_t* h_out, int max_samples,
                        int32_t* out_len);
/* Sizes in bytes of n_files files (-1: no such file), by a pool of host threads: what the folder loop needs to size its rows
 * (08_fit_radial_detector.py:477-503 reads every file whole) without a Python-level stat() per file. */
int amph_wav_sizes(const char* const* paths, int n_files, int64_t* h_sizes, int n_threads);
int amph_wav_read_many(const char* const* paths, int n_files, int expect_sr,
                       int16_t* h_out, int max_samples, int32_t* h_lengths,
                       int32_t* h_status, int n_threads);
/* Pinned host allocation for the ingest buffers (hipHostMalloc / hipHostFree). */
void* amph_host_alloc(size_t bytes);
void amph_host_free(void* p);

/* -------------------------------------------------------------------------
 * Per-kernel timing with HIP events on the launch stream (bench.py's roofline
 * leg).  amph_profile_enable(ctx, 0) = off, 1 = bracket every kernel launch of
 * the encode path with an event pair, 2 + i = bracket only stage i (two event
 * records per batch: cheap enough to leave on inside a timed region).
 * amph_profile_read synchronises, then reports for stage i its launch count and
 * total milliseconds since the last reset, with the stage's algorithmic FLOPs
 * and bytes per chunk.
 * ------------------------------------------------------------------------- */
int amph_profile_enable(amph_ctx* ctx, int on);
int amph_profile_stage_count(const amph_ctx* ctx);
const char* amph_profile_stage_name(const amph_ctx* ctx, int stage);
int amph_profile_read(amph_ctx* ctx, int stage, int64_t* out_launches, double* out_total_ms,
                      double* out_flops_per_chunk, double* out_bytes_per_chunk);
int amph_profile_reset(amph_ctx* ctx);

#ifdef __cplusplus
}
#endif
#endif /* AMPH_H_ */
