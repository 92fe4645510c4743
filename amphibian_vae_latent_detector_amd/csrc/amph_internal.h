// Internal declarations shared by the HIP translation units of libamph.
// Kernel parameter blocks are plain structs passed by value.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace amph {

// ----------------------------------------------------------------------------
// Front end
// ----------------------------------------------------------------------------
constexpr int kFrontThreads = 256;      // 4 waves per workgroup
constexpr int kZPad = 4;                // float2 of padding per 256-point block of Z

struct WaveSrc {
    const void* wave;        // int16_t or float rows
    long long row_stride;    // in samples
    const int32_t* lengths;  // nullable: valid samples per row
    int n_in;                // valid samples per row when lengths == nullptr
    const float* scale;      // nullable: per-row RMS scale (1.0 when gated)
    int requant;             // emulate the PCM16 write/read of 00_normalize (x*32767, rint, /32768)
    int lead;                // > 0 (k_stft_mel_w only): rows already are fix_length'd and centre padded (`lead` samples in front and
                             // behind, n_in = n_samples + 2 lead, lengths == nullptr): every frame is an interior frame
};

// K0's optional second output (PCM16 rows): lead == 0 -> the normalised row as it is (row_stride of the input);
// lead > 0 -> truncated / zero padded to n_fix samples + `lead` samples of centre padding on both sides (mode 0 reflect, 1 zeros)
struct NormPad {
    int lead = 0, n_fix = 0, mode = 0;
    long long out_stride = 0;
};

struct StftParams {
    WaveSrc src;
    int n_samples;           // N after fix_length
    int n_fft, hop, n_frames;
    int pad_mode;            // 0 reflect, 1 constant
    int frames_per_block;    // FT
    int tiles_per_row;       // ceil(n_frames / FT)
    const float* tables;     // packed LDS image: half window | tw_b | tw_post (stft_mel_table_bytes())
    const float2* tw_a;      // [15][64] W_1024^(t*k1), k1 = 1..15 (read from global memory)
    int n_mels;
    const int32_t* mel_start;  // [n_mels] first FFT bin of the filter rounded down to a multiple of 4
    const int32_t* mel_len;    // [n_mels] 16-bin steps of the filter's ROUND (16 consecutive filters share it)
    const int32_t* mel_wofs;   // [n_mels] offset into mel_w (multiple of 4)
    const float* mel_w;        // per filter 16*steps weights, zero padded, 16-byte aligned (staged into LDS)
    int mel_w_floats;          // total, a multiple of 16
    int p_tail;                // floats past bin n_fft/2 the padded rows may read (zeroed per frame)
    float* melpow;           // [B][n_frames][n_mels]
    // fused dB / z-score / crop-pad (k_stft_mel_w only; tiles_per_row == 1, i.e. one workgroup owns a whole row): when db_out is
    // set the mel power of a row stays in LDS and the workgroup writes the standardised image itself (melpow is not written)
    float* db_out;           // nullable: [B][target][n_mels] (layout 1) or [B][n_mels][target] (layout 0)
    int db_layout, db_target_frames;
    // k_stft_pair (two frames per wave as one complex FFT; nullptr: that kernel is not used)
    const float* pair_tables;  // packed LDS image: window pairs [16][64][2] (x 2^-15) | stage-2 twiddles [2][32][2]
    const float2* pair_tw1;    // [31][64] W_2048^(col(lane) k1), k1 = 1..31 (read from global memory)
    const int32_t* pair_mel_meta;   // k_stft_pair (n_fft 2048): [rounds][16 lane quads] x (start bin, steps, weight offset, band or -1) - amph_api.hip build_pair_mel
    const float* pair_mel_w;        // ... and the weights in that kernel's lane order
    int pair_mel_w_floats;
    int pair_waves;            // waves per workgroup of that kernel (8, 10 or 12)
    int pair_tw_lds;           // keep pair_tw1 in LDS when the image still fits 160 KB
    // K0 inside k_stft_pair ("stft_k0_fuse", round 4; k_frontend.hip, K0F): k0_in != nullptr -> the kernel reads the RAW PCM16 rows, folds the
    // per-chunk sums of squares K0's first kernel left in k0_chunk_sums into each row's scale and re-quantises in its decode; src then only
    // describes the geometry (lead = n_fft / 2), its buffer is not read
    const int16_t* k0_in;      // raw PCM16 rows
    long long k0_row_stride;
    const int32_t* k0_lengths; // nullable
    int k0_n_in, k0_scale_f64;
    float* k0_scale;           // [B] out
    int32_t* k0_ok;            // [B] out (nullable)
    const float* k0_chunk_sums;   // [B][k0_chunk_stride] sums of squares per numpy chunk (launch_rms_chunk_sums_pcm16); nullptr: no RMS stage - the rows as they are
    int k0_chunk_stride;
    int pair2_regdb;           // k_stft_pair2 with the fused dB stage: the row's mel power stays in registers ("stft_pair2_regdb")
    int pair_rows2;            // K0F + fused dB stage: two rows per workgroup and pair queue when the batch is large enough ("stft_pair_rows2")
    int pair_cq;               // ... or, > 1: that many rows per workgroup on ONE continuous pair queue, the dB stage of a row run by the wave that completes it ("stft_pair_cq")
    // per-context launch choices (were process-wide globals until round 3)
    int indep;                 // 1: k_stft_mel_w (independent waves), 0: the tile-synchronous k_stft_mel
    int grid_wgs;              // persistent workgroups; 0 = fill the chip
#ifdef AMPH_STAMPS
    unsigned long long* stamps;   // diagnostic build: [grid][16] s_memtime stamps
#endif
};

struct DbNormParams {
    const float* melpow;     // [B][n_frames][n_mels]
    int n_frames, n_mels, target_frames;
    int layout;              // 0: out [B][n_mels][target]; 1: out [B][target][n_mels]
    float* out;
};

// scale_f64: 1 = the scalar arithmetic of numpy 1.26 (the reference's pin: float64 gate compare and scale, one cast), 0 = numpy 2.x (float32)
// norm_out (nullable): also write the normalised, re-quantised PCM16 rows (same row stride)
// chunk_sums (nullable): workspace of n_rows * rms_scale_chunks(n_in) floats; with it, rows longer than one LDS pass take the
// two-kernel path (chunk sums by (row, pass) workgroups, then fold + normalise + pad by (row, tile) workgroups)
int rms_scale_chunks(int n_in);
// chunk_path_over: rows of more numpy chunks than this take the two-kernel path (when chunk_sums is given)
void launch_rms_scale_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in,
                            int n_rows, float* scale, int32_t* ok, int scale_f64, int16_t* norm_out, const NormPad& pad, float* chunk_sums,
                            int chunk_path_over, hipStream_t s);
void launch_rms_chunk_sums_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows, float* chunk_sums, hipStream_t s);
void launch_pad_rows_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows, int16_t* out,
                           const NormPad& pad, hipStream_t s);
void launch_rms_scale_f32(const float* in, long long row_stride, const int32_t* lengths, int n_in,
                          int n_rows, float* scale, int32_t* ok, int scale_f64, hipStream_t s);
void launch_rms_apply_f32(const float* in, int n_rows, int n, const float* scale, const int32_t* ok, float* out,
                          hipStream_t s);
void launch_rms_apply_pcm16(const int16_t* in, int n_rows, int n, const float* scale, const int32_t* ok, int16_t* out,
                            hipStream_t s);
// sample_is_pcm16: 1 -> src.wave is int16_t, 0 -> float
void launch_stft_mel(const StftParams& p, int n_rows, int sample_is_pcm16, int n_waves, hipStream_t s);
size_t stft_mel_lds_bytes(int n_fft, int hop, int frames_per_block, int n_mels, int mel_w_floats, int n_waves);
// LDS of the independent-wave kernel with the fused dB stage (no sample span; + the row's mel power), and whether a
// configuration can run it (every wave must own a frame of every row: the row-end barriers are workgroup barriers)
size_t stft_mel_fused_lds_bytes(int n_fft, int n_frames, int n_mels, int mel_w_floats, int n_waves);
bool stft_mel_can_fuse_db(int n_fft, int n_frames, int n_mels, int mel_w_floats, int n_waves);
int stft_mel_table_bytes(int n_fft);
int stft_mel_scratch_floats(int n_fft);   // floats of per-frame scratch the power spectrum + tail may use
hipError_t stft_mel_configure(int lds_bytes);   // raises the dynamic-LDS limit of the STFT kernels
// the pair kernel (k_stft_pair): LDS image, packed table size, and the configurations it covers
size_t stft_pair_lds_bytes(int n_frames, int n_mels, int mel_w_floats, int n_waves, bool fused, int rows = 1);   // rows: mel-power images kept in LDS (2: the G2 form)
bool stft_pair_can_fuse_k0(int n_in, int n_waves);      // the K0F form of k_stft_pair covers this row length / workgroup shape
bool stft_pair2_can_fuse_k0(int n_in, int n_frames, int n_mels, int n_waves, bool fused_db, bool regdb);
int stft_pair_table_bytes();
size_t stft_pair_tw1_lds_bytes();
bool stft_pair_ok(int n_fft, int hop, int p_tail);
// ... and its n_fft 1024 / hop 256 form (k_stft_pair2: a pair per half wave, four frames per wave iteration; 8 or 12 waves)
size_t stft_pair2_lds_bytes(int n_frames, int n_mels, int mel_w_floats, int n_waves, bool fused, bool regdb = true);   // regdb: the dB stage on registers where it applies
bool stft_pair2_regdb_ok(int n_frames, int n_mels, int n_waves);
int stft_pair2_table_bytes();
size_t stft_pair2_tw1_lds_bytes();
bool stft_pair2_ok(int n_fft, int hop, int p_tail);
void launch_dbnorm(const DbNormParams& p, int n_rows, hipStream_t s);
// the general STFT + mel kernel: any power-of-two n_fft in 64 .. 8192, any hop (win: [n_fft] periodic Hann, tw: [n_fft / 2] W_n^k; melpow only)
void launch_stft_generic(const StftParams& p, int n_rows, int sample_is_pcm16, const float* win, const float2* tw, hipStream_t s);

struct ResampleParams {      // resampy's resample_f for rows of equal length
    const float* x;          // [rows][in_stride]
    long long in_stride;
    int n_in;
    float* y;                // [rows][out_stride]
    long long out_stride;
    int n_out;               // int(n_in * ratio): samples resampy computes
    int n_fix;               // ceil(n_in * ratio): length after librosa's fix_length (zero padded)
    const double* win;       // device [nwin] half window (already scaled by the ratio when downsampling)
    const double* delta;     // device [nwin] diff(win), last element 0
    int nwin, num_table, index_step;
    double scale, time_inc;  // min(1, ratio), 1 / ratio
};
void launch_resample(const ResampleParams& p, int n_rows, hipStream_t s);

// ----------------------------------------------------------------------------
// Encoder
// ----------------------------------------------------------------------------
struct ConvFirstParams {     // C_in == 1 direct convolution, NHWC output
    const float* x;          // [B][H][W]
    const float* w;          // [k*k][C_out]  (tap-major, channel fastest)
    const float* scale;      // [C_out] BN alpha = gamma / sqrt(var + eps)
    const float* shift;      // [C_out] BN beta' = (bias - mean) * alpha + beta
    float* y;                // [B][OH][OW][C_out]
    int H, W, OH, OW, C_out, k, stride, pad;
    int* range_flag;         // nullable: raised when an output is no fp16 split operand (kH2Max below)
};

struct ConvGemmParams {      // implicit GEMM: M = B*OH*OW, N = C_out, K = k*k*C_in (kh, kw, ci)
    const float* x;          // NHWC [B][H][W][C_in]
    const float* w;          // [N_pad][K_pad], k contiguous, zero padded
    const float* scale;      // [N_pad] (epilogue 0) or nullptr
    const float* shift;      // [N_pad]
    float* y;                // epilogue 0: NHWC [M][N]; epilogue 1: partials [splits][M][N]
    int H, W, C_in, OH, OW, N, k, stride, pad;
    int K, K_pad;            // true and padded (multiple of 32) reduction length
    int M;                   // B*OH*OW
    int epilogue;            // 0: y = relu(acc*scale + shift); 1: raw partial sums (split-K)
    int splits;              // gridDim.z; each split reduces K_pad/32/splits slices
    // fused first layer (FUSE1 launches only): x is unused, the input is recomputed from `img`
    const float* img;        // [B][H0][W0] standardised mel image
    const float* w1;         // first layer [k1*k1][C_in] (tap-major)
    const float* scale1;     // [C_in]
    const float* shift1;     // [C_in]
    int H0, W0, k1, s1, p1;
    const float* zeros;      // >= 16 bytes of zeros (16-byte aligned): source of masked loads in the fused kernel
    int x_guard;             // bytes of ZEROS in front of x (same allocation): padded taps of k_conv_gemm_p3 read there
    int y_bf16;              // k_conv_gemm_f3s only: write the output as bf16 NHWC (the bf16 encoder's layer 3 reads it)
    int flags;               // (first-generation kernel only; bit 0: static issue priority from the wave's slot on its SIMD - always 0 since round 3)
    int* range_flag;         // nullable; two-term fp16 split kernels (epilogue 0): set to 1 when an output - the NEXT layer's split operand - is
                             // at or above fp16's 65 504 or NaN (kH2Max below); host-visible word, read after the stream has synchronised
#ifdef AMPH_STAMPS
    unsigned long long* stamps;   // diagnostic build: [workgroup][role][4]
#endif
};

struct HeadReduceParams {    // mu/logvar = sum_s partial[s] + bias
    const float* partial;    // [splits][B][2D]
    const float* bias;       // [2D]
    int splits, D;
    float* mu;               // [B][D]
    float* logvar;           // nullable [B][D]
};

void launch_conv_first(const ConvFirstParams& p, int n_rows, hipStream_t s);
void launch_range_check(const float* x, long long n, float limit, int* flag, hipStream_t s);   // *flag = 1 if any |x[i]| >= limit or NaN
extern int g_num_cus;      // compute units of the device (a property of the hardware, not a knob)
bool conv_gemm_can_fuse(const ConvGemmParams& p, int bm);
size_t conv_gemm_x_guard(int k, int W, int C_in);          // zero guard k_conv_gemm_p3 needs in front of its input
bool conv_gemm_f3_ok(const ConvGemmParams& p, int bm);   // second-generation fused kernel (64-row tiles, 32 first-layer channels)
void launch_conv_gemm(const ConvGemmParams& p, int bm, bool fuse_first, hipStream_t s);
bool conv_gemm_p3_ok(const ConvGemmParams& p, int bm);
void launch_head_reduce(const HeadReduceParams& p, int n_rows, hipStream_t s);
// fp32-accurate GEMM layers on the bf16 matrix cores (k_encoder_split.hip): p.w = split weights [n_pad / 64][K_pad / 32][64][3][32] bf16
bool conv_gemm_s3_ok(const ConvGemmParams& p);
void launch_conv_gemm_s3(const ConvGemmParams& p, hipStream_t s);
bool conv_gemm_s3w_ok(const ConvGemmParams& p);           // 128 x 128 tiles, 512 threads (same weights, bit-identical results)
void launch_conv_gemm_s3w(const ConvGemmParams& p, hipStream_t s);
void conv_gemm_s3_split_weights(const float* w, int n_pad, int K_pad, uint16_t* out);
void conv_gemm_h2_split_weights(const float* w, int n_pad, int K_pad, uint16_t* out);     // two-term fp16 split (k_conv_gemm_s3 / _s3w)
void conv_gemm_h2_fragment_weights(const float* w, int n_pad, int K_pad, uint16_t* out);  // the same in MFMA fragment order (k_conv_gemm_g4 / _g5, fp32 layers)
// ... and the fused first-layer pair in the same arithmetic (p.img etc. as for k_conv_gemm_f3, p.w = split weights)
bool conv_gemm_f3s_ok(const ConvGemmParams& p);
void launch_conv_gemm_f3s(const ConvGemmParams& p, hipStream_t s);
// ... and with the layer-1 activation of a tile resident in LDS, weights in registers (k_encoder_patch.hip):
// wfrag = split weights in MFMA fragment order [n_pad / 64][2][K_pad / 16][3][64 lanes][8] bf16, wgs = workgroups (0: one per CU)
bool conv_gemm_f4s_ok(const ConvGemmParams& p);
void launch_conv_gemm_f4s(const ConvGemmParams& p, const void* wfrag, int wgs, hipStream_t s);
// ... and with one bf16 term per operand for the bf16 encoder (same shapes: conv_gemm_f4s_ok; bf16 NHWC output)
void conv_gemm_f4b_fragment_weights(const float* w, int n_pad, int K_pad, uint16_t* out);
bool conv_gemm_f4b_ok(const ConvGemmParams& p);
void launch_conv_gemm_f4b(const ConvGemmParams& p, const void* wfrag, int wgs, hipStream_t s);
// the 3x3 stride-2 layers behind the pair in the same form: weights in registers, the tile's input image(s) resident in LDS
int conv_gemm_g4_kind(const ConvGemmParams& p);
void launch_conv_gemm_g4(const ConvGemmParams& p, const void* wfrag, int wgs, int terms, hipStream_t s);   // terms 3: fp32 in / out; 1: bf16 in / out (wfrag: conv_gemm_f4b_fragment_weights)
// bf16 encoder path (k_encoder_bf16.hip): x / w are bf16, y bf16 (epilogue 0) or fp32 partials (epilogue 1);
// K, K_pad and C_in count bf16 elements, K_pad % 64 == 0, C_in % 8 == 0
void launch_conv_first_bf16(const ConvFirstParams& p, int n_rows, hipStream_t s);
void launch_conv_gemm_bf16(const ConvGemmParams& p, int bm, hipStream_t s);

#ifdef __HIPCC__
// The two-term fp16 split x = h + 2^-11 l (k_encoder_split.hip) exists for |x| < 65 520 only: h = fp16(x) is inf above and the low
// term -inf.  Every kernel whose output feeds such a split tests its pre-ReLU values with this (true for >= 65 504, +inf and NaN;
// large negative values are zeroed by the ReLU) and raises ConvGemmParams::range_flag; amph_range_status reports it.
constexpr float kH2Max = 65504.0f;
__device__ __forceinline__ bool h2_unsplittable(float t) { return !(t < kH2Max); }

// sum((z - mu)^2) over n <= 128 elements in numpy's float32 pairwise order (n < 8 sequential; else eight strided accumulators
// combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail): np.sum(a * a) of 09_evaluate_wav_detection.py:354-355 and
// each row of l2_norm_rows, 08_fit_radial_detector.py:105-106
__device__ __forceinline__ float np_pairwise_sumsq(const float* z, const float* mu, int n) {
#pragma clang fp contract(off)   // numpy rounds a = z - mu, a * a and every add separately; no FMA
    if (n < 8) {
        float r = 0.0f;
        for (int i = 0; i < n; ++i) { const float a = z[i] - mu[i]; r = r + a * a; }
        return r;
    }
    float r[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float a = z[j] - mu[j]; r[j] = a * a; }
    int i = 8;
    for (; i < n - (n % 8); i += 8) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float a = z[i + j] - mu[i + j]; r[j] = r[j] + a * a; }
    }
    float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
    for (; i < n; ++i) { const float a = z[i] - mu[i]; res = res + a * a; }
    return res;
}


// x = h + m + l exactly, h = bf16(x), m = bf16(x - h), l = bf16(x - h - m) (RNE; every subtraction is exact): packed pairs
typedef __bf16 amph_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void s3_split2(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    const amph_bf16x2 hp = {(__bf16)x0, (__bf16)x1};                  // v_cvt_pk_bf16_f32
    h = __builtin_bit_cast(uint32_t, hp);
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    const amph_bf16x2 mp = {(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(uint32_t, mp);
    const float t0 = r0 - __uint_as_float(m << 16), t1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    const amph_bf16x2 lp = {(__bf16)t0, (__bf16)t1};
    l = __builtin_bit_cast(uint32_t, lp);
}
// the same with the packed pairs made opaque to the optimiser: h << 16 is then a shift of h, not a second conversion of x0 alone
// (8 instructions fewer per 8 values; costs registers where the caller is at its limit - k_conv_gemm_f3s keeps the form above)
__device__ __forceinline__ void s3_split2_o(float x0, float x1, uint32_t& h, uint32_t& m, uint32_t& l) {
    const amph_bf16x2 hp = {(__bf16)x0, (__bf16)x1};
    h = __builtin_bit_cast(uint32_t, hp);
    asm("" : "+v"(h));
    const float r0 = x0 - __uint_as_float(h << 16), r1 = x1 - __uint_as_float(h & 0xFFFF0000u);
    const amph_bf16x2 mp = {(__bf16)r0, (__bf16)r1};
    m = __builtin_bit_cast(uint32_t, mp);
    asm("" : "+v"(m));
    const float t0 = r0 - __uint_as_float(m << 16), t1 = r1 - __uint_as_float(m & 0xFFFF0000u);
    const amph_bf16x2 lp = {(__bf16)t0, (__bf16)t1};
    l = __builtin_bit_cast(uint32_t, lp);
}
#endif

// ----------------------------------------------------------------------------
// Radial decision
// ----------------------------------------------------------------------------
constexpr int kMaxSpecies = 64;
constexpr int kMaxRadialDim = 128;
struct RadialParams {
    const float* z;          // [B][D]
    const float* centroids;  // device [K][D]
    const double* thresholds;// device [K]
    const int32_t* rank;     // device [K]
    int K, D;
    int32_t* pred;
    float* best_d;
};
void launch_decide_radial(const RadialParams& p, long long n_rows, hipStream_t s);

// ----------------------------------------------------------------------------
// Gaussian-MAP decision
// ----------------------------------------------------------------------------
struct MapParams {
    const float* z;          // [B][D]
    const float* means;      // device [K][D]
    const float* prec;       // device [K][D][D] precision matrices (row-major, float32 as the reference reads them)
    const double* logdet;    // device [K] log det of the covariance
    const double* log_prior; // device [K] log(prior + 1e-12)
    double d_log2pi;         // D * log(2 pi)
    int K, D;
    int has_tau;
    double tau;
    int32_t* pred;           // [B] species index, -1 = NO_DETECT
    double* best_score;      // [B]
    double* scores;          // nullable [B][K]
};
void launch_score_map(const MapParams& p, long long n_rows, hipStream_t s);

// ----------------------------------------------------------------------------
// Radial fit on the device (k_fit.hip): 08_fit_radial_detector.py:310-333 on latents resident in HBM
// ----------------------------------------------------------------------------
struct FitParams {
    const float* z;            // [n][D] rows grouped by species (species k = rows offsets[k] .. offsets[k + 1] - 1)
    const long long* offsets;  // device [K + 1]
    int K, D;
    long long n;
    float* mu;                 // device [K][D]
    float* rho;                // device [K][n]: distance of EVERY row to centroid k
    const long long* ranks;    // device [K][16]: order statistics wanted, 8 per class (in, out): q lo, q hi, min, p50 lo, hi, p90 lo, hi, max; -1 = none
    float* stats;              // device [K][16]
};
void launch_fit_radial(const FitParams& p, hipStream_t s);

}  // namespace amph
