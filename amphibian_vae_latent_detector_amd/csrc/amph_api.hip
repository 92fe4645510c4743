// libamph: C-ABI entry points (include/amph.h), context and host-side preparation.
//
// Host preparation restates, in double precision then float32 exactly as librosa
// 0.9.2 does, the tables the reference gets from librosa at
// latent_space_exploration/map_detector_core.py:219-228 (periodic Hann window,
// Slaney mel filterbank) plus FFT twiddles, and re-lays the encoder weights out
// for the NHWC implicit-GEMM kernels.
#include "../../include/amph.h"
#include "amph_internal.h"

#include <atomic>
#include <fcntl.h>
#include <sys/stat.h>
#include <sys/uio.h>
#include <unistd.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

using namespace amph;

#ifdef AMPH_ABL_SKIP
// timing-only diagnostic build (make ablation_skip; tools/sweep.py --lib=... abl_skip=...): stages left out of the batch pipeline, bit 0 K0,
// 1 STFT, 2.. the GEMM layers in order, 6 the head's reduction - what a stage's removal is worth end to end bounds what any faster form of it
// can be worth.  Results are wrong by construction; never shipped.
static int g_abl_skip = 0;
namespace amph { extern int g_abl_skip_k0; }
extern "C" void amph_abl_skip(int mask) { g_abl_skip = mask; amph::g_abl_skip_k0 = (mask >> 7) & 3; }
#endif

namespace {

thread_local std::string g_create_error;

struct Stage {
    std::string name;
    double flops_per_chunk = 0, bytes_per_chunk = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending;
    long long launches = 0;
    double total_ms = 0;
};

// Per-lane scratch: a lane is one HIP stream + the buffers one batch needs, so that
// batches assigned to different lanes overlap on the GPU (VALU-bound front end of one
// batch under the MFMA-bound convolutions of another).
struct Workspace {
    hipStream_t stream = nullptr;          // lane 0 uses the caller's stream
    hipEvent_t done = nullptr;
    float* scale = nullptr;
    int32_t* ok = nullptr;
    float* melpow = nullptr;
    float* image = nullptr;
    std::vector<float*> act;               // one per conv layer; each has act_guard[i] zero bytes in front (same allocation)
    std::vector<void*> act_raw;            // the allocations
    std::vector<int> act_guard;
    float* partial = nullptr;
    int16_t* norm16 = nullptr;             // RMS-normalised PCM16 rows of the batch (what 00_normalize writes to disk), input of the STFT
    size_t norm16_cap = 0;                 // samples
    float* chunk_sums = nullptr;           // K0 on long rows: per-row sums of the numpy chunks (8192 samples each)
    size_t chunk_sums_cap = 0;             // floats
};

struct GemmLayer {
    ConvGemmParams prm{};   // x / y / M / splits filled per call
    ConvGemmParams prm_m(int M) const { ConvGemmParams q = prm; q.M = M; return q; }
    int bm = 128;
    float* d_w = nullptr;
    uint16_t* d_w_s3 = nullptr;     // three-term bf16 split [n_pad / 64][K_pad / 32][64][3][32] for k_conv_gemm_f3s (nullptr: shape not covered)
    uint16_t* d_w_h2 = nullptr;     // two-term fp16 split [n_pad / 64][K_pad / 32][64][2][32] for k_conv_gemm_s3 / _s3w
    uint16_t* d_w_f4 = nullptr;     // the same split in MFMA fragment order for k_conv_gemm_f4s (first GEMM layer) / k_conv_gemm_g4 (C_in 64 / 128)
    uint16_t* d_w_f4b = nullptr;    // bf16 weights in MFMA fragment order for k_conv_gemm_f4b (first GEMM layer, K_pad == 288 only)
    uint16_t* d_w_bf16 = nullptr;   // bf16 copy [n_pad][K_pad_b] for the bf16 encoder path
    int K_pad_b = 0;                // K rounded up to 64 elements
    float* d_scale = nullptr;
    float* d_shift = nullptr;
    int n_pad = 0;
};

}  // namespace

struct amph_ctx {
    int device = 0;
    int max_batch = 0;
    amph_frontend_config cfg{};
    amph_encoder_spec spec{};
    bool has_encoder = false;
    std::string err;

    // front-end tables
    int n_frames = 0, frames_per_block = 8;
    int stft_waves = 4;                    // waves per STFT workgroup: 4 (two workgroups per CU) or 12 (one)
    float* d_tables = nullptr;             // packed LDS image of the STFT tables
    float2* d_tw_a = nullptr;              // stage-A twiddles (global memory)
    std::vector<unsigned char> radial_key; // the detector (K, D, centroids, thresholds, ranks) last uploaded for K6
    bool generic_stft = false;             // n_fft other than 1024 / 2048, or an odd hop: k_stft_generic (+ the separate dB kernel)
    float* d_gen_win = nullptr;            // ... its periodic Hann window [n_fft]
    float2* d_gen_tw = nullptr;            // ... and twiddles W_n^k, k < n_fft / 2
    float* d_pair_tables = nullptr;        // k_stft_pair: window pairs + stage-2 twiddles (packed LDS image); nullptr: configuration not covered
    float2* d_pair_tw1 = nullptr;          // k_stft_pair: stage-1 twiddles (global memory)
    int32_t* d_pair_mel_meta = nullptr;    // k_stft_pair: per (round, lane quad) (start bin, steps, weight offset, output band) - build_pair_mel
    float* d_pair_mel_w = nullptr;         // k_stft_pair: the filters' weights in that kernel's lane order
    int pair_mel_w_floats = 0;
    bool stft_indep = true;                // k_stft_mel_w (independent waves) rather than the tile-synchronous k_stft_mel
    int stft_grid = 0;                     // persistent STFT workgroups (0 = fill the chip)
    int rms_chunk_path_over = 2;           // PCM16 rows of more numpy chunks than this: (row, chunk) + (row, tile) kernel pair
    bool stft_pair = true;                 // use k_stft_pair (two frames per wave as one complex FFT) where it applies
    int stft_pair_waves = 10;              // its workgroup: 8, 10, 11 or 12 waves; amph_create picks the largest whose LDS image, stage-1 twiddles
                                           // included, fits (1 s rows with the fused dB stage: 11 - 29 pairs on 11 waves are 3 rounds, at most 8 pairs per SIMD)
    bool stft_pair_tw_lds = true;          // its stage-1 twiddles in LDS when the image still fits
    bool stft_k0_fuse = true;              // K0's fold + apply + padded copy inside k_stft_pair (its chunk sums stay a launch); same bits
    bool stft_pair2_regdb = true;          // n_fft 1024 pair kernel: fused dB stage on register-resident mel power (no image in LDS: fits 12 waves at 128 bands)
    bool stft_pair_rows2 = true;           // ... with two rows per workgroup and pair queue when the batch has more rows than half the CUs (same bits)
    int stft_pair_cq = 0;                  // ... > 1: that many rows per workgroup on one continuous pair queue, a row's dB stage by the wave that completes it (same bits)
    bool k0_call_sums = false;             // K0's chunk sums for ALL the rows of a call in one launch in front of the batches ("k0_call_sums"; same bits;
                                           // measured 4.10 against 4.16 M chunks/s: the pass over the call's 441 MB runs in front of every lane)
    float* call_sums = nullptr;            // ... [rows of the call][chunks]
    size_t call_sums_cap = 0;
    int mel_w_floats = 0, mel_p_tail = 0;
    int32_t *d_mel_start = nullptr, *d_mel_len = nullptr, *d_mel_wofs = nullptr;
    float* d_mel_w = nullptr;
    std::vector<float> mel_basis;          // dense [n_mels][F] for amph_get_mel_basis

    // encoder
    ConvFirstParams first{};
    float *d_first_w = nullptr, *d_first_scale = nullptr, *d_first_shift = nullptr;
    std::vector<GemmLayer> gemm;           // conv layers 2.. then the head GEMM last
    float* d_head_bias = nullptr;
    float *d_map_means = nullptr, *d_map_prec = nullptr;   // Gaussian-MAP parameters (amph_decide_map)
    double *d_map_logdet = nullptr, *d_map_lprior = nullptr;
    size_t map_prec_cap = 0;
    double *d_rs_win = nullptr, *d_rs_delta = nullptr;     // resampler tables (amph_resample_f32)
    size_t rs_cap = 0;
    std::vector<double> rs_host;
    float* d_zeros = nullptr;              // 256 bytes of zeros: the source of padded taps (k_conv_gemm_p3)
    // radial fit scratch (amph_fit_radial)
    long long* d_fit_off = nullptr;        // [kMaxSpecies + 1] offsets, then [kMaxSpecies][16] ranks
    float *d_fit_mu = nullptr, *d_fit_stats = nullptr, *d_fit_rho = nullptr;
    size_t fit_rho_cap = 0;
    int head_splits = 1;
    int flat = 0, latent = 0;

    // workspaces (each sized for max_batch); lanes.size() >= 1
    std::vector<Workspace> lanes;
    hipEvent_t ev_fork = nullptr;
    unsigned long long* d_stamps = nullptr;   // AMPH_STAMPS builds only
    unsigned long long* d_gstamps = nullptr;
    int gstamp_wgs[16] = {0};
    bool gstamp_f4s = false;
    int gemm_min_wg128 = 512;
    bool bf16_encoder = false;             // BASELINE config 4: bf16 weights/activations, bf16 MFMA, fp32 accumulate
    bool fuse_conv12 = true;               // recompute layer 1 inside layer 2's A staging when the tiling allows
    bool stft_prenorm = true;              // PCM16 + RMS: write the normalised PCM16 rows once, STFT reads those
    bool stft_prepad = true;               // ... fix_length'd and centre padded (independent-wave STFT kernel only)
    bool stft_fuse_db = true;              // dB / z-score / crop-pad inside the STFT kernel when a row's mel power fits its LDS
    bool split_fused = true;               // ... the fused conv1 + conv2 kernel too (k_conv_gemm_f3s)
    int patch_gemm = 1;                    // bit 0 / 1: conv3 / conv4 of the thesis shape with weights in registers and the input resident in LDS (k_conv_gemm_g4);
                                           // conv4 in that form is 2 us faster alone and 1 % slower end to end (two images = 127 KB of LDS per workgroup): off
    int patch_gemm_bf16 = 3;               // the same for the bf16 encoder (bf16 operands: patches of 42 / 44 KB)
    int wide_gemm = -1;                    // staged GEMM layers on 128 x 128 tiles (k_conv_gemm_s3w: a quarter of the workgroups, each ~1.6x as long; same results).
                                           // -1: when the call's batches run on several lanes (a batch then costs the sum of its kernels' CU-time, not of their
                                           // durations: profiles/r03_stage_skip_sweep.txt) and the grid still has a workgroup for a quarter of the CUs; 0 / 1: never / always
    // fp16 range of the two-term split (k_encoder_split.hip): x = h + 2^-11 l needs |x| < 65 520
    int* h_range_flag = nullptr;           // host-coherent word the split kernels raise when an output - the next layer's operand - is out of range
    int* d_range_flag = nullptr;           // its device address
    float img_limit = 0.0f;                // largest |image value| for which the layer-1 activation the fused pair builds in LDS provably stays in range
                                           // (0: the layer-1 weights themselves are not splittable)
    float img_bound = 0.0f;                // what the front end can produce: |z-score| <= sqrt(n - 1) of the T_raw x M values it standardises
    int patch_gemm_wgs = 0;                // workgroups of that kernel (0: four tiles each, at most half the CUs)
    bool bf16_pair = true;                 // bf16 encoder: layers 1 + 2 in bf16 proper (k_conv_gemm_f4b) instead of fp32-accurate with a bf16 epilogue
    bool patch_fused = true;               // ... in its patch-resident form where the shape allows (k_conv_gemm_f4s)
    int patch_wgs = 0;                     // workgroups of that kernel (0: eight tiles each, at most half the CUs)
    bool split_gemm = true;                // GEMM layers on the f16 matrix cores with two-term fp16 split operands, x = h + 2^-11 l (fp32-accurate; 0: the f32-MFMA kernels)
    int rms_f64_scale = 1;                 // RMS gate + scale in the scalar arithmetic of numpy 1.26 (reference pin) instead of numpy 2.x's
    // radial scratch
    float* d_centroids = nullptr;
    double* d_thr = nullptr;
    int32_t* d_rank = nullptr;

    // host staging for the *_host entry point
    hipStream_t copy_stream = nullptr, compute_stream = nullptr;
    int16_t* pin_in[2] = {nullptr, nullptr};
    int16_t* dev_in[2] = {nullptr, nullptr};
    int32_t* dev_len[2] = {nullptr, nullptr};
    hipEvent_t ev_copied[2] = {nullptr, nullptr}, ev_done[2] = {nullptr, nullptr};
    size_t stage_samples = 0;

    bool profiling = false;
    int profile_only = -1;                 // -1: every stage; >= 0: only that stage
    int profile_every = 1;                 // bracket every n-th launch of a profiled stage ("profile_every": the two event records per batch
                                           // of bench.py's in-region bracket cost ~2 % of the headline at 1, a quarter of that at 4)
    unsigned profile_seen = 0;             // ... launches of profiled stages so far
    std::vector<hipEvent_t> event_pool;    // recycled timing events
    std::vector<Stage> stages;
    hipStream_t last_stream = nullptr;
};

namespace {

int fail(amph_ctx* c, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (c) c->err = buf;
    else g_create_error = buf;
    return code;
}

#define HIP_TRY(c, expr)                                                                          \
    do {                                                                                          \
        hipError_t e_ = (expr);                                                                   \
        if (e_ != hipSuccess) return fail((c), AMPH_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

template <typename T>
hipError_t upload(T** dptr, const std::vector<T>& h) {
    hipError_t e = hipMalloc(reinterpret_cast<void**>(dptr), std::max<size_t>(h.size(), 1) * sizeof(T));
    if (e != hipSuccess) return e;
    if (!h.empty()) e = hipMemcpy(*dptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    return e;
}

// ---- Slaney mel scale (librosa.hz_to_mel / mel_to_hz, htk=False) -------------
constexpr double kFsp = 200.0 / 3.0;
constexpr double kMinLogHz = 1000.0;
const double kMinLogMel = kMinLogHz / kFsp;
const double kLogStep = std::log(6.4) / 27.0;
double hz_to_mel(double f) { return f >= kMinLogHz ? kMinLogMel + std::log(f / kMinLogHz) / kLogStep : f / kFsp; }
double mel_to_hz(double m) { return m >= kMinLogMel ? kMinLogHz * std::exp(kLogStep * (m - kMinLogMel)) : kFsp * m; }
// numpy.linspace(start, stop, num, endpoint=True)
std::vector<double> linspace(double start, double stop, int num) {
    std::vector<double> v(num);
    if (num == 1) { v[0] = start; return v; }
    const double step = (stop - start) / (double)(num - 1);
    for (int i = 0; i < num; ++i) v[i] = (double)i * step + start;
    v[num - 1] = stop;
    return v;
}

// librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax, htk=False, norm="slaney", dtype=float32)
std::vector<float> mel_filterbank(int sr, int n_fft, int n_mels, double fmin, double fmax) {
    const int F = 1 + n_fft / 2;
    std::vector<float> w((size_t)n_mels * F, 0.0f);
    const std::vector<double> fftfreqs = linspace(0.0, (double)sr / 2.0, F);
    std::vector<double> mel_f = linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2);
    for (auto& m : mel_f) m = mel_to_hz(m);
    for (int i = 0; i < n_mels; ++i) {
        const double fd0 = mel_f[i + 1] - mel_f[i], fd1 = mel_f[i + 2] - mel_f[i + 1];
        const double enorm = 2.0 / (mel_f[i + 2] - mel_f[i]);
        for (int k = 0; k < F; ++k) {
            const double lower = -(mel_f[i] - fftfreqs[k]) / fd0;
            const double upper = (mel_f[i + 2] - fftfreqs[k]) / fd1;
            const float tri = (float)std::max(0.0, std::min(lower, upper));     // float32 store
            w[(size_t)i * F + k] = (float)((double)tri * enorm);               // weights *= enorm
        }
    }
    return w;
}

// float -> bf16, round to nearest even (NaN stays NaN)
uint16_t f32_to_bf16_host(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}

// bf16 copy of a [n_pad][K_pad] fp32 GEMM weight matrix, rows re-padded to a multiple of 64 elements
hipError_t upload_bf16(uint16_t** dptr, const std::vector<float>& w, int n_pad, int K_pad, int K_pad_b) {
    std::vector<uint16_t> h((size_t)n_pad * K_pad_b, 0);
    for (int n = 0; n < n_pad; ++n)
        for (int k = 0; k < K_pad; ++k) h[(size_t)n * K_pad_b + k] = f32_to_bf16_host(w[(size_t)n * K_pad + k]);
    return upload(dptr, h);
}

int pick_head_splits(int n_slices) {
    int best = 1;
    for (int s = 1; s <= 16; ++s)
        if (n_slices % s == 0) best = s;
    return best;
}

hipEvent_t take_event(amph_ctx* c) {
    hipEvent_t e = nullptr;
    if (!c->event_pool.empty()) { e = c->event_pool.back(); c->event_pool.pop_back(); }
    else (void)hipEventCreate(&e);
    return e;
}

struct StageTimer {
    amph_ctx* c;
    int stage;
    hipStream_t s;
    hipEvent_t a = nullptr, b = nullptr;
    bool on;
    StageTimer(amph_ctx* c_, int stage_, hipStream_t s_) : c(c_), stage(stage_), s(s_) {
        on = c->profiling && (c->profile_only < 0 || c->profile_only == stage);
        if (on && c->profile_every > 1) on = (c->profile_seen++ % (unsigned)c->profile_every) == 0;
        if (on) {
            a = take_event(c);
            b = take_event(c);
            (void)hipEventRecord(a, s);
            c->last_stream = s;
        }
    }
    void cancel() {                            // nothing was launched in this stage after all: give the events back
        if (on) { c->event_pool.push_back(a); c->event_pool.push_back(b); on = false; }
    }
    ~StageTimer() {
        if (on) {
            (void)hipEventRecord(b, s);
            c->stages[stage].pending.emplace_back(a, b);
        }
    }
};

enum { ST_RMS = 0, ST_STFT = 1, ST_DBNORM = 2, ST_CONV0 = 3 };

int stage_head_gemm(const amph_ctx* c) { return ST_CONV0 + c->spec.n_conv; }
int stage_head_reduce(const amph_ctx* c) { return ST_CONV0 + c->spec.n_conv + 1; }

// k_stft_pair's mel tables (n_fft 2048).  The kernel gives a filter to a lane QUAD; per 16-bin step lane q of the quad reads the
// (|A|^2, |B|^2) pairs of bins 2q, 2q + 1 (16 bytes at 16 q of the step's 128 bytes) and 8 + 2q, 8 + 2q + 1 (at 64 + 16 q): a quad's read is
// 64 CONTIGUOUS bytes.  A ds_read_b128 lane group holds four quads - {0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15} - and is
// conflict-free when their four 64-byte windows tile the 256-byte bank row, i.e. when the start bins, in units of 4 bins (32 bytes) mod 8,
// are {0, 2, 4, 6} or {1, 3, 5, 7}.  A filter's start may move down by multiples of 4 bins as long as its round's 16-bin steps still
// cover its last bin, and which filter sits on which quad is free (the band index travels in the table): a small exact search per round
// of 16 filters finds such an arrangement (all four rounds of the 64-band bench preset; tools/mel_read_bank_model.py: 302 -> 104 array
// cycles of power reads per pair of frames).  No arrangement: the filters keep their order and starts (correct, conflicts as before).
// meta: [rounds][16 quads] x (start, steps, weight offset, band or -1); weights: per quad 16 floats per step in lane order
// (q, j) -> bin offset {2q, 2q + 1, 8 + 2q, 8 + 2q + 1}[j], strides as d_mel_w's (an odd number of 64-byte slots per filter).
namespace {
struct PairMelSearch {
    int nf;                       // filters of the round
    std::vector<int> cand[16];    // per filter: allowed residues o = (start / 4) % 8
    int grp_of[16], o_of[16];
    bool used[16];
    bool fits(const int* g, int par, int* o_out) const {
        static const int perms[24][4] = {{0,1,2,3},{0,1,3,2},{0,2,1,3},{0,2,3,1},{0,3,1,2},{0,3,2,1},{1,0,2,3},{1,0,3,2},{1,2,0,3},{1,2,3,0},{1,3,0,2},{1,3,2,0},
                                         {2,0,1,3},{2,0,3,1},{2,1,0,3},{2,1,3,0},{2,3,0,1},{2,3,1,0},{3,0,1,2},{3,0,2,1},{3,1,0,2},{3,1,2,0},{3,2,0,1},{3,2,1,0}};
        for (const auto& pm : perms) {
            bool ok = true;
            for (int j = 0; j < 4 && ok; ++j) {
                const int need = par + 2 * pm[j];
                if (g[j] >= nf) continue;         // a padding slot takes any residue
                ok = std::find(cand[g[j]].begin(), cand[g[j]].end(), need) != cand[g[j]].end();
            }
            if (ok) { for (int j = 0; j < 4; ++j) o_out[j] = par + 2 * pm[j]; return true; }
        }
        return false;
    }
    bool solve(int grp) {
        if (grp == 4) return true;
        int a = 0;
        while (a < 16 && used[a]) ++a;
        int g[4] = {a, 0, 0, 0};
        used[a] = true;
        for (int b = a + 1; b < 16; ++b) if (!used[b])
            for (int cc = b + 1; cc < 16; ++cc) if (!used[cc])
                for (int d = cc + 1; d < 16; ++d) if (!used[d]) {
                    g[1] = b; g[2] = cc; g[3] = d;
                    for (int par = 0; par < 2; ++par) {
                        int o[4];
                        if (!fits(g, par, o)) continue;
                        used[b] = used[cc] = used[d] = true;
                        for (int j = 0; j < 4; ++j) { grp_of[g[j]] = grp; o_of[g[j]] = o[j]; }
                        if (solve(grp + 1)) return true;
                        used[b] = used[cc] = used[d] = false;
                    }
                }
        used[a] = false;
        return false;
    }
};
}  // namespace

static void build_pair_mel(const std::vector<float>& basis, int n_mels, int F, std::vector<int32_t>& meta, std::vector<float>& wv) {
    static const int kGroupQuads[4][4] = {{0, 3, 5, 6}, {1, 2, 4, 7}, {8, 11, 13, 14}, {9, 10, 12, 15}};
    const int n_rounds = (n_mels + 15) / 16;
    meta.assign((size_t)n_rounds * 16 * 4, 0);
    wv.clear();
    for (int r = 0; r < n_rounds; ++r) {
        const int m0 = 16 * r, nf = std::min(16, n_mels - m0);
        int first[16], last[16], round_steps = 1;
        for (int j = 0; j < nf; ++j) {
            first[j] = last[j] = -1;
            for (int k = 0; k < F; ++k)
                if (basis[(size_t)(m0 + j) * F + k] != 0.0f) { if (first[j] < 0) first[j] = k; last[j] = k; }
            if (first[j] >= 0) round_steps = std::max(round_steps, (last[j] - (first[j] & ~3) + 16) / 16);
        }
        PairMelSearch ss;
        ss.nf = nf;
        for (int j = 0; j < 16; ++j) { ss.used[j] = false; ss.grp_of[j] = -1; ss.o_of[j] = 0; }
        for (int j = 0; j < nf; ++j) {
            if (first[j] < 0) { for (int o = 0; o < 8; ++o) ss.cand[j].push_back(o); continue; }       // an empty filter: any start
            const int lo = std::max(0, last[j] - 16 * round_steps + 1);
            for (int st = first[j] & ~3; st >= lo; st -= 4)
                if (std::find(ss.cand[j].begin(), ss.cand[j].end(), (st / 4) % 8) == ss.cand[j].end()) ss.cand[j].push_back((st / 4) % 8);
        }
        int quad_of[16], start_of[16];
        if (ss.solve(0)) {
            int fill[4] = {0, 0, 0, 0};
            for (int j = 0; j < 16; ++j) {
                quad_of[j] = kGroupQuads[ss.grp_of[j]][fill[ss.grp_of[j]]++];
                if (j >= nf) continue;
                int st = first[j] < 0 ? 0 : (first[j] & ~3);
                while (first[j] >= 0 && (st / 4) % 8 != ss.o_of[j]) st -= 4;        // the largest allowed start with the residue (it exists: cand)
                start_of[j] = first[j] < 0 ? 4 * ss.o_of[j] : st;
            }
        } else {
            for (int j = 0; j < 16; ++j) { quad_of[j] = j; start_of[j] = (j < nf && first[j] >= 0) ? (first[j] & ~3) : 0; }
        }
        int filt_of_quad[16];
        for (int j = 0; j < 16; ++j) filt_of_quad[quad_of[j]] = j;
        for (int t = 0; t < 16; ++t) {                 // quads in order: the weight strides keep a lane group's reads on different banks
            const int j = filt_of_quad[t];
            const bool live = j < nf;
            const int st = live ? start_of[j] : 0;
            int32_t* e = meta.data() + ((size_t)r * 16 + t) * 4;
            e[0] = st; e[1] = round_steps; e[2] = (int32_t)wv.size(); e[3] = live ? m0 + j : -1;
            for (int i = 0; i < round_steps; ++i)
                for (int q = 0; q < 4; ++q)
                    for (int u = 0; u < 4; ++u) {
                        const int k = st + 16 * i + (u < 2 ? 2 * q + u : 8 + 2 * q + (u - 2));
                        wv.push_back(live && k < F ? basis[(size_t)(m0 + j) * F + k] : 0.0f);
                    }
            if ((round_steps & 1) == 0) wv.insert(wv.end(), 16, 0.0f);
        }
    }
}

int build_frontend(amph_ctx* c) {
    const amph_frontend_config& f = c->cfg;
    // the tuned kernels cover n_fft 1024 / 2048 with an even hop; every other power of two from 64 to 8192 and any hop take the general
    // kernel (k_stft_generic).  librosa also accepts an n_fft that is no power of two (map_detector_core.py:219-228): not implemented.
    if (f.n_fft < 64 || f.n_fft > 8192 || (f.n_fft & (f.n_fft - 1)))
        return fail(c, AMPH_ERR_UNSUPPORTED, "n_fft=%d: this build implements the powers of two from 64 to 8192", f.n_fft);
    if (f.hop_length <= 0) return fail(c, AMPH_ERR_INVALID_ARG, "hop_length must be positive");
    c->generic_stft = (f.n_fft != 2048 && f.n_fft != 1024) || (f.hop_length & 1);
    if (f.n_samples <= 0 || f.n_mels <= 0 || f.target_frames <= 0 || f.sr <= 0)
        return fail(c, AMPH_ERR_INVALID_ARG, "non-positive front-end parameter");
    if (f.pad_mode == 0 && f.n_samples <= f.n_fft / 2)
        return fail(c, AMPH_ERR_UNSUPPORTED, "reflect padding needs n_samples > n_fft/2");
    if (f.pad_mode != 0 && f.pad_mode != 1) return fail(c, AMPH_ERR_INVALID_ARG, "pad_mode must be 0 or 1");
    c->n_frames = 1 + f.n_samples / f.hop_length;
    c->frames_per_block = 8;

    const int n_fft = f.n_fft, Nc = n_fft / 2, TLO = Nc / 256, L = 16 * TLO;
    const double PI = 3.14159265358979323846;
    if (c->generic_stft) {
        std::vector<float> gw(n_fft);
        std::vector<float2> gt(n_fft / 2);
        for (int n = 0; n < n_fft; ++n) gw[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * (double)n / (double)n_fft));      // scipy get_window("hann", fftbins=True)
        for (int k = 0; k < n_fft / 2; ++k) {
            const double a = -2.0 * PI * (double)k / (double)n_fft;
            gt[k] = make_float2((float)std::cos(a), (float)std::sin(a));
        }
        HIP_TRY(c, upload(&c->d_gen_win, gw));
        HIP_TRY(c, upload(&c->d_gen_tw, gt));
    }
    // packed table image (layout = the kernel's LDS image): half window | tw_b | tw_post; tw_a separately
    std::vector<float> tab(c->generic_stft ? 0 : stft_mel_table_bytes(n_fft) / 4, 0.0f);
    std::vector<float2> twa(c->generic_stft ? 0 : 15 * L);
    if (!c->generic_stft) {
        float* win = tab.data();                                  // w[0..1024]; w[n] == w[N - n] covers the rest
        for (int n = 0; n <= n_fft / 2; ++n) win[n] = (float)(0.5 - 0.5 * std::cos(2.0 * PI * (double)n / (double)n_fft));
        for (int k1 = 1; k1 < 16; ++k1)
            for (int t = 0; t < L; ++t) {
                const double a = -2.0 * PI * (double)((t * k1) % Nc) / (double)Nc;
                twa[(k1 - 1) * L + t] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        float* twb = tab.data() + ((((Nc + 1) * 4 + 15) & ~15) / 4);
        for (int k2a = 0; k2a < 16; ++k2a)
            for (int t = 0; t < TLO; ++t) {
                const double a = -2.0 * PI * (double)((t * k2a) % L) / (double)L;
                twb[2 * (k2a * TLO + t)] = (float)std::cos(a);
                twb[2 * (k2a * TLO + t) + 1] = (float)std::sin(a);
            }
        float* twp = twb + 16 * TLO * 2;
        for (int k = 0; k <= Nc / 2; ++k) {
            const double a = -2.0 * PI * (double)k / (double)n_fft;
            twp[2 * k] = (float)std::cos(a);
            twp[2 * k + 1] = (float)std::sin(a);
        }
    }
    if (!c->generic_stft) {
        HIP_TRY(c, upload(&c->d_tw_a, twa));
        HIP_TRY(c, upload(&c->d_tables, tab));
    }

    const int F = 1 + n_fft / 2;
    c->mel_basis = mel_filterbank(f.sr, n_fft, f.n_mels, f.fmin, f.fmax);
    // mel rows for the kernel: start bin rounded down to a multiple of 4, weights zero padded to whole
    // 16-bin steps (4 lanes x float4) and to the longest row of the filter's round of 16 (so the
    // step count is uniform across a wave); weights and power spectrum are read as aligned float4
    std::vector<int32_t> st(f.n_mels), ln(f.n_mels), wo(f.n_mels);
    std::vector<float> wv;
    int p_tail = 0;
    for (int m0 = 0; m0 < f.n_mels; m0 += 16) {
        int first[16], last[16], round_steps = 0;
        for (int j = 0; j < 16 && m0 + j < f.n_mels; ++j) {
            first[j] = last[j] = -1;
            for (int k = 0; k < F; ++k)
                if (c->mel_basis[(size_t)(m0 + j) * F + k] != 0.0f) { if (first[j] < 0) first[j] = k; last[j] = k; }
            if (first[j] >= 0) round_steps = std::max(round_steps, (last[j] - (first[j] & ~3) + 16) / 16);
        }
        for (int j = 0; j < 16 && m0 + j < f.n_mels; ++j) {
            const int m = m0 + j;
            const int st4 = first[j] < 0 ? 0 : (first[j] & ~3);
            st[m] = st4;
            ln[m] = round_steps;
            wo[m] = (int32_t)wv.size();
            for (int e = 0; e < 16 * round_steps; ++e) {
                const int k = st4 + e;
                wv.push_back(k < F ? c->mel_basis[(size_t)m * F + k] : 0.0f);
            }
            // the 16 filters of a round are read by 16 lane quads, 64 bytes per quad and step: with an even number of steps per
            // filter the quads of one ds_read_b128 lane group (quads 0, 3, 5, 6 / 1, 2, 4, 7) start on the same banks (4-way
            // conflict at 4 and 8 steps); one 64-byte pad makes the filter stride an odd number of 64-byte slots
            if ((round_steps & 1) == 0) wv.insert(wv.end(), 16, 0.0f);
            p_tail = std::max(p_tail, st4 + 16 * round_steps - F);
        }
    }
    c->mel_p_tail = std::max(p_tail, 0);
    if (!c->generic_stft && F + c->mel_p_tail > stft_mel_scratch_floats(n_fft))
        return fail(c, AMPH_ERR_UNSUPPORTED, "mel rows reach past the per-frame scratch");
    HIP_TRY(c, upload(&c->d_mel_start, st));
    HIP_TRY(c, upload(&c->d_mel_len, ln));
    HIP_TRY(c, upload(&c->d_mel_wofs, wo));
    c->mel_w_floats = (int)wv.size();
    HIP_TRY(c, upload(&c->d_mel_w, wv));
    // k_stft_pair (n_fft 2048, hop 384): lane l reads samples 128 j' + 2 l + e and owns column col(l) = 2 (l & 31) + (l >> 5)
    // of the first stage (tools/fft_pair_model.py)
    if (!c->generic_stft && stft_pair_ok(n_fft, f.hop_length, c->mel_p_tail)) {
        std::vector<float> pt(stft_pair_table_bytes() / 4, 0.0f);
        // lane l reads dword dw(l) of every 64-dword sample row (k_stft_pair: lanes l, l + 16 share a column pair)
        auto lane_dw = [](int l) { return (l & 15) + 16 * (l >> 5) + 32 * ((l >> 4) & 1); };
        for (int j = 0; j < 16; ++j)
            for (int l = 0; l < 64; ++l)
                for (int e = 0; e < 2; ++e) {
                    const int n = 128 * j + 2 * lane_dw(l) + e;
                    const float w = (float)(0.5 - 0.5 * std::cos(2.0 * PI * (double)n / (double)n_fft));
                    pt[(size_t)(j * 64 + l) * 2 + e] = w * (1.0f / 32768.0f);      // x / 32768 * w == x * (w / 32768): exact scaling
                }
        // stage-2 twiddles as the kernel multiplies them: per (h, m) the PAIR of mm = 2 m, 2 m + 1: (cos, cos, sin, sin); h = 0: (1, 1, 0, 0)
        float* tw2 = pt.data() + 16 * 64 * 2;
        for (int m = 0; m < 16; ++m) {
            float* e0 = tw2 + 4 * m;
            float* e1 = tw2 + 4 * (17 + m);                 // (row stride 17: k_frontend.hip kPairTw2S)
            e0[0] = e0[1] = 1.0f; e0[2] = e0[3] = 0.0f;
            for (int u = 0; u < 2; ++u) {
                const double a = -2.0 * PI * (double)(2 * m + u) / 64.0;
                e1[u] = (float)std::cos(a); e1[2 + u] = (float)std::sin(a);
            }
        }
        // stage-1 twiddles: per (k, lane) the pair k1 = k, k + 16: (wr_k, wr_k+16, wi_k, wi_k+16), k = 0..15 (k1 = 0: (1, 0))
        std::vector<float2> tw1(2 * 16 * 64);
        for (int k = 0; k < 16; ++k)
            for (int l = 0; l < 64; ++l) {
                const int col = 2 * ((l & 15) + 16 * (l >> 5)) + ((l >> 4) & 1);
                float* e = reinterpret_cast<float*>(tw1.data()) + ((size_t)k * 64 + l) * 4;
                for (int u = 0; u < 2; ++u) {
                    const double a = -2.0 * PI * (double)((col * (k + 16 * u)) % n_fft) / (double)n_fft;
                    e[u] = (float)std::cos(a); e[2 + u] = (float)std::sin(a);
                }
            }
        HIP_TRY(c, upload(&c->d_pair_tables, pt));
        HIP_TRY(c, upload(&c->d_pair_tw1, tw1));
        std::vector<int32_t> pm;
        std::vector<float> pw;
        build_pair_mel(c->mel_basis, f.n_mels, F, pm, pw);
        c->pair_mel_w_floats = (int)pw.size();
        HIP_TRY(c, upload(&c->d_pair_mel_meta, pm));
        HIP_TRY(c, upload(&c->d_pair_mel_w, pw));
    }
    // k_stft_pair2 (n_fft 1024, hop 256): lane' l (0..31) reads samples 64 j' + 2 l + e and owns column 2 (l & 15) + (l >> 4)
    if (!c->generic_stft && stft_pair2_ok(n_fft, f.hop_length, c->mel_p_tail)) {
        std::vector<float> pt(stft_pair2_table_bytes() / 4, 0.0f);
        for (int j = 0; j < 16; ++j)
            for (int l = 0; l < 32; ++l)
                for (int e = 0; e < 2; ++e) {
                    const int n = 64 * j + 2 * l + e;
                    const float w = (float)(0.5 - 0.5 * std::cos(2.0 * PI * (double)n / (double)n_fft));
                    pt[(size_t)(j * 32 + l) * 2 + e] = w * (1.0f / 32768.0f);
                }
        std::vector<float2> tw1(31 * 32);
        for (int k1 = 1; k1 < 32; ++k1)
            for (int l = 0; l < 32; ++l) {
                const int col = 2 * (l & 15) + (l >> 4);
                const double a = -2.0 * PI * (double)((col * k1) % n_fft) / (double)n_fft;
                tw1[(size_t)(k1 - 1) * 32 + l] = make_float2((float)std::cos(a), (float)std::sin(a));
            }
        HIP_TRY(c, upload(&c->d_pair_tables, pt));
        HIP_TRY(c, upload(&c->d_pair_tw1, tw1));
    }
    return AMPH_OK;
}

int build_encoder(amph_ctx* c, const float* hw, size_t n_floats) {
    const amph_encoder_spec& sp = c->spec;
    if (sp.n_conv < 1 || sp.n_conv > AMPH_MAX_CONV_LAYERS) return fail(c, AMPH_ERR_INVALID_ARG, "n_conv out of range");
    if (sp.in_frames != c->cfg.target_frames || sp.in_mels != c->cfg.n_mels)
        return fail(c, AMPH_ERR_INVALID_ARG, "encoder input %dx%d != front end %dx%d", sp.in_frames, sp.in_mels,
                    c->cfg.target_frames, c->cfg.n_mels);
    if (sp.conv[0].c_in != 1) return fail(c, AMPH_ERR_UNSUPPORTED, "first conv layer must have c_in == 1");
    // expected blob size
    size_t need = 0;
    int H = sp.in_frames, W = sp.in_mels;
    std::vector<int> hs, wsz;
    for (int i = 0; i < sp.n_conv; ++i) {
        const amph_conv_layer& l = sp.conv[i];
        if (l.kernel < 1 || l.stride < 1 || l.pad < 0 || l.c_out < 1) return fail(c, AMPH_ERR_INVALID_ARG, "bad conv layer %d", i);
        if (i > 0 && l.c_in != sp.conv[i - 1].c_out) return fail(c, AMPH_ERR_INVALID_ARG, "conv %d c_in mismatch", i);
        if (l.c_out % 4) return fail(c, AMPH_ERR_UNSUPPORTED, "c_out must be a multiple of 4 (layer %d)", i);
        need += (size_t)l.c_out * l.c_in * l.kernel * l.kernel + 5 * (size_t)l.c_out;
        H = (H + 2 * l.pad - l.kernel) / l.stride + 1;
        W = (W + 2 * l.pad - l.kernel) / l.stride + 1;
        if (H < 1 || W < 1) return fail(c, AMPH_ERR_INVALID_ARG, "conv %d output is empty", i);
        hs.push_back(H);
        wsz.push_back(W);
    }
    const int D = sp.latent_dim;
    if (D < 1) return fail(c, AMPH_ERR_INVALID_ARG, "latent_dim < 1");
    const int C_last = sp.conv[sp.n_conv - 1].c_out;
    const int flat = C_last * H * W;
    need += 2 * ((size_t)D * flat + D);
    if (n_floats != need)
        return fail(c, AMPH_ERR_INVALID_ARG, "weight blob has %zu floats, spec needs %zu", n_floats, need);
    c->flat = flat;
    c->latent = D;

    const float* pw = hw;
    int Hin = sp.in_frames, Win = sp.in_mels;
    for (int i = 0; i < sp.n_conv; ++i) {
        const amph_conv_layer& l = sp.conv[i];
        const int kk = l.kernel * l.kernel;
        const float* w = pw;                       pw += (size_t)l.c_out * l.c_in * kk;
        const float* bias = pw;                    pw += l.c_out;
        const float* g = pw;                       pw += l.c_out;
        const float* beta = pw;                    pw += l.c_out;
        const float* mean = pw;                    pw += l.c_out;
        const float* var = pw;                     pw += l.c_out;
        // BatchNorm eval as torch's CPU kernel folds it: alpha = weight / sqrt(var + eps),
        // y = x * alpha + (bias_bn - mean * alpha); the conv bias is folded into the shift.
        const int n_pad = (l.c_out + 63) / 64 * 64;
        std::vector<float> alpha(n_pad, 0.0f), shift(n_pad, 0.0f);
        for (int o = 0; o < l.c_out; ++o) {
            const float invstd = 1.0f / std::sqrt(var[o] + sp.bn_eps);
            alpha[o] = invstd * g[o];
            shift[o] = (beta[o] - mean[o] * alpha[o]) + bias[o] * alpha[o];
        }
        if (i == 0) {
            std::vector<float> wf((size_t)kk * l.c_out);
            for (int o = 0; o < l.c_out; ++o)
                for (int t = 0; t < kk; ++t) wf[(size_t)t * l.c_out + o] = w[(size_t)o * kk + t];
            HIP_TRY(c, upload(&c->d_first_w, wf));
            HIP_TRY(c, upload(&c->d_first_scale, alpha));
            HIP_TRY(c, upload(&c->d_first_shift, shift));
            // fp16 range of the fused pair's layer 1 (k_conv_gemm_f4s splits the image, these weights and the layer-1 activation it builds in
            // LDS into fp16 terms): |a1[c]| <= (sum_t |w1[t][c]|) |alpha_c| X + |shift_c| for |image| <= X.  img_limit = the largest X for which every
            // channel stays below 65 504 (and the image itself does); 0 when a weight is not splittable.
            {
                double lim = 65504.0 * (1.0 - 1e-4);
                bool okw = true;
                for (int o = 0; o < l.c_out; ++o) {
                    double l1 = 0.0;
                    for (int t = 0; t < kk; ++t) { const double a = std::fabs((double)w[(size_t)o * kk + t]); l1 += a; okw = okw && a < 65504.0; }
                    const double gain = l1 * std::fabs((double)alpha[o]), sh = std::fabs((double)shift[o]);
                    okw = okw && std::isfinite(gain) && std::isfinite(sh) && sh < 65504.0;
                    if (okw && gain > 0.0) lim = std::min(lim, (65504.0 * (1.0 - 1e-4) - sh) / gain);
                }
                c->img_limit = okw && lim > 0.0 ? (float)lim : 0.0f;
            }
            ConvFirstParams& p = c->first;
            p.w = c->d_first_w; p.scale = c->d_first_scale; p.shift = c->d_first_shift;
            p.H = Hin; p.W = Win; p.OH = hs[i]; p.OW = wsz[i]; p.C_out = l.c_out;
            p.k = l.kernel; p.stride = l.stride; p.pad = l.pad;
        } else {
            if (l.c_in % 4) return fail(c, AMPH_ERR_UNSUPPORTED, "c_in must be a multiple of 4 (layer %d)", i);
            GemmLayer gl;
            const int K = kk * l.c_in, K_pad = (K + 31) / 32 * 32;
            std::vector<float> wg((size_t)n_pad * K_pad, 0.0f);
            for (int o = 0; o < l.c_out; ++o)
                for (int ci = 0; ci < l.c_in; ++ci)
                    for (int t = 0; t < kk; ++t)
                        wg[(size_t)o * K_pad + (size_t)t * l.c_in + ci] = w[((size_t)o * l.c_in + ci) * kk + t];
            HIP_TRY(c, upload(&gl.d_w, wg));
            // two-term fp16 weights exist only for |w| < 65 504 (h = fp16(w) is inf above): a layer with larger or non-finite weights keeps
            // d_w_h2 / d_w_f4 null and runs on the f32-MFMA kernels (the three-term bf16 split of the fused pair has fp32's range)
            bool h2_ok = true;
            for (const float v : wg) h2_ok = h2_ok && std::fabs(v) < 65504.0f;      // (false for NaN too)
            if (l.c_in % 32 == 0) {
                if (i == 1) {                                       // k_conv_gemm_f3s: the fused pair only
                    std::vector<uint16_t> w3((size_t)n_pad * K_pad * 3);
                    conv_gemm_s3_split_weights(wg.data(), n_pad, K_pad, w3.data());
                    HIP_TRY(c, upload(&gl.d_w_s3, w3));
                }
                if (h2_ok) {
                    std::vector<uint16_t> w2((size_t)n_pad * K_pad * 2);
                    conv_gemm_h2_split_weights(wg.data(), n_pad, K_pad, w2.data());
                    HIP_TRY(c, upload(&gl.d_w_h2, w2));
                }
                if (i == 1 && K_pad == 288) {
                    std::vector<uint16_t> wb((size_t)n_pad * K_pad);
                    conv_gemm_f4b_fragment_weights(wg.data(), n_pad, K_pad, wb.data());
                    HIP_TRY(c, upload(&gl.d_w_f4b, wb));
                }
                if (i > 1 && (l.c_in == 64 || l.c_in == 128) && K_pad == 9 * l.c_in) {                                 // k_conv_gemm_g4, bf16 encoder
                    std::vector<uint16_t> wb((size_t)n_pad * K_pad);
                    conv_gemm_f4b_fragment_weights(wg.data(), n_pad, K_pad, wb.data());
                    HIP_TRY(c, upload(&gl.d_w_f4b, wb));
                }
                if (h2_ok && ((i == 1 && K_pad == 288) || (i > 1 && (l.c_in == 64 || l.c_in == 128) && K_pad == 9 * l.c_in))) {   // k_conv_gemm_f4s / _g4 / _g5: two fp16 terms
                    std::vector<uint16_t> wf((size_t)n_pad * K_pad * 2);
                    conv_gemm_h2_fragment_weights(wg.data(), n_pad, K_pad, wf.data());
                    HIP_TRY(c, upload(&gl.d_w_f4, wf));
                }
            }
            gl.K_pad_b = (K + 63) / 64 * 64;
            HIP_TRY(c, upload_bf16(&gl.d_w_bf16, wg, n_pad, K_pad, gl.K_pad_b));
            HIP_TRY(c, upload(&gl.d_scale, alpha));
            HIP_TRY(c, upload(&gl.d_shift, shift));
            gl.n_pad = n_pad;
            ConvGemmParams& p = gl.prm;
            p.w = gl.d_w; p.scale = gl.d_scale; p.shift = gl.d_shift;
            p.H = Hin; p.W = Win; p.C_in = l.c_in; p.OH = hs[i]; p.OW = wsz[i]; p.N = l.c_out;
            p.k = l.kernel; p.stride = l.stride; p.pad = l.pad; p.K = K; p.K_pad = K_pad;
            p.epilogue = 0; p.splits = 1;
            c->gemm.push_back(gl);
        }
        Hin = hs[i];
        Win = wsz[i];
    }
    // head: [fc_mu; fc_logvar] as one [2D][flat] matrix, columns permuted from the NCHW
    // flatten index c*H*W + h*W + w to the NHWC one (h*W + w)*C + c.
    {
        const float* wmu = pw;  pw += (size_t)D * flat;
        const float* bmu = pw;  pw += D;
        const float* wlv = pw;  pw += (size_t)D * flat;
        const float* blv = pw;  pw += D;
        if (flat % 4) return fail(c, AMPH_ERR_UNSUPPORTED, "flatten size must be a multiple of 4");
        const int N = 2 * D, n_pad = (N + 63) / 64 * 64, K_pad = (flat + 31) / 32 * 32;
        std::vector<float> wg((size_t)n_pad * K_pad, 0.0f), bias(N);
        const int HW = H * W;
        for (int n = 0; n < N; ++n) {
            const float* src = n < D ? wmu + (size_t)n * flat : wlv + (size_t)(n - D) * flat;
            for (int ch = 0; ch < C_last; ++ch)
                for (int px = 0; px < HW; ++px) wg[(size_t)n * K_pad + (size_t)px * C_last + ch] = src[(size_t)ch * HW + px];
            bias[n] = n < D ? bmu[n] : blv[n - D];
        }
        GemmLayer gl;
        HIP_TRY(c, upload(&gl.d_w, wg));
        bool h2_ok = true;
        for (const float v : wg) h2_ok = h2_ok && std::fabs(v) < 65504.0f;
        if (flat % 32 == 0 && h2_ok) {                   // the head only runs through k_conv_gemm_s3: the two-term fp16 weights
            std::vector<uint16_t> w2((size_t)n_pad * K_pad * 2);
            conv_gemm_h2_split_weights(wg.data(), n_pad, K_pad, w2.data());
            HIP_TRY(c, upload(&gl.d_w_h2, w2));
        }
        gl.K_pad_b = (flat + 63) / 64 * 64;
        HIP_TRY(c, upload_bf16(&gl.d_w_bf16, wg, n_pad, K_pad, gl.K_pad_b));
        HIP_TRY(c, upload(&c->d_head_bias, bias));
        HIP_TRY(c, upload(&c->d_zeros, std::vector<float>(64, 0.0f)));
        gl.n_pad = n_pad;
        ConvGemmParams& p = gl.prm;
        p.w = gl.d_w; p.scale = nullptr; p.shift = nullptr;
        p.H = 1; p.W = 1; p.C_in = flat; p.OH = 1; p.OW = 1; p.N = N; p.k = 1; p.stride = 1; p.pad = 0;
        p.K = flat; p.K_pad = K_pad; p.epilogue = 1;
        c->head_splits = pick_head_splits(K_pad / 32);
        p.splits = c->head_splits;
        gl.bm = 64;
        c->gemm.push_back(gl);
    }
    return AMPH_OK;
}

// the normalised / padded PCM16 rows K0 (or the copy kernel) writes for the STFT kernel: sized at create for the padded
// form (n_samples + n_fft per row); only a caller whose rows are longer than that AND who turned "stft_prepad" off grows it
int ensure_norm16(amph_ctx* c, Workspace& w, size_t need, hipStream_t s) {
    if (w.norm16_cap >= need) return AMPH_OK;
    if (w.norm16) {
        HIP_TRY(c, hipStreamSynchronize(s));               // an earlier batch of this lane may still read the old rows
        (void)hipFree(w.norm16);
    }
    w.norm16 = nullptr; w.norm16_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&w.norm16, need * sizeof(int16_t)));
    w.norm16_cap = need;
    return AMPH_OK;
}

// K0's per-row chunk sums (rows longer than one LDS pass take the two-kernel path): max_batch x ceil(n_in / 8192) floats
int ensure_chunk_sums(amph_ctx* c, Workspace& w, int n_in, hipStream_t s) {
    const size_t need = (size_t)c->max_batch * (size_t)rms_scale_chunks(n_in);
    if (w.chunk_sums_cap >= need) return AMPH_OK;
    if (w.chunk_sums) { HIP_TRY(c, hipStreamSynchronize(s)); (void)hipFree(w.chunk_sums); }
    w.chunk_sums = nullptr; w.chunk_sums_cap = 0;
    HIP_TRY(c, hipMalloc((void**)&w.chunk_sums, need * sizeof(float)));
    w.chunk_sums_cap = need;
    return AMPH_OK;
}

int alloc_lane(amph_ctx* c, Workspace& w, bool own_stream) {
    const int MB = c->max_batch;
    const amph_frontend_config& f = c->cfg;
    if (own_stream) HIP_TRY(c, hipStreamCreateWithFlags(&w.stream, hipStreamNonBlocking));
    HIP_TRY(c, hipEventCreateWithFlags(&w.done, hipEventDisableTiming));
    HIP_TRY(c, hipMalloc((void**)&w.scale, (size_t)MB * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**)&w.ok, (size_t)MB * sizeof(int32_t)));
    HIP_TRY(c, hipMalloc((void**)&w.melpow, (size_t)MB * c->n_frames * f.n_mels * sizeof(float)));
    HIP_TRY(c, hipMalloc((void**)&w.image, (size_t)MB * f.target_frames * f.n_mels * sizeof(float)));
    {
        const int rc = ensure_norm16(c, w, (size_t)MB * (size_t)(((long long)f.n_samples + f.n_fft + 1) & ~1ll), w.stream);
        if (rc != AMPH_OK) return rc;
    }
    if (c->has_encoder) {
        int H = c->spec.in_frames, W = c->spec.in_mels;
        for (int i = 0; i < c->spec.n_conv; ++i) {
            const amph_conv_layer& l = c->spec.conv[i];
            H = (H + 2 * l.pad - l.kernel) / l.stride + 1;
            W = (W + 2 * l.pad - l.kernel) / l.stride + 1;
            // the layer that READS this buffer (conv i+1, or the head as a H x W "convolution") takes its padded taps
            // from a zero guard in front of it (k_conv_gemm_p3)
            const size_t guard = i + 1 < c->spec.n_conv ? amph::conv_gemm_x_guard(c->spec.conv[i + 1].kernel, W, l.c_out)
                                                        : amph::conv_gemm_x_guard(1, 1, H * W * l.c_out);     // the head: 1 x 1 over the flattened map
            char* raw = nullptr;
            HIP_TRY(c, hipMalloc((void**)&raw, guard + (size_t)MB * H * W * l.c_out * sizeof(float)));
            HIP_TRY(c, hipMemset(raw, 0, guard));
            w.act_raw.push_back(raw);
            w.act_guard.push_back((int)guard);
            w.act.push_back(reinterpret_cast<float*>(raw + guard));
        }
        HIP_TRY(c, hipMalloc((void**)&w.partial, (size_t)c->head_splits * MB * 2 * c->latent * sizeof(float)));
    }
    return AMPH_OK;
}

void free_lane(Workspace& w) {
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    fr(w.scale); fr(w.ok); fr(w.melpow); fr(w.image); fr(w.chunk_sums);
    for (auto p : w.act_raw) fr(p);
    fr(w.partial);
    fr(w.norm16);
    if (w.done) (void)hipEventDestroy(w.done);
    if (w.stream) (void)hipStreamDestroy(w.stream);
    w = Workspace();
}

int set_lanes(amph_ctx* c, int n) {
    if (n < 1 || n > 8) return fail(c, AMPH_ERR_INVALID_ARG, "lanes must be in 1..8");
    (void)hipDeviceSynchronize();
    while ((int)c->lanes.size() > n) { free_lane(c->lanes.back()); c->lanes.pop_back(); }
    while ((int)c->lanes.size() < n) {
        c->lanes.emplace_back();
        const int rc = alloc_lane(c, c->lanes.back(), c->lanes.size() > 1);
        if (rc != AMPH_OK) return rc;
    }
    return AMPH_OK;
}

int alloc_workspace(amph_ctx* c) {
    HIP_TRY(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
    return set_lanes(c, 4);
}

void build_stages(amph_ctx* c) {
    const amph_frontend_config& f = c->cfg;
    const double N = f.n_samples, T = c->n_frames, Fb = 1 + f.n_fft / 2, M = f.n_mels;
    auto add = [&](const char* name, double flops, double bytes) {
        Stage s;
        s.name = name;
        s.flops_per_chunk = flops;
        s.bytes_per_chunk = bytes;
        c->stages.push_back(s);
    };
    add("rms_scale", 2.0 * N, 2.0 * N);
    add("stft_mel", T * (f.n_fft + 2.5 * f.n_fft * std::log2((double)f.n_fft)) + 3.0 * Fb * T + 4.0 * Fb * T,
        2.0 * N + 4.0 * T * M);
    add("dbnorm", 12.0 * M * T, 4.0 * T * M + 4.0 * f.target_frames * M);
    if (c->has_encoder) {
        int H = c->spec.in_frames, W = c->spec.in_mels;
        for (int i = 0; i < c->spec.n_conv; ++i) {
            const amph_conv_layer& l = c->spec.conv[i];
            const double in_b = 4.0 * H * W * l.c_in;
            H = (H + 2 * l.pad - l.kernel) / l.stride + 1;
            W = (W + 2 * l.pad - l.kernel) / l.stride + 1;
            char nm[32];
            snprintf(nm, sizeof nm, "conv%d", i + 1);
            add(nm, 2.0 * l.kernel * l.kernel * l.c_in * l.c_out * H * W, in_b + 4.0 * H * W * l.c_out);
        }
        add("head_gemm", 2.0 * c->flat * 2.0 * c->latent, 4.0 * c->flat + 8.0 * c->latent * c->head_splits);
        add("head_reduce", 2.0 * c->latent * c->head_splits, 8.0 * c->latent * (c->head_splits + 1));
    }
}

// encoder on ws_image-like input; x is [nb][T][M]
// lanes: how many lanes the CALL spreads its batches over (picks the tile form of the staged GEMM layers: "wide_gemm");
// img_checked: the image is known to stay below c->img_limit (the front end's own images when img_bound < img_limit; for a caller's
// tensor amph_encoder_forward runs k_range_check first and the fused pair raises nothing itself for layer 1)
int run_encoder(amph_ctx* c, Workspace& w, const float* d_x, int nb, float* d_mu, float* d_logvar, hipStream_t s, int lanes, bool img_checked) {
    const int n_gemm = (int)c->gemm.size();
    if (c->bf16_encoder) {
        // layers 1 + 2: the fused split-operand kernel (fp32 layer 1 recomputed in the staging, fp32-accurate layer 2 on the bf16
        // matrix cores) with a bf16 epilogue - the layer-1 activation, the largest tensor of the path, never exists; the
        // unfused bf16 pair (k_conv_first_bf16 24 us + its GEMM 19 us per 256 chunks) only where that kernel does not apply
        int gi0 = 0;
        if (c->fuse_conv12 && c->split_fused && n_gemm >= 2 && c->gemm[0].d_w_s3) {
            ConvGemmParams p = c->gemm[0].prm;
            p.x = w.act[0]; p.x_guard = w.act_guard[0]; p.M = nb * p.OH * p.OW; p.zeros = c->d_zeros; p.y = w.act[1];
            p.img = d_x; p.w1 = c->first.w; p.scale1 = c->first.scale; p.shift1 = c->first.shift;
            p.H0 = c->first.H; p.W0 = c->first.W; p.k1 = c->first.k; p.s1 = c->first.stride; p.p1 = c->first.pad;
            if (c->patch_fused && c->bf16_pair && c->gemm[0].d_w_f4b && conv_gemm_f4b_ok(p)) {
                // layers 1 + 2 in bf16 proper: image, weights and the layer-1 activation rounded to bf16, fp32 accumulate / BN
                StageTimer t(c, ST_CONV0 + 1, s);
                p.y_bf16 = 1;
                launch_conv_gemm_f4b(p, c->gemm[0].d_w_f4b, c->patch_wgs, s);
                gi0 = 1;
            } else if (c->patch_fused && c->gemm[0].d_w_f4 && img_checked && conv_gemm_f4s_ok(p)) {
                StageTimer t(c, ST_CONV0 + 1, s);
                p.y_bf16 = 1;
                launch_conv_gemm_f4s(p, c->gemm[0].d_w_f4, c->patch_wgs, s);
                gi0 = 1;
            } else if (conv_gemm_f3s_ok(p)) {
                StageTimer t(c, ST_CONV0 + 1, s);
                p.w = reinterpret_cast<const float*>(c->gemm[0].d_w_s3);
                p.y_bf16 = 1;
                launch_conv_gemm_f3s(p, s);
                gi0 = 1;
            }
        }
        if (gi0 == 0) {
            StageTimer t(c, ST_CONV0, s);
            ConvFirstParams p = c->first;
            p.x = d_x;
            p.y = w.act[0];                            // bf16 NHWC in the (fp32-sized) activation buffer
            launch_conv_first_bf16(p, nb, s);
        }
        for (int gi = gi0; gi < n_gemm; ++gi) {
            const bool head = gi == n_gemm - 1;
            ConvGemmParams p = c->gemm[gi].prm;
            p.x = w.act[gi];
            p.w = reinterpret_cast<const float*>(c->gemm[gi].d_w_bf16);
            p.K_pad = c->gemm[gi].K_pad_b;
            p.M = nb * p.OH * p.OW;
            p.img = nullptr;
            int bm = 128;
            if (head) {
                p.y = w.partial;
                bm = 64;
                int sp = 1;                            // split-K over 64-element slices, at most the fp32 path's count
                for (int q = 1; q <= c->head_splits; ++q) if ((p.K_pad / 64) % q == 0) sp = q;
                p.splits = sp;
            } else {
                p.y = w.act[gi + 1];
                const long long wg128 = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64);
                if (wg128 < c->gemm_min_wg128) bm = 64;
            }
            {
                StageTimer t(c, head ? stage_head_gemm(c) : ST_CONV0 + 1 + gi, s);
                const int kind = (!head && c->gemm[gi].d_w_f4b) ? conv_gemm_g4_kind(c->gemm[gi].prm_m(p.M)) : 0;
                if (((kind == 3 || kind == 5 || kind == 6) && (c->patch_gemm_bf16 & 1)) || ((kind == 4 || kind == 7) && (c->patch_gemm_bf16 & 2))) {
                    ConvGemmParams pg = c->gemm[gi].prm;              // (fp32 element counts: K, K_pad as the layer has them)
                    pg.x = w.act[gi]; pg.y = w.act[gi + 1]; pg.M = p.M; pg.y_bf16 = 1;
#ifdef AMPH_STAMPS
                    if (!c->d_gstamps) (void)hipMalloc((void**)&c->d_gstamps, (size_t)n_gemm * 8192 * 8 * 8);
                    pg.stamps = c->d_gstamps + (size_t)gi * 8192 * 8;
                    c->gstamp_wgs[gi] = c->patch_gemm_wgs > 0 ? c->patch_gemm_wgs : 128;
#endif
                    launch_conv_gemm_g4(pg, c->gemm[gi].d_w_f4b, c->patch_gemm_wgs, 1, s);
                } else launch_conv_gemm_bf16(p, bm, s);
            }
            if (head) {
                StageTimer t2(c, stage_head_reduce(c), s);
                HeadReduceParams hp;
                hp.partial = w.partial; hp.bias = c->d_head_bias; hp.splits = p.splits; hp.D = c->latent;
                hp.mu = d_mu; hp.logvar = d_logvar;
                launch_head_reduce(hp, nb, s);
            }
        }
        return AMPH_OK;
    }
    // tile height per GEMM layer, then: can layer 2 recompute layer 1 on the fly?
    std::vector<ConvGemmParams> prm(n_gemm);
    std::vector<int> bms(n_gemm);
    for (int gi = 0; gi < n_gemm; ++gi) {
        const bool head = gi == n_gemm - 1;
        ConvGemmParams& p = prm[gi];
        p = c->gemm[gi].prm;
        p.x = w.act[gi];
        p.x_guard = w.act_guard[gi];
        p.M = nb * p.OH * p.OW;
        p.img = nullptr;
        p.zeros = c->d_zeros;
        p.range_flag = head ? nullptr : c->d_range_flag;       // (only the two-term fp16 kernels look at it; the head's output is not split again)
        int bm = 128;
        if (head) { p.y = w.partial; bm = 64; }
        else {
            p.y = w.act[gi + 1];
            // 128-row tiles move 25 % fewer operand bytes than 64-row ones (the kernels are bound by
            // operand delivery) as long as they still give every CU a workgroup
            const long long wg128 = (long long)((p.M + 127) / 128) * ((p.N + 63) / 64);
            if (wg128 < c->gemm_min_wg128) bm = 64;
            // (that rule was written for the first-generation kernel; the second-generation one only has 64-row tiles and
            // beats the 128-row first-generation tile at every batch size)
            if (conv_gemm_p3_ok(p, 64)) bm = 64;
        }
        bms[gi] = bm;
    }
    bool fuse = false;
    if (c->fuse_conv12 && n_gemm >= 2) {
        ConvGemmParams& p = prm[0];
        p.img = d_x; p.w1 = c->first.w; p.scale1 = c->first.scale; p.shift1 = c->first.shift;
        p.H0 = c->first.H; p.W0 = c->first.W; p.k1 = c->first.k; p.s1 = c->first.stride; p.p1 = c->first.pad;
        if (conv_gemm_f3_ok(p, 64)) { bms[0] = 64; fuse = true; }      // second-generation fused kernel
        else {
            fuse = conv_gemm_can_fuse(p, bms[0]);
            if (!fuse && bms[0] == 128 && conv_gemm_can_fuse(p, 64)) { bms[0] = 64; fuse = true; }
        }
        if (!fuse) p.img = nullptr;
    }
    if (!fuse) {
        StageTimer t(c, ST_CONV0, s);
        ConvFirstParams p = c->first;
        p.x = d_x;
        p.y = w.act[0];
        p.range_flag = c->split_gemm ? c->d_range_flag : nullptr;     // its output may be the next layer's split operand
        launch_conv_first(p, nb, s);
    }
    for (int gi = 0; gi < n_gemm; ++gi) {
        const bool head = gi == n_gemm - 1;
#ifdef AMPH_ABL_SKIP
        if (g_abl_skip & (4 << gi)) continue;
#endif
#ifdef AMPH_STAMPS
        if (!c->d_gstamps) (void)hipMalloc((void**)&c->d_gstamps, (size_t)n_gemm * 8192 * 8 * 8);
        prm[gi].stamps = c->d_gstamps + (size_t)gi * 8192 * 8;
        c->gstamp_wgs[gi] = ((prm[gi].M + bms[gi] - 1) / bms[gi]) * ((prm[gi].N + 63) / 64) * prm[gi].splits;
        if (fuse && gi == 0 && conv_gemm_f3_ok(prm[gi], bms[gi])) c->gstamp_wgs[gi] = std::min(c->gstamp_wgs[gi], 2 * amph::g_num_cus);
#endif
        StageTimer t(c, head ? stage_head_gemm(c) : ST_CONV0 + 1 + gi, s);
        const bool fused_here = fuse && gi == 0;
        if (c->split_gemm && c->split_fused && c->patch_fused && fused_here && c->gemm[gi].d_w_f4 && img_checked && conv_gemm_f4s_ok(prm[gi])) {
#ifdef AMPH_STAMPS
            c->gstamp_wgs[gi] = std::min(prm[gi].M / 64, c->patch_wgs > 0 ? c->patch_wgs : std::max(1, std::min(amph::g_num_cus / 4, (prm[gi].M / 64 + 15) / 16)));
            c->gstamp_f4s = true;                               // rows of 16: consumer wave 0 in [0..7], producer wave 4 in [8..15]
#endif
            launch_conv_gemm_f4s(prm[gi], c->gemm[gi].d_w_f4, c->patch_wgs, s);
            continue;
        }
        if (c->split_gemm && c->split_fused && fused_here && c->gemm[gi].d_w_s3 && conv_gemm_f3s_ok(prm[gi])) {
            ConvGemmParams ps = prm[gi];
            ps.w = reinterpret_cast<const float*>(c->gemm[gi].d_w_s3);
            launch_conv_gemm_f3s(ps, s);
            continue;
        }
        // GEMM layers on the f16 matrix cores with two-term fp16 split operands (fp32-accurate): k_encoder_split.hip, k_encoder_patch.hip
        if (c->split_gemm && !fused_here && !head && c->gemm[gi].d_w_f4) {
            const int kind = conv_gemm_g4_kind(prm[gi]);
            if (((kind == 3 || kind == 5 || kind == 6) && (c->patch_gemm & 1)) || (kind == 4 && (c->patch_gemm & 2))) {
#ifdef AMPH_STAMPS
                c->gstamp_wgs[gi] = c->patch_gemm_wgs > 0 ? c->patch_gemm_wgs : 64;
#endif
                launch_conv_gemm_g4(prm[gi], c->gemm[gi].d_w_f4, c->patch_gemm_wgs, 3, s);
                continue;
            }
        }
        if (c->split_gemm && !fused_here && c->gemm[gi].d_w_h2 && conv_gemm_s3_ok(prm[gi])) {
            ConvGemmParams ps = prm[gi];
            ps.w = reinterpret_cast<const float*>(c->gemm[gi].d_w_h2);
            const bool wide = c->wide_gemm > 0 || (c->wide_gemm < 0 && lanes > 1 && (long long)((ps.M + 127) / 128) * (ps.N / 128) * 4 >= amph::g_num_cus);
            if (wide && !head && conv_gemm_s3w_ok(ps)) launch_conv_gemm_s3w(ps, s);
            else launch_conv_gemm_s3(ps, s);
            continue;
        }
        launch_conv_gemm(prm[gi], bms[gi], fused_here, s);
    }
#ifdef AMPH_ABL_SKIP
    if (g_abl_skip & 64) return AMPH_OK;
#endif
    {
        StageTimer t(c, stage_head_reduce(c), s);
        HeadReduceParams p;
        p.partial = w.partial; p.bias = c->d_head_bias; p.splits = c->head_splits; p.D = c->latent;
        p.mu = d_mu; p.logvar = d_logvar;
        launch_head_reduce(p, nb, s);
    }
    return AMPH_OK;
}

// can this context run the dB / z-score stage inside the STFT kernel (one workgroup per row, the row's mel power in LDS)?
// LDS image of the pair kernel this context would use (k_stft_pair at n_fft 2048, k_stft_pair2 at 1024)
size_t pair_lds(const amph_ctx* c, int nw, bool fused) {
    return c->cfg.n_fft == 2048 ? stft_pair_lds_bytes(c->n_frames, c->cfg.n_mels, c->pair_mel_w_floats, nw, fused)
                                : stft_pair2_lds_bytes(c->n_frames, c->cfg.n_mels, c->mel_w_floats, nw, fused, c->stft_pair2_regdb);
}
size_t pair_tw1_lds(const amph_ctx* c) { return c->cfg.n_fft == 2048 ? stft_pair_tw1_lds_bytes() : stft_pair2_tw1_lds_bytes(); }
// do PCM16 rows that K0 / the pad kernel wrote go through a pair kernel?
bool pair_path(const amph_ctx* c, bool pcm16) {
    return pcm16 && c->stft_pair && c->d_pair_tables && c->stft_prenorm && c->stft_prepad && c->stft_indep;
}
// does the context's pair kernel have a K0F form (raw rows, RMS scale and re-quantisation inside the kernel)?
bool pair_can_fuse_k0(const amph_ctx* c, int n_in, bool fused_db) {
    if (c->cfg.n_fft == 2048) return stft_pair_can_fuse_k0(n_in, c->stft_pair_waves);
    return c->cfg.n_fft == 1024 && c->cfg.hop_length == 256 &&
           stft_pair2_can_fuse_k0(n_in, c->n_frames, c->cfg.n_mels, c->stft_pair_waves, fused_db, c->stft_pair2_regdb);
}
// pcm16_rows: the call's input is PCM16 (those rows take a pair kernel when the context has one)
bool can_fuse_db(const amph_ctx* c, bool pcm16_rows = false) {
    if (!c->stft_fuse_db || !c->stft_indep || c->generic_stft) return false;
    if (pair_path(c, pcm16_rows)) return pair_lds(c, c->stft_pair_waves, true) <= 160 * 1024;
    return stft_mel_can_fuse_db(c->cfg.n_fft, c->n_frames, c->cfg.n_mels, c->mel_w_floats, c->stft_waves);
}

// db_out (nullable): the standardised image is written by the STFT kernel itself (fused dB stage), melpow is not
StftParams make_stft(amph_ctx* c, const WaveSrc& src, float* melpow, float* db_out = nullptr, int db_layout = 1) {
    StftParams p{};
    p.src = src;
    p.n_samples = c->cfg.n_samples; p.n_fft = c->cfg.n_fft; p.hop = c->cfg.hop_length; p.n_frames = c->n_frames;
    p.pad_mode = c->cfg.pad_mode; p.frames_per_block = c->frames_per_block;
    p.tiles_per_row = (c->n_frames + c->frames_per_block - 1) / c->frames_per_block;
    p.db_out = db_out; p.db_layout = db_layout; p.db_target_frames = c->cfg.target_frames;
    p.indep = c->stft_indep ? 1 : 0; p.grid_wgs = c->stft_grid;
    if (db_out) {                                  // one tile = one whole row: its frames go round-robin over the waves
        const int round = c->stft_waves * (c->cfg.n_fft == 2048 ? 1 : 2);
        p.frames_per_block = (c->n_frames + round - 1) / round * round;
        p.tiles_per_row = 1;
    }
    p.tables = c->d_tables; p.tw_a = c->d_tw_a; p.mel_w_floats = c->mel_w_floats; p.p_tail = c->mel_p_tail;
    p.n_mels = c->cfg.n_mels; p.mel_start = c->d_mel_start; p.mel_len = c->d_mel_len; p.mel_wofs = c->d_mel_wofs;
    p.mel_w = c->d_mel_w; p.melpow = melpow;
    // the pair kernel where it applies (launch_stft_mel checks the source rows); with the fused dB stage its LDS image must fit
    if (c->stft_pair && c->d_pair_tables && pair_lds(c, c->stft_pair_waves, db_out != nullptr) <= 160 * 1024) {
        p.pair_tables = c->d_pair_tables; p.pair_tw1 = c->d_pair_tw1; p.pair_waves = c->stft_pair_waves;
        p.pair_mel_meta = c->d_pair_mel_meta; p.pair_mel_w = c->d_pair_mel_w; p.pair_mel_w_floats = c->pair_mel_w_floats;
        p.pair_tw_lds = c->stft_pair_tw_lds ? 1 : 0;
        p.pair2_regdb = c->stft_pair2_regdb ? 1 : 0;
    }
#ifdef AMPH_STAMPS
    if (!c->d_stamps) (void)hipMalloc((void**)&c->d_stamps, (size_t)(c->max_batch * 64 + 6 * 4096) * 16 * 8);   // + the pair kernel's timeline region
    p.stamps = c->d_stamps;
#endif
    return p;
}

// front end for one batch: wave rows -> melpow (-> image)
// db_out != nullptr (only with can_fuse_db): the kernel writes the standardised image there and melpow stays untouched
// does a batch of this call take K0's fused form (chunk sums launched, fold + apply inside the pair kernel)?
bool k0_fused_ok(const amph_ctx* c, bool pcm16, int nb, int n_in, const void* d_wave, bool fused_db) {
    return pcm16 && c->stft_prenorm && !c->generic_stft && c->stft_indep && c->stft_prepad && c->stft_k0_fuse && pair_path(c, true) &&
           pair_lds(c, c->stft_pair_waves, fused_db) <= 160 * 1024 && pair_can_fuse_k0(c, n_in, fused_db) &&
           (n_in & 1) == 0 && (reinterpret_cast<uintptr_t>(d_wave) & 3u) == 0 &&    // (its dword loads: every row 4-byte aligned)
           (long long)nb * n_in * 2 < (1ll << 31);                                  // (32-bit buffer offsets)
}

// pre_sums (nullable): the batch's chunk sums, already computed by a launch in front of the call's batches (encode_common)
int run_frontend(amph_ctx* c, Workspace& w, const void* d_wave, bool pcm16, int nb, int n_in, const int32_t* d_len,
                 int apply_rms, int32_t* d_rms_ok, float* melpow, hipStream_t s, float* db_out = nullptr, int db_layout = 1,
                 const float* pre_sums = nullptr) {
    WaveSrc src{};
    src.wave = d_wave; src.row_stride = n_in; src.lengths = d_len; src.n_in = n_in;
    src.scale = nullptr; src.requant = 0;
    bool k0 = false;                               // K0 runs inside the STFT kernel ("stft_k0_fuse")
    const float* k0_sums = nullptr;                // ... except its chunk sums (mode 2)
    int32_t* k0_ok = nullptr;
    if (apply_rms) {
        StageTimer t(c, ST_RMS, s);
        int32_t* okp = d_rms_ok ? d_rms_ok : w.ok;
        k0_ok = okp;
        int16_t* norm = nullptr;
        // PCM16 rows: materialise the normalised, re-quantised samples once (the file 00_normalize_dataset_rms.py would have
        // written), inside the K0 launch, instead of normalising inside the STFT kernel, whose independent waves would each
        // redo it for the 81 % of a frame they share with their neighbours (6 VALU instructions per sample against 2 for a
        // plain PCM16 decode)
        NormPad pad{};
        if (pcm16 && c->stft_prenorm && !c->generic_stft) {
            // with the independent-wave STFT kernel the rows are written as that kernel wants to read them: fix_length'd and
            // centre padded, so that no frame needs index reflection
            if (c->stft_indep && c->stft_prepad) {
                pad.lead = c->cfg.n_fft / 2; pad.n_fix = c->cfg.n_samples; pad.mode = c->cfg.pad_mode;
                pad.out_stride = ((long long)pad.n_fix + 2 * pad.lead + 1) & ~1ll;
            }
            const int rc = ensure_norm16(c, w, (size_t)c->max_batch * (pad.lead > 0 ? (size_t)pad.out_stride : (size_t)n_in), s);
            if (rc != AMPH_OK) return rc;
            norm = w.norm16;
        }
        // K0's fold + apply inside the pair kernel (k_frontend.hip, K0F): only the per-chunk sums of squares are launched here; the kernel
        // reads the raw rows and re-quantises in its decode - no padded normalised copy
        const bool k0_fused = norm && pad.lead > 0 && k0_fused_ok(c, pcm16, nb, n_in, d_wave, db_out != nullptr);
        if (k0_fused && pre_sums) {
            k0 = true;
            k0_sums = pre_sums;
        } else if (k0_fused) {
            const int rcs = ensure_chunk_sums(c, w, n_in, s);
            if (rcs != AMPH_OK) return rcs;
            launch_rms_chunk_sums_pcm16((const int16_t*)d_wave, n_in, d_len, n_in, nb, w.chunk_sums, s);
            k0 = true;
            k0_sums = w.chunk_sums;
        } else if (pcm16) {
            const int rcs = ensure_chunk_sums(c, w, n_in, s);
            if (rcs != AMPH_OK) return rcs;
#ifdef AMPH_ABL_SKIP
            if (!(g_abl_skip & 1))
#endif
            launch_rms_scale_pcm16((const int16_t*)d_wave, n_in, d_len, n_in, nb, w.scale, okp, c->rms_f64_scale, norm, pad, w.chunk_sums, c->rms_chunk_path_over, s);
        }
        else launch_rms_scale_f32((const float*)d_wave, n_in, d_len, n_in, nb, w.scale, okp, c->rms_f64_scale, s);
        src.scale = w.scale;
        src.requant = 1;
        if (norm) {
            src.wave = norm; src.scale = nullptr; src.requant = 0;
            if (pad.lead > 0) { src.lead = pad.lead; src.row_stride = pad.out_stride; src.lengths = nullptr; src.n_in = pad.n_fix + 2 * pad.lead; }
        }
    } else {
        if (d_rms_ok) (void)hipMemsetAsync(d_rms_ok, 0, (size_t)nb * sizeof(int32_t), s);
        // PCM16 rows used as they are: the same fix_length'd, centre-padded copy, by a copy kernel (timed with the RMS stage) - or, for the
        // shapes the K0F form of k_stft_pair covers, no copy at all: the kernel reads the raw rows (K0F 2)
        const bool raw_rows = pcm16 && c->stft_prenorm && c->stft_prepad && c->stft_k0_fuse && pair_path(c, true) &&
                              pair_lds(c, c->stft_pair_waves, db_out != nullptr) <= 160 * 1024 && pair_can_fuse_k0(c, n_in, db_out != nullptr) &&
                              (n_in & 1) == 0 && (reinterpret_cast<uintptr_t>(d_wave) & 3u) == 0 && (long long)nb * n_in * 2 < (1ll << 31);
        if (raw_rows) {
            k0 = true;                                      // (k0_sums stays nullptr: no RMS stage)
            src.lead = c->cfg.n_fft / 2; src.lengths = nullptr;          // the geometry launch_stft_mel keys the pair kernel on; src.wave is not read
        } else if (pcm16 && c->stft_prenorm && c->stft_prepad && c->stft_indep && !c->generic_stft) {
            StageTimer t(c, ST_RMS, s);
            NormPad pad{};
            pad.lead = c->cfg.n_fft / 2; pad.n_fix = c->cfg.n_samples; pad.mode = c->cfg.pad_mode;
            pad.out_stride = ((long long)pad.n_fix + 2 * pad.lead + 1) & ~1ll;
            const int rc = ensure_norm16(c, w, (size_t)c->max_batch * (size_t)pad.out_stride, s);
            if (rc != AMPH_OK) return rc;
            launch_pad_rows_pcm16((const int16_t*)d_wave, n_in, d_len, n_in, nb, w.norm16, pad, s);
            src.wave = w.norm16; src.lead = pad.lead; src.row_stride = pad.out_stride; src.lengths = nullptr; src.n_in = pad.n_fix + 2 * pad.lead;
        }
    }
    {
        StageTimer t(c, ST_STFT, s);
#ifdef AMPH_ABL_SKIP
        if (!(g_abl_skip & 2))
#endif
        {
            StftParams sp = make_stft(c, src, melpow, db_out, db_layout);
            if (k0) {
                sp.k0_in = (const int16_t*)d_wave; sp.k0_row_stride = n_in; sp.k0_lengths = d_len; sp.k0_n_in = n_in;
                sp.k0_scale_f64 = c->rms_f64_scale; sp.k0_scale = w.scale; sp.k0_ok = k0_ok;
                sp.k0_chunk_sums = k0_sums; sp.k0_chunk_stride = rms_scale_chunks(n_in);
                sp.pair_rows2 = c->stft_pair_rows2 ? 1 : 0;
                sp.pair_cq = c->stft_pair_cq;
            }
            if (c->generic_stft) launch_stft_generic(sp, nb, pcm16 ? 1 : 0, c->d_gen_win, c->d_gen_tw, s);
            else launch_stft_mel(sp, nb, pcm16 ? 1 : 0, c->stft_waves, s);
        }
    }
    return AMPH_OK;
}

int run_dbnorm(amph_ctx* c, const float* melpow, int nb, int layout, float* out, hipStream_t s) {
    StageTimer t(c, ST_DBNORM, s);
    DbNormParams p;
    p.melpow = melpow; p.n_frames = c->n_frames; p.n_mels = c->cfg.n_mels; p.target_frames = c->cfg.target_frames;
    p.layout = layout; p.out = out;
    launch_dbnorm(p, nb, s);
    return AMPH_OK;
}

int check_launch(amph_ctx* c, const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(c, AMPH_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
    return AMPH_OK;
}

int encode_common(amph_ctx* c, const void* d_wave, bool pcm16, int64_t n_rows, int n_in, const int32_t* d_len,
                  int apply_rms, float* d_mu, float* d_logvar, int32_t* d_rms_ok, hipStream_t s) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (!c->has_encoder) return fail(c, AMPH_ERR_INVALID_ARG, "context was created without encoder weights");
    if (n_rows < 0 || n_in <= 0 || (n_rows > 0 && (!d_wave || !d_mu))) return fail(c, AMPH_ERR_INVALID_ARG, "bad encode arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t esz = pcm16 ? 2 : 4;
    const int D = c->latent;
    // fork: batches go round-robin over the lanes; lane 0 runs on the caller's stream
    const int64_t n_batches = (n_rows + c->max_batch - 1) / c->max_batch;
    const int L = (int)std::min<int64_t>((int64_t)c->lanes.size(), std::max<int64_t>(n_batches, 1));
    // K0's chunk sums of EVERY row of the call in ONE launch in front of the batches (round 4, off: -1.4 %): one launch less at the head of
    // every batch's chain, but a serial pass over all the call's rows in front of every lane, and the rows come from HBM twice
    const float* call_sums = nullptr;
    const int n_chunks = rms_scale_chunks(n_in);
    if (apply_rms && c->k0_call_sums && n_batches > 1 && !c->profiling && (long long)n_rows * n_chunks < (1ll << 28) &&
        k0_fused_ok(c, pcm16, (int)std::min<int64_t>(c->max_batch, n_rows), n_in, d_wave, can_fuse_db(c, pcm16))) {
        const size_t need = (size_t)n_rows * (size_t)n_chunks;
        if (c->call_sums_cap < need) {
            if (c->call_sums) { HIP_TRY(c, hipDeviceSynchronize()); (void)hipFree(c->call_sums); }
            c->call_sums = nullptr; c->call_sums_cap = 0;
            HIP_TRY(c, hipMalloc((void**)&c->call_sums, need * sizeof(float)));
            c->call_sums_cap = need;
        }
        launch_rms_chunk_sums_pcm16((const int16_t*)d_wave, n_in, d_len, n_in, (int)n_rows, c->call_sums, s);
        call_sums = c->call_sums;
    }
    if (L > 1) {
        HIP_TRY(c, hipEventRecord(c->ev_fork, s));
        for (int l = 1; l < L; ++l) HIP_TRY(c, hipStreamWaitEvent(c->lanes[l].stream, c->ev_fork, 0));
    }
    // a partial batch goes FIRST (10 000 rows at max_batch 256: 16 + 39 x 256): as the last one it cost its lane a whole batch
    // latency for a few rows while the other lanes had already drained; per-chunk results do not depend on the batch they are in
    const int first = (int)(n_rows % c->max_batch);
    int64_t bi = 0;
    int rc = AMPH_OK;
    for (int64_t off = 0, nb_ = 0; off < n_rows && rc == AMPH_OK; off += nb_, ++bi) {
        const int nb = (bi == 0 && first > 0) ? first : (int)std::min<int64_t>(c->max_batch, n_rows - off);
        nb_ = nb;
        Workspace& w = c->lanes[bi % L];
        hipStream_t ls = (bi % L == 0) ? s : w.stream;
        const char* wv = (const char*)d_wave + (size_t)off * n_in * esz;
        // a stage that fails (an allocation, a launch) stops the call: the later stages of the batch would read stale buffers
        const bool fdb = can_fuse_db(c, pcm16);        // dB / z-score / crop-pad inside the STFT kernel: no mel power in HBM
        // (lanes on CU-masked streams - halves, quarters, a front-end / encoder split of the CUs - were measured in round 3 and run
        // 5-7x slower than this: profiles/r03_cu_partition_sweep.txt)
        rc = run_frontend(c, w, wv, pcm16, nb, n_in, d_len ? d_len + off : nullptr, apply_rms, d_rms_ok ? d_rms_ok + off : nullptr,
                          w.melpow, ls, fdb ? w.image : nullptr, 1, call_sums ? call_sums + (size_t)off * n_chunks : nullptr);
        if (rc == AMPH_OK && !fdb) rc = run_dbnorm(c, w.melpow, nb, 1, w.image, ls);
        if (rc == AMPH_OK) rc = run_encoder(c, w, w.image, nb, d_mu + off * D, d_logvar ? d_logvar + off * D : nullptr, ls, L,
                                            c->img_bound < c->img_limit);
        if (rc == AMPH_OK) rc = check_launch(c, "encode");
    }
    for (int l = 1; l < L; ++l) {           // join (on failure too: the caller's stream must not run ahead of the lanes)
        hipError_t e = hipEventRecord(c->lanes[l].done, c->lanes[l].stream);
        if (e == hipSuccess) e = hipStreamWaitEvent(s, c->lanes[l].done, 0);
        if (e != hipSuccess && rc == AMPH_OK) rc = fail(c, AMPH_ERR_HIP, "encode: joining lane %d: %s", l, hipGetErrorString(e));
    }
    return rc;
}

}  // namespace

// ============================================================================
// C-ABI
// ============================================================================
extern "C" {

int amph_abi_version(void) { return AMPH_ABI_VERSION; }

const char* amph_last_error(const amph_ctx* ctx) { return ctx ? ctx->err.c_str() : g_create_error.c_str(); }

int amph_create(const amph_frontend_config* cfg, const amph_encoder_spec* spec, const float* h_weights,
                size_t n_weight_floats, int device, int max_batch, amph_ctx** out_ctx) {
    if (!cfg || !out_ctx || max_batch < 1) return fail(nullptr, AMPH_ERR_INVALID_ARG, "amph_create: null config/out or max_batch < 1");
    *out_ctx = nullptr;
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev == 0) return fail(nullptr, AMPH_ERR_NO_DEVICE, "no HIP device visible");
    if (device < 0 || device >= n_dev) return fail(nullptr, AMPH_ERR_NO_DEVICE, "device %d out of range (%d visible)", device, n_dev);
    if (hipSetDevice(device) != hipSuccess) return fail(nullptr, AMPH_ERR_NO_DEVICE, "hipSetDevice(%d) failed", device);
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, device) != hipSuccess) return fail(nullptr, AMPH_ERR_NO_DEVICE, "hipGetDeviceProperties failed");
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
        return fail(nullptr, AMPH_ERR_NO_DEVICE, "device %d is %s; libamph is built for gfx950 only", device, prop.gcnArchName);
    if (prop.multiProcessorCount > 0) amph::g_num_cus = prop.multiProcessorCount;

    amph_ctx* c = new amph_ctx();
    c->device = device;
    c->max_batch = max_batch;
    c->cfg = *cfg;
    int rc = build_frontend(c);
    if (rc == AMPH_OK && h_weights) {
        if (!spec) rc = fail(c, AMPH_ERR_INVALID_ARG, "weights given without an encoder spec");
        else {
            c->spec = *spec;
            c->has_encoder = true;
            rc = build_encoder(c, h_weights, n_weight_floats);
        }
    }
    if (rc == AMPH_OK) rc = alloc_workspace(c);
    if (rc == AMPH_OK) {
        // the word the two-term fp16 kernels raise (host-coherent memory: visible to the host once the raising kernel's stream has synchronised)
        hipError_t e = hipHostMalloc((void**)&c->h_range_flag, 64, hipHostMallocMapped);
        if (e == hipSuccess) { *c->h_range_flag = 0; e = hipHostGetDevicePointer((void**)&c->d_range_flag, c->h_range_flag, 0); }
        if (e != hipSuccess) rc = fail(c, AMPH_ERR_HIP, "range flag: %s", hipGetErrorString(e));
        // power_to_db -> z-score over the T_raw x M array (map_detector_core.py:229-232): |z| <= sqrt(n - 1) whatever the signal
        c->img_bound = std::sqrt((float)c->n_frames * (float)c->cfg.n_mels);
    }
    if (rc == AMPH_OK) {
        build_stages(c);
        // STFT workgroup shape: 12 waves (one 768-thread workgroup per CU, three waves per SIMD) with one
        // wave-iteration of frames per tile when that fits the 160 KB of LDS, else 4 waves x 8 frames.
        // The kernel is bound by the LDS round trips of a frame, so the third wave per SIMD is worth
        // 73 -> 52 us per 256 one-second chunks (tools/sweep.py frames_per_block=8,12 stft_waves=4,12).
        if (!c->generic_stft) {
            const int slots = c->cfg.n_fft == 2048 ? 1 : 2;
            const size_t l12 = stft_mel_lds_bytes(c->cfg.n_fft, c->cfg.hop_length, 12 * slots, c->cfg.n_mels, c->mel_w_floats, 12);
            if (l12 <= 160 * 1024) { c->stft_waves = 12; c->frames_per_block = 12 * slots; }
        }
        int lds = c->generic_stft ? 0 : (int)stft_mel_lds_bytes(c->cfg.n_fft, c->cfg.hop_length, c->frames_per_block, c->cfg.n_mels, c->mel_w_floats, c->stft_waves);
        if (c->generic_stft) {
            if ((size_t)(c->cfg.n_fft / 2 + 1 + c->mel_p_tail) * 4 > (size_t)c->cfg.n_fft * 8) rc = fail(c, AMPH_ERR_UNSUPPORTED, "mel rows reach past the frame's LDS");
            else if (amph::stft_mel_configure(1) != hipSuccess) rc = fail(c, AMPH_ERR_HIP, "hipFuncSetAttribute failed");       // (K0's kernels: their LDS limit is raised there too)
        }
        else if (lds > 160 * 1024) rc = fail(c, AMPH_ERR_UNSUPPORTED, "frame tile needs %d bytes of LDS", lds);
        else {
            for (int nw : {4, 12})                  // the fused dB stage keeps a row's mel power in LDS too
                if (stft_mel_can_fuse_db(c->cfg.n_fft, c->n_frames, c->cfg.n_mels, c->mel_w_floats, nw))
                    lds = std::max(lds, (int)stft_mel_fused_lds_bytes(c->cfg.n_fft, c->n_frames, c->cfg.n_mels, c->mel_w_floats, nw));
            if (c->d_pair_tables) {
                // workgroup of the pair kernel: the largest whose LDS image fits WITH the fused dB stage and the stage-1 twiddles
                // (1 s rows at n_fft 2048: 12 waves; the 128-band rows at n_fft 1024, 44.5 KB of mel power per row: 8), else the
                // largest that fits without
                static const int opts2048[4] = {8, 10, 11, 12}, opts1024[2] = {8, 12};
                const int* opts = c->cfg.n_fft == 2048 ? opts2048 : opts1024;
                const int n_opts = c->cfg.n_fft == 2048 ? 4 : 2;
                int pick = 0;
                for (int fused = 1; fused >= 0 && !pick; --fused)
                    for (int i = 0; i < n_opts; ++i)
                        if (pair_lds(c, opts[i], fused != 0) + pair_tw1_lds(c) <= 160 * 1024) pick = opts[i];
                if (pick) c->stft_pair_waves = pick;
                else c->stft_pair_waves = opts[0];
                // n_fft 1024, 128 bands: 12 waves + the separate dB kernel beat 8 waves with the stage fused (22 four-frame slots per row:
                // 2 rounds against 3; 29.3 + 14.6 us against 41.0 us per 256 rows alone, 2.61 against 2.39 M chunks/s end to end in bf16 mode)
                if (c->cfg.n_fft == 1024 && pair_lds(c, 12, false) + pair_tw1_lds(c) <= 160 * 1024) c->stft_pair_waves = 12;
                for (int fused = 0; fused < 2; ++fused) {
                    size_t lp = pair_lds(c, c->stft_pair_waves, fused != 0);
                    if (lp + pair_tw1_lds(c) <= 160 * 1024) lp += pair_tw1_lds(c);
                    if (lp <= 160 * 1024) lds = std::max(lds, (int)lp);
                }
            }
            rc = amph::stft_mel_configure(lds) == hipSuccess ? AMPH_OK : fail(c, AMPH_ERR_HIP, "hipFuncSetAttribute failed");
        }
    }
    if (rc != AMPH_OK) {
        g_create_error = c->err;
        amph_destroy(c);
        return rc;
    }
    *out_ctx = c;
    return AMPH_OK;
}

void amph_destroy(amph_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    auto fr = [](void* p) { if (p) (void)hipFree(p); };
    fr(c->d_gen_win); fr(c->d_gen_tw);
    fr(c->d_tables); fr(c->d_tw_a); fr(c->d_pair_tables); fr(c->d_pair_tw1); fr(c->d_pair_mel_meta); fr(c->d_pair_mel_w);
    fr(c->d_mel_start); fr(c->d_mel_len); fr(c->d_mel_wofs); fr(c->d_mel_w);
    fr(c->d_first_w); fr(c->d_first_scale); fr(c->d_first_shift);
    for (auto& g : c->gemm) { fr(g.d_w); fr(g.d_w_s3); fr(g.d_w_h2); fr(g.d_w_f4); fr(g.d_w_f4b); fr(g.d_w_bf16); fr(g.d_scale); fr(g.d_shift); }
    fr(c->d_head_bias); fr(c->d_zeros); fr(c->call_sums);
    for (auto& w : c->lanes) free_lane(w);
    if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
    fr(c->d_centroids); fr(c->d_thr); fr(c->d_rank);
    fr(c->d_rs_win); fr(c->d_rs_delta);
    fr(c->d_fit_off); fr(c->d_fit_mu); fr(c->d_fit_stats); fr(c->d_fit_rho);
    fr(c->d_map_means); fr(c->d_map_prec); fr(c->d_map_logdet); fr(c->d_map_lprior);
    if (c->h_range_flag) (void)hipHostFree(c->h_range_flag);
    for (int i = 0; i < 2; ++i) {
        if (c->pin_in[i]) (void)hipHostFree(c->pin_in[i]);
        fr(c->dev_in[i]); fr(c->dev_len[i]);
        if (c->ev_copied[i]) (void)hipEventDestroy(c->ev_copied[i]);
        if (c->ev_done[i]) (void)hipEventDestroy(c->ev_done[i]);
    }
    if (c->copy_stream) (void)hipStreamDestroy(c->copy_stream);
    if (c->compute_stream) (void)hipStreamDestroy(c->compute_stream);
    for (auto& st : c->stages)
        for (auto& pr : st.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (auto e : c->event_pool) (void)hipEventDestroy(e);
    delete c;
}

int amph_configure(amph_ctx* c, const char* key, int value) {
    if (!c || !key) return AMPH_ERR_INVALID_ARG;
    HIP_TRY(c, hipSetDevice(c->device));
    if (!std::strcmp(key, "lanes")) return set_lanes(c, value);
    if (!std::strcmp(key, "fuse_conv12")) { c->fuse_conv12 = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_prenorm")) { c->stft_prenorm = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_prepad")) { c->stft_prepad = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_fuse_db")) { c->stft_fuse_db = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_pair")) { c->stft_pair = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_pair_tw_lds")) { c->stft_pair_tw_lds = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_k0_fuse")) { c->stft_k0_fuse = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_pair_rows2")) { c->stft_pair_rows2 = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "k0_call_sums")) { c->k0_call_sums = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "profile_every")) { if (value < 1) return AMPH_ERR_INVALID_ARG; c->profile_every = value; c->profile_seen = 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_pair_cq")) { if (value < 0 || value > 64) return AMPH_ERR_INVALID_ARG; c->stft_pair_cq = value; return AMPH_OK; }
    if (!std::strcmp(key, "stft_pair2_regdb")) { c->stft_pair2_regdb = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_pair_waves")) {
        const bool okv = c->cfg.n_fft == 2048 ? (value == 8 || value == 10 || value == 11 || value == 12) : (value == 8 || value == 12);
        if (!okv) return fail(c, AMPH_ERR_INVALID_ARG, "stft_pair_waves must be 8, 10, 11 or 12 (n_fft 2048) / 8 or 12 (n_fft 1024)");
        if (c->d_pair_tables) {
            size_t need = pair_lds(c, value, false);
            if (need > 160 * 1024) return fail(c, AMPH_ERR_UNSUPPORTED, "stft_pair_waves=%d needs %zu bytes of LDS", value, need);
            if (pair_lds(c, value, true) <= 160 * 1024) need = pair_lds(c, value, true);
            if (need + pair_tw1_lds(c) <= 160 * 1024) need += pair_tw1_lds(c);
            if (amph::stft_mel_configure((int)need) != hipSuccess) return fail(c, AMPH_ERR_HIP, "hipFuncSetAttribute failed");
        }
        c->stft_pair_waves = value;
        return AMPH_OK;
    }
    if (!std::strcmp(key, "split_gemm")) { c->split_gemm = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "split_fused")) { c->split_fused = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "patch_gemm")) { c->patch_gemm = value & 3; return AMPH_OK; }
    if (!std::strcmp(key, "wide_gemm")) { c->wide_gemm = value < 0 ? -1 : (value ? 1 : 0); return AMPH_OK; }
    if (!std::strcmp(key, "patch_gemm_bf16")) { c->patch_gemm_bf16 = value & 3; return AMPH_OK; }
    if (!std::strcmp(key, "patch_gemm_wgs")) { if (value < 0) return AMPH_ERR_INVALID_ARG; c->patch_gemm_wgs = value; return AMPH_OK; }
    if (!std::strcmp(key, "rms_chunk_path_over")) { if (value < 1) return AMPH_ERR_INVALID_ARG; c->rms_chunk_path_over = value; return AMPH_OK; }
    if (!std::strcmp(key, "bf16_pair")) { c->bf16_pair = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "patch_fused")) { c->patch_fused = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "patch_wgs")) { if (value < 0) return AMPH_ERR_INVALID_ARG; c->patch_wgs = value; return AMPH_OK; }
    if (!std::strcmp(key, "rms_numpy2")) { c->rms_f64_scale = value ? 0 : 1; return AMPH_OK; }
#ifdef AMPH_STAMPS
    if (!std::strcmp(key, "dump_gemm_stamps")) {
        (void)hipDeviceSynchronize();
        for (int gi = 0; gi < (int)c->gemm.size(); ++gi) {
            if (gi == 0 && c->gstamp_f4s) {
                const int nw = std::min(c->gstamp_wgs[0], 4096);
                std::vector<unsigned long long> h((size_t)nw * 16);
                (void)hipMemcpy(h.data(), c->d_gstamps, h.size() * 8, hipMemcpyDeviceToHost);
                double m[16] = {0};
                for (int w = 0; w < nw; ++w) for (int k = 0; k < 16; ++k) m[k] += (double)h[(size_t)w * 16 + k] / nw;
                fprintf(stderr, "  f4s (%d workgroups) mean cycles per workgroup: consumer start-up %.0f multiply %.0f partial+X %.0f Y+epilogue %.0f | lifetime %.0f ticks\n"
                                "      producer build %.0f (image fetch issue %.0f, groups %.0f, image write %.0f) barriers %.0f\n"
                                "      start-up: consumer loads issued at %.0f, landed %.0f, barrier A (image staged) passed %.0f, first patch built %.0f\n"
                                "      producer groups: window reads %.0f, layer-1 MFMAs %.0f, BN + split + stores %.0f\n",
                        nw, m[0], m[1], m[2], m[3], m[5], m[8], m[10], m[11], m[8] - m[10] - m[11], m[9], m[4], m[6], m[12], m[0], m[13], m[14], m[15]);
                c->gstamp_f4s = false;
                continue;
            }
            const int nwg = std::min(c->gstamp_wgs[gi], 8192);
            std::vector<unsigned long long> h((size_t)nwg * 8);
            (void)hipMemcpy(h.data(), c->d_gstamps + (size_t)gi * 8192 * 8, h.size() * 8, hipMemcpyDeviceToHost);
            double s8[8] = {0};
            for (int w = 0; w < nwg; ++w) for (int k = 0; k < 8; ++k) s8[k] += (double)h[(size_t)w * 8 + k];
            if (gi == 0) {                                  // fused kernel: o[6] packs three 20-bit sums (mid-tile patch work, masks, epilogue)
                double m3[3] = {0, 0, 0};
                for (int w = 0; w < nwg; ++w) {
                    const unsigned long long v = h[(size_t)w * 8 + 6];
                    m3[0] += (double)(v >> 40); m3[1] += (double)((v >> 20) & 0xfffff); m3[2] += (double)(v & 0xfffff);
                }
                double pr[4] = {0, 0, 0, 0};
                for (int w = 0; w < nwg; ++w) {
                    const unsigned long long v = h[(size_t)w * 8 + 1];
                    for (int k = 0; k < 4; ++k) pr[k] += 2.0 * (double)((v >> (48 - 16 * k)) & 0xffff);
                }
                fprintf(stderr, "  gemm 0 prologue cycles per workgroup: setup %.0f, patch fetch + write + masks %.0f, B loads + barrier %.0f, two stages + barrier %.0f\n",
                        pr[0] / nwg, pr[1] / nwg, pr[2] / nwg, pr[3] / nwg);
                fprintf(stderr, "  gemm 0 out-of-step cycles per workgroup: patch write+fetch %.0f, masks %.0f, epilogue %.0f\n",
                        m3[0] / nwg, m3[1] / nwg, m3[2] / nwg);
            }
            fprintf(stderr, "  gemm %d (%d workgroups) mean cycles per workgroup:", gi, nwg);
            for (int k = 0; k < 8; ++k) fprintf(stderr, " %8.0f", s8[k] / nwg);
            unsigned long long t0 = ~0ull, t1 = 0, l0 = ~0ull, l1 = 0;
            for (int w = 0; w < nwg; ++w) {
                t0 = std::min(t0, h[(size_t)w * 8 + 7]); t1 = std::max(t1, h[(size_t)w * 8 + 7]);
                l0 = std::min(l0, h[(size_t)w * 8 + 5]); l1 = std::max(l1, h[(size_t)w * 8 + 5]);
            }
            fprintf(stderr, "   | lifetime ticks min %llu max %llu, start spread %llu ticks (100 MHz)\n", l0, l1, t1 - t0);
            if (t1 > t0 && t1 - t0 < 100000) {                 // start-time histogram, 2 us bins; end = start + lifetime
                int hist[40] = {0}, hend[40] = {0};
                for (int w = 0; w < nwg; ++w) {
                    const unsigned long long st = h[(size_t)w * 8 + 7] - t0, en = st + h[(size_t)w * 8 + 5];
                    hist[std::min<unsigned long long>(st / 200, 39)]++;
                    hend[std::min<unsigned long long>(en / 200, 39)]++;
                }
                fprintf(stderr, "      starts per 2 us:");
                for (int b = 0; b < 40; ++b) if (hist[b]) fprintf(stderr, " [%d]=%d", 2 * b, hist[b]);
                fprintf(stderr, "\n      ends   per 2 us:");
                for (int b = 0; b < 40; ++b) if (hend[b]) fprintf(stderr, " [%d]=%d", 2 * b, hend[b]);
                fprintf(stderr, "\n");
            }
        }
        return AMPH_OK;
    }
    if (!std::strcmp(key, "dump_timeline")) {    // k_stft_pair's workgroup timeline; value = workgroups of the last launch
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)value * 16);
        (void)hipMemcpy(h.data(), c->d_stamps + 16 * 4096, h.size() * 8, hipMemcpyDeviceToHost);
        double d[8] = {0}, ticks = 0;
        int n = 0;
        for (int b = 0; b < value; ++b) {
            const unsigned long long* t = h.data() + (size_t)b * 16;
            if (!t[7]) continue;
            ++n;
            d[0] += (double)(t[1] - t[0]);
            unsigned long long prev = t[1];
            for (int r = 0; r < 4; ++r) if (t[2 + r]) { d[1 + r] += (double)(t[2 + r] - prev); prev = t[2 + r]; }
            d[5] += (double)(t[6] - prev); d[6] += (double)(t[7] - t[6]); d[7] += (double)(t[7] - t[0]); ticks += (double)t[8];
        }
        {
            std::vector<unsigned long long> w((size_t)value * 16 * 8);
            (void)hipMemcpy(w.data(), c->d_stamps + 32 * 4096, w.size() * 8, hipMemcpyDeviceToHost);
            for (int wv = 0; wv < 12; ++wv) {
                double st[3] = {0}, du[3] = {0};
                int cn[3] = {0};
                for (int b = 0; b < value; ++b) {
                    const unsigned long long* t = w.data() + ((size_t)b * 16 + wv) * 8;
                    const unsigned long long t1 = h[(size_t)b * 16 + 1];
                    for (int r = 0; r < 3; ++r) if (t[2 * r] && t[2 * r + 1] > t[2 * r]) { st[r] += (double)(t[2 * r] - t1); du[r] += (double)(t[2 * r + 1] - t[2 * r]); ++cn[r]; }
                }
                fprintf(stderr, "    wave %2d: pair rounds start (after tables) / duration:", wv);
                for (int r = 0; r < 3; ++r) if (cn[r]) fprintf(stderr, "  %6.0f / %6.0f", st[r] / cn[r], du[r] / cn[r]);
                fprintf(stderr, "\n");
            }
        }
        {
            double dd[6] = {0};
            int nn = 0;
            for (int b = 0; b < value; ++b) {
                const unsigned long long* t = h.data() + (size_t)b * 16;
                if (!t[7] || !t[9]) continue;
                ++nn;
                dd[0] += (double)(t[9] - t[6]); dd[1] += (double)(t[10] - t[9]); dd[2] += (double)(t[11] - t[10]); dd[3] += (double)(t[12] - t[11]);
                dd[4] += (double)(t[13] - t[12]); dd[5] += (double)(t[7] - t[13]);
            }
            if (nn) fprintf(stderr, "    dB stage (thread 0): max + barrier %.0f | log10 + sums %.0f | reduce + barrier %.0f | mean, squares, reduce, barriers %.0f | std %.0f | z + stores %.0f\n",
                            dd[0] / nn, dd[1] / nn, dd[2] / nn, dd[3] / nn, dd[4] / nn, dd[5] / nn);
        }
        {
            double a = 0, b = 0; int m = 0;
            for (int bb = 0; bb < value; ++bb) {
                const unsigned long long* t = h.data() + (size_t)bb * 16;
                if (t[7] && t[9] && t[10] > t[9] && c->stft_pair_cq > 1) { a += (double)(t[9] - t[1]); b += (double)(t[10] - t[9]); ++m; }
            }
            if (m) fprintf(stderr, "    continuous queue: row 0's single-wave dB stage starts %.0f cycles after the tables, takes %.0f\n", a / m, b / m);
        }
        if (n) fprintf(stderr, "  k_stft_pair timeline, mean cycles over %d workgroups: tables staged %.0f | pair rounds (last wave) %.0f %.0f %.0f %.0f | to the barrier %.0f | dB stage %.0f | "
                               "total %.0f cycles = %.0f ticks of 10 ns (%.2f GHz)\n", n, d[0] / n, d[1] / n, d[2] / n, d[3] / n, d[4] / n, d[5] / n, d[6] / n, d[7] / n, ticks / n,
                       ticks > 0 ? d[7] / ticks / 10.0 : 0.0);
        return AMPH_OK;
    }
    if (!std::strcmp(key, "dump_stamps")) {      // value = number of workgroups of the last stft launch
        (void)hipDeviceSynchronize();
        std::vector<unsigned long long> h((size_t)value * 16);
        (void)hipMemcpy(h.data(), c->d_stamps, h.size() * 8, hipMemcpyDeviceToHost);
        double sum[16] = {0};
        for (int b = 0; b < value; ++b) for (int k = 1; k < 10; ++k) sum[k] += (double)(h[(size_t)b * 16 + k] - h[(size_t)b * 16 + k - 1]);
        const char* nm_w[10] = {"", "staging+sync", "window+pack", "stageA+exwrite", "exread", "dft16B", "stageC+Zwrite", "Zread", "post+Pwrite", "mel"};
        const char* nm_p[10] = {"", "decode+window", "stage-1 DFT32", "twiddle+exchange", "radix-2+twiddle", "stage-2 DFT32", "split passes", "power+Pwrite", "fetch issue", "mel"};
        const char* const* nm = (c->stft_pair && c->d_pair_tables) ? nm_p : nm_w;      // (k_stft_pair: wave 0's first pair)
        for (int k = 1; k < 10; ++k) fprintf(stderr, "  stamp %-16s %9.0f cycles (mean over %d workgroups, first frame of wave 0)\n", nm[k], sum[k] / value, value);
        double w[6] = {0};
        for (int b = 0; b < value; ++b) for (int k = 0; k < 6; ++k) w[k] += (double)h[(size_t)b * 16 + 10 + k];
        fprintf(stderr, "  workgroup (wave 0) sums over its tiles: stage+sync %.0f (own staging work %.0f)  frames %.0f  end barrier %.0f | kernel %.0f cycles = %.0f ticks of 10 ns (clock %.2f GHz)\n",
                w[0] / value, w[5] / value, w[1] / value, w[2] / value, w[3] / value, w[4] / value, w[4] > 0 ? w[3] / w[4] / 10.0 : 0.0);
        return AMPH_OK;
    }
#endif
    if (!std::strcmp(key, "gemm_min_wg128")) { c->gemm_min_wg128 = value; return AMPH_OK; }
    if (!std::strcmp(key, "bf16_encoder")) {
        if (value) {
            if (!c->has_encoder) return fail(c, AMPH_ERR_INVALID_ARG, "no encoder in this context");
            for (auto& g : c->gemm)
                if (g.prm.C_in % 8) return fail(c, AMPH_ERR_UNSUPPORTED, "bf16 path needs every GEMM layer's c_in to be a multiple of 8");
        }
        c->bf16_encoder = value != 0;
        return AMPH_OK;
    }
    if (!std::strcmp(key, "stft_grid")) {
        if (value < 0 || value > 65536) return fail(c, AMPH_ERR_INVALID_ARG, "stft_grid must be in 0..65536 (0 = fill the chip)");
        c->stft_grid = value;
        return AMPH_OK;
    }
    if (!std::strcmp(key, "stft_indep")) { c->stft_indep = value != 0; return AMPH_OK; }
    if (!std::strcmp(key, "stft_waves")) {
        if (value != 4 && value != 12) return fail(c, AMPH_ERR_INVALID_ARG, "stft_waves must be 4 or 12");
        const int lds = (int)stft_mel_lds_bytes(c->cfg.n_fft, c->cfg.hop_length, c->frames_per_block, c->cfg.n_mels, c->mel_w_floats, value);
        if (lds > 160 * 1024) return fail(c, AMPH_ERR_UNSUPPORTED, "%d waves x %d frames need %d bytes of LDS", value, c->frames_per_block, lds);
        if (amph::stft_mel_configure(lds) != hipSuccess) return fail(c, AMPH_ERR_HIP, "hipFuncSetAttribute failed");
        c->stft_waves = value;
        return AMPH_OK;
    }
    if (!std::strcmp(key, "frames_per_block")) {
        if (value < 4 || value > 32 || (value & 3)) return fail(c, AMPH_ERR_INVALID_ARG, "frames_per_block must be a multiple of 4 in 4..32");
        const int lds = (int)stft_mel_lds_bytes(c->cfg.n_fft, c->cfg.hop_length, value, c->cfg.n_mels, c->mel_w_floats, c->stft_waves);
        if (lds > 160 * 1024) return fail(c, AMPH_ERR_UNSUPPORTED, "frame tile needs %d bytes of LDS", lds);
        if (amph::stft_mel_configure(lds) != hipSuccess) return fail(c, AMPH_ERR_HIP, "hipFuncSetAttribute failed");
        c->frames_per_block = value;
        return AMPH_OK;
    }
    return fail(c, AMPH_ERR_INVALID_ARG, "unknown option '%s'", key);
}

int amph_raw_frames(const amph_ctx* c) { return c ? c->n_frames : -1; }
int amph_latent_dim(const amph_ctx* c) { return c ? c->latent : -1; }
int amph_max_batch(const amph_ctx* c) { return c ? c->max_batch : -1; }

int amph_get_mel_basis(const amph_ctx* c, float* h_out, size_t n_floats) {
    if (!c || !h_out || n_floats != c->mel_basis.size()) return AMPH_ERR_INVALID_ARG;
    std::memcpy(h_out, c->mel_basis.data(), n_floats * sizeof(float));
    return AMPH_OK;
}

int amph_rms_normalize_f32(amph_ctx* c, const float* d_in, int n_rows, int n, float* d_out, int32_t* d_ok, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || n <= 0 || (n_rows > 0 && (!d_in || !d_out))) return fail(c, AMPH_ERR_INVALID_ARG, "bad rms arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    for (int off = 0; off < n_rows; off += c->max_batch) {
        const int nb = std::min(c->max_batch, n_rows - off);
        Workspace& w = c->lanes[0];
        int32_t* okp = d_ok ? d_ok + off : w.ok;
        launch_rms_scale_f32(d_in + (size_t)off * n, n, nullptr, n, nb, w.scale, okp, c->rms_f64_scale, s);
        launch_rms_apply_f32(d_in + (size_t)off * n, nb, n, w.scale, okp, d_out + (size_t)off * n, s);
    }
    return check_launch(c, "rms_normalize_f32");
}

int amph_rms_normalize_pcm16(amph_ctx* c, const int16_t* d_in, int n_rows, int n, int16_t* d_out, int32_t* d_ok, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || n <= 0 || (n_rows > 0 && (!d_in || !d_out))) return fail(c, AMPH_ERR_INVALID_ARG, "bad rms arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    for (int off = 0; off < n_rows; off += c->max_batch) {
        const int nb = std::min(c->max_batch, n_rows - off);
        Workspace& w = c->lanes[0];
        int32_t* okp = d_ok ? d_ok + off : w.ok;
        const int rcs = ensure_chunk_sums(c, w, n, s);
        if (rcs != AMPH_OK) return rcs;
        // (rows longer than one LDS pass: the two-kernel path writes the normalised rows itself)
        if (rms_scale_chunks(n) > 4) {
            launch_rms_scale_pcm16(d_in + (size_t)off * n, n, nullptr, n, nb, w.scale, okp, c->rms_f64_scale, d_out + (size_t)off * n, NormPad{}, w.chunk_sums, c->rms_chunk_path_over, s);
            continue;
        }
        launch_rms_scale_pcm16(d_in + (size_t)off * n, n, nullptr, n, nb, w.scale, okp, c->rms_f64_scale, nullptr, NormPad{}, nullptr, c->rms_chunk_path_over, s);
        launch_rms_apply_pcm16(d_in + (size_t)off * n, nb, n, w.scale, okp, d_out + (size_t)off * n, s);
    }
    return check_launch(c, "rms_normalize_pcm16");
}

int amph_mel_power_f32(amph_ctx* c, const float* d_wave, int n_rows, int n_in, const int32_t* d_lengths,
                       float* d_melpow_out, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || n_in <= 0 || (n_rows > 0 && (!d_wave || !d_melpow_out))) return fail(c, AMPH_ERR_INVALID_ARG, "bad mel arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t per = (size_t)c->n_frames * c->cfg.n_mels;
    for (int off = 0; off < n_rows; off += c->max_batch) {
        const int nb = std::min(c->max_batch, n_rows - off);
        const int rc = run_frontend(c, c->lanes[0], d_wave + (size_t)off * n_in, false, nb, n_in, d_lengths ? d_lengths + off : nullptr, 0,
                                    nullptr, d_melpow_out + (size_t)off * per, s);
        if (rc != AMPH_OK) return rc;
    }
    return check_launch(c, "mel_power_f32");
}

int amph_mel_f32(amph_ctx* c, const float* d_wave, int n_rows, int n_in, const int32_t* d_lengths, int layout,
                 float* d_mel_out, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || n_in <= 0 || (layout != 0 && layout != 1) || (n_rows > 0 && (!d_wave || !d_mel_out)))
        return fail(c, AMPH_ERR_INVALID_ARG, "bad mel arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t per = (size_t)c->cfg.target_frames * c->cfg.n_mels;
    for (int off = 0; off < n_rows; off += c->max_batch) {
        const int nb = std::min(c->max_batch, n_rows - off);
        const bool fdb = can_fuse_db(c);
        int rc = run_frontend(c, c->lanes[0], d_wave + (size_t)off * n_in, false, nb, n_in, d_lengths ? d_lengths + off : nullptr, 0,
                              nullptr, c->lanes[0].melpow, s, fdb ? d_mel_out + (size_t)off * per : nullptr, layout);
        if (rc == AMPH_OK && !fdb) rc = run_dbnorm(c, c->lanes[0].melpow, nb, layout, d_mel_out + (size_t)off * per, s);
        if (rc != AMPH_OK) return rc;
    }
    return check_launch(c, "mel_f32");
}

int amph_mel_pcm16(amph_ctx* c, const int16_t* d_pcm, int n_rows, int n_in, const int32_t* d_lengths, int apply_rms, int layout,
                   float* d_mel_out, float* d_melpow_out, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || n_in <= 0 || (layout != 0 && layout != 1) || (n_rows > 0 && (!d_pcm || (!d_mel_out && !d_melpow_out))))
        return fail(c, AMPH_ERR_INVALID_ARG, "bad mel arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t per = (size_t)c->cfg.target_frames * c->cfg.n_mels, per_pow = (size_t)c->n_frames * c->cfg.n_mels;
    for (int off = 0; off < n_rows; off += c->max_batch) {
        const int nb = std::min(c->max_batch, n_rows - off);
        Workspace& w = c->lanes[0];
        // the mel power is wanted: it has to reach memory, so the dB stage runs as its own kernel
        const bool fdb = !d_melpow_out && can_fuse_db(c, true);
        float* mp = d_melpow_out ? d_melpow_out + (size_t)off * per_pow : w.melpow;
        int rc = run_frontend(c, w, d_pcm + (size_t)off * n_in, true, nb, n_in, d_lengths ? d_lengths + off : nullptr, apply_rms, nullptr, mp, s,
                              fdb ? d_mel_out + (size_t)off * per : nullptr, layout);
        if (rc == AMPH_OK && !fdb && d_mel_out) rc = run_dbnorm(c, mp, nb, layout, d_mel_out + (size_t)off * per, s);
        if (rc != AMPH_OK) return rc;
    }
    return check_launch(c, "mel_pcm16");
}

int amph_encoder_forward(amph_ctx* c, const float* d_x, int n_rows, float* d_mu, float* d_logvar, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (!c->has_encoder) return fail(c, AMPH_ERR_INVALID_ARG, "context was created without encoder weights");
    if (n_rows < 0 || (n_rows > 0 && (!d_x || !d_mu))) return fail(c, AMPH_ERR_INVALID_ARG, "bad encoder arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    const size_t per = (size_t)c->cfg.target_frames * c->cfg.n_mels;
    const int D = c->latent;
    // the caller's tensor, not the front end's z-scored image: its values are tested against the limit the fused pair's fp16 terms need
    const bool split_first = c->split_gemm && c->img_limit > 0.0f;
    if (split_first && n_rows > 0) launch_range_check(d_x, (long long)n_rows * (long long)per, c->img_limit, c->d_range_flag, s);
    for (int off = 0; off < n_rows; off += c->max_batch) {
        const int nb = std::min(c->max_batch, n_rows - off);
        const int rc = run_encoder(c, c->lanes[0], d_x + (size_t)off * per, nb, d_mu + (size_t)off * D, d_logvar ? d_logvar + (size_t)off * D : nullptr, s, 1,
                                   split_first);
        if (rc != AMPH_OK) return rc;
    }
    return check_launch(c, "encoder_forward");
}

int amph_range_status(amph_ctx* c, int clear) {
    if (!c || !c->h_range_flag) return 0;
    const int v = *reinterpret_cast<volatile int*>(c->h_range_flag) != 0 ? 1 : 0;
    if (clear) *reinterpret_cast<volatile int*>(c->h_range_flag) = 0;
    return v;
}

int amph_encode_pcm16(amph_ctx* c, const int16_t* d_pcm, int64_t n_rows, int n_in, const int32_t* d_lengths, int apply_rms,
                      float* d_mu, float* d_logvar, int32_t* d_rms_ok, void* stream) {
    return encode_common(c, d_pcm, true, n_rows, n_in, d_lengths, apply_rms, d_mu, d_logvar, d_rms_ok, (hipStream_t)stream);
}

int amph_encode_f32(amph_ctx* c, const float* d_wave, int64_t n_rows, int n_in, const int32_t* d_lengths, int apply_rms,
                    float* d_mu, float* d_logvar, int32_t* d_rms_ok, void* stream) {
    return encode_common(c, d_wave, false, n_rows, n_in, d_lengths, apply_rms, d_mu, d_logvar, d_rms_ok, (hipStream_t)stream);
}

static int encode_pcm16_host_once(amph_ctx* c, const int16_t* h_pcm, int64_t n_rows, int n_in, const int32_t* h_lengths,
                                  int apply_rms, float* h_mu, float* h_logvar, int32_t* h_rms_ok);

int amph_encode_pcm16_host(amph_ctx* c, const int16_t* h_pcm, int64_t n_rows, int n_in, const int32_t* h_lengths,
                           int apply_rms, float* h_mu, float* h_logvar, int32_t* h_rms_ok) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    const int before = amph_range_status(c, 0);
    int rc = encode_pcm16_host_once(c, h_pcm, n_rows, n_in, h_lengths, apply_rms, h_mu, h_logvar, h_rms_ok);
    // this entry point synchronises, so it can act on the fp16-range flag itself: an activation the two-term split cannot hold
    // -> the call runs again on the f32-MFMA kernels, and the context stays on them (its data are outside the split's range)
    if (rc == AMPH_OK && !before && amph_range_status(c, 1)) {
        c->split_gemm = false;
        c->split_fused = false;
        rc = encode_pcm16_host_once(c, h_pcm, n_rows, n_in, h_lengths, apply_rms, h_mu, h_logvar, h_rms_ok);
    }
    return rc;
}

static int encode_pcm16_host_once(amph_ctx* c, const int16_t* h_pcm, int64_t n_rows, int n_in, const int32_t* h_lengths,
                                  int apply_rms, float* h_mu, float* h_logvar, int32_t* h_rms_ok) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (!c->has_encoder) return fail(c, AMPH_ERR_INVALID_ARG, "context was created without encoder weights");
    if (n_rows < 0 || n_in <= 0 || (n_rows > 0 && (!h_pcm || !h_mu))) return fail(c, AMPH_ERR_INVALID_ARG, "bad encode arguments");
    HIP_TRY(c, hipSetDevice(c->device));
    const int MB = c->max_batch, D = c->latent;
    const size_t need = (size_t)MB * n_in;
    if (!c->copy_stream) {
        HIP_TRY(c, hipStreamCreateWithFlags(&c->copy_stream, hipStreamNonBlocking));
        HIP_TRY(c, hipStreamCreateWithFlags(&c->compute_stream, hipStreamNonBlocking));
        for (int i = 0; i < 2; ++i) {
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_copied[i], hipEventDisableTiming));
            HIP_TRY(c, hipEventCreateWithFlags(&c->ev_done[i], hipEventDisableTiming));
            HIP_TRY(c, hipMalloc((void**)&c->dev_len[i], (size_t)MB * sizeof(int32_t)));
        }
    }
    if (c->stage_samples < need) {
        for (int i = 0; i < 2; ++i) {
            if (c->pin_in[i]) (void)hipHostFree(c->pin_in[i]);
            if (c->dev_in[i]) (void)hipFree(c->dev_in[i]);
            HIP_TRY(c, hipHostMalloc((void**)&c->pin_in[i], need * sizeof(int16_t), hipHostMallocDefault));
            HIP_TRY(c, hipMalloc((void**)&c->dev_in[i], need * sizeof(int16_t)));
        }
        c->stage_samples = need;
    }
    float *d_mu = nullptr, *d_lv = nullptr;
    int32_t* d_ok = nullptr;
    int rc = AMPH_OK;
    // rows the caller already holds in pinned (page-locked / registered) memory are copied to the device straight from there:
    // the pageable -> pinned memcpy below is a full extra pass over the input on one host core
    bool pinned_src = false;
    if (n_rows > 0) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, h_pcm) == hipSuccess) pinned_src = attr.type == hipMemoryTypeHost;
        else (void)hipGetLastError();                  // an ordinary pageable pointer: not an error
    }
    {
        hipError_t e = hipMalloc((void**)&d_mu, std::max<size_t>(1, (size_t)n_rows * D) * sizeof(float));
        if (e == hipSuccess && h_logvar) e = hipMalloc((void**)&d_lv, std::max<size_t>(1, (size_t)n_rows * D) * sizeof(float));
        if (e == hipSuccess && h_rms_ok) e = hipMalloc((void**)&d_ok, std::max<size_t>(1, (size_t)n_rows) * sizeof(int32_t));
        if (e != hipSuccess) rc = fail(c, AMPH_ERR_HIP, "encode_pcm16_host: result buffers: %s", hipGetErrorString(e));
    }
    int slot = 0;
    bool used[2] = {false, false};
    for (int64_t off = 0; off < n_rows && rc == AMPH_OK; off += MB, slot ^= 1) {
        const int nb = (int)std::min<int64_t>(MB, n_rows - off);
        if (used[slot]) (void)hipEventSynchronize(c->ev_done[slot]);          // staging slot free again
        const int16_t* src = h_pcm + (size_t)off * n_in;
        if (!pinned_src) {
            std::memcpy(c->pin_in[slot], src, (size_t)nb * n_in * sizeof(int16_t));
            src = c->pin_in[slot];
        }
        (void)hipMemcpyAsync(c->dev_in[slot], src, (size_t)nb * n_in * sizeof(int16_t), hipMemcpyHostToDevice, c->copy_stream);
        if (h_lengths)
            (void)hipMemcpyAsync(c->dev_len[slot], h_lengths + off, (size_t)nb * sizeof(int32_t), hipMemcpyHostToDevice, c->copy_stream);
        (void)hipEventRecord(c->ev_copied[slot], c->copy_stream);
        (void)hipStreamWaitEvent(c->compute_stream, c->ev_copied[slot], 0);
        rc = encode_common(c, c->dev_in[slot], true, nb, n_in, h_lengths ? c->dev_len[slot] : nullptr, apply_rms,
                           d_mu + (size_t)off * D, d_lv ? d_lv + (size_t)off * D : nullptr, d_ok ? d_ok + off : nullptr,
                           c->compute_stream);
        (void)hipEventRecord(c->ev_done[slot], c->compute_stream);
        used[slot] = true;
    }
    if (rc == AMPH_OK) {
        hipError_t e = hipStreamSynchronize(c->compute_stream);
        if (e == hipSuccess && n_rows > 0) e = hipMemcpy(h_mu, d_mu, (size_t)n_rows * D * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess && h_logvar && n_rows > 0) e = hipMemcpy(h_logvar, d_lv, (size_t)n_rows * D * sizeof(float), hipMemcpyDeviceToHost);
        if (e == hipSuccess && h_rms_ok && n_rows > 0) e = hipMemcpy(h_rms_ok, d_ok, (size_t)n_rows * sizeof(int32_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) rc = fail(c, AMPH_ERR_HIP, "encode_pcm16_host: %s", hipGetErrorString(e));
    } else {
        (void)hipStreamSynchronize(c->compute_stream);
    }
    if (d_mu) (void)hipFree(d_mu);
    if (d_lv) (void)hipFree(d_lv);
    if (d_ok) (void)hipFree(d_ok);
    return rc;
}

int amph_decide_radial(amph_ctx* c, const float* d_z, int64_t n_rows, int dim, const float* h_centroids,
                       const double* h_thresholds, const int32_t* h_rank, int n_species, int32_t* d_pred, float* d_best_d,
                       void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || dim < 1 || dim > kMaxRadialDim || n_species < 0 || n_species > kMaxSpecies ||
        (n_species > 0 && (!h_centroids || !h_thresholds || !h_rank)) || (n_rows > 0 && (!d_z || !d_pred || !d_best_d)))
        return fail(c, AMPH_ERR_INVALID_ARG, "bad radial arguments (dim <= %d, species <= %d)", kMaxRadialDim, kMaxSpecies);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (!c->d_centroids) {
        HIP_TRY(c, hipMalloc((void**)&c->d_centroids, (size_t)kMaxSpecies * kMaxRadialDim * sizeof(float)));
        HIP_TRY(c, hipMalloc((void**)&c->d_thr, kMaxSpecies * sizeof(double)));
        HIP_TRY(c, hipMalloc((void**)&c->d_rank, kMaxSpecies * sizeof(int32_t)));
    }
    if (n_species > 0) {
        // the detector is the same from call to call (a benchmark run decides every batch against one fit): its three small
        // uploads - synchronous copies of pageable memory, 3 % of the encode + decide loop - happen only when the bytes change
        const size_t nc = (size_t)n_species * dim * sizeof(float), nt = (size_t)n_species * sizeof(double), nr = (size_t)n_species * sizeof(int32_t);
        std::vector<unsigned char> key(2 * sizeof(int) + nc + nt + nr);
        const int hdr[2] = {n_species, dim};
        std::memcpy(key.data(), hdr, sizeof hdr);
        std::memcpy(key.data() + sizeof hdr, h_centroids, nc);
        std::memcpy(key.data() + sizeof hdr + nc, h_thresholds, nt);
        std::memcpy(key.data() + sizeof hdr + nc + nt, h_rank, nr);
        if (key != c->radial_key) {
            // (ordered behind earlier launches on this stream that still read the buffers; a caller alternating streams and
            // detectors must synchronise them itself, as with any other reuse of the context)
            HIP_TRY(c, hipMemcpyAsync(c->d_centroids, h_centroids, nc, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(c->d_thr, h_thresholds, nt, hipMemcpyHostToDevice, s));
            HIP_TRY(c, hipMemcpyAsync(c->d_rank, h_rank, nr, hipMemcpyHostToDevice, s));
            c->radial_key.swap(key);
        }
    }
    RadialParams p;
    p.z = d_z; p.centroids = c->d_centroids; p.thresholds = c->d_thr; p.rank = c->d_rank; p.K = n_species; p.D = dim;
    p.pred = d_pred; p.best_d = d_best_d;
    if (n_rows > 0) launch_decide_radial(p, n_rows, s);
    return check_launch(c, "decide_radial");
}

int amph_decide_map(amph_ctx* c, const float* d_z, int64_t n_rows, int dim, const float* h_means, const float* h_precision,
                    const double* h_logdet, const double* h_log_prior, int n_species, int has_tau, double tau,
                    int32_t* d_pred, double* d_best_score, double* d_scores, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || dim < 1 || dim > kMaxRadialDim || n_species < 0 || n_species > kMaxSpecies ||
        (n_species > 0 && (!h_means || !h_precision || !h_logdet || !h_log_prior)) ||
        (n_rows > 0 && (!d_z || !d_pred || !d_best_score)))
        return fail(c, AMPH_ERR_INVALID_ARG, "bad MAP arguments (dim <= %d, species <= %d)", kMaxRadialDim, kMaxSpecies);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (!c->d_map_means) {
        HIP_TRY(c, hipMalloc((void**)&c->d_map_means, (size_t)kMaxSpecies * kMaxRadialDim * sizeof(float)));
        HIP_TRY(c, hipMalloc((void**)&c->d_map_logdet, kMaxSpecies * sizeof(double)));
        HIP_TRY(c, hipMalloc((void**)&c->d_map_lprior, kMaxSpecies * sizeof(double)));
    }
    const size_t need = (size_t)std::max(n_species, 1) * dim * dim * sizeof(float);
    if (need > c->map_prec_cap) {
        HIP_TRY(c, hipStreamSynchronize(s));                 // an earlier launch may still read the old buffer
        if (c->d_map_prec) (void)hipFree(c->d_map_prec);
        c->d_map_prec = nullptr; c->map_prec_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_map_prec, need));
        c->map_prec_cap = need;
    }
    if (n_species > 0) {
        HIP_TRY(c, hipMemcpyAsync(c->d_map_means, h_means, (size_t)n_species * dim * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->d_map_prec, h_precision, (size_t)n_species * dim * dim * sizeof(float), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->d_map_logdet, h_logdet, (size_t)n_species * sizeof(double), hipMemcpyHostToDevice, s));
        HIP_TRY(c, hipMemcpyAsync(c->d_map_lprior, h_log_prior, (size_t)n_species * sizeof(double), hipMemcpyHostToDevice, s));
    }
    MapParams p;
    p.z = d_z; p.means = c->d_map_means; p.prec = c->d_map_prec; p.logdet = c->d_map_logdet; p.log_prior = c->d_map_lprior;
    p.d_log2pi = (double)dim * 1.8378770664093453;          // d * float(np.log(2.0 * np.pi)), map_detector_core.py:323
    p.K = n_species; p.D = dim; p.has_tau = has_tau; p.tau = tau;
    p.pred = d_pred; p.best_score = d_best_score; p.scores = d_scores;
    if (n_rows > 0) launch_score_map(p, n_rows, s);
    return check_launch(c, "decide_map");
}

// np.quantile's interpolation between the order statistics a = x_(lo), b = x_(lo + 1) (method "linear"), in the scalar
// arithmetic of numpy 1.26 (float64 virtual index / weight / sum, the difference b - a in float32) or numpy 2.x (all float32)
static void quantile_ranks(long long n, double q, int numpy2, long long* lo, long long* hi) {
#pragma clang fp contract(off)
    double vi;
    if (numpy2) vi = (double)((float)(n - 1) * (float)q);
    else vi = (double)(n - 1) * q;
    if (vi >= (double)(n - 1)) { *lo = *hi = n - 1; return; }
    if (vi < 0) { *lo = *hi = 0; return; }
    *lo = (long long)std::floor(vi);
    *hi = *lo + 1;
}
static double quantile_lerp(float a, float b, long long n, double q, int numpy2) {
#pragma clang fp contract(off)
    const float diff = b - a;
    if (numpy2) {
        const float vi = (float)(n - 1) * (float)q;
        const float t = vi - std::floor(vi);
        if (t >= 0.5f) { const float u = 1.0f - t; const float m = diff * u; return (double)(b - m); }
        const float m = diff * t;
        return (double)(a + m);
    }
    const double vi = (double)(n - 1) * q;
    const double t = vi - std::floor(vi);
    if (t >= 0.5) return (double)b - (double)diff * (1.0 - t);
    return (double)a + (double)diff * t;
}

int amph_fit_radial(amph_ctx* c, const float* d_z, const int64_t* h_offsets, int n_species, int dim, double q_in, double q_out,
                    int numpy2, float* h_centroids, double* h_rk_in, double* h_rk_out, double* h_rk, double* h_summaries, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_species < 1 || n_species > kMaxSpecies || dim < 1 || dim > kMaxRadialDim || !d_z || !h_offsets || !h_centroids || !h_rk_in ||
        !h_rk_out || !h_rk || !(q_in >= 0.0 && q_in <= 1.0) || !(q_out >= 0.0 && q_out <= 1.0))
        return fail(c, AMPH_ERR_INVALID_ARG, "bad radial-fit arguments (dim <= %d, species <= %d, quantiles in [0, 1])", kMaxRadialDim, kMaxSpecies);
    for (int k = 0; k < n_species; ++k)
        if (h_offsets[k + 1] <= h_offsets[k] || h_offsets[0] != 0)
            return fail(c, AMPH_ERR_INVALID_ARG, "radial fit: species %d has no rows (offsets must start at 0 and increase)", k);
    const long long n = h_offsets[n_species];
    if (n >= (1ll << 24)) return fail(c, AMPH_ERR_UNSUPPORTED, "radial fit: more than 2^24 rows");     // (float)(n - 1) must be exact
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    if (!c->d_fit_off) {
        HIP_TRY(c, hipMalloc((void**)&c->d_fit_off, (size_t)(kMaxSpecies + 1 + 16 * kMaxSpecies) * sizeof(long long)));
        HIP_TRY(c, hipMalloc((void**)&c->d_fit_mu, (size_t)kMaxSpecies * kMaxRadialDim * sizeof(float)));
        HIP_TRY(c, hipMalloc((void**)&c->d_fit_stats, (size_t)kMaxSpecies * 16 * sizeof(float)));
    }
    const size_t need = (size_t)n_species * (size_t)n;
    if (c->fit_rho_cap < need) {
        HIP_TRY(c, hipStreamSynchronize(s));
        if (c->d_fit_rho) (void)hipFree(c->d_fit_rho);
        c->d_fit_rho = nullptr; c->fit_rho_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_fit_rho, need * sizeof(float)));
        c->fit_rho_cap = need;
    }
    std::vector<long long> meta((size_t)kMaxSpecies + 1 + 16 * kMaxSpecies, -1);
    for (int k = 0; k <= n_species; ++k) meta[k] = h_offsets[k];
    auto class_ranks = [&](long long* r, long long cnt, double q) {      // q lo, hi | min | p50 lo, hi | p90 lo, hi | max
        quantile_ranks(cnt, q, numpy2, &r[0], &r[1]);
        r[2] = 0;
        quantile_ranks(cnt, 0.50, numpy2, &r[3], &r[4]);
        quantile_ranks(cnt, 0.90, numpy2, &r[5], &r[6]);
        r[7] = cnt - 1;
    };
    for (int k = 0; k < n_species; ++k) {
        long long* r = meta.data() + kMaxSpecies + 1 + 16 * k;
        const long long n_in = h_offsets[k + 1] - h_offsets[k], n_out = n - n_in;
        class_ranks(r, n_in, q_in);
        if (n_out > 0) class_ranks(r + 8, n_out, q_out);
    }
    HIP_TRY(c, hipMemcpyAsync(c->d_fit_off, meta.data(), meta.size() * sizeof(long long), hipMemcpyHostToDevice, s));
    FitParams p;
    p.z = d_z; p.offsets = c->d_fit_off; p.K = n_species; p.D = dim; p.n = n; p.mu = c->d_fit_mu; p.rho = c->d_fit_rho;
    p.ranks = c->d_fit_off + kMaxSpecies + 1; p.stats = c->d_fit_stats;
    launch_fit_radial(p, s);
    std::vector<float> st((size_t)n_species * 16);
    HIP_TRY(c, hipMemcpyAsync(h_centroids, c->d_fit_mu, (size_t)n_species * dim * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipMemcpyAsync(st.data(), c->d_fit_stats, st.size() * sizeof(float), hipMemcpyDeviceToHost, s));
    HIP_TRY(c, hipStreamSynchronize(s));               // host outputs (and `meta`, a pageable source) are complete
    for (int k = 0; k < n_species; ++k) {
        const long long n_in = h_offsets[k + 1] - h_offsets[k], n_out = n - n_in;
        const float* t = st.data() + 16 * k;
        h_rk_in[k] = quantile_lerp(t[0], t[1], n_in, q_in, numpy2);
        h_rk_out[k] = n_out > 0 ? quantile_lerp(t[8], t[9], n_out, q_out, numpy2) : INFINITY;
        h_rk[k] = std::min(h_rk_in[k], h_rk_out[k]);
        if (h_summaries) {                              // summarize_dist of rho_in / rho_out: min, p50, p90, max (NaN x 4: empty)
            double* o = h_summaries + 8 * k;
            o[0] = t[2]; o[1] = quantile_lerp(t[3], t[4], n_in, 0.50, numpy2); o[2] = quantile_lerp(t[5], t[6], n_in, 0.90, numpy2); o[3] = t[7];
            for (int u = 0; u < 4; ++u) o[4 + u] = NAN;
            if (n_out > 0) {
                o[4] = t[10]; o[5] = quantile_lerp(t[11], t[12], n_out, 0.50, numpy2); o[6] = quantile_lerp(t[13], t[14], n_out, 0.90, numpy2);
                o[7] = t[15];
            }
        }
    }
    return check_launch(c, "fit_radial");
}

int amph_resample_length(int n_in, int orig_sr, int target_sr) {
    if (n_in < 0 || orig_sr <= 0 || target_sr <= 0) return -1;
    return (int)std::ceil((double)n_in * ((double)target_sr / (double)orig_sr));
}

int amph_resample_f32(amph_ctx* c, const float* d_in, int n_rows, int n_in, int orig_sr, int target_sr,
                      const double* h_half_window, int n_window, int num_table, float* d_out, int out_stride, void* stream) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    if (n_rows < 0 || n_in <= 0 || orig_sr <= 0 || target_sr <= 0 || !h_half_window || n_window < 2 || num_table < 1 ||
        (n_rows > 0 && (!d_in || !d_out)))
        return fail(c, AMPH_ERR_INVALID_ARG, "bad resample arguments");
    const double ratio = (double)target_sr / (double)orig_sr;
    const int n_out = (int)((double)n_in * ratio), n_fix = (int)std::ceil((double)n_in * ratio);
    if (n_out < 1) return fail(c, AMPH_ERR_INVALID_ARG, "input signal length is too small to resample");
    if (out_stride < n_fix) return fail(c, AMPH_ERR_INVALID_ARG, "out_stride %d < resampled length %d", out_stride, n_fix);
    const double scale = std::min(1.0, ratio);
    const int index_step = (int)(scale * (double)num_table);
    if (index_step < 1) return fail(c, AMPH_ERR_UNSUPPORTED, "ratio %.6g too small for a table of %d entries per zero crossing", ratio, num_table);
    HIP_TRY(c, hipSetDevice(c->device));
    hipStream_t s = (hipStream_t)stream;
    HIP_TRY(c, hipStreamSynchronize(s));                      // the previous call's table upload / kernel are done
    if ((size_t)n_window > c->rs_cap) {
        if (c->d_rs_win) (void)hipFree(c->d_rs_win);
        if (c->d_rs_delta) (void)hipFree(c->d_rs_delta);
        c->d_rs_win = c->d_rs_delta = nullptr; c->rs_cap = 0;
        HIP_TRY(c, hipMalloc((void**)&c->d_rs_win, (size_t)n_window * sizeof(double)));
        HIP_TRY(c, hipMalloc((void**)&c->d_rs_delta, (size_t)n_window * sizeof(double)));
        c->rs_cap = n_window;
    }
    // resampy.core.resample: interp_win *= ratio when downsampling; interp_delta[:-1] = diff(interp_win)
    c->rs_host.assign((size_t)2 * n_window, 0.0);
    double* win = c->rs_host.data();
    double* delta = win + n_window;
    for (int i = 0; i < n_window; ++i) win[i] = ratio < 1.0 ? ratio * h_half_window[i] : h_half_window[i];
    for (int i = 0; i + 1 < n_window; ++i) delta[i] = win[i + 1] - win[i];
    HIP_TRY(c, hipMemcpyAsync(c->d_rs_win, win, (size_t)n_window * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipMemcpyAsync(c->d_rs_delta, delta, (size_t)n_window * sizeof(double), hipMemcpyHostToDevice, s));
    HIP_TRY(c, hipStreamSynchronize(s));                      // rs_host may be rewritten by the next call (any stream)
    ResampleParams p;
    p.x = d_in; p.in_stride = n_in; p.n_in = n_in; p.y = d_out; p.out_stride = out_stride; p.n_out = n_out; p.n_fix = n_fix;
    p.win = c->d_rs_win; p.delta = c->d_rs_delta; p.nwin = n_window; p.num_table = num_table; p.index_step = index_step;
    p.scale = scale; p.time_inc = 1.0 / ratio;
    launch_resample(p, n_rows, s);
    return check_launch(c, "resample");
}

// ---------------------------------------------------------------------------- WAV ingest
// POSIX open / read / pread / close (no stdio buffer, no fstat, no whole-row memset): the ingest is bound by per-file
// host work - the GPU path is ~20 x faster than any loader - so every avoided syscall and pass over the row counts.
// The chunk list is parsed from the first 4 KB (one read); the samples are read straight into the caller's row.
// The general RIFF walk on an open descriptor (any chunk order, WAVE_FORMAT_EXTENSIBLE, odd chunk sizes); closes fd.
static int read_wav_fd(const int fd, int expect_sr, int16_t* out, int max_samples, int32_t* out_len) {
    if (out_len) *out_len = 0;
    auto zero_from = [&](size_t first) { if (first < (size_t)max_samples) std::memset(out + first, 0, ((size_t)max_samples - first) * sizeof(int16_t)); };
    if (fd < 0) { zero_from(0); return AMPH_ERR_IO; }
    unsigned char hdr[4096];
    const ssize_t have = ::pread(fd, hdr, sizeof hdr, 0);
    int rc = AMPH_ERR_IO;
    bool have_fmt = false;
    size_t filled = 0;
    if (have >= 12 && !std::memcmp(hdr, "RIFF", 4) && !std::memcmp(hdr + 8, "WAVE", 4)) {
        long long pos = 12;                            // file offset of the next chunk header
        unsigned char ch[8];
        for (;;) {
            if (pos + 8 <= have) std::memcpy(ch, hdr + pos, 8);
            else if (::pread(fd, ch, 8, pos) != 8) break;
            const uint32_t sz = ch[4] | (ch[5] << 8) | (ch[6] << 16) | ((uint32_t)ch[7] << 24);
            if (!std::memcmp(ch, "fmt ", 4)) {
                unsigned char fm[40] = {0};
                const size_t rd = std::min<size_t>(sz, sizeof fm);
                if (rd < 16) break;
                if (pos + 8 + (long long)rd <= have) std::memcpy(fm, hdr + pos + 8, rd);
                else if (::pread(fd, fm, rd, pos + 8) != (ssize_t)rd) break;
                int fmt = fm[0] | (fm[1] << 8);
                const int channels = fm[2] | (fm[3] << 8);
                const uint32_t sr = fm[4] | (fm[5] << 8) | (fm[6] << 16) | ((uint32_t)fm[7] << 24);
                const int bits = fm[14] | (fm[15] << 8);
                if (fmt == 0xFFFE && rd >= 26) fmt = fm[24] | (fm[25] << 8);      // WAVE_FORMAT_EXTENSIBLE sub-format
                if (fmt != 1 || bits != 16 || channels != 1 || (expect_sr > 0 && (int)sr != expect_sr)) { rc = AMPH_ERR_UNSUPPORTED; break; }
                have_fmt = true;
            } else if (!std::memcmp(ch, "data", 4)) {
                if (!have_fmt) break;
                const size_t avail = sz / 2;
                const size_t want = std::min<size_t>(avail, (size_t)max_samples);
                const long long data_pos = pos + 8;
                size_t got_bytes = 0;
                if (data_pos < have) {                 // the head of the samples came with the header read
                    got_bytes = std::min<size_t>((size_t)(have - data_pos), want * 2);
                    std::memcpy(out, hdr + data_pos, got_bytes);
                }
                while (got_bytes < want * 2) {
                    const ssize_t r = ::pread(fd, reinterpret_cast<char*>(out) + got_bytes, want * 2 - got_bytes, data_pos + (long long)got_bytes);
                    if (r <= 0) break;
                    got_bytes += (size_t)r;
                }
                const size_t got = got_bytes / 2;
                filled = got;
                if (out_len) *out_len = (int32_t)got;
                rc = (got == want || sz == 0xFFFFFFFFu) ? AMPH_OK : AMPH_ERR_IO;
                if (rc != AMPH_OK && got > 0 && got < want) rc = AMPH_OK;           // truncated file: keep what is there (libsndfile does)
                break;
            }
            pos += 8 + (long long)sz + (sz & 1);
        }
    }
    ::close(fd);
    zero_from(rc == AMPH_OK ? filled : 0);
    if (rc != AMPH_OK && out_len) *out_len = 0;
    return rc;
}

// One file, relative to an open directory (AT_FDCWD for a path as given).  The canonical file - what soundfile / the wave module write: a
// 16-byte PCM fmt chunk at 12, the data chunk at 36 - costs three system calls: openat, ONE preadv that puts the 44 header bytes on the
// stack and the samples straight into the caller's row, close.  Anything else (LIST chunks in front, WAVE_FORMAT_EXTENSIBLE, ...) takes
// the general walk on the same descriptor.  openat on a directory descriptor instead of open(full path): with 8 - 32 reader threads in ONE
// directory the full-path lookups do not scale at all on this kernel (tools/probe/file_ingest_probe.c: 89 k files/s at 1, 8 and 32
// threads, against 88 k -> 590 k with openat) - every thread walks the same three ancestors' dentries.
static int read_wav_at(const int dirfd, const char* name, int expect_sr, int16_t* out, int max_samples, int32_t* out_len) {
    if (out_len) *out_len = 0;
    if (!name || !out || max_samples < 0) return AMPH_ERR_INVALID_ARG;
    const int fd = ::openat(dirfd, name, O_RDONLY | O_CLOEXEC);
    if (fd < 0) { if (max_samples) std::memset(out, 0, (size_t)max_samples * sizeof(int16_t)); return AMPH_ERR_IO; }
    unsigned char h[44];
    struct iovec iov[2] = {{h, sizeof h}, {out, (size_t)max_samples * sizeof(int16_t)}};
    const ssize_t n = ::preadv(fd, iov, 2, 0);
    auto u32 = [&](int o) { return (uint32_t)h[o] | ((uint32_t)h[o + 1] << 8) | ((uint32_t)h[o + 2] << 16) | ((uint32_t)h[o + 3] << 24); };
    const bool canonical = n >= 44 && !std::memcmp(h, "RIFF", 4) && !std::memcmp(h + 8, "WAVEfmt ", 8) && u32(16) == 16 && !std::memcmp(h + 36, "data", 4);
    if (!canonical) return read_wav_fd(fd, expect_sr, out, max_samples, out_len);        // (overwrites the row; closes fd)
    const int fmt = h[20] | (h[21] << 8), channels = h[22] | (h[23] << 8), bits = h[34] | (h[35] << 8);
    const uint32_t sr = u32(24), sz = u32(40);
    int rc = AMPH_OK;
    size_t got = 0;
    if (fmt != 1 || bits != 16 || channels != 1 || (expect_sr > 0 && (int)sr != expect_sr)) rc = AMPH_ERR_UNSUPPORTED;
    else {
        const size_t want = std::min<size_t>(sz / 2, (size_t)max_samples);               // (0xFFFFFFFF, a streamed file: whatever is there)
        size_t got_bytes = std::min<size_t>((size_t)(n - 44), want * 2);
        while (got_bytes < want * 2) {                                                    // a short read: the rest, as the general reader does
            const ssize_t r = ::pread(fd, reinterpret_cast<char*>(out) + got_bytes, want * 2 - got_bytes, 44 + (long long)got_bytes);
            if (r <= 0) break;
            got_bytes += (size_t)r;
        }
        got = got_bytes / 2;
        if (got < want && got == 0 && want > 0 && sz != 0xFFFFFFFFu) rc = AMPH_ERR_IO;      // (a truncated file keeps what is there, as libsndfile does)
    }
    ::close(fd);
    const size_t keep = rc == AMPH_OK ? got : 0;
    if (keep < (size_t)max_samples) std::memset(out + keep, 0, ((size_t)max_samples - keep) * sizeof(int16_t));
    if (out_len) *out_len = rc == AMPH_OK ? (int32_t)got : 0;
    return rc;
}

static int read_wav_file(const char* path, int expect_sr, int16_t* out, int max_samples, int32_t* out_len) {
    if (!path) { if (out_len) *out_len = 0; return AMPH_ERR_INVALID_ARG; }
    return read_wav_at(AT_FDCWD, path, expect_sr, out, max_samples, out_len);
}

int amph_wav_read_pcm16(const char* path, int expect_sr, int16_t* h_out, int max_samples, int32_t* out_len) {
    if (!h_out || max_samples < 0) { if (out_len) *out_len = 0; return AMPH_ERR_INVALID_ARG; }
    return read_wav_file(path, expect_sr, h_out, max_samples, out_len);
}

// a reader thread's directory descriptors: the files of a call come from a handful of directories (one per species), consecutive files from the same one
namespace {
struct DirCache {
    std::string dir[4];
    int fd[4] = {-1, -1, -1, -1};
    int next = 0;
    ~DirCache() { for (int f : fd) if (f >= 0) ::close(f); }
    // -> (directory descriptor or AT_FDCWD, name relative to it)
    int split(const char* path, const char** name) {
        const char* slash = std::strrchr(path, '/');
        if (!slash || slash == path) { *name = path; return AT_FDCWD; }
        const size_t len = (size_t)(slash - path);
        for (int k = 0; k < 4; ++k)
            if (fd[k] >= 0 && dir[k].size() == len && !std::memcmp(dir[k].data(), path, len)) { *name = slash + 1; return fd[k]; }
        const int k = next;
        next = (next + 1) & 3;
        if (fd[k] >= 0) ::close(fd[k]);
        dir[k].assign(path, len);
        fd[k] = ::open(dir[k].c_str(), O_RDONLY | O_DIRECTORY | O_CLOEXEC);
        if (fd[k] < 0) { dir[k].clear(); *name = path; return AT_FDCWD; }     // (unreadable directory: the full path reports the error)
        *name = slash + 1;
        return fd[k];
    }
};
}  // namespace

int amph_wav_read_many(const char* const* paths, int n_files, int expect_sr, int16_t* h_out, int max_samples,
                       int32_t* h_lengths, int32_t* h_status, int n_threads) {
    if (n_files < 0 || (n_files > 0 && (!paths || !h_out || !h_lengths || !h_status))) return AMPH_ERR_INVALID_ARG;
    n_threads = std::max(1, std::min(n_threads, std::max(1, n_files)));
    std::atomic<int> next{0};
    auto work = [&]() {
        DirCache dirs;
        for (;;) {
            // (runs of 8: neighbouring files are neighbouring rows of h_out - a thread then writes 8 consecutive rows)
            const int i0 = next.fetch_add(8);
            if (i0 >= n_files) break;
            for (int i = i0; i < std::min(n_files, i0 + 8); ++i) {
                if (!paths[i]) { h_status[i] = AMPH_ERR_INVALID_ARG; h_lengths[i] = 0; continue; }
                const char* name = nullptr;
                const int dfd = dirs.split(paths[i], &name);
                h_status[i] = read_wav_at(dfd, name, expect_sr, h_out + (size_t)i * max_samples, max_samples, &h_lengths[i]);
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return AMPH_OK;
}

int amph_wav_sizes(const char* const* paths, int n_files, int64_t* h_sizes, int n_threads) {
    if (n_files < 0 || (n_files > 0 && (!paths || !h_sizes))) return AMPH_ERR_INVALID_ARG;
    n_threads = std::max(1, std::min(n_threads, std::max(1, n_files / 64)));
    std::atomic<int> next{0};
    auto work = [&]() {
        DirCache dirs;
        for (;;) {
            const int i0 = next.fetch_add(64);
            if (i0 >= n_files) break;
            for (int i = i0; i < std::min(n_files, i0 + 64); ++i) {
                struct stat st;
                const char* name = nullptr;
                h_sizes[i] = -1;
                if (!paths[i]) continue;
                const int dfd = dirs.split(paths[i], &name);
                if (::fstatat(dfd, name, &st, 0) == 0) h_sizes[i] = (int64_t)st.st_size;
            }
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work);
    work();
    for (auto& t : pool) t.join();
    return AMPH_OK;
}

void* amph_host_alloc(size_t bytes) {
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) return nullptr;
    return p;
}
void amph_host_free(void* p) { if (p) (void)hipHostFree(p); }

// ---------------------------------------------------------------------------- profiling
int amph_profile_enable(amph_ctx* c, int on) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    c->profiling = on != 0;
    c->profile_only = on >= 2 ? on - 2 : -1;
    if (c->profile_only >= (int)c->stages.size()) { c->profiling = false; return fail(c, AMPH_ERR_INVALID_ARG, "no such stage"); }
    return AMPH_OK;
}
int amph_profile_stage_count(const amph_ctx* c) { return c ? (int)c->stages.size() : 0; }
const char* amph_profile_stage_name(const amph_ctx* c, int stage) {
    if (!c || stage < 0 || stage >= (int)c->stages.size()) return "";
    return c->stages[stage].name.c_str();
}
int amph_profile_read(amph_ctx* c, int stage, int64_t* out_launches, double* out_total_ms, double* out_flops_per_chunk,
                      double* out_bytes_per_chunk) {
    if (!c || stage < 0 || stage >= (int)c->stages.size()) return AMPH_ERR_INVALID_ARG;
    Stage& st = c->stages[stage];
    for (auto& pr : st.pending) {
        (void)hipEventSynchronize(pr.second);
        float ms = 0.0f;
        if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) { st.total_ms += ms; st.launches += 1; }
        c->event_pool.push_back(pr.first);
        c->event_pool.push_back(pr.second);
    }
    st.pending.clear();
    if (out_launches) *out_launches = st.launches;
    if (out_total_ms) *out_total_ms = st.total_ms;
    if (out_flops_per_chunk) *out_flops_per_chunk = st.flops_per_chunk;
    if (out_bytes_per_chunk) *out_bytes_per_chunk = st.bytes_per_chunk;
    return AMPH_OK;
}
int amph_profile_reset(amph_ctx* c) {
    if (!c) return AMPH_ERR_INVALID_ARG;
    for (auto& st : c->stages) {
        for (auto& pr : st.pending) {
            (void)hipEventSynchronize(pr.second);
            c->event_pool.push_back(pr.first);
            c->event_pool.push_back(pr.second);
        }
        st.pending.clear();
        st.launches = 0;
        st.total_ms = 0;
    }
    return AMPH_OK;
}

}  // extern "C"
