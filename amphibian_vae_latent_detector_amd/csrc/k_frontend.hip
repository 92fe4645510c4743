// Front-end kernels for gfx950: RMS scale (K0), STFT + mel projection (K1/K2),
// dB / standardise / crop-pad (K3).
//
// Reference arithmetic being replaced (file:line in the reference repo):
//   K0   rms_normalize                     00_normalize_dataset_rms.py:29-38
//        PCM16 write/read hand-off         00_normalize_dataset_rms.py:57 -> map_detector_core.py:210
//   K1   fix length, librosa stft          map_detector_core.py:212-228
//   K2   mel filterbank projection         map_detector_core.py:219-228
//   K3   power_to_db(ref=max), z-score,
//        crop_or_pad_time, [T,M] layout    map_detector_core.py:229-237, 185-195, 268
//
// K1/K2 design (n_fft = 2048 -> 1024-point complex FFT of the even/odd packed frame):
//   one wave (64 lanes) owns one frame; each lane holds 16 complex points.
//   stage A  radix-16 in registers (stride 64)          + twiddle W_1024^(t*k1)
//   exchange through a padded LDS tile [16][68] float2 (conflict-free both ways)
//   stage B  radix-16 in registers                      + twiddle W_64^(t_lo*k2a)
//   stage C  radix-4 across the 4 lanes of a quad with DPP quad_perm butterflies
//   real-FFT split of (k, 1024-k) pairs -> |X|^2 into LDS, mel rows by 4 lanes
//   per filter with a DPP quad reduction.  tools/fft_lane_model.py is the numpy
//   model of exactly this data movement.
//   A workgroup (4 waves) stages the sample span of its frame tile in LDS once,
//   applying PCM16 decode, RMS scale, clip and the PCM16 re-quantisation on load.
#include "amph_internal.h"

namespace amph {

// ----------------------------------------------------------------------------
// small helpers
// ----------------------------------------------------------------------------
__device__ __forceinline__ float dpp_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));
}
__device__ __forceinline__ float dpp_xor2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));
}
// Orders this wave's LDS writes before its later LDS reads (LDS ops of one wave
// execute in order; this only stops the compiler from moving them).
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_down(v, o);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_down(v, o));
    return v;
}
// Block-wide reductions for 256-thread blocks; result valid in every thread.
__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__device__ __forceinline__ float block_max(float v, float* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    return fmaxf(fmaxf(sh[0], sh[1]), fmaxf(sh[2], sh[3]));
}

// ----------------------------------------------------------------------------
// K0: per-row RMS scale.  rms = sqrt(mean(y^2)); rms < 1e-4 -> unchanged;
// else scale = 0.05 / (rms + 1e-8), all float32 (00_normalize_dataset_rms.py:30-36).
//
// The scale feeds a PCM16 re-quantisation (00 writes PCM16, 07 reads it back), which
// is discontinuous: a last-bit difference in the scale can move every occurrence of
// one sample value by an LSB (measured: 1.7e-4 on the latents of a quiet chunk).  So
// mean(y^2) is evaluated in EXACTLY numpy's float32 order (tools/pairwise_model.py):
//   the row is consumed in ufunc-buffer chunks of 8192 elements, chunk sums are
//   accumulated sequentially; inside a chunk numpy recurses n -> (n/2 - (n/2)%8, rest)
//   down to blocks of <= 128, each summed with 8 strided accumulators combined as
//   ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail.
// Mapping: 8 lanes per block (one per accumulator, xor-1/2/4 butterflies give the
// combine order), block results into a per-chunk heap in LDS, heap levels summed
// left+right deepest first, thread 0 folds the chunk sums in order.
// ----------------------------------------------------------------------------
constexpr int kNpChunk = 8192;     // numpy ufunc buffer size (elements)
constexpr int kNpBlock = 128;      // PW_BLOCKSIZE
constexpr int kChunksPerPass = 4;

__device__ __forceinline__ int np_split(int n) { const int h = n >> 1; return h - (h & 7); }

// depth of the deepest leaf of a chunk of n elements (the right child is never smaller)
__device__ __forceinline__ int np_depth(int n) {
    int d = 0;
    while (n > kNpBlock) { n -= np_split(n); ++d; }
    return d;
}
// node reached from the root along `path` (MSB first) after at most `level` steps
__device__ __forceinline__ void np_walk(int n, int path, int level, int& off, int& size, int& reached) {
    off = 0;
    int l = 0;
    for (; l < level && n > kNpBlock; ++l) {
        const int n2 = np_split(n);
        if ((path >> (level - 1 - l)) & 1) { off += n2; n -= n2; }
        else n = n2;
    }
    size = n;
    reached = l;
}

// NOTE: `__fmul_rn` / `__fadd_rn` are device-library functions whose fmul / fadd carry the
// library's own `contract` flag: hipcc fused them into v_fmac (seen in the ISA; 1-ulp
// differences from numpy).  Code that must round each operation separately is written
// with plain operators under `#pragma clang fp contract(off)` and checked in the ISA.
template <bool PCM16>
__device__ __forceinline__ float sq_at(const void* row, int i) {
#pragma clang fp contract(off)
    float y;
    if (PCM16) y = (float)reinterpret_cast<const int16_t*>(row)[i] * (1.0f / 32768.0f);
    else y = reinterpret_cast<const float*>(row)[i];
    return y * y;
}

template <bool PCM16>
__global__ __launch_bounds__(256) void k_rms_scale(const void* __restrict__ in, long long row_stride,
                                                   const int32_t* __restrict__ lengths, int n_in,
                                                   float* __restrict__ scale, int32_t* __restrict__ ok) {
#pragma clang fp contract(off)
    __shared__ float heap[kChunksPerPass][2 * 64];       // chunk depth <= 6: 64 leaves, heap of 128
    const int row = blockIdx.x, tid = threadIdx.x;
    const char* x = reinterpret_cast<const char*>(in) + (long long)row * row_stride * (PCM16 ? 2 : 4);
    const int n = lengths ? min(lengths[row], n_in) : n_in;
    const int n_chunks = (n + kNpChunk - 1) / kNpChunk;
    float total = 0.0f;                                  // thread 0 only
    for (int c0 = 0; c0 < n_chunks; c0 += kChunksPerPass) {
        // ---- leaves: task = (chunk, leaf slot, accumulator lane)
        for (int task = tid; task < kChunksPerPass * 64 * 8; task += 256) {
            const int cl = task >> 9, slot = (task >> 3) & 63, j = task & 7;
            const int chunk = c0 + cl;
            bool owner = false;
            int off = 0, size = 0, lvl = 0, rem = 0;
            if (chunk < n_chunks) {
                const int cn = min(kNpChunk, n - chunk * kNpChunk);
                const int D = np_depth(cn);
                if (slot < (1 << D)) {
                    np_walk(cn, slot, D, off, size, lvl);
                    rem = D - lvl;
                    owner = (slot & ((1 << rem) - 1)) == 0;   // an early leaf belongs to the slot with zero low bits
                    off += chunk * kNpChunk;
                }
            }
            float r = 0.0f;
            const int body = size - (size & 7);
            if (owner && size >= 8) {
                r = sq_at<PCM16>(x, off + j);
                for (int i = 8; i < body; i += 8) r = (r + sq_at<PCM16>(x, off + i + j));
            }
            r = (r + dpp_xor1(r));
            r = (r + dpp_xor2(r));
            r = (r + __shfl_xor(r, 4));
            if (owner && j == 0) {
                if (size < 8) {                           // numpy: res = -0.0; res += a[i]
                    r = -0.0f;
                    for (int i = 0; i < size; ++i) r = (r + sq_at<PCM16>(x, off + i));
                } else {
                    for (int i = body; i < size; ++i) r = (r + sq_at<PCM16>(x, off + i));
                }
                heap[cl][(1 << lvl) + (slot >> rem)] = r;
            }
        }
        __syncthreads();
        // ---- internal nodes, deepest level first: value = left + right
        for (int lvl = 5; lvl >= 0; --lvl) {
            for (int t = tid; t < kChunksPerPass << lvl; t += 256) {
                const int cl = t >> lvl, path = t & ((1 << lvl) - 1);
                const int chunk = c0 + cl;
                if (chunk < n_chunks) {
                    const int cn = min(kNpChunk, n - chunk * kNpChunk);
                    int off, size, reached;
                    np_walk(cn, path, lvl, off, size, reached);
                    if (reached == lvl && size > kNpBlock) {
                        const int idx = (1 << lvl) + path;
                        heap[cl][idx] = (heap[cl][2 * idx] + heap[cl][2 * idx + 1]);
                    }
                }
            }
            __syncthreads();
        }
        if (tid == 0) {
            for (int cl = 0; cl < kChunksPerPass && c0 + cl < n_chunks; ++cl)
                total = (c0 + cl == 0) ? heap[cl][1] : (total + heap[cl][1]);
        }
        __syncthreads();
    }
    if (tid == 0) {
        float sc = 1.0f;
        int good = 0;
        if (n > 0) {
            const float rms = sqrtf(total / (float)n);           // np.mean = sum / n in float32; IEEE sqrt
            if (!(rms < 1e-4f)) { sc = 0.05f / (rms + 1e-8f); good = 1; }
        }
        scale[row] = sc;
        if (ok) ok[row] = good;
    }
}

// y = clip(x * scale, -1, 1) for scaled rows; gated rows (scale == 1 and not ok)
// are copied unchanged exactly as the reference returns `y` itself.
__global__ __launch_bounds__(256) void k_rms_apply_f32(const float* __restrict__ in, int n, const float* __restrict__ scale,
                                                       const int32_t* __restrict__ ok, float* __restrict__ out) {
    const int row = blockIdx.y;
    const float sc = scale[row];
    const bool scaled = ok[row] != 0;
    const long long base = (long long)row * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float x = in[base + i];
        out[base + i] = scaled ? fminf(fmaxf(x * sc, -1.0f), 1.0f) : x;
    }
}
__global__ __launch_bounds__(256) void k_rms_apply_pcm16(const int16_t* __restrict__ in, int n, const float* __restrict__ scale,
                                                         const int32_t* __restrict__ ok, int16_t* __restrict__ out) {
    const int row = blockIdx.y;
    const float sc = scale[row];
    const bool scaled = ok[row] != 0;
    const long long base = (long long)row * n;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        float x = (float)in[base + i] * (1.0f / 32768.0f);
        if (scaled) x = fminf(fmaxf(x * sc, -1.0f), 1.0f);
        out[base + i] = (int16_t)(int)rintf(x * 32767.0f);     // libsndfile float -> PCM_16
    }
}

// ----------------------------------------------------------------------------
// K1/K2: STFT power + mel
// ----------------------------------------------------------------------------
// forward 4-point DFT in place (W4 = -i)
#define AMPH_BFLY4(r0, i0, r1, i1, r2, i2, r3, i3)                     \
    {                                                                  \
        const float ar = r0 + r2, ai = i0 + i2;                        \
        const float br = r0 - r2, bi = i0 - i2;                        \
        const float cr = r1 + r3, ci = i1 + i3;                        \
        const float dr = i1 - i3, di = r3 - r1;                        \
        r0 = ar + cr; i0 = ai + ci;                                    \
        r1 = br + dr; i1 = bi + di;                                    \
        r2 = ar - cr; i2 = ai - ci;                                    \
        r3 = br - dr; i3 = bi - di;                                    \
    }

// (r, i) *= (wr, wi)
#define AMPH_CMUL(r, i, wr, wi)                                        \
    {                                                                  \
        const float tr_ = r * (wr) - i * (wi);                         \
        i = r * (wi) + i * (wr);                                       \
        r = tr_;                                                       \
    }

// 16-point forward DFT in registers (4x4).  Output X[k] ends up at position
// DFT16_POS(k) = (k >> 2) | ((k & 3) << 2).
#define DFT16_POS(k) ((((k) >> 2) & 3) | (((k) & 3) << 2))
__device__ __forceinline__ void dft16(float (&re)[16], float (&im)[16]) {
    constexpr float C1 = 0.92387953251128675613f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f;   // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;   // sqrt(1/2)
#pragma unroll
    for (int j0 = 0; j0 < 4; ++j0)
        AMPH_BFLY4(re[j0], im[j0], re[j0 + 4], im[j0 + 4], re[j0 + 8], im[j0 + 8], re[j0 + 12], im[j0 + 12]);
    // y[j0][ka] sits at j0 + 4*ka; multiply by W16^(j0*ka) = (cos, -sin)(2 pi j0 ka / 16)
    AMPH_CMUL(re[1 + 4], im[1 + 4], C1, -S1);        // p = 1
    AMPH_CMUL(re[2 + 4], im[2 + 4], R2, -R2);        // p = 2
    AMPH_CMUL(re[3 + 4], im[3 + 4], S1, -C1);        // p = 3
    AMPH_CMUL(re[1 + 8], im[1 + 8], R2, -R2);        // p = 2
    { const float t = re[2 + 8]; re[2 + 8] = im[2 + 8]; im[2 + 8] = -t; }   // p = 4: * (-i)
    AMPH_CMUL(re[3 + 8], im[3 + 8], -R2, -R2);       // p = 6
    AMPH_CMUL(re[1 + 12], im[1 + 12], S1, -C1);      // p = 3
    AMPH_CMUL(re[2 + 12], im[2 + 12], -R2, -R2);     // p = 6
    AMPH_CMUL(re[3 + 12], im[3 + 12], -C1, S1);      // p = 9
#pragma unroll
    for (int ka = 0; ka < 4; ++ka)
        AMPH_BFLY4(re[4 * ka], im[4 * ka], re[4 * ka + 1], im[4 * ka + 1], re[4 * ka + 2], im[4 * ka + 2],
                   re[4 * ka + 3], im[4 * ka + 3]);
}

template <bool PCM16>
__device__ __forceinline__ float fetch_sample(const WaveSrc& src, const void* rowp, int s, int len, float sc) {
    if (s >= len) return 0.0f;                       // zero pad of fix_length (map_detector_core.py:214-215)
    float x;
    if (PCM16) x = (float)reinterpret_cast<const int16_t*>(rowp)[s] * (1.0f / 32768.0f);
    else x = reinterpret_cast<const float*>(rowp)[s];
    if (src.scale) {
        x = fminf(fmaxf(x * sc, -1.0f), 1.0f);
        if (src.requant) x = rintf(x * 32767.0f) * (1.0f / 32768.0f);
    }
    return x;
}

constexpr int kNc = 1024;                             // complex FFT length (n_fft = 2048)
constexpr int kWaveExBytes = 16 * kExStride * 8;      // 8704: exchange tile, reused as Z[1024 + 4*kZPad]
constexpr int kWavePBytes = 4112;                     // 1025 floats of power, rounded to 16
constexpr int kWaveScratch = kWaveExBytes + kWavePBytes;

template <bool PCM16>
__global__ __launch_bounds__(256) void k_stft_mel(StftParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row = blockIdx.x / p.tiles_per_row;
    const int tile = blockIdx.x - row * p.tiles_per_row;
    const int FT = p.frames_per_block;
    const int f_base = tile * FT;
    const int n_fft = 2 * kNc;
    const int span_len = (FT - 1) * p.hop + n_fft;
    float* span = reinterpret_cast<float*>(smem);
    const int span_bytes = (span_len * 4 + 15) & ~15;
    char* wbase = smem + span_bytes + wave * kWaveScratch;
    float2* ex = reinterpret_cast<float2*>(wbase);
    float* P = reinterpret_cast<float*>(wbase + kWaveExBytes);

    // ---- stage the sample span of this frame tile (decode / normalise on load)
    {
        const char* rowp = reinterpret_cast<const char*>(p.src.wave) +
                           (long long)row * p.src.row_stride * (PCM16 ? 2 : 4);
        int len = p.src.lengths ? min(p.src.lengths[row], p.src.n_in) : p.src.n_in;
        len = min(len, p.n_samples);                 // truncation of fix_length (:217)
        const float sc = p.src.scale ? p.src.scale[row] : 1.0f;
        const int N = p.n_samples;
        const int s0 = f_base * p.hop - (n_fft >> 1);
        for (int i = tid; i < span_len; i += 256) {
            int s = s0 + i;
            float v = 0.0f;
            if (p.pad_mode == 0) {                    // reflect (np.pad mode="reflect")
                if (s < 0) s = -s;
                if (s >= N) s = 2 * (N - 1) - s;
            }
            if (s >= 0 && s < N) v = fetch_sample<PCM16>(p.src, rowp, s, len, sc);
            span[i] = v;
        }
    }
    __syncthreads();

    const int k1b = lane >> 2, t_lo = lane & 3;
    for (int fl = wave; fl < FT; fl += 4) {
        const int frame = f_base + fl;
        if (frame >= p.n_frames) break;               // wave-uniform
        float re[16], im[16];
        // ---- window + even/odd packing: z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1], n = lane + 64 j
        {
            const float2* xs = reinterpret_cast<const float2*>(span + fl * p.hop);
            const float2* wn = reinterpret_cast<const float2*>(p.window);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 x = xs[lane + 64 * j];
                const float2 w = wn[lane + 64 * j];
                re[j] = x.x * w.x;
                im[j] = x.y * w.y;
            }
        }
        // ---- stage A
        dft16(re, im);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            float r = re[DFT16_POS(k1)], i = im[DFT16_POS(k1)];
            if (k1 > 0) {
                const float2 w = p.tw_a[k1 * 64 + lane];
                AMPH_CMUL(r, i, w.x, w.y);
            }
            ex[k1 * kExStride + lane] = make_float2(r, i);
        }
        wave_lds_sync();
        // ---- stage B: lane (k1b, t_lo) gathers t = t_lo + 4 m
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 v = ex[k1b * kExStride + t_lo + 4 * m];
            re[m] = v.x;
            im[m] = v.y;
        }
        dft16(re, im);
        float zr[16], zi[16];
#pragma unroll
        for (int k2a = 0; k2a < 16; ++k2a) {
            float r = re[DFT16_POS(k2a)], i = im[DFT16_POS(k2a)];
            if (k2a > 0) {
                const float2 w = p.tw_b[k2a * 4 + t_lo];
                AMPH_CMUL(r, i, w.x, w.y);
            }
            // ---- stage C: radix-4 over the quad's t_lo (xor-2 then xor-1 butterflies)
            {
                const float pr = dpp_xor2(r), pi = dpp_xor2(i);
                float nr, ni;
                if (t_lo < 2) { nr = r + pr; ni = i + pi; }
                else {
                    const float dr = pr - r, di = pi - i;          // v[t_lo-2] - v[t_lo]
                    if (t_lo == 3) { nr = di; ni = -dr; }          // * (-i)
                    else { nr = dr; ni = di; }
                }
                const float qr = dpp_xor1(nr), qi = dpp_xor1(ni);
                if ((t_lo & 1) == 0) { r = nr + qr; i = ni + qi; }
                else { r = qr - nr; i = qi - ni; }
            }
            zr[k2a] = r;
            zi[k2a] = i;
        }
        wave_lds_sync();                              // all stage-B reads of `ex` are done
        {
            // lane holds X[k1b + 16 k2a + 256 k2b], k2b = bit-reversed t_lo
            const int k2b = ((t_lo & 1) << 1) | (t_lo >> 1);
            float2* Z = ex;
#pragma unroll
            for (int k2a = 0; k2a < 16; ++k2a)
                Z[k1b + 16 * k2a + (256 + kZPad) * k2b] = make_float2(zr[k2a], zi[k2a]);
        }
        wave_lds_sync();
        // ---- real-FFT split: bins k and 1024 - k from Z[k], Z[1024 - k]
        {
            const float2* Z = ex;
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int k = lane + 64 * it;
                if (k <= kNc / 2) {
                    const int k2 = (kNc - k) & (kNc - 1);
                    const float2 a = Z[k + kZPad * (k >> 8)];
                    const float2 b = Z[k2 + kZPad * (k2 >> 8)];
                    const float2 w = p.tw_post[k];
                    const float xer = 0.5f * (a.x + b.x), xei = 0.5f * (a.y - b.y);      // (Z[k] + conj Z[N-k]) / 2
                    const float xor_ = 0.5f * (a.y + b.y), xoi = -0.5f * (a.x - b.x);    // -i (Z[k] - conj Z[N-k]) / 2
                    const float tr = xor_ * w.x - xoi * w.y, ti = xor_ * w.y + xoi * w.x;
                    const float pr = xer + tr, pi = xei + ti, mr = xer - tr, mi = xei - ti;
                    P[k] = pr * pr + pi * pi;
                    if (k != kNc - k) P[kNc - k] = mr * mr + mi * mi;
                }
            }
        }
        wave_lds_sync();
        // ---- mel rows: 4 lanes per filter, 16 filters per round
        {
            float* out = p.melpow + ((long long)row * p.n_frames + frame) * p.n_mels;
            const int q = lane & 3;
            for (int m0 = 0; m0 < p.n_mels; m0 += 16) {
                const int m = m0 + (lane >> 2);
                float acc = 0.0f;
                if (m < p.n_mels) {
                    const int st = p.mel_start[m], len = p.mel_len[m];
                    const float* w = p.mel_w + p.mel_wofs[m];
                    for (int i = q; i < len; i += 4) acc = fmaf(w[i], P[st + i], acc);
                }
                acc += dpp_xor1(acc);
                acc += dpp_xor2(acc);
                if (q == 0 && m < p.n_mels) out[m] = acc;
            }
        }
        wave_lds_sync();                              // P / ex are rewritten by the next frame
    }
}

size_t stft_mel_lds_bytes(int n_fft, int hop, int frames_per_block) {
    const size_t span = (((size_t)(frames_per_block - 1) * hop + n_fft) * 4 + 15) & ~(size_t)15;
    return span + 4 * (size_t)kWaveScratch;
}

hipError_t stft_mel_configure(int lds_bytes) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stft_mel<true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    if (e != hipSuccess) return e;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_stft_mel<false>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
}

void launch_stft_mel(const StftParams& p, int n_rows, int sample_is_pcm16, hipStream_t s) {
    const size_t lds = stft_mel_lds_bytes(p.n_fft, p.hop, p.frames_per_block);
    const dim3 grid((unsigned)(n_rows * p.tiles_per_row));
    if (sample_is_pcm16) hipLaunchKernelGGL(k_stft_mel<true>, grid, dim3(256), lds, s, p);
    else hipLaunchKernelGGL(k_stft_mel<false>, grid, dim3(256), lds, s, p);
}

// ----------------------------------------------------------------------------
// K3: power_to_db(ref=max, amin=1e-10, top_db=80) -> z-score over the whole
// [M, T_raw] array (population std) -> centre crop / zero pad -> [T, M] or [M, T].
// One 256-thread block per chunk; dB values are recomputed per pass.
// ----------------------------------------------------------------------------
__device__ __forceinline__ float to_db(float s, float ref_db) {
#pragma clang fp contract(off)
    // separately rounded multiply and subtract (numpy order): the max element gives exactly 0
    const float db = 10.0f * log10f(fmaxf(1e-10f, s));
    return fmaxf(db - ref_db, -80.0f);
}

__global__ __launch_bounds__(256) void k_dbnorm(DbNormParams p) {
    __shared__ double shd[4];
    __shared__ float shf[4];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = p.n_frames * p.n_mels;
    const float* S = p.melpow + (long long)row * n;
    float mx = -INFINITY;
    for (int i = tid; i < n; i += 256) mx = fmaxf(mx, S[i]);
    mx = block_max(mx, shf);
    const float ref_db = 10.0f * log10f(fmaxf(1e-10f, mx));             // same two operations as to_db (no FMA)
    double acc = 0.0;
    for (int i = tid; i < n; i += 256) acc += (double)to_db(S[i], ref_db);
    const float mean = (float)(block_sum(acc, shd) / (double)n);
    acc = 0.0;
    for (int i = tid; i < n; i += 256) { const float d = to_db(S[i], ref_db) - mean; acc += (double)(d * d); }
    const float sd = sqrtf((float)(block_sum(acc, shd) / (double)n));
    const float den = sd + 1e-8f;
    // crop_or_pad_time (map_detector_core.py:185-195)
    const int T = p.n_frames, TT = p.target_frames, M = p.n_mels;
    const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);   // source frame = tt + shift
    float* out = p.out + (long long)row * TT * M;
    for (int i = tid; i < TT * M; i += 256) {
        const int tt = i / M, m = i - tt * M;
        const int src = tt + shift;
        float v = 0.0f;
        if (src >= 0 && src < T) v = (to_db(S[src * M + m], ref_db) - mean) / den;
        if (p.layout == 1) out[i] = v;
        else out[m * TT + tt] = v;
    }
}

void launch_dbnorm(const DbNormParams& p, int n_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_dbnorm, dim3(n_rows), dim3(256), 0, s, p);
}

void launch_rms_scale_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows,
                            float* scale, int32_t* ok, hipStream_t s) {
    hipLaunchKernelGGL(k_rms_scale<true>, dim3(n_rows), dim3(256), 0, s, (const void*)in, row_stride, lengths, n_in, scale, ok);
}
void launch_rms_scale_f32(const float* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows,
                          float* scale, int32_t* ok, hipStream_t s) {
    hipLaunchKernelGGL(k_rms_scale<false>, dim3(n_rows), dim3(256), 0, s, (const void*)in, row_stride, lengths, n_in, scale, ok);
}
void launch_rms_apply_f32(const float* in, int n_rows, int n, const float* scale, const int32_t* ok, float* out,
                          hipStream_t s) {
    const int bx = max(1, min(64, (n + 255) / 256));
    hipLaunchKernelGGL(k_rms_apply_f32, dim3(bx, n_rows), dim3(256), 0, s, in, n, scale, ok, out);
}
void launch_rms_apply_pcm16(const int16_t* in, int n_rows, int n, const float* scale, const int32_t* ok, int16_t* out,
                            hipStream_t s) {
    const int bx = max(1, min(64, (n + 255) / 256));
    hipLaunchKernelGGL(k_rms_apply_pcm16, dim3(bx, n_rows), dim3(256), 0, s, in, n, scale, ok, out);
}

}  // namespace amph
