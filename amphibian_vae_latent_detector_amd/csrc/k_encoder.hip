// Encoder kernels for gfx950: conv2d + BatchNorm(eval) + ReLU stack and the
// dense mu / log-variance heads (K4/K5), plus the radial decision (K6).
//
// Replaces `encoder(x)` at latent_space_exploration/map_detector_core.py:270
// (module built at :175-179, eval mode => BN uses running statistics only) and
// the decision block of 10_benchmark_folder_detection.py:175-199.
//
// Layout: activations are NHWC in HBM so that the implicit-GEMM reduction index
// k = (kh, kw, ci) reads 128-byte contiguous channel runs.  The host permutes
// the head weights once so that the NHWC flatten equals PyTorch's NCHW flatten.
//
// Arithmetic: fp32 throughout.  The GEMM-shaped layers use
// v_mfma_f32_32x32x2_f32, which is bit-for-bit a k-ordered fp32 fma chain
// (one rounding per product), so precision equals a VALU fp32 convolution.
#include "amph_internal.h"

namespace amph {

typedef float f32x16 __attribute__((ext_vector_type(16)));

// ----------------------------------------------------------------------------
// First layer (C_in == 1): direct convolution.  One thread = one output pixel x
// 4 output channels; the 8/16/... threads of a pixel write one contiguous
// C_out*4-byte run, so a wave's store is fully coalesced.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_conv_first(ConvFirstParams p, long long total) {
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
    if (g >= total) return;
    const int CG = p.C_out >> 2;
    const int cg = (int)(g % CG);
    const long long pix = g / CG;
    const int ow = (int)(pix % p.OW);
    const long long t = pix / p.OW;
    const int oh = (int)(t % p.OH);
    const long long b = t / p.OH;
    const float* x = p.x + b * (long long)p.H * p.W;
    const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int kh = 0; kh < p.k; ++kh) {
        const int ih = ih0 + kh;
        for (int kw = 0; kw < p.k; ++kw) {
            const int iw = iw0 + kw;
            float v = 0.0f;
            if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W) v = x[ih * p.W + iw];
            const float4 w = *reinterpret_cast<const float4*>(p.w + (kh * p.k + kw) * p.C_out + 4 * cg);
            acc.x = fmaf(v, w.x, acc.x);
            acc.y = fmaf(v, w.y, acc.y);
            acc.z = fmaf(v, w.z, acc.z);
            acc.w = fmaf(v, w.w, acc.w);
        }
    }
    const float4 sc = *reinterpret_cast<const float4*>(p.scale + 4 * cg);
    const float4 sh = *reinterpret_cast<const float4*>(p.shift + 4 * cg);
    float4 o;
    o.x = fmaxf(fmaf(acc.x, sc.x, sh.x), 0.0f);
    o.y = fmaxf(fmaf(acc.y, sc.y, sh.y), 0.0f);
    o.z = fmaxf(fmaf(acc.z, sc.z, sh.z), 0.0f);
    o.w = fmaxf(fmaf(acc.w, sc.w, sh.w), 0.0f);
    *reinterpret_cast<float4*>(p.y + pix * p.C_out + 4 * cg) = o;
}

void launch_conv_first(const ConvFirstParams& p, int n_rows, hipStream_t s) {
    const long long total = (long long)n_rows * p.OH * p.OW * (p.C_out >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_conv_first, dim3(blocks), dim3(256), 0, s, p, total);
}

// ----------------------------------------------------------------------------
// Implicit-GEMM convolution / dense layer on the f32 matrix cores.
//   C[m][n] = sum_k A[m][k] * Wt[n][k]
//   m = (b, oh, ow), n = c_out, k = (kh, kw, ci);  A is gathered from NHWC input.
// Block tile BM x 64 x 32, 256 threads = 4 waves, double-buffered LDS with one
// barrier per K slice; operands are staged global -> registers -> LDS so that
// the global loads of slice s+1 fly under the MFMAs of slice s.
// LDS rows are padded to 36 floats: ds_read_b128 of 16 rows x 16 B then covers
// all 64 banks (row stride 9 slots of 16 B is odd) -> conflict-free.
// Each lane feeds the MFMA with a float4 (4 consecutive k); MFMA step s of a
// k-group of 8 consumes k = {s, 4 + s}: a fixed permutation of the k order.
// ----------------------------------------------------------------------------
constexpr int kBN = 64, kBK = 32, kLdsRow = 36;

template <int BM>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemmParams p) {
    constexpr int A_ROWS = BM / 32;            // rows of the A tile each thread stages
    constexpr int NT = (BM == 128) ? 2 : 1;    // 32x32 tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [2][BM][36]
    float* Bs = As + 2 * BM * kLdsRow;                                // [2][64][36]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int m_blk = blockIdx.x * BM, n_blk = blockIdx.y * kBN;

    // per-thread A rows: output pixel -> input window origin
    int ih0[A_ROWS], iw0[A_ROWS];
    long long boff[A_ROWS];
    bool mval[A_ROWS];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m_blk + r0 + 32 * i;
        mval[i] = m < p.M;
        const int mm = mval[i] ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        ih0[i] = oh * p.stride - p.pad;
        iw0[i] = ow * p.stride - p.pad;
        boff[i] = (long long)b * p.H * p.W * p.C_in;
    }
    const float* wrow0 = p.w + (long long)(n_blk + r0) * p.K_pad + 4 * c4;
    const float* wrow1 = p.w + (long long)(n_blk + r0 + 32) * p.K_pad + 4 * c4;

    const int n_slices = p.K_pad / kBK;
    const int sps = n_slices / p.splits;
    const int ks_begin = blockIdx.z * sps, ks_end = ks_begin + sps;

    // staging registers (named scalars / fully unrolled arrays so they stay in VGPRs)
    float4 ra[A_ROWS];
    float4 rb0, rb1;
#define AMPH_LOAD_SLICE(ks_)                                                                              \
    {                                                                                                     \
        const int kflat_ = (ks_) * kBK + 4 * c4;                                                          \
        const bool kval_ = kflat_ < p.K;                                                                  \
        const int kpos_ = kflat_ / p.C_in, ci_ = kflat_ - kpos_ * p.C_in;                                 \
        const int kh_ = kpos_ / p.k, kw_ = kpos_ - kh_ * p.k;                                             \
        _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                              \
            const int ih_ = ih0[i] + kh_, iw_ = iw0[i] + kw_;                                             \
            float4 v_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
            if (kval_ && mval[i] && ih_ >= 0 && ih_ < p.H && iw_ >= 0 && iw_ < p.W)                       \
                v_ = *reinterpret_cast<const float4*>(p.x + boff[i] + ((long long)ih_ * p.W + iw_) * p.C_in + ci_); \
            ra[i] = v_;                                                                                   \
        }                                                                                                 \
        rb0 = *reinterpret_cast<const float4*>(wrow0 + (ks_) * kBK);                                      \
        rb1 = *reinterpret_cast<const float4*>(wrow1 + (ks_) * kBK);                                      \
    }
#define AMPH_STORE_SLICE(buf_)                                                                            \
    {                                                                                                     \
        float* a_ = As + (buf_) * BM * kLdsRow;                                                           \
        float* b_ = Bs + (buf_) * kBN * kLdsRow;                                                          \
        _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i)                                                \
            *reinterpret_cast<float4*>(a_ + (r0 + 32 * i) * kLdsRow + 4 * c4) = ra[i];                    \
        *reinterpret_cast<float4*>(b_ + r0 * kLdsRow + 4 * c4) = rb0;                                     \
        *reinterpret_cast<float4*>(b_ + (r0 + 32) * kLdsRow + 4 * c4) = rb1;                              \
    }

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int row_tile = (BM == 128) ? wave : (wave >> 1);
    const int col_tile0 = (BM == 128) ? 0 : (wave & 1);
    const int a_off = (row_tile * 32 + (lane & 31)) * kLdsRow + 4 * (lane >> 5);
    const int b_off = (col_tile0 * 32 + (lane & 31)) * kLdsRow + 4 * (lane >> 5);

    AMPH_LOAD_SLICE(ks_begin);
    AMPH_STORE_SLICE(0);
    __syncthreads();
    int buf = 0;
    for (int ks = ks_begin; ks < ks_end; ++ks) {
        const bool more = ks + 1 < ks_end;
        if (more) AMPH_LOAD_SLICE(ks + 1);
        const float* a = As + buf * BM * kLdsRow + a_off;
        const float* b = Bs + buf * kBN * kLdsRow + b_off;
#pragma unroll
        for (int kg = 0; kg < kBK / 8; ++kg) {
            const float4 fa = *reinterpret_cast<const float4*>(a + 8 * kg);
            float4 fb[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) fb[t] = *reinterpret_cast<const float4*>(b + t * 32 * kLdsRow + 8 * kg);
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.x, fb[t].x, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.y, fb[t].y, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.z, fb[t].z, acc[t], 0, 0, 0);
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa.w, fb[t].w, acc[t], 0, 0, 0);
            }
        }
        if (more) AMPH_STORE_SLICE(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: C/D map of the 32x32 tile: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n_blk + (col_tile0 + t) * 32 + (lane & 31);
        if (n >= p.N) continue;
        float sc = 1.0f, sh = 0.0f;
        if (p.epilogue == 0) { sc = p.scale[n]; sh = p.shift[n]; }
        float* ybase = p.y + (p.epilogue == 1 ? (long long)blockIdx.z * p.M * p.N : 0ll);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_blk + row_tile * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < p.M) {
                float v = acc[t][r];
                if (p.epilogue == 0) v = fmaxf(fmaf(v, sc, sh), 0.0f);
                ybase[(long long)m * p.N + n] = v;
            }
        }
    }
}

static size_t conv_gemm_lds(int bm) { return (size_t)2 * (bm + kBN) * kLdsRow * sizeof(float); }

void launch_conv_gemm(const ConvGemmParams& p, int bm, hipStream_t s) {
    const dim3 grid((unsigned)((p.M + bm - 1) / bm), (unsigned)((p.N + kBN - 1) / kBN), (unsigned)p.splits);
    if (bm == 128) hipLaunchKernelGGL(k_conv_gemm<128>, grid, dim3(256), conv_gemm_lds(128), s, p);
    else hipLaunchKernelGGL(k_conv_gemm<64>, grid, dim3(256), conv_gemm_lds(64), s, p);
}

// ----------------------------------------------------------------------------
// Head: mu / logvar = sum over split-K partials (fixed order) + bias.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_reduce(HeadReduceParams p, int n_rows) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int N = 2 * p.D;
    if (g >= n_rows * N) return;
    const int b = g / N, n = g - b * N;
    float acc = 0.0f;
    for (int s = 0; s < p.splits; ++s) acc += p.partial[((long long)s * n_rows + b) * N + n];
    acc += p.bias[n];
    if (n < p.D) p.mu[(long long)b * p.D + n] = acc;
    else if (p.logvar) p.logvar[(long long)b * p.D + (n - p.D)] = acc;
}

void launch_head_reduce(const HeadReduceParams& p, int n_rows, hipStream_t s) {
    const int total = n_rows * 2 * p.D;
    hipLaunchKernelGGL(k_head_reduce, dim3((total + 255) / 256), dim3(256), 0, s, p, n_rows);
}

// ----------------------------------------------------------------------------
// K6: radial decision.  d_k = float32 sqrt(np.sum(a*a)) with a = z - mu_k, using
// numpy's float32 pairwise summation order (n < 8 sequential; 8 <= n <= 128 eight
// strided accumulators), accept if (double)d_k <= thr_k; tie -> smallest rank.
// ----------------------------------------------------------------------------
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

__global__ __launch_bounds__(256) void k_decide_radial(RadialParams p, long long n_rows) {
    const long long b = (long long)blockIdx.x * 256 + threadIdx.x;
    if (b >= n_rows) return;
    const float* z = p.z + b * p.D;
    float best = INFINITY;
    int pred = -1, pred_rank = 0x7fffffff;
    for (int k = 0; k < p.K; ++k) {
        const float d = sqrtf(np_pairwise_sumsq(z, p.centroids + (long long)k * p.D, p.D));   // IEEE-rounded (not __fsqrt_rn: raw v_sqrt_f32, 1 ulp)
        best = fminf(best, d);
        if ((double)d <= p.thresholds[k] && p.rank[k] < pred_rank) { pred = k; pred_rank = p.rank[k]; }
    }
    p.pred[b] = pred;
    p.best_d[b] = best;
}

void launch_decide_radial(const RadialParams& p, long long n_rows, hipStream_t s) {
    hipLaunchKernelGGL(k_decide_radial, dim3((unsigned)((n_rows + 255) / 256)), dim3(256), 0, s, p, n_rows);
}

}  // namespace amph
