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

#include <algorithm>

namespace amph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

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
    o.x = fmaf(acc.x, sc.x, sh.x); o.y = fmaf(acc.y, sc.y, sh.y); o.z = fmaf(acc.z, sc.z, sh.z); o.w = fmaf(acc.w, sc.w, sh.w);
    // the next layer may take this tensor as an fp16 split operand (amph_internal.h kH2Max)
    if (p.range_flag && (h2_unsplittable(o.x) | h2_unsplittable(o.y) | h2_unsplittable(o.z) | h2_unsplittable(o.w))) *p.range_flag = 1;
    o.x = fmaxf(o.x, 0.0f); o.y = fmaxf(o.y, 0.0f); o.z = fmaxf(o.z, 0.0f); o.w = fmaxf(o.w, 0.0f);
    *reinterpret_cast<float4*>(p.y + pix * p.C_out + 4 * cg) = o;
}

void launch_conv_first(const ConvFirstParams& p, int n_rows, hipStream_t s) {
    const long long total = (long long)n_rows * p.OH * p.OW * (p.C_out >> 2);
    const unsigned blocks = (unsigned)((total + 255) / 256);
    hipLaunchKernelGGL(k_conv_first, dim3(blocks), dim3(256), 0, s, p, total);
}

// amph_encoder_forward's input is the CALLER's tensor: before a kernel splits it into fp16 terms (and builds a layer-1 activation from
// it whose bound `limit` was derived at create), raise the flag for any value at or beyond the limit, or NaN
__global__ __launch_bounds__(256) void k_range_check(const float* __restrict__ x, long long n4, long long n, float limit, int* flag) {
    bool bad = false;
    const long long g = (long long)blockIdx.x * 256 + threadIdx.x, step = (long long)gridDim.x * 256;
    for (long long i = g; i < n4; i += step) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        bad |= !(fabsf(v.x) < limit) | !(fabsf(v.y) < limit) | !(fabsf(v.z) < limit) | !(fabsf(v.w) < limit);
    }
    for (long long i = 4 * n4 + g; i < n; i += step) bad |= !(fabsf(x[i]) < limit);      // the tail (everything when x is not 16-byte aligned)
    if (bad) *flag = 1;
}
void launch_range_check(const float* x, long long n, float limit, int* flag, hipStream_t s) {
    if (n <= 0 || !flag) return;
    const long long n4 = (reinterpret_cast<uintptr_t>(x) & 15) == 0 ? n / 4 : 0;
    const unsigned blocks = (unsigned)std::min<long long>(1024, std::max<long long>(1, (std::max(n4, n - 4 * n4) + 255) / 256));
    hipLaunchKernelGGL(k_range_check, dim3(blocks), dim3(256), 0, s, x, n4, n, limit, flag);
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

// FUSE1: the layer's input is not read from HBM.  It is the output of the FIRST layer
// (C_in == 1 direct 3x3 conv + BN + ReLU), recomputed per A-tile element from an image patch
// held in LDS, with exactly k_conv_first's FMA order (so fused == unfused bitwise).  The
// first layer's 128 KB/chunk activation is then never written nor re-read 2.25x; its VALU
// work is interleaved with this kernel's MFMAs.  Requires the BM-row tile to be whole output
// rows of one chunk (host checks (OH*OW) % BM == 0 and BM % OW == 0).
//
// Pipeline: a 3-stage REGISTER ring.  Global loads for slice ks+3 are issued while slice ks is
// multiplied; the ring stage holding slice ks+1 is written to the spare LDS buffer at the end of
// the iteration (the compiler's counted vmcnt wait leaves two slices in flight), one barrier per
// slice, double-buffered LDS.
//
// What bounds it (profiles/r01_gemm_*; DESIGN.md "K4"): operand delivery.  In-kernel stamps of a
// wave-specialised variant (4 loader + 4 MFMA waves) showed the loaders busy 44k of 53k cycles and
// never waiting, the MFMA waves waiting 19k: per launch the 64x64 tiles pull 150 MB (im2col repeats
// every input 2.25x, every A tile is re-read per N tile, the weights per M tile) = 13 B/clk/CU,
// the fabric rate.  Prefetch depth, address-math trimming and wave roles therefore did not move it;
// the roles variant was dropped again (it starved the fused layer-1 recompute of issue slots).
template <int BM, bool FUSE1>
__global__ __launch_bounds__(256) void k_conv_gemm(ConvGemmParams p) {
    constexpr int A_ROWS = BM / 32;            // rows of the A tile each thread stages
    constexpr int NT = (BM == 128) ? 2 : 1;    // 32x32 tiles per wave
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                       // [2][BM][36]
    float* Bs = As + 2 * BM * kLdsRow;                                // [2][64][36]
    // FUSE1 only: first-layer tables and the image patch of this tile
    float* w1s = Bs + 2 * kBN * kLdsRow;                              // [9][C_in] then scale1, shift1 [C_in]
    float* patch = w1s + 11 * p.C_in;                                 // [patch_rows][W0 + 2 p1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int m_blk = blockIdx.x * BM, n_blk = blockIdx.y * kBN;
    // Co-resident waves of a SIMD run this same loop and fall into lockstep: with equal priority they
    // alternate MFMAs, finish their 16 together and then do their loads / address math / barrier
    // together with the matrix pipe idle (PMC: issue-stall = 2x the wave's own MFMA time).  A static
    // priority taken from the wave's slot on its SIMD (HW_ID[3:0]; the waves of one workgroup
    // normally hold the same slot on all four SIMDs) lets one wave run its MFMAs while the other
    // is in its non-matrix phase.
    if (p.flags & 1) {
        const unsigned slot = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 3u;
        if (slot == 1) __builtin_amdgcn_s_setprio(1);
        else if (slot == 2) __builtin_amdgcn_s_setprio(2);
        else if (slot == 3) __builtin_amdgcn_s_setprio(3);
    }

    // per-thread A rows: output pixel -> input window origin, base pointer of the window in the
    // NHWC input (never dereferenced for padded taps) and a bitmask of the k*k taps that are in range
    int ih0[A_ROWS], iw0[A_ROWS];
    const float* pa[A_ROWS];
    unsigned vmask[A_ROWS];
    const int ohw = p.OH * p.OW;
#pragma unroll
    for (int i = 0; i < A_ROWS; ++i) {
        const int m = m_blk + r0 + 32 * i;
        const bool mv = m < p.M;
        const int mm = mv ? m : 0;
        const int b = mm / ohw, rem = mm - b * ohw;
        const int oh = rem / p.OW, ow = rem - oh * p.OW;
        ih0[i] = oh * p.stride - p.pad;
        iw0[i] = ow * p.stride - p.pad;
        pa[i] = p.x + (long long)b * p.H * p.W * p.C_in + ((long long)ih0[i] * p.W + iw0[i]) * p.C_in;
        unsigned vm = 0;
        for (int kh = 0; kh < p.k; ++kh)
            for (int kw = 0; kw < p.k; ++kw)
                if (mv && ih0[i] + kh >= 0 && ih0[i] + kh < p.H && iw0[i] + kw >= 0 && iw0[i] + kw < p.W)
                    vm |= 1u << (kh * p.k + kw);
        vmask[i] = vm;
    }
    const int PW = p.W0 + 2 * p.p1;
    int patch_lo = 0;                          // image row held in patch row 0
    if (FUSE1) {
        const int b = m_blk / ohw, oh_first = (m_blk - b * ohw) / p.OW, n_oh = BM / p.OW;
        const int r1_lo = oh_first * p.stride - p.pad;
        const int r1_cnt = (n_oh - 1) * p.stride + p.k;
        patch_lo = r1_lo * p.s1 - p.p1;
        const int rows = (r1_cnt - 1) * p.s1 + 3;
        const float* img = p.img + (long long)b * p.H0 * p.W0;
        for (int i = tid; i < rows * PW; i += 256) {
            const int r = i / PW, cc = i - r * PW;
            const int ir = patch_lo + r, ic = cc - p.p1;
            patch[i] = (ir >= 0 && ir < p.H0 && ic >= 0 && ic < p.W0) ? img[ir * p.W0 + ic] : 0.0f;
        }
        const int nw = 9 * p.C_in;
        for (int i = tid; i < nw + 2 * p.C_in; i += 256)
            w1s[i] = i < nw ? p.w1[i] : (i < nw + p.C_in ? p.scale1[i - nw] : p.shift1[i - nw - p.C_in]);
        __syncthreads();
    }
    const float* wrow0 = p.w + (long long)(n_blk + r0) * p.K_pad + 4 * c4;
    const float* wrow1 = p.w + (long long)(n_blk + r0 + 32) * p.K_pad + 4 * c4;

    const int n_slices = p.K_pad / kBK;
    const int sps = n_slices / p.splits;
    const int ks_begin = blockIdx.z * sps, ks_end = ks_begin + sps;

    // ---- register ring: stage q holds the A rows / B rows of one K slice
    float4 sa0[A_ROWS], sa1[A_ROWS], sa2[A_ROWS];
    float4 sb0a, sb0b, sb1a, sb1b, sb2a, sb2b;
    // position of this thread's float4 inside the K slices, advanced slice by slice without divisions:
    // flat k = (kh*k + kw)*C_in + ci.  `ld` is the state of the next slice the ring will LOAD (plain
    // path), `fd` of the next slice the fused path will recompute.
    struct KPos { int kflat, kpos, kh, kw, ci; };
    auto kpos_init = [&](int ks) {
        KPos d;
        d.kflat = ks * kBK + 4 * c4;
        d.kpos = d.kflat / p.C_in;
        d.ci = d.kflat - d.kpos * p.C_in;
        d.kh = d.kpos / p.k;
        d.kw = d.kpos - d.kh * p.k;
        return d;
    };
    auto kpos_next = [&](KPos& d) {
        d.kflat += kBK;
        d.ci += kBK;
        while (d.ci >= p.C_in) {
            d.ci -= p.C_in;
            ++d.kpos;
            if (++d.kw == p.k) { d.kw = 0; ++d.kh; }
        }
    };
    KPos ld = kpos_init(ks_begin), fd = ld;
    int ld_left = ks_end - ks_begin, fd_left = ld_left;     // slices not yet consumed by each state
    const float* wp0 = wrow0 + ks_begin * kBK;
    const float* wp1 = wrow1 + ks_begin * kBK;
#ifdef AMPH_ABL_NOLOAD
#define AMPH_A_FETCH(ptr_) make_float4((float)koff_, 1.f, 1.f, 1.f)
#else
#define AMPH_A_FETCH(ptr_) (*reinterpret_cast<const float4*>(ptr_))
#endif
    // issue the global loads of the NEXT slice into ring stage q (past the end: reload the last slice)
#ifdef AMPH_ABL_NOLOAD   /* timing-only diagnostic build: no global operand loads */
#define AMPH_LOAD_B(q) { sb##q##a = make_float4(1.f, 1.f, 1.f, 1.f); sb##q##b = sb##q##a; }
#else
#define AMPH_LOAD_B(q)                                                                                    \
    {                                                                                                     \
        sb##q##a = *reinterpret_cast<const float4*>(wp0);                                                 \
        sb##q##b = *reinterpret_cast<const float4*>(wp1);                                                 \
    }
#endif
#define AMPH_LOAD_A(q)                                                                                    \
    {                                                                                                     \
        const bool kval_ = ld.kflat < p.K;                                                                \
        const int koff_ = (ld.kh * p.W + ld.kw) * p.C_in + ld.ci;                                         \
        _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i) {                                              \
            float4 a_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                  \
            if (kval_ && ((vmask[i] >> ld.kpos) & 1u)) a_ = AMPH_A_FETCH(pa[i] + koff_);                  \
            sa##q[i] = a_;                                                                                \
        }                                                                                                 \
    }
    // advance the load state by one slice (stays on the last slice once the end is reached)
#define AMPH_LOAD_ADVANCE()                                                                               \
    { if (--ld_left > 0) { kpos_next(ld); wp0 += kBK; wp1 += kBK; } else { ld_left = 1; } }
#define AMPH_STORE(q, buf_)                                                                               \
    {                                                                                                     \
        float* a_ = As + (buf_) * BM * kLdsRow;                                                           \
        float* b_ = Bs + (buf_) * kBN * kLdsRow;                                                          \
        _Pragma("unroll") for (int i = 0; i < A_ROWS; ++i)                                                \
            *reinterpret_cast<float4*>(a_ + (r0 + 32 * i) * kLdsRow + 4 * c4) = sa##q[i];                 \
        *reinterpret_cast<float4*>(b_ + r0 * kLdsRow + 4 * c4) = sb##q##a;                                \
        *reinterpret_cast<float4*>(b_ + (r0 + 32) * kLdsRow + 4 * c4) = sb##q##b;                         \
    }
    // FUSE1, software pipelined across the k-groups of one slice: F_DECODE fetches the slice's 9 tap
    // weights + BN, F_PATCH(i) the 3x3 image window of row i, F_ROW(q, i) turns the window fetched one
    // k-group earlier into layer-1 output (conv + BN + ReLU) in ring stage q
    float4 sc1 = make_float4(0.f, 0.f, 0.f, 0.f), sh1 = sc1, w1r[9];
    float pv[9];
    bool pok = false, f_kval = false;
    int f_kh = 0, f_kw = 0;
#pragma unroll
    for (int t = 0; t < 9; ++t) { w1r[t] = sc1; pv[t] = 0.0f; }
#define AMPH_F_DECODE()                                                                                   \
    {                                                                                                     \
        f_kval = fd.kflat < p.K; f_kh = fd.kh; f_kw = fd.kw;                                              \
        const float* wt_ = w1s + (f_kval ? fd.ci : 0);                                                    \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) w1r[t] = *reinterpret_cast<const float4*>(wt_ + t * p.C_in); \
        sc1 = *reinterpret_cast<const float4*>(wt_ + 9 * p.C_in);                                         \
        sh1 = *reinterpret_cast<const float4*>(wt_ + 10 * p.C_in);                                        \
    }
#define AMPH_F_ADVANCE()                                                                                  \
    { if (--fd_left > 0) { kpos_next(fd); } else { fd_left = 1; } }
#define AMPH_F_PATCH(i)                                                                                   \
    {                                                                                                     \
        const int ih_ = ih0[i] + f_kh, iw_ = iw0[i] + f_kw;                                               \
        pok = f_kval && ((vmask[i] >> (f_kh * p.k + f_kw)) & 1u);                                         \
        const float* pp_ = patch + (pok ? (ih_ * p.s1 - p.p1 - patch_lo) * PW + iw_ * p.s1 : 0);          \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) pv[t] = pp_[(t / 3) * PW + (t % 3)];                \
    }
#define AMPH_F_ROW(q, i)                                                                                  \
    {                                                                                                     \
        float4 a_ = make_float4(0.f, 0.f, 0.f, 0.f);                                                      \
        _Pragma("unroll") for (int t = 0; t < 9; ++t) {                                                   \
            a_.x = fmaf(pv[t], w1r[t].x, a_.x);                                                           \
            a_.y = fmaf(pv[t], w1r[t].y, a_.y);                                                           \
            a_.z = fmaf(pv[t], w1r[t].z, a_.z);                                                           \
            a_.w = fmaf(pv[t], w1r[t].w, a_.w);                                                           \
        }                                                                                                 \
        a_.x = pok ? fmaxf(fmaf(a_.x, sc1.x, sh1.x), 0.0f) : 0.0f;                                        \
        a_.y = pok ? fmaxf(fmaf(a_.y, sc1.y, sh1.y), 0.0f) : 0.0f;                                        \
        a_.z = pok ? fmaxf(fmaf(a_.z, sc1.z, sh1.z), 0.0f) : 0.0f;                                        \
        a_.w = pok ? fmaxf(fmaf(a_.w, sc1.w, sh1.w), 0.0f) : 0.0f;                                        \
        sa##q[i] = a_;                                                                                    \
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

#ifdef AMPH_ABL_NOMFMA   /* timing-only diagnostic build: operands stay live, no matrix instruction */
#define AMPH_MFMA(acc_, a_, b_) { asm volatile("" ::"v"(a_), "v"(b_)); }
#else
#define AMPH_MFMA(acc_, a_, b_) acc_ = __builtin_amdgcn_mfma_f32_32x32x2f32(a_, b_, acc_, 0, 0, 0);
#endif
    // one slice: multiply LDS buffer `buf_` (slice ks_), load slice ks_+3 into stage qload_,
    // stage slice ks_+1 (ring stage qstore_) into the other buffer
#ifdef AMPH_STAMPS
#define AMPH_GSTAMP(v_)                                                                                   \
    {                                                                                                     \
        __builtin_amdgcn_sched_barrier(0);                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");                        \
        __builtin_amdgcn_sched_barrier(0);                                                                \
    }
    unsigned long long t_a = 0, t_b = 0, t_c = 0, sum_work = 0, sum_wait = 0, t_start = 0;
    AMPH_GSTAMP(t_start)
#else
#define AMPH_GSTAMP(v_)
#endif
#define AMPH_STEP(buf_, qload_, qstore_)                                                                  \
    {                                                                                                     \
        AMPH_GSTAMP(t_a)                                                                                  \
        AMPH_LOAD_B(qload_);                                                                              \
        if (!FUSE1) AMPH_LOAD_A(qload_)                                                                   \
        else { AMPH_F_DECODE(); AMPH_F_PATCH(0); }                                                        \
        AMPH_LOAD_ADVANCE()                                                                               \
        const float* a = As + (buf_) * BM * kLdsRow + a_off;                                              \
        const float* b = Bs + (buf_) * kBN * kLdsRow + b_off;                                             \
        _Pragma("unroll") for (int kg = 0; kg < kBK / 8; ++kg) {                                          \
            const float4 fa = *reinterpret_cast<const float4*>(a + 8 * kg);                               \
            float4 fb[NT];                                                                                \
            _Pragma("unroll") for (int t = 0; t < NT; ++t)                                                \
                fb[t] = *reinterpret_cast<const float4*>(b + t * 32 * kLdsRow + 8 * kg);                  \
            const float fav[4] = {fa.x, fa.y, fa.z, fa.w};                                                \
            _Pragma("unroll") for (int st = 0; st < 4; ++st)                                              \
                _Pragma("unroll") for (int t = 0; t < NT; ++t) {                                          \
                    const float fbv = st == 0 ? fb[t].x : st == 1 ? fb[t].y : st == 2 ? fb[t].z : fb[t].w; \
                    AMPH_MFMA(acc[t], fav[st], fbv)                                                       \
                }                                                                                         \
            if (FUSE1 && kg < A_ROWS) {                                                                   \
                /* layer-1 output of A row kg of the NEXT slice from the window read one k-group       */ \
                /* earlier, and the window of row kg + 1; the group barriers spread this VALU / LDS    */ \
                /* work between this k-group's MFMAs (a wave issues in order)                          */ \
                AMPH_F_ROW(qstore_, kg);                                                                  \
                if (kg + 1 < A_ROWS) AMPH_F_PATCH(kg + 1);                                                \
                _Pragma("unroll") for (int g = 0; g < 4 * NT; ++g) {                                      \
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);                                    \
                    __builtin_amdgcn_sched_group_barrier(0x002, 96 / (4 * NT), 0);                        \
                    __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);                                    \
                }                                                                                         \
            }                                                                                             \
        }                                                                                                 \
        if (FUSE1) AMPH_F_ADVANCE()                                                                       \
        AMPH_STORE(qstore_, (buf_) ^ 1);                                                                  \
        AMPH_GSTAMP(t_b)                                                                                  \
        __syncthreads();                                                                                  \
        AMPH_GSTAMP(t_c)                                                                                  \
        AMPH_GACC()                                                                                       \
    }
#ifdef AMPH_STAMPS
#define AMPH_GACC() { sum_work += t_b - t_a; sum_wait += t_c - t_b; }
#else
#define AMPH_GACC()
#endif

    // prologue: slices 0, 1, 2 into the ring (B always; A gathered, or slice 0 recomputed), slice 0 to LDS.
    // Plain path: the load state runs three slices ahead of the multiply.  Fused path: B runs three
    // ahead, the recompute state `fd` one ahead (it fills the ring stage that is stored next).
    AMPH_LOAD_B(0);
    if (!FUSE1) AMPH_LOAD_A(0)
    AMPH_LOAD_ADVANCE()
    AMPH_LOAD_B(1);
    if (!FUSE1) AMPH_LOAD_A(1)
    AMPH_LOAD_ADVANCE()
    AMPH_LOAD_B(2);
    if (!FUSE1) AMPH_LOAD_A(2)
    AMPH_LOAD_ADVANCE()
    if (FUSE1) {
        AMPH_F_DECODE();
#pragma unroll
        for (int i = 0; i < A_ROWS; ++i) { AMPH_F_PATCH(i); AMPH_F_ROW(0, i); }
        AMPH_F_ADVANCE()
    }
    AMPH_STORE(0, 0);
    __syncthreads();
    // main loop, unrolled by the ring length so that every stage index is static; the LDS buffer
    // parity alternates 0,1,0 / 1,0,1 across the two halves of a 6-slice period
    for (int ks = ks_begin; ks < ks_end; ks += 6) {
        AMPH_STEP(0, 0, 1)
        if (ks + 1 < ks_end) AMPH_STEP(1, 1, 2)
        if (ks + 2 < ks_end) AMPH_STEP(0, 2, 0)
        if (ks + 3 < ks_end) AMPH_STEP(1, 0, 1)
        if (ks + 4 < ks_end) AMPH_STEP(0, 1, 2)
        if (ks + 5 < ks_end) AMPH_STEP(1, 2, 0)
    }
#ifdef AMPH_STAMPS
    if (tid == 0 && p.stamps) {
        unsigned long long t_end;
        AMPH_GSTAMP(t_end)
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
        o[0] = sum_work; o[1] = sum_wait; o[2] = t_end - t_start; o[3] = 1;
        o[4] = o[5] = o[6] = o[7] = 0;
    }
#endif

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

// ----------------------------------------------------------------------------
// Second-generation plain (non-fused) implicit GEMM, used when C_in % 32 == 0 (then a 32-deep K slice
// lies inside ONE tap (kh, kw), the tap state is wave-uniform scalar work and K needs no padding).
//
// What the first kernel lost (PMC, profiles/r01_pmc_sq_v1.txt): its waves spent 55 % of their life
// stalled at MFMA issue and 23 % at waits, because the compiler clusters the 16 MFMAs of a slice and
// puts every load / address / LDS instruction, the vmcnt wait, the LDS stores and the barrier in one
// block between the clusters - with the matrix pipe idle meanwhile - and the co-resident waves of a
// SIMD run the same program in phase.  Here the instruction order is pinned (sched_barrier) so that
// all of that issues in the 64-cycle shadows of the MFMAs:
//   * THREE LDS slice buffers: slice s+2 is stored in the middle of iteration s (its global loads were
//     issued at the top of iteration s-1), so no store sits in front of the barrier;
//   * the first fragments of slice s+1 (stored in iteration s-1, published by the previous barrier)
//     are read BEFORE the barrier that ends iteration s, under the last MFMAs: the matrix pipe does
//     not wait for an LDS round trip after the barrier;
//   * A loads are unconditional (padded taps read a zero line), so the loop body is one basic block
//     and the compiler's vmcnt counts are exact.
// Same MFMA k order as k_conv_gemm (step s of a k-group of 8 consumes k = {s, 4 + s}) => bit-identical.
// ----------------------------------------------------------------------------
#define AMPH_SB() __builtin_amdgcn_sched_barrier(0)

// floor(n / d) for 0 <= n < 2^22, d >= 1, rcp = 1.0f / d: float estimate (off by at most one) + correction.  The
// compiler's 32-bit integer division is ~40 VALU instructions; the fused kernel's per-thread setup had a dozen of them.
__device__ __forceinline__ int div_small(int n, int d, float rcp) {
    int q = (int)((float)n * rcp);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

template <int BM, bool ACC2>
__global__ __launch_bounds__(256) void k_conv_gemm_p3(ConvGemmParams p) {
    static_assert(BM == 64, "one 32x32 tile per wave");
    constexpr int kABuf = BM * kLdsRow, kBBuf = kBN * kLdsRow;          // floats per slice buffer
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                         // [3][BM][36]
    float* Bs = As + 3 * kABuf;                                         // [3][64][36]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int m_blk = blockIdx.x * BM, n_blk = blockIdx.y * kBN;
    const int n_slices = p.K_pad / kBK;
    const int sps = n_slices / p.splits;
    const int ks_begin = blockIdx.z * sps, ks_end = ks_begin + sps;

    // Addressing: buffer loads - a resource (SGPRs) + a 32-bit per-lane byte offset that never changes + a wave-uniform
    // offset in an SGPR that scalar instructions advance (`buffer_load_dwordx4 v, v_off, s[rsrc], s_off offen`).  f32 MFMAs
    // and VALU instructions time-share the SIMD (tools/probe/mfma_valu_samewave.hip), so every per-slice VALU instruction
    // is MFMA time lost, and a load with one 32-bit address VGPR costs 11 cycles of issue in an MFMA gap against 26 for the
    // 64-bit-VGPR-address form.  (Written as global loads from `uniform pointer + lane offset`, the compiler strength-
    // reduces the sum back into a per-lane 64-bit pointer.)
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) - p.x_guard), 0, -1, 0x00020000);
#define AMPH_P3_BUFLOAD(rs_, voff_, soff_) __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_, voff_, soff_, 0))
    // register stages (plain named variables: a struct passed to a lambda ends up in scratch memory)
    float4 sa0x, sa0y, sa1x, sa1y, sa2x, sa2y;           // A rows r0 and r0 + 32 of the three stages
    float4 sb0a, sb0b, sb1a, sb1b, sb2a, sb2b;
    // B (weights [N][K_pad]): the three prologue slices are requested before any setup arithmetic
    int w_soff = ks_begin * kBK * 4;                     // byte offset of the next B slice (wave-uniform)
    const unsigned woff0 = ((unsigned)(n_blk + r0) * (unsigned)p.K_pad + 4u * c4) * 4u, woff1 = woff0 + 32u * (unsigned)p.K_pad * 4u;
    {
        const int o1 = sps > 1 ? kBK * 4 : 0, o2 = sps > 2 ? 2 * kBK * 4 : o1;          // slice min(s, sps - 1), as AMPH_P3_ADV1 clamps
        sb0a = AMPH_P3_BUFLOAD(rs_w, woff0, w_soff);      sb0b = AMPH_P3_BUFLOAD(rs_w, woff1, w_soff);
        sb1a = AMPH_P3_BUFLOAD(rs_w, woff0, w_soff + o1); sb1b = AMPH_P3_BUFLOAD(rs_w, woff1, w_soff + o1);
        sb2a = AMPH_P3_BUFLOAD(rs_w, woff0, w_soff + o2); sb2b = AMPH_P3_BUFLOAD(rs_w, woff1, w_soff + o2);
    }
    // A (NHWC input): base = x - guard; lane offset = guard + (window origin, channel 4 c4) for a tap inside the image,
    // 0 for a padded one - the address then falls into the zero guard the host keeps in front of x (conv_gemm_x_guard)
    const int ohw = p.OH * p.OW;
    const float rcp_ohw = 1.0f / (float)ohw, rcp_ow = 1.0f / (float)p.OW;
    unsigned aoff0, aoff1, vmask0, vmask1;
#define AMPH_P3_ROW(i)                                                                     \
    {                                                                                      \
        const int m = m_blk + r0 + 32 * i;                                                 \
        const bool mv = m < p.M;                                                           \
        const int mm = mv ? m : 0;                                                         \
        const int b = div_small(mm, ohw, rcp_ohw), rem = mm - b * ohw;                     \
        const int oh = div_small(rem, p.OW, rcp_ow), ow = rem - oh * p.OW;                 \
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;                \
        aoff##i = (unsigned)p.x_guard + 4u * (unsigned)(((b * p.H + ih0) * p.W + iw0) * p.C_in + 4 * c4); \
        unsigned vm = 0;                                                                   \
        if (p.k == 3) {                                                                    \
            const unsigned hb = ((unsigned)iw0 < (unsigned)p.W ? 0x049u : 0u) | ((unsigned)(iw0 + 1) < (unsigned)p.W ? 0x092u : 0u) | \
                                ((unsigned)(iw0 + 2) < (unsigned)p.W ? 0x124u : 0u);       \
            const unsigned rb = ((unsigned)ih0 < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ih0 + 1) < (unsigned)p.H ? 0x038u : 0u) | \
                                ((unsigned)(ih0 + 2) < (unsigned)p.H ? 0x1c0u : 0u);       \
            vm = hb & rb;                                                                  \
        } else {                                                                           \
            for (int kh = 0; kh < p.k; ++kh)                                               \
                for (int kw = 0; kw < p.k; ++kw)                                           \
                    if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) vm |= 1u << (kh * p.k + kw); \
        }                                                                                  \
        vmask##i = mv ? vm : 0u;                                                           \
    }
    AMPH_P3_ROW(0)
    AMPH_P3_ROW(1)
#undef AMPH_P3_ROW

    // wave-uniform position of the next slice to LOAD: tap (kh, kw) = kpos, first channel ci0 - kept as the two things the
    // loads use: the byte offset of (tap, ci0) inside a window, 4 ((kh W + kw) C_in + ci0), and the tap's mask bit.  In NHWC
    // the next channel slice and the next tap of a row are both +128 bytes; only a new tap row adds 4 (W - k) C_in.
    // (readfirstlane pins them to SGPRs: left alone, the compiler carries the tap index in a VGPR for the mask test and
    // then wraps every buffer load whose scalar offset derives from it in a waterfall loop.)
    int l_left = sps;                                   // slices not yet loaded (stays on the last one at the end)
    int l_ci0, l_kw, x_soff;
    unsigned l_bit;
    {
        const int kpos = (ks_begin * kBK) / p.C_in;
        l_ci0 = ks_begin * kBK - kpos * p.C_in;
        const int kh = kpos / p.k;
        l_kw = kpos - kh * p.k;
        x_soff = 4 * ((kh * p.W + l_kw) * p.C_in + l_ci0);
        l_bit = 1u << kpos;
    }
    const int row_skip = 4 * (p.W - p.k) * p.C_in;

// the load of one slice in pieces small enough for one 64-cycle MFMA shadow each (one wave per SIMD has no partner to
// cover a long gap: with everything behind the first MFMA the second waited ~190 cycles for the issue stream)
#define AMPH_P3_LOADB(q)                                                                   \
    {                                                                                      \
        sb##q##a = AMPH_P3_BUFLOAD(rs_w, woff0, w_soff);                                   \
        sb##q##b = AMPH_P3_BUFLOAD(rs_w, woff1, w_soff);                                   \
    }
#define AMPH_P3_LOADA0(q)                                                                  \
    {                                                                                      \
        sa##q##x = AMPH_P3_BUFLOAD(rs_x, (vmask0 & l_bit) ? aoff0 : 0u, x_soff);           \
    }
#define AMPH_P3_LOADA1(q)                                                                  \
    {                                                                                      \
        sa##q##y = AMPH_P3_BUFLOAD(rs_x, (vmask1 & l_bit) ? aoff1 : 0u, x_soff);           \
    }
    /* advance with selects only: the loop body stays one basic block */
#define AMPH_P3_ADV1()                                                                     \
    {                                                                                      \
        const int adv_ = l_left > 1 ? 1 : 0;                                               \
        l_left -= adv_;                                                                    \
        w_soff += kBK * 4 * adv_;                                                          \
        x_soff += kBK * 4 * adv_;                                                          \
        l_ci0 += kBK * adv_;                                                               \
    }
#define AMPH_P3_ADV2()                                                                     \
    {                                                                                      \
        const int wrap_ = l_ci0 >= p.C_in ? 1 : 0;                                         \
        l_ci0 -= wrap_ * p.C_in;                                                           \
        l_bit = __builtin_amdgcn_readfirstlane(l_bit << wrap_);                            \
        l_kw += wrap_;                                                                     \
        const int wrap2_ = l_kw >= p.k ? 1 : 0;                                            \
        l_kw -= wrap2_ * p.k;                                                              \
        x_soff = __builtin_amdgcn_readfirstlane(x_soff + wrap2_ * row_skip);               \
        w_soff = __builtin_amdgcn_readfirstlane(w_soff);                                   \
    }
#define AMPH_P3_LOAD(q) { AMPH_P3_LOADB(q) AMPH_P3_LOADA0(q) AMPH_P3_LOADA1(q) AMPH_P3_ADV1() AMPH_P3_ADV2() }
#define AMPH_P3_LOADA(q) { AMPH_P3_LOADA0(q) AMPH_P3_LOADA1(q) AMPH_P3_ADV1() AMPH_P3_ADV2() }
    const int st_off = r0 * kLdsRow + 4 * c4;
#define AMPH_P3_STORE(q, buf_)                                                             \
    {                                                                                      \
        float* a_ = As + (buf_) * kABuf + st_off;                                          \
        float* b_ = Bs + (buf_) * kBBuf + st_off;                                          \
        *reinterpret_cast<float4*>(a_) = sa##q##x;                                         \
        *reinterpret_cast<float4*>(a_ + 32 * kLdsRow) = sa##q##y;                          \
        *reinterpret_cast<float4*>(b_) = sb##q##a;                                         \
        *reinterpret_cast<float4*>(b_ + 32 * kLdsRow) = sb##q##b;                          \
    }

    f32x16 acc, acc2;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = acc2[r] = 0.0f;
    const int row_tile = wave >> 1, col_tile = wave & 1;
    // BN constants of the epilogue, requested now: loaded in the epilogue they cost every workgroup a global round trip
    // with nothing left to hide it (columns past N read the vector's first element, the split-K form the weights': unused)
    float4 e_sc0, e_sc1, e_sc2, e_sc3, e_sh0, e_sh1, e_sh2, e_sh3;
    {
        const int nb = n_blk + col_tile * 32 + 4 * (lane >> 5);
        const bool bn = p.epilogue == 0;
#define AMPH_P3_BN(g)                                                                      \
        e_sc##g = *reinterpret_cast<const float4*>((bn ? p.scale : p.w) + ((bn && nb + 8 * g < p.N) ? nb + 8 * g : 0)); \
        e_sh##g = *reinterpret_cast<const float4*>((bn ? p.shift : p.w) + ((bn && nb + 8 * g < p.N) ? nb + 8 * g : 0));
        AMPH_P3_BN(0) AMPH_P3_BN(1) AMPH_P3_BN(2) AMPH_P3_BN(3)
#undef AMPH_P3_BN
    }
    const float* a_rd = As + (row_tile * 32 + (lane & 31)) * kLdsRow + 4 * (lane >> 5);
    const float* b_rd = Bs + (col_tile * 32 + (lane & 31)) * kLdsRow + 4 * (lane >> 5);

#define AMPH_P3_READ(fa_, fb_, buf_, kg_)                                                  \
    fa_ = *reinterpret_cast<const float4*>(a_rd + (buf_) * kABuf + 8 * (kg_));            \
    fb_ = *reinterpret_cast<const float4*>(b_rd + (buf_) * kBBuf + 8 * (kg_));
// The B fragment goes in as the MFMA's FIRST operand: the accumulator then holds the TRANSPOSED tile (lane = output
// row m, registers 4g..4g+3 = four consecutive channels n), so the epilogue is 4 x 16-byte stores per lane instead of
// 16 dword stores (store issue was ~7k cycles per tile).  Products commute: the sums are bit-identical.
#define AMPH_P3_MFMA(x_, y_) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y_, x_, acc, 0, 0, 0);
// ACC2: the odd k-steps go to a second accumulator, so that no MFMA waits on the one in flight
#define AMPH_P3_MFMB(x_, y_)                                                               \
    if constexpr (ACC2) acc2 = __builtin_amdgcn_mfma_f32_32x32x2f32(y_, x_, acc2, 0, 0, 0); \
    else acc = __builtin_amdgcn_mfma_f32_32x32x2f32(y_, x_, acc, 0, 0, 0);
#ifdef AMPH_STAMPS
#define AMPH_P3_T(v_) AMPH_GSTAMP(v_)
#define AMPH_P3_WAITLOADS() asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
#define AMPH_P3_ACC() { q_p1 += ts1 - ts0; q_ld += ts2 - ts1; q_p2 += ts3 - ts2; q_bar += ts4 - ts3; }
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, q_p1 = 0, q_ld = 0, q_p2 = 0, q_bar = 0, q_begin = 0;
    AMPH_GSTAMP(q_begin)
    const unsigned long long q_rt0 = __builtin_amdgcn_s_memrealtime();
#else
#define AMPH_P3_T(v_)
#define AMPH_P3_WAITLOADS()
#define AMPH_P3_ACC()
#endif
    // one slice from LDS buffer R; stage R receives slice s+3, stage (R+2)%3 goes to buffer (R+2)%3
#define AMPH_P3_STEP(R, QL, QS)                                                            \
    {                                                                                      \
        float4 f1a, f1b, f2a, f2b, f3a, f3b, n0a, n0b;                                     \
        AMPH_P3_T(ts0)                                                                     \
        AMPH_P3_READ(f1a, f1b, R, 1)                                                       \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f0a.x, f0b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_LOADB(QL)                                                                  \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMB(f0a.y, f0b.y)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_READ(f2a, f2b, R, 2)                                                       \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f0a.z, f0b.z)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_LOADA0(QL)                                                                 \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMB(f0a.w, f0b.w)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_LOADA1(QL)                                                                 \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f1a.x, f1b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_READ(f3a, f3b, R, 3)                                                       \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMB(f1a.y, f1b.y)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_ADV1()                                                                     \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f1a.z, f1b.z)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_ADV2()                                                                     \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMB(f1a.w, f1b.w)                                                         \
        AMPH_P3_MFMA(f2a.x, f2b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_T(ts1)                                                                     \
        AMPH_P3_WAITLOADS()                                                                \
        AMPH_P3_T(ts2)                                                                     \
        AMPH_P3_STORE(QS, (R + 2) % 3)                                                     \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMB(f2a.y, f2b.y)                                                         \
        AMPH_P3_MFMA(f2a.z, f2b.z)                                                         \
        AMPH_P3_MFMB(f2a.w, f2b.w)                                                         \
        AMPH_P3_MFMA(f3a.x, f3b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_READ(n0a, n0b, (R + 1) % 3, 0)                                             \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMB(f3a.y, f3b.y)                                                         \
        AMPH_P3_MFMA(f3a.z, f3b.z)                                                         \
        AMPH_P3_MFMB(f3a.w, f3b.w)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_T(ts3)                                                                     \
        __syncthreads();                                                                   \
        AMPH_P3_T(ts4)                                                                     \
        AMPH_P3_ACC()                                                                      \
        f0a = n0a; f0b = n0b;                                                              \
    }

    // prologue: slices 0, 1, 2 into the register stages, 0 and 1 on to LDS
    AMPH_P3_LOADA(0)                                     // (their B halves were requested at the top)
    AMPH_P3_LOADA(1)
    AMPH_P3_LOADA(2)
    AMPH_P3_STORE(0, 0)
    AMPH_P3_STORE(1, 1)
    __syncthreads();
    float4 f0a, f0b;
    AMPH_P3_READ(f0a, f0b, 0, 0)
    // three slices per trip in ONE basic block (the compiler's waitcnt scoreboard is exact inside a block;
    // across the block boundaries of a conditionally executed step it merges conservatively and drains
    // the load ring early), then the 0..2 remaining slices
    int ks = ks_begin;
    for (; ks + 3 <= ks_end; ks += 3) {
        AMPH_P3_STEP(0, 0, 2)
        AMPH_P3_STEP(1, 1, 0)
        AMPH_P3_STEP(2, 2, 1)
    }
    if (ks < ks_end) {
        AMPH_P3_STEP(0, 0, 2)
        if (ks + 1 < ks_end) AMPH_P3_STEP(1, 1, 0)
    }

#ifdef AMPH_STAMPS
    if (tid == 0 && p.stamps) {
        unsigned long long q_end;
        AMPH_GSTAMP(q_end)
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
        o[0] = q_p1; o[1] = q_ld; o[2] = q_p2; o[3] = q_bar; o[4] = q_end - q_begin;
        o[5] = __builtin_amdgcn_s_memrealtime() - q_rt0;   // 100 MHz ticks: clock = o[4] / o[5] * 100 MHz
        o[6] = o[7] = 0;
    }
#endif
    if constexpr (ACC2) {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] += acc2[r];
    }
    // epilogue on the transposed accumulator: lane = row m, register group g = channels n0 .. n0 + 3
    {
        const int m = m_blk + row_tile * 32 + (lane & 31);
        float* ybase = p.y + (p.epilogue == 1 ? (long long)blockIdx.z * p.M * p.N : 0ll);
        if (m < p.M) {
#define AMPH_P3_OUT(g)                                                                     \
            {                                                                              \
                const int n0 = n_blk + col_tile * 32 + 8 * g + 4 * (lane >> 5);            \
                if (n0 < p.N) {                                                      /* N % 4 == 0 (host check) */ \
                    float4 v = make_float4(acc[4 * g], acc[4 * g + 1], acc[4 * g + 2], acc[4 * g + 3]); \
                    if (p.epilogue == 0) {                                                 \
                        v.x = fmaxf(fmaf(v.x, e_sc##g.x, e_sh##g.x), 0.0f); v.y = fmaxf(fmaf(v.y, e_sc##g.y, e_sh##g.y), 0.0f); \
                        v.z = fmaxf(fmaf(v.z, e_sc##g.z, e_sh##g.z), 0.0f); v.w = fmaxf(fmaf(v.w, e_sc##g.w, e_sh##g.w), 0.0f); \
                    }                                                                      \
                    *reinterpret_cast<float4*>(ybase + (long long)m * p.N + n0) = v;       \
                }                                                                          \
            }
            AMPH_P3_OUT(0) AMPH_P3_OUT(1) AMPH_P3_OUT(2) AMPH_P3_OUT(3)
#undef AMPH_P3_OUT
        }
    }
}

// ----------------------------------------------------------------------------
// Fused conv1 -> conv2 in the second-generation structure ("f3"): k_conv_gemm_p3 whose A stage is not
// loaded but RECOMPUTED - layer 1 (C_in == 1 direct 3x3 conv + BN + ReLU, exactly k_conv_first's FMA
// order, so fused == unfused bitwise) evaluated from an image patch in LDS for the 2 x 4 A-tile values
// a thread stages per slice.  Requires layer 1 to have 32 channels (= one K slice per conv2 tap: the
// thread's 4 channels, hence its 9 x 4 tap weights + BN constants, are the same for every slice and
// live in registers) and a 3x3 first layer.  The ~100 VALU + 18 LDS reads per slice are placed in the
// shadows of MFMAs 1..10, the LDS store of the finished stage after MFMA 11.
// LDS 3 x 36.9 KB + patch = 62 KB: two workgroups per CU, so the 1024 workgroups of a 256-chunk batch
// run as two full rounds (the first-generation fused kernel: 43 KB, three per CU, 768 + 256).
// ----------------------------------------------------------------------------
template <int BM>
__global__ __launch_bounds__(256) void k_conv_gemm_f3(ConvGemmParams p) {
    static_assert(BM == 64, "one 32x32 tile per wave");
    constexpr int kABuf = BM * kLdsRow, kBBuf = kBN * kLdsRow;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* As = reinterpret_cast<float*>(smem);                         // [3][BM][36]
    float* Bs = As + 3 * kABuf;                                         // [3][64][36]
    float* patch = Bs + 3 * kBBuf;                                      // [rows][W0 + 2 p1]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int n_blk = blockIdx.y * kBN;
    const int ohw = p.OH * p.OW;
    const int PW = p.W0 + 2 * p.p1;
    const float rcp_pw = 1.0f / (float)PW, rcp_ow = 1.0f / (float)p.OW, rcp_ohw = 1.0f / (float)ohw;
    const int n_tiles = (p.M + BM - 1) / BM;
#ifdef AMPH_STAMPS
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, ts4 = 0, q_p1 = 0, q_ld = 0, q_p2 = 0, q_bar = 0, q_begin = 0, q_loop = 0,
                       q_tile = 0, q_pro = 0, q_main = 0;
    AMPH_GSTAMP(q_begin)
    const unsigned long long q_rt0 = __builtin_amdgcn_s_memrealtime();
#endif
    // this thread's 4 first-layer channels: 9 tap weights + BN constants, in registers for the whole kernel
    float4 w1_0, w1_1, w1_2, w1_3, w1_4, w1_5, w1_6, w1_7, w1_8, sc1, sh1;
    {
        const float* wt = p.w1 + 4 * c4;
        w1_0 = *reinterpret_cast<const float4*>(wt + 0 * p.C_in); w1_1 = *reinterpret_cast<const float4*>(wt + 1 * p.C_in);
        w1_2 = *reinterpret_cast<const float4*>(wt + 2 * p.C_in); w1_3 = *reinterpret_cast<const float4*>(wt + 3 * p.C_in);
        w1_4 = *reinterpret_cast<const float4*>(wt + 4 * p.C_in); w1_5 = *reinterpret_cast<const float4*>(wt + 5 * p.C_in);
        w1_6 = *reinterpret_cast<const float4*>(wt + 6 * p.C_in); w1_7 = *reinterpret_cast<const float4*>(wt + 7 * p.C_in);
        w1_8 = *reinterpret_cast<const float4*>(wt + 8 * p.C_in);
        sc1 = *reinterpret_cast<const float4*>(p.scale1 + 4 * c4);
        sh1 = *reinterpret_cast<const float4*>(p.shift1 + 4 * c4);
    }
    // the same constants as channel PAIRS: layer 1 is evaluated with v_pk_fma_f32 (two IEEE fmas per instruction, so the
    // per-element FMA order - and every bit of the result - is k_conv_first's).  f32 MFMAs and VALU instructions time-share
    // a SIMD (tools/probe/mfma_valu_samewave.hip: every VALU instruction adds its 4 cycles to the MFMA time, and every
    // MFMA -> VALU -> MFMA switch 12 more), so the recompute is paid in full: halve its instruction count, keep it in few gaps.
#define AMPH_F3_PAIR(t) const f32x2 wl_##t = {w1_##t.x, w1_##t.y}, wh_##t = {w1_##t.z, w1_##t.w};
    AMPH_F3_PAIR(0) AMPH_F3_PAIR(1) AMPH_F3_PAIR(2) AMPH_F3_PAIR(3) AMPH_F3_PAIR(4) AMPH_F3_PAIR(5) AMPH_F3_PAIR(6) AMPH_F3_PAIR(7) AMPH_F3_PAIR(8)
    const f32x2 sc1l = {sc1.x, sc1.y}, sc1h = {sc1.z, sc1.w}, sh1l = {sh1.x, sh1.y}, sh1h = {sh1.z, sh1.w};
    // PERSISTENT over the M tiles (grid.x = two workgroups per CU): a second dispatch round of this kernel started
    // 5-8 us after the workgroups it replaced had ended (start / end histograms of the stamps build); one resident
    // round that walks its tiles has no such gap
    // patch slots of this thread (the same for every tile): element i = tid + 256 u of the [rows][PW] patch
    constexpr int PU = 8;
    const int n_oh = div_small(BM, p.OW, rcp_ow);
    const int r1_cnt = (n_oh - 1) * p.stride + p.k;
    const int n_patch = ((r1_cnt - 1) * p.s1 + 3) * PW;            // <= 256 * PU (host check)
    int p_r[PU], p_ic[PU];
#pragma unroll
    for (int u = 0; u < PU; ++u) {
        const int i = tid + 256 * u;
        const int r = div_small(i, PW, rcp_pw), ic = i - r * PW - p.p1;
        p_r[u] = i < n_patch ? r : -(1 << 20);                       // out-of-patch slots never pass the row test
        p_ic[u] = (ic >= 0 && ic < p.W0) ? ic : -1;
    }
    float pf[PU];                                                    // the NEXT tile's patch values, prefetched
#define AMPH_F3_FETCH_PATCH(tile_)                                                         \
    {                                                                                      \
        const int mb_ = (tile_) * BM;                                                      \
        const int b_ = div_small(mb_, ohw, rcp_ohw), ohf_ = div_small(mb_ - b_ * ohw, p.OW, rcp_ow); \
        const int plo_ = (ohf_ * p.stride - p.pad) * p.s1 - p.p1;                          \
        const float* img_ = p.img + (long long)b_ * p.H0 * p.W0;                           \
        _Pragma("unroll") for (int u = 0; u < PU; ++u) {                                   \
            const int ir_ = plo_ + p_r[u];                                                 \
            pf[u] = *((ir_ >= 0 && ir_ < p.H0 && p_ic[u] >= 0) ? img_ + ir_ * p.W0 + p_ic[u] : p.zeros); \
        }                                                                                  \
    }
    // every load of the prologue is in flight before anything waits: the first tile's patch here, the B slices below
    if ((int)blockIdx.x < n_tiles) AMPH_F3_FETCH_PATCH(blockIdx.x)
    // per-thread A rows r0 and r0 + 32 inside a tile (whole output rows of one chunk, so tile-independent): local output
    // row / column, window origin in the patch for tap (0, 0) and the horizontally valid taps
    const int ohl0 = div_small(r0, p.OW, rcp_ow), owl0 = r0 - ohl0 * p.OW, ohl1 = div_small(r0 + 32, p.OW, rcp_ow), owl1 = (r0 + 32) - ohl1 * p.OW;
    const int rowbase0 = (ohl0 * p.stride * p.s1) * PW + (owl0 * p.stride - p.pad) * p.s1;
    const int rowbase1 = (ohl1 * p.stride * p.s1) * PW + (owl1 * p.stride - p.pad) * p.s1;
    unsigned hm0, hm1;                                   // k == 3 (host check): bit kh * 3 + kw
    {
        const int iwA = owl0 * p.stride - p.pad, iwB = owl1 * p.stride - p.pad;
        hm0 = ((unsigned)iwA < (unsigned)p.W ? 0x049u : 0u) | ((unsigned)(iwA + 1) < (unsigned)p.W ? 0x092u : 0u) |
              ((unsigned)(iwA + 2) < (unsigned)p.W ? 0x124u : 0u);
        hm1 = ((unsigned)iwB < (unsigned)p.W ? 0x049u : 0u) | ((unsigned)(iwB + 1) < (unsigned)p.W ? 0x092u : 0u) |
              ((unsigned)(iwB + 2) < (unsigned)p.W ? 0x124u : 0u);
    }
    // B (weight) slices do not depend on the tile: the load pointer runs CYCLICALLY over the K slices, so the last three
    // steps of a tile already fetch slices 0, 1, 2 for the next one (n_slices % 3 == 0: the stage rotation lines up) and
    // only the first tile's prologue loads them.  BN constants of the epilogue: tile-independent too, loaded once.
    const int n_slices = p.K_pad / kBK;                  // == k * k taps (C_in == 32), splits == 1
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, -1, 0x00020000);
    const unsigned woff0 = ((unsigned)(n_blk + r0) * (unsigned)p.K_pad + 4u * c4) * 4u, woff1 = woff0 + 32u * (unsigned)p.K_pad * 4u;
    int w_soff = 0;                                      // byte offset of the next B slice (wave-uniform; buffer loads as in p3)
    int l_left = n_slices;                               // slices until the offset wraps
    float4 sb0a, sb0b, sb1a, sb1b, sb2a, sb2b;           // B register stages
    float4 e_sc0, e_sc1, e_sc2, e_sc3, e_sh0, e_sh1, e_sh2, e_sh3;
    {
        // the select is on the ADDRESS (columns past N read the zero line): a select on the loaded value makes the
        // prologue wait for these loads - a full cold global round trip - before the patch fetch is even issued
        const int nb = n_blk + (wave & 1) * 32 + 4 * (lane >> 5);
        e_sc0 = *reinterpret_cast<const float4*>(nb + 0 < p.N ? p.scale + nb + 0 : p.zeros);
        e_sc1 = *reinterpret_cast<const float4*>(nb + 8 < p.N ? p.scale + nb + 8 : p.zeros);
        e_sc2 = *reinterpret_cast<const float4*>(nb + 16 < p.N ? p.scale + nb + 16 : p.zeros);
        e_sc3 = *reinterpret_cast<const float4*>(nb + 24 < p.N ? p.scale + nb + 24 : p.zeros);
        e_sh0 = *reinterpret_cast<const float4*>(nb + 0 < p.N ? p.shift + nb + 0 : p.zeros);
        e_sh1 = *reinterpret_cast<const float4*>(nb + 8 < p.N ? p.shift + nb + 8 : p.zeros);
        e_sh2 = *reinterpret_cast<const float4*>(nb + 16 < p.N ? p.shift + nb + 16 : p.zeros);
        e_sh3 = *reinterpret_cast<const float4*>(nb + 24 < p.N ? p.shift + nb + 24 : p.zeros);
    }
#ifdef AMPH_STAMPS
    unsigned long long pq0 = 0, pq1 = 0, pq2 = 0, pq3 = 0;
    AMPH_GSTAMP(pq0)
#endif
    // ---- compute-side state (the slice whose A stage is being recomputed): its tile's valid-tap masks and patch buffer.
    //      The slice stream is CONTINUOUS across this workgroup's tiles: slices 9 and 10 of a tile are slices 0 and 1 of
    //      its next tile (the B pointer is cyclic, the LDS ring and the stage rotation line up because 9 % 3 == 0), so
    //      steps 7 and 8 - which used to recompute all-zero stages - prepare the next tile from the SECOND patch buffer
    //      and only the first tile pays a prologue.
    float* patch_a = patch;
    float* patch_b = patch + ((n_patch + 3) & ~3);
    float* patch_c = patch_a;                            // patch of the tile being COMPUTED
    unsigned vmc0 = 0, vmc1 = 0;
#define AMPH_F3_MASKS(t_)                                                                  \
    {                                                                                      \
        const int mb_ = (t_) * BM;                                                         \
        const int b_ = div_small(mb_, ohw, rcp_ohw), ohf_ = div_small(mb_ - b_ * ohw, p.OW, rcp_ow); \
        const int ihA = (ohf_ + ohl0) * p.stride - p.pad, ihB = (ohf_ + ohl1) * p.stride - p.pad; \
        const unsigned rbA = ((unsigned)(ihA) < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ihA + 1) < (unsigned)p.H ? 0x038u : 0u) | \
                             ((unsigned)(ihA + 2) < (unsigned)p.H ? 0x1c0u : 0u);         /* k == 3 (host check) */ \
        const unsigned rbB = ((unsigned)(ihB) < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ihB + 1) < (unsigned)p.H ? 0x038u : 0u) | \
                             ((unsigned)(ihB + 2) < (unsigned)p.H ? 0x1c0u : 0u);          \
        vmc0 = mb_ + r0 < p.M ? hm0 & rbA : 0u;                                            \
        vmc1 = mb_ + r0 + 32 < p.M ? hm1 & rbB : 0u;                                       \
    }
#define AMPH_F3_WRITE_PATCH(dst_)                                                          \
    {                                                                                      \
        _Pragma("unroll") for (int u = 0; u < PU; ++u)                                     \
            if (tid + 256 * u < n_patch) (dst_)[tid + 256 * u] = pf[u];                    \
    }
    int c_kpos = 0, c_kh = 0, c_kw = 0;                  // tap of the next slice to COMPUTE (wave-uniform)
    f32x2 a0l, a0h, a1l, a1h;                            // the A stage being computed (rows r0, r0 + 32; channel pairs)
    float pv0_0, pv0_1, pv0_2, pv0_3, pv0_4, pv0_5, pv0_6, pv0_7, pv0_8;
    float pv1_0, pv1_1, pv1_2, pv1_3, pv1_4, pv1_5, pv1_6, pv1_7, pv1_8;
    float cap0 = 0.0f, cap1 = 0.0f;                      // +inf for a tap inside the image, 0 for a padded one
#define AMPH_F3_LOADB(q)                                                                   \
    {                                                                                      \
        sb##q##a = AMPH_P3_BUFLOAD(rs_w, woff0, w_soff);                                   \
        sb##q##b = AMPH_P3_BUFLOAD(rs_w, woff1, w_soff);                                   \
        --l_left;                                                                          \
        w_soff = __builtin_amdgcn_readfirstlane(l_left == 0 ? 0 : w_soff + kBK * 4);   /* wrap to slice 0 */ \
        l_left = l_left == 0 ? n_slices : l_left;                                          \
    }
    // window of row i for the tap being computed (taps past the end have no mask bit: all zero)
#define AMPH_F3_WINDOW(i)                                                                  \
    {                                                                                      \
        const bool ok_ = (vmc##i >> c_kpos) & 1u;                                          \
        cap##i = ok_ ? __builtin_inff() : 0.0f;                                            \
        const float* pp_ = patch_c + (ok_ ? rowbase##i + (c_kh * p.s1 * PW + c_kw * p.s1) : 0); \
        pv##i##_0 = pp_[0]; pv##i##_1 = pp_[1]; pv##i##_2 = pp_[2];                        \
        pv##i##_3 = pp_[PW]; pv##i##_4 = pp_[PW + 1]; pv##i##_5 = pp_[PW + 2];             \
        pv##i##_6 = pp_[2 * PW]; pv##i##_7 = pp_[2 * PW + 1]; pv##i##_8 = pp_[2 * PW + 2]; \
    }
#define AMPH_F3_TAP(i, t)                                                                  \
    a##i##l = __builtin_elementwise_fma(f32x2{pv##i##_##t, pv##i##_##t}, wl_##t, a##i##l); \
    a##i##h = __builtin_elementwise_fma(f32x2{pv##i##_##t, pv##i##_##t}, wh_##t, a##i##h);
#define AMPH_F3_ZERO(i) a##i##l = f32x2{0.f, 0.f}; a##i##h = f32x2{0.f, 0.f};
    // BN, then ReLU and the padding mask in one v_med3_f32: med3(x, 0, +inf) = max(x, 0), med3(x, 0, 0) = 0
#define AMPH_F3_FIN(i)                                                                     \
    a##i##l = __builtin_elementwise_fma(a##i##l, sc1l, sh1l);                              \
    a##i##h = __builtin_elementwise_fma(a##i##h, sc1h, sh1h);                              \
    a##i##l.x = __builtin_amdgcn_fmed3f(a##i##l.x, 0.0f, cap##i); a##i##l.y = __builtin_amdgcn_fmed3f(a##i##l.y, 0.0f, cap##i); \
    a##i##h.x = __builtin_amdgcn_fmed3f(a##i##h.x, 0.0f, cap##i); a##i##h.y = __builtin_amdgcn_fmed3f(a##i##h.y, 0.0f, cap##i);
#define AMPH_F3_NEXT_TAP()                                                                 \
    {                                                                                      \
        ++c_kpos; ++c_kw;                                                                  \
        const int wrap_ = c_kw >= p.k ? 1 : 0;                                             \
        c_kw -= wrap_ * p.k;                                                               \
        c_kh += wrap_;                                                                     \
    }
    const int st_off = r0 * kLdsRow + 4 * c4;
#define AMPH_F3_STORE(q, buf_)                                                             \
    {                                                                                      \
        float* a_ = As + (buf_) * kABuf + st_off;                                          \
        float* b_ = Bs + (buf_) * kBBuf + st_off;                                          \
        *reinterpret_cast<float4*>(a_) = make_float4(a0l.x, a0l.y, a0h.x, a0h.y);          \
        *reinterpret_cast<float4*>(a_ + 32 * kLdsRow) = make_float4(a1l.x, a1l.y, a1h.x, a1h.y); \
        *reinterpret_cast<float4*>(b_) = sb##q##a;                                         \
        *reinterpret_cast<float4*>(b_ + 32 * kLdsRow) = sb##q##b;                          \
    }
#define AMPH_F3_COMPUTE_ALL()                                                              \
    {                                                                                      \
        AMPH_F3_WINDOW(0) AMPH_F3_WINDOW(1)                                                \
        AMPH_F3_ZERO(0) AMPH_F3_ZERO(1)                                                    \
        AMPH_F3_TAP(0, 0) AMPH_F3_TAP(0, 1) AMPH_F3_TAP(0, 2) AMPH_F3_TAP(0, 3) AMPH_F3_TAP(0, 4) \
        AMPH_F3_TAP(0, 5) AMPH_F3_TAP(0, 6) AMPH_F3_TAP(0, 7) AMPH_F3_TAP(0, 8) AMPH_F3_FIN(0) \
        AMPH_F3_TAP(1, 0) AMPH_F3_TAP(1, 1) AMPH_F3_TAP(1, 2) AMPH_F3_TAP(1, 3) AMPH_F3_TAP(1, 4) \
        AMPH_F3_TAP(1, 5) AMPH_F3_TAP(1, 6) AMPH_F3_TAP(1, 7) AMPH_F3_TAP(1, 8) AMPH_F3_FIN(1) \
        AMPH_F3_NEXT_TAP()                                                                 \
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int row_tile = wave >> 1, col_tile = wave & 1;
    const float* a_rd = As + (row_tile * 32 + (lane & 31)) * kLdsRow + 4 * (lane >> 5);
    const float* b_rd = Bs + (col_tile * 32 + (lane & 31)) * kLdsRow + 4 * (lane >> 5);

    // one slice from LDS buffer R: B stage QL receives slice s+3; slice s+2 (A computed here, B stage QS) goes
    // to buffer (R+2)%3
#define AMPH_F3_STEP(R, QL, QS)                                                            \
    {                                                                                      \
        float4 f1a, f1b, f2a, f2b, f3a, f3b, n0a, n0b;                                     \
        AMPH_P3_T(ts0)                                                                     \
        AMPH_P3_READ(f1a, f1b, R, 1)                                                       \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f0a.x, f0b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_F3_LOADB(QL)                                                                  \
        AMPH_F3_WINDOW(0)                                                                  \
        AMPH_F3_WINDOW(1)                                                                  \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f0a.y, f0b.y)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_READ(f2a, f2b, R, 2)                                                       \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f0a.z, f0b.z)                                                         \
        AMPH_P3_MFMA(f0a.w, f0b.w)                                                         \
        AMPH_P3_MFMA(f1a.x, f1b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_READ(f3a, f3b, R, 3)                                                       \
        AMPH_F3_ZERO(0) AMPH_F3_ZERO(1)                                                    \
        AMPH_F3_TAP(0, 0) AMPH_F3_TAP(1, 0) AMPH_F3_TAP(0, 1) AMPH_F3_TAP(1, 1) AMPH_F3_TAP(0, 2) AMPH_F3_TAP(1, 2) \
        AMPH_F3_TAP(0, 3) AMPH_F3_TAP(1, 3) AMPH_F3_TAP(0, 4) AMPH_F3_TAP(1, 4) AMPH_F3_TAP(0, 5) AMPH_F3_TAP(1, 5) \
        AMPH_F3_TAP(0, 6) AMPH_F3_TAP(1, 6) AMPH_F3_TAP(0, 7) AMPH_F3_TAP(1, 7) AMPH_F3_TAP(0, 8) AMPH_F3_TAP(1, 8) \
        AMPH_F3_FIN(0) AMPH_F3_FIN(1)                                                      \
        AMPH_F3_NEXT_TAP()                                                                 \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f1a.y, f1b.y)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_T(ts1)                                                                     \
        AMPH_P3_T(ts2)                                                                     \
        AMPH_F3_STORE(QS, (R + 2) % 3)                                                     \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f1a.z, f1b.z)                                                         \
        AMPH_P3_MFMA(f1a.w, f1b.w)                                                         \
        AMPH_P3_MFMA(f2a.x, f2b.x)                                                         \
        AMPH_P3_MFMA(f2a.y, f2b.y)                                                         \
        AMPH_P3_MFMA(f2a.z, f2b.z)                                                         \
        AMPH_P3_MFMA(f2a.w, f2b.w)                                                         \
        AMPH_P3_MFMA(f3a.x, f3b.x)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_READ(n0a, n0b, (R + 1) % 3, 0)                                             \
        AMPH_SB();                                                                         \
        AMPH_P3_MFMA(f3a.y, f3b.y)                                                         \
        AMPH_P3_MFMA(f3a.z, f3b.z)                                                         \
        AMPH_P3_MFMA(f3a.w, f3b.w)                                                         \
        AMPH_SB();                                                                         \
        AMPH_P3_T(ts3)                                                                     \
        __syncthreads();                                                                   \
        AMPH_P3_T(ts4)                                                                     \
        AMPH_P3_ACC()                                                                      \
        f0a = n0a; f0b = n0b;                                                              \
    }


    // ---- first tile: patch, masks, B slices 0..2, A of slices 0 and 1, on to LDS buffers 0 and 1
    AMPH_F3_LOADB(0)                                     // issued before the patch is waited for
    AMPH_F3_LOADB(1)
    AMPH_F3_LOADB(2)
    if ((int)blockIdx.x < n_tiles) {
        AMPH_F3_WRITE_PATCH(patch_a)
        AMPH_F3_MASKS(blockIdx.x)
    }
    AMPH_GSTAMP(pq1)
    __syncthreads();                                     // the patch is staged
    AMPH_GSTAMP(pq2)
    if ((int)(blockIdx.x + gridDim.x) < n_tiles) AMPH_F3_FETCH_PATCH(blockIdx.x + gridDim.x)
    AMPH_F3_COMPUTE_ALL()
    AMPH_F3_STORE(0, 0)
    AMPH_F3_COMPUTE_ALL()
    AMPH_F3_STORE(1, 1)
    __syncthreads();
    float4 f0a, f0b;
    AMPH_P3_READ(f0a, f0b, 0, 0)
#ifdef AMPH_STAMPS
    unsigned long long tq0 = 0, tq1 = 0, q_mid = 0, q_msk = 0, q_epi = 0;
    AMPH_GSTAMP(tq0)
    q_pro = tq0 - q_begin;
    pq3 = tq0;
    q_pro = (((pq0 - q_begin) >> 1) << 48) | (((pq1 - pq0) >> 1) << 32) | (((pq2 - pq1) >> 1) << 16) | ((pq3 - pq2) >> 1);
#endif

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m_blk = tile * BM;
        const int nt = tile + (int)gridDim.x;
        const bool has_next = nt < n_tiles;
        float* patch_n = patch_c == patch_a ? patch_b : patch_a;
        AMPH_F3_STEP(0, 0, 2)                            // slices 0..2 multiplied, 2..4 recomputed
        AMPH_F3_STEP(1, 1, 0)
        AMPH_F3_STEP(2, 2, 1)
        // the next tile's patch goes to the spare buffer (last read at step 6 of the previous tile; first read at
        // step 7 of this one, several barriers from here), the patch after it into the registers
        AMPH_GSTAMP(tq0)
        if (has_next) AMPH_F3_WRITE_PATCH(patch_n)
        if (nt + (int)gridDim.x < n_tiles) AMPH_F3_FETCH_PATCH(nt + gridDim.x)
        AMPH_GSTAMP(tq1)
#ifdef AMPH_STAMPS
        q_mid += tq1 - tq0;
#endif
        AMPH_F3_STEP(0, 0, 2)
        AMPH_F3_STEP(1, 1, 0)
        AMPH_F3_STEP(2, 2, 1)
        AMPH_F3_STEP(0, 0, 2)                            // slice 6 multiplied, 8 recomputed
        // the compute side moves on to the next tile (no tile: empty masks -> all-zero stages, never multiplied)
        AMPH_GSTAMP(tq0)
        c_kpos = 0; c_kh = 0; c_kw = 0;
        if (has_next) { AMPH_F3_MASKS(nt) patch_c = patch_n; }
        else { vmc0 = 0; vmc1 = 0; }
        AMPH_GSTAMP(tq1)
#ifdef AMPH_STAMPS
        q_msk += tq1 - tq0;
#endif
        AMPH_F3_STEP(1, 1, 0)                            // slices 7, 8 multiplied; the NEXT tile's 0, 1 recomputed
        AMPH_F3_STEP(2, 2, 1)
        // epilogue on the transposed accumulator: lane = row m, register group g = channels n0 .. n0 + 3.  (A version that
        // transposed each wave's tile through LDS so that every store instruction wrote 8 whole 128-byte lines instead of 32
        // lines x 32 bytes was SLOWER, 2.65 k -> 4.1 k cycles per tile: the cost is the burst itself - all 512 workgroups
        // store their 16 KB at about the same time - not the shape of the store instructions.)
        AMPH_GSTAMP(tq0)
        {
            const int m = m_blk + row_tile * 32 + (lane & 31);
            if (m < p.M) {
#define AMPH_F3_OUT(g)                                                                     \
                {                                                                          \
                    const int n0 = n_blk + col_tile * 32 + 8 * g + 4 * (lane >> 5);        \
                    if (n0 < p.N) {                                                        \
                        float4 v;                                                          \
                        v.x = fmaxf(fmaf(acc[4 * g], e_sc##g.x, e_sh##g.x), 0.0f);         \
                        v.y = fmaxf(fmaf(acc[4 * g + 1], e_sc##g.y, e_sh##g.y), 0.0f);     \
                        v.z = fmaxf(fmaf(acc[4 * g + 2], e_sc##g.z, e_sh##g.z), 0.0f);     \
                        v.w = fmaxf(fmaf(acc[4 * g + 3], e_sc##g.w, e_sh##g.w), 0.0f);     \
                        *reinterpret_cast<float4*>(p.y + (long long)m * p.N + n0) = v;     \
                    }                                                                      \
                }
                AMPH_F3_OUT(0) AMPH_F3_OUT(1) AMPH_F3_OUT(2) AMPH_F3_OUT(3)
#undef AMPH_F3_OUT
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        AMPH_GSTAMP(tq1)
#ifdef AMPH_STAMPS
        q_epi += tq1 - tq0;
#endif
    }
#ifdef AMPH_STAMPS
    if (tid == 0 && p.stamps) {
        unsigned long long q_end;
        AMPH_GSTAMP(q_end)
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = q_p1; o[1] = q_pro; o[2] = q_p2; o[3] = q_bar; o[4] = q_end - q_begin;   // sums over this workgroup's tiles
        o[5] = __builtin_amdgcn_s_memrealtime() - q_rt0;
        o[6] = (q_mid << 40) | (q_msk << 20) | q_epi;                                  // three 20-bit fields
        o[7] = q_rt0 & 0xffffffffull;
    }
#endif
}

static size_t conv_gemm_f3_lds(const ConvGemmParams& p, int bm) {
    const int n_oh = bm / p.OW;
    const int rows = ((n_oh - 1) * p.stride + p.k - 1) * p.s1 + p.k1;
    const size_t patch = ((size_t)rows * (p.W0 + 2 * p.p1) + 3) & ~(size_t)3;          // two buffers (current / next tile)
    return ((size_t)3 * (bm + kBN) * kLdsRow + 2 * patch) * sizeof(float);
}
bool conv_gemm_f3_ok(const ConvGemmParams& p, int bm) {
    const int ohw = p.OH * p.OW;
    return bm == 64 && p.img != nullptr && p.k1 == 3 && p.C_in == kBK && p.K == p.K_pad && p.k == 3 && p.splits == 1 && (p.N % 4) == 0 &&
           p.M < (1 << 22) && p.zeros != nullptr && (unsigned long long)p.N * p.K_pad * 4ull < (1ull << 32) &&
           p.epilogue == 0 && ohw % bm == 0 && bm % p.OW == 0 && conv_gemm_f3_lds(p, bm) <= 80 * 1024 && (p.K_pad / kBK) % 3 == 0 &&
           (((bm / p.OW - 1) * p.stride + p.k - 1) * p.s1 + 3) * (p.W0 + 2 * p.p1) <= 256 * 8;
}

// Zero bytes k_conv_gemm_p3 needs in front of its input: a padded tap's address is (x - guard) + tap offset, and a valid
// tap's lane offset guard + window origin must not be negative (origin >= -(pad * W + pad) * C_in floats, pad < k).
size_t conv_gemm_x_guard(int k, int W, int C_in) {
    return ((size_t)k * (W + 1) * C_in * sizeof(float) + 64 + 255) & ~(size_t)255;
}

bool conv_gemm_p3_ok(const ConvGemmParams& p, int bm) {
    const unsigned long long span = (unsigned long long)p.M / (p.OH * p.OW) * p.H * p.W * p.C_in * 4ull + p.x_guard;     // 32-bit lane offsets
    return bm == 64 && (p.C_in % kBK) == 0 && p.K == p.K_pad && p.k * p.k <= 32 && (p.N % 4) == 0 && p.pad < p.k &&
           (size_t)p.x_guard >= conv_gemm_x_guard(p.k, p.W, p.C_in) && span < (1ull << 32) &&
           (unsigned long long)p.N * p.K_pad * 4ull < (1ull << 32) && p.M < (1 << 22);
}

// first-generation fused tile: the 64-row tile needs 43 KB (three workgroups per CU: 768 of a 256-chunk batch's 1024
// workgroups are resident, a second, one-third-full round follows); the 128-row tile (66 KB, one round of 512) was
// measured no faster (48.4 vs 49.1 us) and stays excluded
constexpr size_t kFuseLdsMax = 64 * 1024;

static size_t conv_gemm_lds(const ConvGemmParams& p, int bm, bool fuse) {
    size_t b = (size_t)2 * (bm + kBN) * kLdsRow * sizeof(float);
    if (fuse) {
        const int n_oh = bm / p.OW;
        const int rows = ((n_oh - 1) * p.stride + p.k - 1) * p.s1 + p.k1;
        b += ((size_t)11 * p.C_in + (size_t)rows * (p.W0 + 2 * p.p1)) * sizeof(float);
    }
    return b;
}

bool conv_gemm_can_fuse(const ConvGemmParams& p, int bm) {
    const int ohw = p.OH * p.OW;
    return p.img != nullptr && p.k1 == 3 && p.k * p.k <= 32 && ohw % bm == 0 && bm % p.OW == 0 && (p.C_in % 4) == 0 &&
           conv_gemm_lds(p, bm, true) <= kFuseLdsMax;
}

int g_num_cus = 256;      // compute units of the device (amph_create sets it)

void launch_conv_gemm(const ConvGemmParams& p_in, int bm, bool fuse, hipStream_t s) {
    ConvGemmParams p = p_in;
    p.flags = 0;
    const dim3 grid((unsigned)((p.M + bm - 1) / bm), (unsigned)((p.N + kBN - 1) / kBN), (unsigned)p.splits);
    const size_t lds = conv_gemm_lds(p, bm, fuse);
    if (!fuse && conv_gemm_p3_ok(p, bm)) {
        const size_t lds3 = (size_t)3 * (bm + kBN) * kLdsRow * sizeof(float);
        hipLaunchKernelGGL((k_conv_gemm_p3<64, false>), grid, dim3(256), lds3, s, p);
        return;
    }
    if (fuse && conv_gemm_f3_ok(p, bm)) {
        static bool raised3[64] = {false};
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !raised3[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_f3<64>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
            raised3[dev] = true;
        }
        const dim3 pgrid(std::min(grid.x, (unsigned)(2 * g_num_cus)), grid.y, 1);      // persistent: one resident round
        hipLaunchKernelGGL((k_conv_gemm_f3<64>), pgrid, dim3(256), conv_gemm_f3_lds(p, bm), s, p);
        return;
    }
    if (fuse) {
        static bool raised[64] = {false};          // per device: dynamic LDS above the 64 KB default
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (dev >= 0 && dev < 64 && !raised[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm<128, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFuseLdsMax);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm<64, true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kFuseLdsMax);
            raised[dev] = true;
        }
        if (bm == 128) hipLaunchKernelGGL((k_conv_gemm<128, true>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((k_conv_gemm<64, true>), grid, dim3(256), lds, s, p);
    } else {
        if (bm == 128) hipLaunchKernelGGL((k_conv_gemm<128, false>), grid, dim3(256), lds, s, p);
        else hipLaunchKernelGGL((k_conv_gemm<64, false>), grid, dim3(256), lds, s, p);
    }
}

// ----------------------------------------------------------------------------
// Head: mu / logvar = sum over split-K partials (fixed order) + bias.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_head_reduce(HeadReduceParams p, int n_rows) {
    const int g = blockIdx.x * 256 + threadIdx.x;
    const int N = 2 * p.D;
    if (g >= n_rows * N) return;
    const int b = g / N, n = g - b * N;
    // all partials requested at once, then added in split order (the loop form waited for one global round trip per split)
    float acc = 0.0f;
    const float* src = p.partial + (long long)b * N + n;
    const long long step = (long long)n_rows * N;
    for (int s0 = 0; s0 < p.splits; s0 += 16) {
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = s0 + u < p.splits ? src[(long long)(s0 + u) * step] : 0.0f;
#pragma unroll
        for (int u = 0; u < 16; ++u) if (s0 + u < p.splits) acc += v[u];
    }
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

// ----------------------------------------------------------------------------
// K7: Gaussian-MAP scoring + decision (10b_benchmark_folder_detection_map.py:143-169 with
// gaussian_logpdf_from_precision, map_detector_core.py:319-323):
//   diff = z - mu_k (float32);  quad = diff^T P_k diff (float32; the reference's is numpy / BLAS
//   sgemv + sdot, whose summation order is the library's own);  in float64:
//   score_k = -0.5 * ((quad + logdet_k) + D log 2pi) + log(prior_k + 1e-12)
//   pred = first k with the largest score (strict >), NO_DETECT (-1) if best < tau.
// A group of Dp = pow2 >= D threads owns one chunk: thread j forms y_j = sum_i diff_i P[i][j] (fma
// chain, i ascending), the products y_j diff_j are summed by a fixed LDS tree: deterministic and
// independent of the batch composition.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_score_map(MapParams p, long long n_rows, int Dp) {
#pragma clang fp contract(off)
    __shared__ float sh_diff[256];
    __shared__ float sh_part[256];
    const int tid = threadIdx.x;
    const int cpb = 256 / Dp;                               // chunks per block
    const int cl = tid / Dp, j = tid - cl * Dp;
    const long long row = (long long)blockIdx.x * cpb + cl;
    const bool rv = row < n_rows, jv = j < p.D;
    const float zj = (rv && jv) ? p.z[row * p.D + j] : 0.0f;
    double best = -INFINITY;
    int pred = -1;
    for (int k = 0; k < p.K; ++k) {
        const float dj = jv ? zj - p.means[(long long)k * p.D + j] : 0.0f;
        __syncthreads();                                    // previous species' tree is done with sh_part / sh_diff
        sh_diff[tid] = dj;
        __syncthreads();
        float y = 0.0f;
        if (jv) {
            const float* P = p.prec + (long long)k * p.D * p.D + j;
            const float* df = sh_diff + cl * Dp;
            for (int i = 0; i < p.D; ++i) y = __builtin_fmaf(df[i], P[(long long)i * p.D], y);
        }
        sh_part[tid] = y * dj;
        __syncthreads();
        for (int o = Dp >> 1; o > 0; o >>= 1) {
            if (j < o) sh_part[tid] = sh_part[tid] + sh_part[tid + o];
            __syncthreads();
        }
        if (j == 0 && rv) {
            const double quad = (double)sh_part[tid];
            const double sc = -0.5 * ((quad + p.logdet[k]) + p.d_log2pi) + p.log_prior[k];
            if (p.scores) p.scores[row * p.K + k] = sc;
            if (sc > best) { best = sc; pred = k; }
        }
    }
    if (j == 0 && rv) {
        if (pred >= 0 && p.has_tau && best < p.tau) pred = -1;
        p.pred[row] = pred;
        p.best_score[row] = best;
    }
}

void launch_score_map(const MapParams& p, long long n_rows, hipStream_t s) {
    int Dp = 8;
    while (Dp < p.D) Dp <<= 1;
    const int cpb = 256 / Dp;
    hipLaunchKernelGGL(k_score_map, dim3((unsigned)((n_rows + cpb - 1) / cpb)), dim3(256), 0, s, p, n_rows, Dp);
}

}  // namespace amph
