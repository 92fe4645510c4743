// Fused conv1 -> conv2 with the layer-1 activation of a tile RESIDENT in LDS ("f4s"), wave-specialised.
//
// Replaces, like k_conv_gemm_f3 / _f3s, layers 1 and 2 of `encoder(x)` at
// latent_space_exploration/map_detector_core.py:270 (module built at :175-179): C_in == 1 3x3 stride-2 conv + BN + ReLU, then
// 3x3 stride-2 conv (32 -> N channels) + BN + ReLU.  Same arithmetic as k_conv_gemm_f3s - layer 1 in fp32 with
// k_conv_first's FMA order, three-term bf16 split of both operands, six v_mfma_f32_32x32x16_bf16 per K = 16 in the same
// order - except that the K = 288 sum is formed as (k-steps 0..8) + (k-steps 9..17) by two waves instead of in one chain.
//
// Why another form: f3 / f3s recompute layer 1 for every (tap, output pixel) pair a K slice needs - 576 pixel evaluations
// per 64-pixel tile for 297 distinct pixels -, push each value through LDS staging once per tap and re-stage the weight
// slice for every tile; in-kernel stamps show the three resources of a step (VALU recompute, LDS staging, MFMA) used one
// after the other (profiles/r02_f3s_stamps.txt).  Here a 512-thread workgroup, one per CU, has two roles:
//   producer waves 4..7  compute the tile's layer-1 patch ONCE - (2 n_oh + 1) rows x (2 OW + 1) columns x 32 channels -
//                        from image rows staged in LDS (as three bf16 term planes), on the matrix cores too: per group of 32 patch
//                        pixels a K = 16 product (9 taps + zero weights) with both operands split, six MFMAs; BN + ReLU + split of
//                        the result go to LDS as [column parity][row][column / 2][term][32 ch] bf16 (208-byte pixels: odd
//                        multiple of 16, conflict-free for the gather below);
//   consumer waves 0..3  = (channel tile ct, K half th).  A wave keeps ITS HALF OF THE WEIGHTS IN REGISTERS for the whole
//                        kernel (9 k-steps x 3 terms x 16 B per lane = 108 VGPRs) and reads the MFMA activation fragments
//                        STRAIGHT from the patch: for output pixel (oh, ow) and tap (kh, kw) the layer-1 pixel is patch row
//                        2 oh + kh, parity kw & 1, index ow + (kw >> 1): a per-lane base plus a wave-uniform constant per
//                        k-step - no im2col staging, no weight traffic at all in the tile loop.  It multiplies all 64 pixels
//                        of the tile (two accumulators), hands the partial sum of the row tile it does not own to its partner
//                        wave through LDS and writes the other one out.
// Two patch buffers: the producers build tile i + 1 while the consumers multiply tile i.  Barriers per tile: X (patch i + 1
// complete, patch i free, partial sums published) and Y (partial sums consumed).
//
// Where the time goes (in-kernel stamps, profiles/r02_f4s_stamps_*.txt; issue-cost probe profiles/r02_probe_mfma_bf16_valu.txt),
// cycles per 64-pixel tile: the consumers multiply in 3.95 k (3.46 k = MFMA bound) and then wait for the producers, who build in
// 5.0 k: window reads 0.6 k, the eighteen layer-1 MFMAs 0.64 k (behind the consumer wave's MFMA stream on the same SIMD), BN + split
// + stores 2.7 k, image fetch + staging 1.4 k.  A wave beside an MFMA-streaming wave gets what that wave leaves of the SIMD's vector
// issue (a gap's budget is 32 cycles: MFMA 8, independent VALU 4, dependent VALU / v_cvt_pk / v_accvgpr 8, an LDS access 8..17), and a
// lone wave pays ~8 cycles per DEPENDENT instruction: hence all window reads first, three interleaved MFMA chains, and the splits
// done stage by stage over eight pairs (7.0 k -> 5.0 k; layer 1 by scalar FMAs instead of MFMAs built in the same 7.0 k before that).
// Raising the producers' s_setprio moves the wait from one role to the other, same period.  Every wave in both roles in ONE
// instruction stream (4 waves, 512 registers, the patch work placed in the MFMA gaps with sched_group_barrier; tried, removed): 6.7 k
// cycles per tile - its ~650 VALU and ~120 LDS instructions per tile sum to more issue cycles than the 126 MFMAs leave.
// Start-up 11 k cycles (weights 1.7 k, first image staged 5 k, first patch 11 k): eight tiles per workgroup.  21 us per 256 chunks with one
// workgroup per CU against 30.7 for k_conv_gemm_f3s; the default (128 workgroups, 31 us alone) leaves half the chip to the other lanes.
#include "amph_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstring>

namespace amph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 amph_h2v __attribute__((ext_vector_type(2)));
// operand fragments of the register-resident-weight kernels: bf16 (TERMS 1: the bf16 encoder) or fp16 (TERMS 2: fp32 layers as the two-term
// split x = h + 2^-11 l of k_encoder_split.hip - three MFMAs per K = 16, the cross terms in a second accumulator)
template <int TERMS> struct G4Frag { typedef f16x8 type; };
template <> struct G4Frag<1> { typedef bf16x8 type; };
__device__ __forceinline__ f32x16 g4_mfma(const bf16x8& w, const bf16x8& x, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, x, c, 0, 0, 0); }
__device__ __forceinline__ f32x16 g4_mfma(const f16x8& w, const f16x8& x, const f32x16& c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(w, x, c, 0, 0, 0); }
constexpr float kG4Lo = 1.0f / 2048.0f;
// (h2_lo_terms / h2_split2 of k_encoder_split.hip: l = fp16((x - h) 2^11) as one v_fma_mix per value, exact before its one rounding)
__device__ __forceinline__ uint32_t g4_lo_terms(uint32_t h, float t0, float t1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "s"(-2048.0f), "v"(t0), "v"(t1));
    return l;
}
__device__ __forceinline__ void g4_split2(float x0, float x1, uint32_t& h, uint32_t& l) {
    const amph_h2v hp = {(_Float16)x0, (_Float16)x1};
    h = __builtin_bit_cast(uint32_t, hp);
    l = g4_lo_terms(h, x0 * 2048.0f, x1 * 2048.0f);
}

constexpr int kPPix = 144;               // bytes per patch pixel: [h | l] x 32 fp16 (the two-term split of k_encoder_split.hip) + 16 B pad (9 slots: odd)
constexpr int kPImgU = 6;                // staged image elements per producer lane (host check)
constexpr int kPGroupU = 3;              // 32-pixel groups of a tile's patch per producer wave (host check)

struct PatchGeo {                        // derived on the host, passed by value
    int n_oh;                            // output rows per 64-pixel tile (64 / OW)
    int PR, PC;                          // patch rows 2 n_oh + 1, columns 2 OW + 1
    int RS;                              // bytes per plane row: OW + 1 pixels, rounded up to 128
    int plane;                           // bytes per parity plane: PR * RS + 64 (pixels xi, xi + 1 of one store on different banks)
    int img_row;                         // floats per staged image row: 2 PC + 1 rounded up to even (element c = image column c - 3)
    int n_img;                           // staged image elements of a tile: (2 PR + 1) rows
    int n_pix;                           // patch pixels of a tile: PR * PC
    int tiles_per_image, n_tiles;
    int rt1_off;                         // patch byte offset of output pixel m + 32 relative to pixel m
};

__device__ __forceinline__ int f4_div_small(int n, int d, float rcp) {
    int q = (int)((float)n * rcp);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

#ifdef AMPH_STAMPS
#define AMPH_F4_T(v_)                                                                      \
    {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
#else
#define AMPH_F4_T(v_)
#endif

__global__ __launch_bounds__(512, 2) void k_conv_gemm_f4s(ConvGemmParams p, PatchGeo g, const uint4* __restrict__ wfrag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch0 = smem;                                                       // [2 parity][PR][RS]
    char* const patch1 = smem + 2 * g.plane;
    uint16_t* const img_stage = reinterpret_cast<uint16_t*>(smem + 4 * g.plane);     // [2][2 terms][2 PR + 1][img_row] fp16
    const int img_pl = (g.n_img + 8) & ~7;                                           // elements per term plane
    char* const part = smem + 4 * g.plane + 2 * 2 * img_pl * 2;                      // [2 tile parities][4 consumer waves][4][64] float4
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);                       // the role branch below is scalar
    const int n_blk = blockIdx.y * 64;
    const float rcp_tpi = 1.0f / (float)g.tiles_per_image;
#ifdef AMPH_STAMPS
    unsigned long long q_t0 = 0, q_a = 0, q_b = 0, q_c = 0, q_d = 0, q_s0 = 0, q_s1 = 0, q_s2 = 0, q_s3 = 0, q_p0 = 0, q_p1 = 0, q_p2 = 0;
    const unsigned long long q_rt0 = __builtin_amdgcn_s_memrealtime();
    AMPH_F4_T(q_t0)
#endif

    if (wave >= 4) {
        // ================================================================ producers
        const int ptid = tid - 256, pw = wave - 4;
        const int px = lane & 31, kh2 = lane >> 5;                 // pixel within a group of 32, K half of the MFMA operands
        // Layer 1 as a K = 16 product on the matrix cores: 32 channels x 32 patch pixels per group, both operands split in three
        // bf16 terms (six MFMAs, as in layer 2).  k = 8 kh2 + j pairs with the dwords a lane reads from the staged image: lanes < 32
        // read columns (c0, c1) of the pixel's three window rows, lanes >= 32 (c2, c3): k 0..5 = taps (r, 0), (r, 1), k 8 / 10 / 12
        // = taps (r, 2); the slots that see c3 or nothing carry zero weights.
        f16x8 w1h, w1l;
        {
            uint32_t hh[4], ll[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int tap0 = d < 3 ? (kh2 ? 3 * d + 2 : 3 * d) : -1, tap1 = (d < 3 && !kh2) ? 3 * d + 1 : -1;
                const float v0 = tap0 >= 0 ? p.w1[tap0 * p.C_in + px] : 0.0f, v1 = tap1 >= 0 ? p.w1[tap1 * p.C_in + px] : 0.0f;
                g4_split2(v0, v1, hh[d], ll[d]);
            }
            w1h = __builtin_bit_cast(f16x8, make_uint4(hh[0], hh[1], hh[2], hh[3]));
            w1l = __builtin_bit_cast(f16x8, make_uint4(ll[0], ll[1], ll[2], ll[3]));
        }
        float4 sc1[4], sh1[4];                                     // BN of this lane's accumulator channels 8 q + 4 kh2 + r
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc1[q] = *reinterpret_cast<const float4*>(p.scale1 + 8 * q + 4 * kh2);
            sh1[q] = *reinterpret_cast<const float4*>(p.shift1 + 8 * q + 4 * kh2);
        }
        // (the constants above are needed from the first group on: wait for them HERE and re-define the registers, or the
        // compiler's wait-count bookkeeping carries their loads into the tile loop and waits for the image prefetch in every group)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(sc1[q].x), "+v"(sc1[q].y), "+v"(sc1[q].z), "+v"(sc1[q].w));
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(sh1[q].x), "+v"(sh1[q].y), "+v"(sh1[q].z), "+v"(sh1[q].w));
        }
        // this wave's pixel groups pw, pw + 4, pw + 8 of ppg <= 32 patch pixels each (the same for every tile; 12 groups cover the
        // patch): byte offset of the lane's window in a staged term plane, patch byte offset of its pixel, and whether that pixel
        // is layer-2 padding (column -1 always; row -1 on an image's first tile)
        int it_src[kPGroupU], it_dst[kPGroupU];
        unsigned it_live = 0, it_pad_always = 0, it_pad_top = 0;
        {
            const float rcp_pc = 1.0f / (float)g.PC;
            const int ppg = (g.n_pix + 11) / 12;
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u) {
                const int pix = (pw + 4 * u) * ppg + px;
                const bool live = px < ppg && pix < g.n_pix;
                const int pc = live ? pix : 0;
                const int yl = f4_div_small(pc, g.PC, rcp_pc), xi = pc - yl * g.PC;
                it_src[u] = 2 * (2 * yl * g.img_row + 2 * xi + 2 * kh2);
                it_dst[u] = (xi & 1) * g.plane + yl * g.RS + (xi >> 1) * kPPix + 8 * kh2;
                if (live) it_live |= 1u << u;
                if (xi == 0) it_pad_always |= 1u << u;
                if (yl == 0) it_pad_top |= 1u << u;
            }
        }
        // staged row r of a tile starting at output row oh0 = image row 4 oh0 - 3 + r (conv1 row 2 oh0 - 1 + yl reads staged
        // rows 2 yl + {0, 1, 2}); staged column c = image column c - 3 (patch column xi = conv1 column xi - 1 reads 2 xi + {0, 1, 2});
        // staged as three bf16 term planes.  This lane's elements (the same for every tile): row, and row * W0 + column or -1
        int io_r[kPImgU], io_off[kPImgU], io_w[kPImgU];
        {
            const float rcp_ir = 1.0f / (float)g.img_row;
#pragma unroll
            for (int u = 0; u < kPImgU; ++u) {
                const int e = ptid + 256 * u;
                const int r = f4_div_small(e, g.img_row, rcp_ir), ic = e - r * g.img_row - 3;
                io_r[u] = r;
                io_off[u] = (e < g.n_img && ic >= 0 && ic < p.W0) ? r * p.W0 + ic : -1;
                io_w[u] = e < g.n_img ? e : g.n_img;                 // lanes past the last element store to the plane's pad element
            }
        }
        float pf[kPImgU];
#define AMPH_F4_FETCH_IMG(tile_)                                                           \
        {                                                                                  \
            const int t_ = (tile_);                                                        \
            const int b_ = f4_div_small(t_, g.tiles_per_image, rcp_tpi), tr_ = t_ - b_ * g.tiles_per_image; \
            const int ir0_ = 4 * tr_ * g.n_oh - 3;                                         \
            const float* img_ = p.img + ((long long)b_ * p.H0 + ir0_) * p.W0;              \
            _Pragma("unroll") for (int u = 0; u < kPImgU; ++u)                             \
                pf[u] = *((io_off[u] >= 0 && (unsigned)(ir0_ + io_r[u]) < (unsigned)p.H0) ? img_ + io_off[u] : p.zeros); \
        }
        // the three-term split of n pairs, stage by stage over all the pairs: every instruction's inputs are n instructions old
        // (one after the other the eleven instructions of a pair form a dependent chain at ~8 cycles apiece)
#define AMPH_F4_SPLIT_PAIRS(n_, x_, h_, l_)                                                \
        {                                                                                  \
            float r_[2 * (n_)];                                                            \
            _Pragma("unroll") for (int k = 0; k < (n_); ++k) {                             \
                const amph_h2v hp_ = {(_Float16)x_[2 * k], (_Float16)x_[2 * k + 1]};       \
                h_[k] = __builtin_bit_cast(uint32_t, hp_);                                 \
            }                                                                              \
            _Pragma("unroll") for (int k = 0; k < (n_); ++k) {                             \
                r_[2 * k] = x_[2 * k] * 2048.0f;                                           \
                r_[2 * k + 1] = x_[2 * k + 1] * 2048.0f;                                   \
            }                                                                              \
            _Pragma("unroll") for (int k = 0; k < (n_); ++k) l_[k] = g4_lo_terms(h_[k], r_[2 * k], r_[2 * k + 1]); \
        }
#define AMPH_F4_WRITE_IMG(dst_)                                                            \
        {                                                                                  \
            uint32_t h_[kPImgU / 2], l_[kPImgU / 2];                                       \
            AMPH_F4_SPLIT_PAIRS(kPImgU / 2, pf, h_, l_)                                    \
            _Pragma("unroll") for (int k = 0; k < kPImgU / 2; ++k) {                       \
                uint16_t* d0_ = (dst_) + io_w[2 * k];                                      \
                uint16_t* d1_ = (dst_) + io_w[2 * k + 1];                                  \
                d0_[0] = (uint16_t)h_[k]; d0_[img_pl] = (uint16_t)l_[k];                   \
                d1_[0] = (uint16_t)(h_[k] >> 16); d1_[img_pl] = (uint16_t)(l_[k] >> 16);   \
            }                                                                              \
        }
        int tile = blockIdx.x;
        AMPH_F4_FETCH_IMG(tile)
        AMPH_F4_WRITE_IMG(img_stage)
        __syncthreads();                                             // A: the first tile's image rows are staged
        for (int i = 0; tile < g.n_tiles; tile += gridDim.x, ++i) {
            const int nt = tile + (int)gridDim.x;
            AMPH_F4_T(q_a)
            // (round 3: only the second tile's image rows are staged here - from the third on the CONSUMERS do it, in the half of a tile
            // period they used to wait for this role, which bounds the tile since the two-term split halved their MFMA stream)
            if (i == 0) AMPH_F4_FETCH_IMG(nt < g.n_tiles ? nt : tile)  // lands under the arithmetic below
#ifdef AMPH_STAMPS
            unsigned long long q_f, q_g, q_g1, q_g2;
            AMPH_F4_T(q_f)
#endif
            const char* sbuf = reinterpret_cast<const char*>(img_stage + (i & 1) * 2 * img_pl);
            char* patch = (i & 1) ? patch1 : patch0;
            const int tr = tile - f4_div_small(tile, g.tiles_per_image, rcp_tpi) * g.tiles_per_image;
            const unsigned it_pad = it_pad_always | (tr == 0 ? it_pad_top : 0u);
            // (conv1 rows 2 oh0 - 1 + yl, yl < PR, and columns xi - 1, xi < PC, never pass the bottom / right edge: host check)
            // all the window reads of the three groups, then their MFMAs interleaved (three independent accumulation chains)
            uint32_t xw[kPGroupU][2][3];                             // [group][term][window row]
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u) {
                const char* sp = sbuf + it_src[u];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 3; ++r) xw[u][t][r] = *reinterpret_cast<const uint32_t*>(sp + 2 * (t * img_pl + r * g.img_row));
            }
#ifdef AMPH_STAMPS
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            AMPH_F4_T(q_g1)
#endif
            f32x16 acc[kPGroupU], accl[kPGroupU];                     // h h' | the cross terms (x 2^-11 below)
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) { acc[u][r] = 0.0f; accl[u][r] = 0.0f; }
#define AMPH_F4_XT(u_, t_) __builtin_bit_cast(f16x8, make_uint4(xw[u_][t_][0], xw[u_][t_][1], xw[u_][t_][2], 0u))
#define AMPH_F4_M3(a_, w_, t_)                                                             \
            a_[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_, AMPH_F4_XT(0, t_), a_[0], 0, 0, 0); \
            a_[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_, AMPH_F4_XT(1, t_), a_[1], 0, 0, 0); \
            a_[2] = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_, AMPH_F4_XT(2, t_), a_[2], 0, 0, 0);
            AMPH_F4_M3(accl, w1l, 0) AMPH_F4_M3(accl, w1h, 1) AMPH_F4_M3(acc, w1h, 0)
#undef AMPH_F4_M3
#undef AMPH_F4_XT
#ifdef AMPH_STAMPS
            asm volatile("v_mov_b32 %0, %0" : "+v"(acc[2][0]));      // waits for the last MFMA
            AMPH_F4_T(q_g2)
#endif
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u) {
                const float cap = ((it_pad >> u) & 1u) ? 0.0f : __builtin_inff();
                float a[16];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a[4 * q] = fmaf(fmaf(accl[u][4 * q], kG4Lo, acc[u][4 * q]), sc1[q].x, sh1[q].x);
                    a[4 * q + 1] = fmaf(fmaf(accl[u][4 * q + 1], kG4Lo, acc[u][4 * q + 1]), sc1[q].y, sh1[q].y);
                    a[4 * q + 2] = fmaf(fmaf(accl[u][4 * q + 2], kG4Lo, acc[u][4 * q + 2]), sc1[q].z, sh1[q].z);
                    a[4 * q + 3] = fmaf(fmaf(accl[u][4 * q + 3], kG4Lo, acc[u][4 * q + 3]), sc1[q].w, sh1[q].w);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) a[r] = __builtin_amdgcn_fmed3f(a[r], 0.0f, cap);
                uint32_t hh[8], ll[8];
                AMPH_F4_SPLIT_PAIRS(8, a, hh, ll)
                if ((it_live >> u) & 1u) {
                    char* dst = patch + it_dst[u];
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        *reinterpret_cast<uint2*>(dst + 16 * q) = make_uint2(hh[2 * q], hh[2 * q + 1]);
                        *reinterpret_cast<uint2*>(dst + 16 * q + 64) = make_uint2(ll[2 * q], ll[2 * q + 1]);
                    }
                }
            }
#ifdef AMPH_STAMPS
            AMPH_F4_T(q_g)
            q_s2 += q_f - q_a; q_s3 += q_g - q_f;
            q_p0 += q_g1 - q_f; q_p1 += q_g2 - q_g1; q_p2 += q_g - q_g2;
#endif
            if (i == 0) AMPH_F4_WRITE_IMG(img_stage + ((i + 1) & 1) * 2 * img_pl)   // that buffer's last read came before the previous X
            AMPH_F4_T(q_b)
            __syncthreads();                                         // X: patch i complete (the consumers are done with patch i - 1)
#ifdef AMPH_STAMPS
            AMPH_F4_T(q_c)
            q_s0 += q_b - q_a; q_s1 += q_c - q_b;
#endif
        }
        __syncthreads();                                             // the consumers' last X
#ifdef AMPH_STAMPS
        if (tid == 256 && p.stamps) { unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16; o[8] = q_s0; o[9] = q_s1; o[10] = q_s2; o[11] = q_s3; o[13] = q_p0; o[14] = q_p1; o[15] = q_p2; }   // producer: build | at the barriers | build: image fetch issue | build: groups | groups: window reads | MFMAs | BN + split + stores
#endif
#undef AMPH_F4_SPLIT_PAIRS
#undef AMPH_F4_FETCH_IMG
#undef AMPH_F4_WRITE_IMG
        return;
    }

    // ==================================================================== consumers
    const int ct = wave & 1, th = wave >> 1;                         // channel tile, K half (k-steps 9 th .. 9 th + 8); owner of row tile th
    uint4 breg[18];                                                  // [k-step][term]: this wave's weights, for the whole kernel
    {
        const uint4* wf = wfrag + (((size_t)blockIdx.y * 2 + ct) * 18 + 9 * th) * (2 * 64) + lane;
#pragma unroll
        for (int f = 0; f < 18; ++f) breg[f] = wf[f * 64];
    }
    int aoff[9];                                                     // patch offset of k-step 9 th + i: tap (kh, kw), channel half
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int s = 9 * th + i, tap = s >> 1, kh = tap / 3, kw = tap - 3 * kh;
        aoff[i] = (kw & 1) * g.plane + kh * g.RS + (kw >> 1) * kPPix + 32 * (s & 1);
    }
    const int m0 = lane & 31;
    const int ohl = m0 / p.OW, ow = m0 - ohl * p.OW;
    const int a_base = ohl * 2 * g.RS + ow * kPPix + 16 * (lane >> 5);
    float4 e_sc0, e_sc1, e_sc2, e_sc3, e_sh0, e_sh1, e_sh2, e_sh3;
    {
        const int nb = n_blk + ct * 32 + 4 * (lane >> 5);
        e_sc0 = *reinterpret_cast<const float4*>(nb + 0 < p.N ? p.scale + nb + 0 : p.zeros);
        e_sc1 = *reinterpret_cast<const float4*>(nb + 8 < p.N ? p.scale + nb + 8 : p.zeros);
        e_sc2 = *reinterpret_cast<const float4*>(nb + 16 < p.N ? p.scale + nb + 16 : p.zeros);
        e_sc3 = *reinterpret_cast<const float4*>(nb + 24 < p.N ? p.scale + nb + 24 : p.zeros);
        e_sh0 = *reinterpret_cast<const float4*>(nb + 0 < p.N ? p.shift + nb + 0 : p.zeros);
        e_sh1 = *reinterpret_cast<const float4*>(nb + 8 < p.N ? p.shift + nb + 8 : p.zeros);
        e_sh2 = *reinterpret_cast<const float4*>(nb + 16 < p.N ? p.shift + nb + 16 : p.zeros);
        e_sh3 = *reinterpret_cast<const float4*>(nb + 24 < p.N ? p.shift + nb + 24 : p.zeros);
    }
    // the inbox has a slot per tile parity (round 4): tile i's partial sums are read behind barrier X(i), the same slot is written again
    // behind X(i + 1) - no second barrier per tile for "partial sums consumed" (it also held the producers: 2 barriers -> 1)
    float4* const part_out0 = reinterpret_cast<float4*>(part + (wave ^ 2) * 4096) + lane;     // the partner's inbox
    const float4* const part_in0 = reinterpret_cast<const float4*>(part + wave * 4096) + lane;
#ifdef AMPH_STAMPS
    unsigned long long q_u0 = 0, q_u1 = 0, q_u2 = 0;
    AMPH_F4_T(q_u0)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMPH_F4_T(q_u1)
#endif
    __syncthreads();                                                 // A
#ifdef AMPH_STAMPS
    AMPH_F4_T(q_u2)
#endif
    __syncthreads();                                                 // X: the first patch is complete
#ifdef AMPH_STAMPS
    AMPH_F4_T(q_d)
    const unsigned long long q_start = q_d - q_t0;
#endif
    int i = 0;
    bool bad = false;                                                // an output conv3's fp16 split cannot hold (amph_internal.h kH2Max)
    for (int tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x, ++i) {
        const char* pb = ((i & 1) ? patch1 : patch0) + a_base;
        AMPH_F4_T(q_a)
        // the image rows of tile i + 2 (the producers build patch i + 1 from the other stage buffer meanwhile): requested here, split and
        // stored behind the multiply.  Element e = tid + 256 u of the [2 PR + 1][img_row] stage: row r = image row 4 oh0 - 3 + r, column c - 3
        const int st_tile = tile + 2 * (int)gridDim.x;
        const bool st_on = st_tile < g.n_tiles;                      // wave-uniform
        float pf[6];
        if (st_on) {
            const int b_ = f4_div_small(st_tile, g.tiles_per_image, rcp_tpi), tr_ = st_tile - b_ * g.tiles_per_image;
            const int ir0_ = 4 * tr_ * g.n_oh - 3;
            const float* img_ = p.img + ((long long)b_ * p.H0 + ir0_) * p.W0;
            const float rcp_ir = 1.0f / (float)g.img_row;
#pragma unroll
            for (int u = 0; u < 6; ++u) {
                const int e = tid + 256 * u;
                const int r = f4_div_small(e, g.img_row, rcp_ir), ic = e - r * g.img_row - 3;
                const bool ok_ = e < g.n_img && ic >= 0 && ic < p.W0 && (unsigned)(ir0_ + r) < (unsigned)p.H0;
                pf[u] = *(ok_ ? img_ + r * p.W0 + ic : p.zeros);
            }
        }
        f32x16 acc0, acc1, accl0, accl1;                             // per row tile: h h' | the cross terms
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = 0.0f; acc1[r] = 0.0f; accl0[r] = 0.0f; accl1[r] = 0.0f; }
#define AMPH_F4_FR(ptr_) __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ptr_))
        f16x8 c0h, c0l, c1h, c1l, n0h, n0l, n1h, n1l;
        {
            const char* pa = pb + aoff[0];
            c0h = AMPH_F4_FR(pa); c0l = AMPH_F4_FR(pa + 64);
            c1h = AMPH_F4_FR(pa + g.rt1_off); c1l = AMPH_F4_FR(pa + g.rt1_off + 64);
        }
#pragma unroll
        for (int s = 0; s < 9; ++s) {
            if (s < 8) {
                const char* pa = pb + aoff[s + 1];
                n0h = AMPH_F4_FR(pa); n0l = AMPH_F4_FR(pa + 64);
                n1h = AMPH_F4_FR(pa + g.rt1_off); n1l = AMPH_F4_FR(pa + g.rt1_off + 64);
            }
            const f16x8 wh = __builtin_bit_cast(f16x8, breg[2 * s]), wl = __builtin_bit_cast(f16x8, breg[2 * s + 1]);
            accl0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, c0h, accl0, 0, 0, 0);
            accl1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wl, c1h, accl1, 0, 0, 0);
            accl0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, c0l, accl0, 0, 0, 0);
            accl1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, c1l, accl1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, c0h, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(wh, c1h, acc1, 0, 0, 0);
            if (s < 8) { c0h = n0h; c0l = n0l; c1h = n1h; c1l = n1l; }
        }
#undef AMPH_F4_FR
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] = fmaf(accl0[r], kG4Lo, acc0[r]); acc1[r] = fmaf(accl1[r], kG4Lo, acc1[r]); }
        if (st_on) {
            uint16_t* dst = img_stage + (i & 1) * 2 * img_pl;        // tile i + 2's stage buffer: last read while patch i was built
#pragma unroll
            for (int k2 = 0; k2 < 3; ++k2) {
                uint32_t h_, l_;
                g4_split2(pf[2 * k2], pf[2 * k2 + 1], h_, l_);
                const int e0 = tid + 256 * (2 * k2), e1 = e0 + 256;
                uint16_t* d0_ = dst + (e0 < g.n_img ? e0 : g.n_img);   // lanes past the last element store to the plane's pad element
                uint16_t* d1_ = dst + (e1 < g.n_img ? e1 : g.n_img);
                d0_[0] = (uint16_t)h_; d0_[img_pl] = (uint16_t)l_;
                d1_[0] = (uint16_t)(h_ >> 16); d1_[img_pl] = (uint16_t)(l_ >> 16);
            }
        }
        // the partial sum of the row tile this wave does not own goes to the partner (same ct, other K half)
        AMPH_F4_T(q_b)
        f32x16 keep, send;
        if (th) { keep = acc1; send = acc0; } else { keep = acc0; send = acc1; }
        float4* const part_out = part_out0 + (i & 1) * 1024;         // (4 x 4096 bytes per parity)
        const float4* const part_in = part_in0 + (i & 1) * 1024;
#pragma unroll
        for (int q = 0; q < 4; ++q) part_out[q * 64] = make_float4(send[4 * q], send[4 * q + 1], send[4 * q + 2], send[4 * q + 3]);
        __syncthreads();                                             // X: partial sums published; patch i + 1 complete; patch i free
        float4 rv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) rv[q] = part_in[q * 64];
        AMPH_F4_T(q_c)
        AMPH_F4_T(q_d)
        {
            const int m = tile * 64 + th * 32 + m0;
            if (m < p.M) {
                // K halves added low + high whichever wave holds which
#define AMPH_F4_SUM(q_, c_) (th ? rv[q_].c_ + keep[4 * q_ + AMPH_F4_IX_##c_] : keep[4 * q_ + AMPH_F4_IX_##c_] + rv[q_].c_)
#define AMPH_F4_IX_x 0
#define AMPH_F4_IX_y 1
#define AMPH_F4_IX_z 2
#define AMPH_F4_IX_w 3
#define AMPH_F4_OUT(gq)                                                                    \
                {                                                                          \
                    const int n0 = n_blk + ct * 32 + 8 * gq + 4 * (lane >> 5);             \
                    if (n0 < p.N) {                                                        \
                        float4 v;                                                          \
                        v.x = fmaf(AMPH_F4_SUM(gq, x), e_sc##gq.x, e_sh##gq.x);            \
                        v.y = fmaf(AMPH_F4_SUM(gq, y), e_sc##gq.y, e_sh##gq.y);            \
                        v.z = fmaf(AMPH_F4_SUM(gq, z), e_sc##gq.z, e_sh##gq.z);            \
                        v.w = fmaf(AMPH_F4_SUM(gq, w), e_sc##gq.w, e_sh##gq.w);            \
                        bad |= h2_unsplittable(v.x) | h2_unsplittable(v.y) | h2_unsplittable(v.z) | h2_unsplittable(v.w); \
                        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); \
                        if (p.y_bf16) {                                                    \
                            const amph_bf16x2 lo_ = {(__bf16)v.x, (__bf16)v.y}, hi_ = {(__bf16)v.z, (__bf16)v.w};   /* RNE */ \
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.N + n0) =       \
                                make_uint2(__builtin_bit_cast(uint32_t, lo_), __builtin_bit_cast(uint32_t, hi_));          \
                        } else *reinterpret_cast<float4*>(p.y + (long long)m * p.N + n0) = v; \
                    }                                                                      \
                }
                AMPH_F4_OUT(0) AMPH_F4_OUT(1) AMPH_F4_OUT(2) AMPH_F4_OUT(3)
#undef AMPH_F4_OUT
#undef AMPH_F4_SUM
            }
        }
#ifdef AMPH_STAMPS
        {
            unsigned long long q_e;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            AMPH_F4_T(q_e)
            q_s0 += q_b - q_a; q_s1 += q_c - q_b; q_s2 += q_d - q_c; q_s3 += q_e - q_d;
        }
#endif
    }
    if (bad && p.range_flag && !p.y_bf16) *p.range_flag = 1;         // (a bf16 output has fp32's range)
#ifdef AMPH_STAMPS
    if (tid == 0 && p.stamps) {       // consumer wave 0: start-up | multiply | partial sums + barrier X | barrier Y | (producer: build) | lifetime ticks | (producer: barriers) | start
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 16;
        o[0] = q_start; o[1] = q_s0; o[2] = q_s1; o[3] = q_s2 + q_s3;
        o[4] = q_u0 - q_t0; o[6] = q_u1 - q_t0; o[12] = q_u2 - q_t0;      // start-up: own loads issued | landed | barrier A passed
        o[5] = __builtin_amdgcn_s_memrealtime() - q_rt0; o[7] = q_rt0;
    }
#endif
}

static PatchGeo f4s_geo(const ConvGemmParams& p) {
    PatchGeo g{};
    g.n_oh = 64 / p.OW;
    g.PR = 2 * g.n_oh + 1;
    g.PC = 2 * p.OW + 1;
    g.RS = ((p.OW + 1) * kPPix + 127) & ~127;
    g.plane = g.PR * g.RS + 64;
    g.img_row = (2 * g.PC + 2) & ~1;
    g.n_img = (2 * g.PR + 1) * g.img_row;
    g.n_pix = g.PR * g.PC;
    g.tiles_per_image = p.OH / g.n_oh;
    g.n_tiles = p.M / 64;
    g.rt1_off = (32 / p.OW) * 2 * g.RS;
    return g;
}
static size_t f4s_lds(const PatchGeo& g) { return (size_t)4 * g.plane + (size_t)2 * 2 * ((g.n_img + 8) & ~7) * sizeof(uint16_t) + 2 * 4 * 4096; }

// the shapes it covers: 3x3 stride-2 pad-1 twice, 32 layer-1 channels, whole output rows per 64-pixel tile (OW 16 or 32)
bool conv_gemm_f4s_ok(const ConvGemmParams& p) {
    if (!(p.img && p.zeros && p.k == 3 && p.stride == 2 && p.pad == 1 && p.k1 == 3 && p.s1 == 2 && p.p1 == 1 && p.C_in == 32 &&
          p.K == 288 && p.K_pad == 288 && p.splits == 1 && p.epilogue == 0 && (p.N % 4) == 0))
        return false;
    if ((p.OW != 16 && p.OW != 32) || (p.OH % (64 / p.OW)) || (p.M % 64) || p.M <= 0) return false;
    if (p.H != (p.H0 + 2 - 3) / 2 + 1 || p.W != (p.W0 + 2 - 3) / 2 + 1 || p.OH != (p.H + 2 - 3) / 2 + 1 || p.OW != (p.W + 2 - 3) / 2 + 1) return false;
    const PatchGeo g = f4s_geo(p);
    return g.n_img <= 256 * kPImgU && g.n_pix <= 12 * 32 && (kPImgU % 2) == 0 && g.n_img <= 256 * kPImgU - 256 * (kPImgU & 1) && p.W == 2 * p.OW && p.H == 2 * p.OH && f4s_lds(g) <= 160 * 1024 && p.M < (1 << 22);
}

// wfrag = conv_gemm_h2_fragment_weights' output on the device (two fp16 terms in MFMA fragment order); wgs = workgroups to launch (0: the rule below)
void launch_conv_gemm_f4s(const ConvGemmParams& p, const void* wfrag, int wgs, hipStream_t s) {
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_f4s), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised[dev] = true;
    }
    const PatchGeo g = f4s_geo(p);
    // eight tiles per workgroup amortise its start-up (weights, first patch: ~10 k cycles, 2.5 tiles' worth); half the CUs at most -
    // the other lanes' kernels fill the rest (tools/sweep.py patch_wgs=...: 64..128 workgroups 2.41 M chunks/s, 256: 2.37 M)
    // (round 4: sixteen tiles each, a quarter of the CUs at most - 64 workgroups at a 256-chunk batch, with 64 for conv3 as well: 4.09 against
    // 4.04 M chunks/s with 128 + 128, seven alternating rounds on one box, profiles/r04_stft_cq.txt; 48: 3.93 M.  The bf16 encoder keeps 128.)
    if (wgs <= 0) wgs = std::max(1, std::min(g_num_cus / 4, (g.n_tiles + 15) / 16));
    const dim3 grid((unsigned)std::min(g.n_tiles, wgs), (unsigned)((p.N + 63) / 64), 1);      // persistent
    hipLaunchKernelGGL(k_conv_gemm_f4s, grid, dim3(512), f4s_lds(g), s, p, g, reinterpret_cast<const uint4*>(wfrag));
}

// ----------------------------------------------------------------------------
// The same pair for the bf16 encoder (BASELINE configs[3]: bf16 weights and activations, bf16 MFMA, fp32 accumulate / BN): "f4b".
// k_conv_gemm_f4s with ONE bf16 term per operand: image and layer-1 weights rounded to bf16 (RNE), layer 1 as one MFMA per group
// of 32 patch pixels, BN + ReLU in fp32, the activation rounded to bf16 into an 80-byte-pixel patch (5 slots: odd), layer 2 as one
// MFMA per k-step with ALL 18 k-steps of a wave's 32 channels in 72 registers - no K split, so no partial-sum exchange: one barrier
// per tile.  bf16 NHWC output.  56 KB of LDS, 160 registers (72 of them weights): one workgroup per CU.
// ----------------------------------------------------------------------------
constexpr int kBPix = 80;                // bytes per patch pixel: 32 bf16 + 16 B pad

struct PatchGeoB {
    int n_oh, PR, PC, RS, plane, img_row, n_img, n_pix, tiles_per_image, n_tiles, rt1_off;
};

__global__ __launch_bounds__(512, 2) void k_conv_gemm_f4b(ConvGemmParams p, PatchGeoB g, const uint4* __restrict__ wfrag) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch0 = smem;                                                       // [2 parity][PR][RS]
    char* const patch1 = smem + 2 * g.plane;
    const int img_pl = (g.n_img + 8) & ~7;
    uint16_t* const img_stage = reinterpret_cast<uint16_t*>(smem + 4 * g.plane);     // [2][2 PR + 1][img_row] bf16
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n_blk = blockIdx.y * 64;
    const float rcp_tpi = 1.0f / (float)g.tiles_per_image;
    const int px = lane & 31, kh2 = lane >> 5;

    if (wave >= 4) {
        // ================================================================ producers
        const int ptid = tid - 256, pw = wave - 4;
        bf16x8 w1f;                                                  // layer-1 weights as an MFMA fragment (k <-> window dwords: k_conv_gemm_f4s)
        {
            uint32_t d4[4];
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const int tap0 = d < 3 ? (kh2 ? 3 * d + 2 : 3 * d) : -1, tap1 = (d < 3 && !kh2) ? 3 * d + 1 : -1;
                const float v0 = tap0 >= 0 ? p.w1[tap0 * p.C_in + px] : 0.0f, v1 = tap1 >= 0 ? p.w1[tap1 * p.C_in + px] : 0.0f;
                const amph_bf16x2 pk = {(__bf16)v0, (__bf16)v1};
                d4[d] = __builtin_bit_cast(uint32_t, pk);
            }
            w1f = __builtin_bit_cast(bf16x8, make_uint4(d4[0], d4[1], d4[2], d4[3]));
        }
        float4 sc1[4], sh1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            sc1[q] = *reinterpret_cast<const float4*>(p.scale1 + 8 * q + 4 * kh2);
            sh1[q] = *reinterpret_cast<const float4*>(p.shift1 + 8 * q + 4 * kh2);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (see k_conv_gemm_f4s: keeps these loads out of the tile loop's wait counts)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(sc1[q].x), "+v"(sc1[q].y), "+v"(sc1[q].z), "+v"(sc1[q].w));
            asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3" : "+v"(sh1[q].x), "+v"(sh1[q].y), "+v"(sh1[q].z), "+v"(sh1[q].w));
        }
        int it_src[kPGroupU], it_dst[kPGroupU];
        unsigned it_live = 0, it_pad_always = 0, it_pad_top = 0;
        {
            const float rcp_pc = 1.0f / (float)g.PC;
            const int ppg = (g.n_pix + 11) / 12;
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u) {
                const int pix = (pw + 4 * u) * ppg + px;
                const bool live = px < ppg && pix < g.n_pix;
                const int pc = live ? pix : 0;
                const int yl = f4_div_small(pc, g.PC, rcp_pc), xi = pc - yl * g.PC;
                it_src[u] = 2 * (2 * yl * g.img_row + 2 * xi + 2 * kh2);
                it_dst[u] = (xi & 1) * g.plane + yl * g.RS + (xi >> 1) * kBPix + 8 * kh2;
                if (live) it_live |= 1u << u;
                if (xi == 0) it_pad_always |= 1u << u;
                if (yl == 0) it_pad_top |= 1u << u;
            }
        }
        int io_r[kPImgU], io_off[kPImgU], io_w[kPImgU];
        {
            const float rcp_ir = 1.0f / (float)g.img_row;
#pragma unroll
            for (int u = 0; u < kPImgU; ++u) {
                const int e = ptid + 256 * u;
                const int r = f4_div_small(e, g.img_row, rcp_ir), ic = e - r * g.img_row - 3;
                io_r[u] = r;
                io_off[u] = (e < g.n_img && ic >= 0 && ic < p.W0) ? r * p.W0 + ic : -1;
                io_w[u] = e < g.n_img ? e : g.n_img;
            }
        }
        float pf[kPImgU];
#define AMPH_F4B_FETCH_IMG(tile_)                                                          \
        {                                                                                  \
            const int t_ = (tile_);                                                        \
            const int b_ = f4_div_small(t_, g.tiles_per_image, rcp_tpi), tr_ = t_ - b_ * g.tiles_per_image; \
            const int ir0_ = 4 * tr_ * g.n_oh - 3;                                         \
            const float* img_ = p.img + ((long long)b_ * p.H0 + ir0_) * p.W0;              \
            _Pragma("unroll") for (int u = 0; u < kPImgU; ++u)                             \
                pf[u] = *((io_off[u] >= 0 && (unsigned)(ir0_ + io_r[u]) < (unsigned)p.H0) ? img_ + io_off[u] : p.zeros); \
        }
#define AMPH_F4B_WRITE_IMG(dst_)                                                           \
        {                                                                                  \
            _Pragma("unroll") for (int k = 0; k < kPImgU / 2; ++k) {                       \
                const amph_bf16x2 pk_ = {(__bf16)pf[2 * k], (__bf16)pf[2 * k + 1]};        \
                const uint32_t w_ = __builtin_bit_cast(uint32_t, pk_);                     \
                (dst_)[io_w[2 * k]] = (uint16_t)w_; (dst_)[io_w[2 * k + 1]] = (uint16_t)(w_ >> 16); \
            }                                                                              \
        }
        int tile = blockIdx.x;
        AMPH_F4B_FETCH_IMG(tile)
        AMPH_F4B_WRITE_IMG(img_stage)
        __syncthreads();                                             // A
        for (int i = 0; tile < g.n_tiles; tile += gridDim.x, ++i) {
            const int nt = tile + (int)gridDim.x;
            AMPH_F4B_FETCH_IMG(nt < g.n_tiles ? nt : tile)
            const char* sbuf = reinterpret_cast<const char*>(img_stage + (i & 1) * img_pl);
            char* patch = (i & 1) ? patch1 : patch0;
            const int tr = tile - f4_div_small(tile, g.tiles_per_image, rcp_tpi) * g.tiles_per_image;
            const unsigned it_pad = it_pad_always | (tr == 0 ? it_pad_top : 0u);
            uint32_t xw[kPGroupU][3];
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u)
#pragma unroll
                for (int r = 0; r < 3; ++r) xw[u][r] = *reinterpret_cast<const uint32_t*>(sbuf + it_src[u] + 2 * r * g.img_row);
            f32x16 acc[kPGroupU];
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.0f;
                acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w1f, __builtin_bit_cast(bf16x8, make_uint4(xw[u][0], xw[u][1], xw[u][2], 0u)), acc[u], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < kPGroupU; ++u) {
                const float cap = ((it_pad >> u) & 1u) ? 0.0f : __builtin_inff();
                uint32_t o[8];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float a0 = __builtin_amdgcn_fmed3f(fmaf(acc[u][4 * q], sc1[q].x, sh1[q].x), 0.0f, cap);
                    const float a1 = __builtin_amdgcn_fmed3f(fmaf(acc[u][4 * q + 1], sc1[q].y, sh1[q].y), 0.0f, cap);
                    const float a2 = __builtin_amdgcn_fmed3f(fmaf(acc[u][4 * q + 2], sc1[q].z, sh1[q].z), 0.0f, cap);
                    const float a3 = __builtin_amdgcn_fmed3f(fmaf(acc[u][4 * q + 3], sc1[q].w, sh1[q].w), 0.0f, cap);
                    const amph_bf16x2 p0 = {(__bf16)a0, (__bf16)a1}, p1 = {(__bf16)a2, (__bf16)a3};
                    o[2 * q] = __builtin_bit_cast(uint32_t, p0); o[2 * q + 1] = __builtin_bit_cast(uint32_t, p1);
                }
                if ((it_live >> u) & 1u) {
                    char* dst = patch + it_dst[u];
#pragma unroll
                    for (int q = 0; q < 4; ++q) *reinterpret_cast<uint2*>(dst + 16 * q) = make_uint2(o[2 * q], o[2 * q + 1]);
                }
            }
            AMPH_F4B_WRITE_IMG(img_stage + ((i + 1) & 1) * img_pl)
            __syncthreads();                                         // X: patch i complete (the consumers are done with patch i - 1)
        }
        __syncthreads();                                             // the consumers' last X
#undef AMPH_F4B_FETCH_IMG
#undef AMPH_F4B_WRITE_IMG
        return;
    }

    // ==================================================================== consumers: (row tile, channel tile), all of K
    const int rt = wave >> 1, ct = wave & 1;
    uint4 breg[18];
    {
        const uint4* wf = wfrag + ((size_t)blockIdx.y * 2 + ct) * 18 * 64 + lane;
#pragma unroll
        for (int f = 0; f < 18; ++f) breg[f] = wf[f * 64];
    }
    const int m_local = rt * 32 + px;
    const int ohl = m_local / p.OW, ow = m_local - ohl * p.OW;
    const int a_base = ohl * 2 * g.RS + ow * kBPix + 16 * kh2;
    float4 e_sc[4], e_sh[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int nb = n_blk + ct * 32 + 4 * kh2 + 8 * q;
        e_sc[q] = *reinterpret_cast<const float4*>(nb < p.N ? p.scale + nb : p.zeros);
        e_sh[q] = *reinterpret_cast<const float4*>(nb < p.N ? p.shift + nb : p.zeros);
    }
    __syncthreads();                                                 // A
    __syncthreads();                                                 // X: the first patch is complete
    int i = 0;
    for (int tile = blockIdx.x; tile < g.n_tiles; tile += gridDim.x, ++i) {
        const char* pb = ((i & 1) ? patch1 : patch0) + a_base;
        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        bf16x8 xf[18];
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) {
            const int tap = s_ >> 1, kh = tap / 3, kw = tap - 3 * kh;
            xf[s_] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(pb + (kw & 1) * g.plane + kh * g.RS + (kw >> 1) * kBPix + 32 * (s_ & 1)));
        }
#pragma unroll
        for (int s_ = 0; s_ < 18; ++s_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, breg[s_]), xf[s_], acc, 0, 0, 0);
        {
            const int m = tile * 64 + m_local;
            if (m < p.M) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int n0 = n_blk + ct * 32 + 8 * q + 4 * kh2;
                    if (n0 < p.N) {
                        const float v0 = fmaxf(fmaf(acc[4 * q], e_sc[q].x, e_sh[q].x), 0.0f), v1 = fmaxf(fmaf(acc[4 * q + 1], e_sc[q].y, e_sh[q].y), 0.0f);
                        const float v2 = fmaxf(fmaf(acc[4 * q + 2], e_sc[q].z, e_sh[q].z), 0.0f), v3 = fmaxf(fmaf(acc[4 * q + 3], e_sc[q].w, e_sh[q].w), 0.0f);
                        const amph_bf16x2 lo_ = {(__bf16)v0, (__bf16)v1}, hi_ = {(__bf16)v2, (__bf16)v3};
                        *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.N + n0) =
                            make_uint2(__builtin_bit_cast(uint32_t, lo_), __builtin_bit_cast(uint32_t, hi_));
                    }
                }
            }
        }
        __syncthreads();                                             // X: patch i + 1 complete; patch i free
    }
}

static PatchGeoB f4b_geo(const ConvGemmParams& p) {
    PatchGeoB g{};
    g.n_oh = 64 / p.OW;
    g.PR = 2 * g.n_oh + 1;
    g.PC = 2 * p.OW + 1;
    g.RS = ((p.OW + 1) * kBPix + 127) & ~127;
    g.plane = g.PR * g.RS + 64;
    g.img_row = (2 * g.PC + 2) & ~1;
    g.n_img = (2 * g.PR + 1) * g.img_row;
    g.n_pix = g.PR * g.PC;
    g.tiles_per_image = p.OH / g.n_oh;
    g.n_tiles = p.M / 64;
    g.rt1_off = (32 / p.OW) * 2 * g.RS;
    return g;
}
static size_t f4b_lds(const PatchGeoB& g) { return (size_t)4 * g.plane + (size_t)2 * ((g.n_img + 8) & ~7) * sizeof(uint16_t); }

// bf16 weights of the layer in MFMA fragment order: [n tile][channel tile][k-step][lane][8 bf16] (RNE)
void conv_gemm_f4b_fragment_weights(const float* w, int n_pad, int K_pad, uint16_t* out) {
    auto rne = [](float f) {
        uint32_t u;
        std::memcpy(&u, &f, 4);
        if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
        u += 0x7fffu + ((u >> 16) & 1u);
        return (uint16_t)(u >> 16);
    };
    const int ksteps = K_pad / 16;
    for (int nt = 0; nt < n_pad / 64; ++nt)
        for (int ct = 0; ct < 2; ++ct)
            for (int ks = 0; ks < ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j)
                        out[((((size_t)(nt * 2 + ct) * ksteps + ks) * 64) + lane) * 8 + j] =
                            rne(w[(size_t)(nt * 64 + ct * 32 + (lane & 31)) * K_pad + 16 * ks + 8 * (lane >> 5) + j]);
}

// the shapes k_conv_gemm_f4b covers: k_conv_gemm_f4s's (3x3 stride-2 pad-1 twice, 32 layer-1 channels, whole output rows per
// 64-pixel tile: OW 16 or 32) with ITS LDS image - 80-byte instead of 208-byte patch pixels, one image plane instead of three, no
// partial-sum inbox: OW = 32 (the 128-band geometry of BASELINE configs[3], 24 x 32 output pixels per image) fits in 60 KB where
// the fp32-accurate kernel would need 172
bool conv_gemm_f4b_ok(const ConvGemmParams& p) {
    if (!(p.img && p.zeros && p.k == 3 && p.stride == 2 && p.pad == 1 && p.k1 == 3 && p.s1 == 2 && p.p1 == 1 && p.C_in == 32 &&
          p.K == 288 && p.K_pad == 288 && p.splits == 1 && p.epilogue == 0 && (p.N % 4) == 0))
        return false;
    if ((p.OW != 16 && p.OW != 32) || (p.OH % (64 / p.OW)) || (p.M % 64) || p.M <= 0) return false;
    if (p.H != (p.H0 + 2 - 3) / 2 + 1 || p.W != (p.W0 + 2 - 3) / 2 + 1 || p.OH != (p.H + 2 - 3) / 2 + 1 || p.OW != (p.W + 2 - 3) / 2 + 1) return false;
    const PatchGeoB g = f4b_geo(p);
    return g.n_img <= 256 * kPImgU && g.n_pix <= 12 * 32 && p.W == 2 * p.OW && p.H == 2 * p.OH && f4b_lds(g) <= 80 * 1024 && p.M < (1 << 22);
}

// wfrag = conv_gemm_f4b_fragment_weights' output; y is bf16 NHWC
void launch_conv_gemm_f4b(const ConvGemmParams& p, const void* wfrag, int wgs, hipStream_t s) {
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_f4b), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        raised[dev] = true;
    }
    const PatchGeoB g = f4b_geo(p);
    if (wgs <= 0) wgs = std::max(1, std::min(g_num_cus / 2, (g.n_tiles + 7) / 8));          // as k_conv_gemm_f4s (bf16 bench: 128 workgroups 3.27 M, 256: 3.23 M)
    const dim3 grid((unsigned)std::min(g.n_tiles, wgs), (unsigned)((p.N + 63) / 64), 1);
    hipLaunchKernelGGL(k_conv_gemm_f4b, grid, dim3(512), f4b_lds(g), s, p, g, reinterpret_cast<const uint4*>(wfrag));
}

// ----------------------------------------------------------------------------
// 3x3 stride-2 layers behind the fused pair with the WEIGHTS IN REGISTERS and the input image(s) of a tile resident in LDS ("g4"):
// k_conv_gemm_f4s's consumer side as a whole kernel.  A 512-thread workgroup owns a slice of the output channels for all its tiles:
// wave = (channel tile ct, K part kq) keeps its 9 k-steps x 3 terms of the split weights in 108 registers for the whole kernel, so the
// multiply loop has no weight traffic, no staging and NO barrier: 6 MFMAs per k-step and row tile on fragments read straight from the
// patch.  Per tile: (1) all eight waves split the fp32 NHWC input of the tile into the patch [image][row][column parity][column / 2]
// [term][C_in] bf16 (once per element, not once per tap); (2) multiply; (3) the K parts are added through LDS (the retired patch) in
// part order, BN + ReLU, out.  conv3 (16 x 16 x 64 -> 8 x 8 x 128): tile = one image, 64 channels per workgroup, K in 4 parts;
// conv4 (8 x 8 x 128 -> 4 x 4 x 256): tile = two images, 64 channels per workgroup, K in 4 parts of 18 k-steps (round 4; was 32 channels / 8 parts).
// ----------------------------------------------------------------------------
struct G4Geo {
    int PXB, PL, RS, IMG_B;   // bytes per patch pixel (3 terms x C_in bf16 + 16), per even-column plane of a row, per row, per image
    int n_tiles;              // tiles of IM images
    int patch_bytes;
};

// C_IN input channels; KW K parts x CT channel tiles of 32 = 8 waves; RT row tiles of 32 output pixels per tile; IM images per tile;
// OWL = log2(output width), OH output rows (default: square); the input is (2 OH) x (2 OW) x C_IN per image
// TERMS = 2: fp32 NHWC input split in two fp16 terms (x = h + 2^-11 l), fp32 output; TERMS = 1 (bf16 encoder): bf16 NHWC input copied, bf16 weights, bf16 output
// (round 4: OH != OW and RT = 3 - conv4 of the 128-band geometry of BASELINE configs[3], 12 x 16 x 128 -> 6 x 8 x 256 in bf16: two images = 96
// output pixels = three row tiles per tile, a row tile may straddle the two images; one workgroup per CU, so up to 256 registers)
template <int C_IN, int KW, int CT, int RT, int IM, int OWL, int TERMS, int OH = (1 << OWL)>
__global__ __launch_bounds__(512, RT >= 3 ? 1 : 2) void k_conv_gemm_g4(ConvGemmParams p, G4Geo g, const uint4* __restrict__ wfrag) {
    constexpr int OW = 1 << OWL, W = 2 * OW, H = 2 * OH, PR = H + 1, PC = W + 1;
    constexpr int KSTEPS = 9 * C_IN / 16, KPW = KSTEPS / KW;            // k-steps of 16 in all, per wave (9; 18 in the 12 x 16 x 128 form)
    constexpr int SPT = C_IN / 16;                                       // k-steps per tap
    constexpr int TERM_B = 2 * C_IN;
    constexpr int CQ = TERMS != 1 ? C_IN / 4 : C_IN / 8;                 // 16-byte units per input pixel: float4 (fp32) / 8 bf16
    constexpr int NV = IM * H * W * CQ / 512;                            // units per thread and tile (8 / 4; 12 for the 12 x 16 x 128 bf16 images)
    constexpr int RPO = 16 / KW;                                         // accumulator registers per owner and row tile
    static_assert(KW * CT == 8 && KPW * KW == KSTEPS && RT * 32 == IM * OH * OW && NV * 512 == IM * H * W * CQ, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#ifdef AMPH_STAMPS
    unsigned long long q_t0 = 0, q_a = 0, q_b = 0, q_c = 0, q_d = 0, q_e = 0, q_f = 0, q_s0 = 0, q_s1 = 0, q_s2 = 0, q_s3 = 0, q_s4 = 0, q_start = 0;
    const unsigned long long q_rt0 = __builtin_amdgcn_s_memrealtime();
    AMPH_F4_T(q_t0)
#endif
    const int ct = CT == 2 ? (wave & 1) : 0, kq = CT == 2 ? (wave >> 1) : wave;
    const int px = lane & 31, kh2 = lane >> 5;
    // ---- this wave's weights: k-steps 9 kq .. 9 kq + 8 of channel tile (64-wide n tile, half) in fragment order
    const int n_tile = CT == 2 ? blockIdx.y : (blockIdx.y >> 1), half = CT == 2 ? ct : (blockIdx.y & 1);
    uint4 breg[KPW * TERMS];
    {
        const uint4* wf = wfrag + (((size_t)n_tile * 2 + half) * KSTEPS + KPW * kq) * (TERMS * 64) + lane;
#pragma unroll
        for (int f = 0; f < KPW * TERMS; ++f) breg[f] = wf[f * 64];
    }
    int aoff[KPW];                                                       // patch offset of k-step 9 kq + i: tap, channel group
#pragma unroll
    for (int i = 0; i < KPW; ++i) {
        const int s = KPW * kq + i, tap = s / SPT, kh = tap / 3, kw = tap - 3 * kh;
        aoff[i] = kh * g.RS + (kw & 1) * g.PL + (kw >> 1) * g.PXB + 32 * (s % SPT);
    }
    // lane -> output pixel 32 t + px of row tile t: image, row, column
    int a_base[RT];
#pragma unroll
    for (int t = 0; t < RT; ++t) {
        const int pp = 32 * t + px, im_ = pp / (OH * OW), rem_ = pp - im_ * (OH * OW);
        a_base[t] = im_ * g.IMG_B + (rem_ >> OWL) * 2 * g.RS + (rem_ & (OW - 1)) * g.PXB + 16 * kh2;
    }
    const int n_base = blockIdx.y * (32 * CT) + ct * 32;                 // first output channel of this wave
    // BN constants of the RPO channels this wave writes out (accumulator register j of a lane = channel 8 (j / 4) + 4 kh2 + (j % 4))
    const int n_own = n_base + 8 * ((RPO * kq) >> 2) + 4 * kh2 + ((RPO * kq) & 3);
    float e_sc[RPO], e_sh[RPO];
#pragma unroll
    for (int r = 0; r < RPO; ++r) { e_sc[r] = n_own < p.N ? p.scale[n_own + r] : 0.0f; e_sh[r] = n_own < p.N ? p.shift[n_own + r] : 0.0f; }
    // ---- input elements of this thread (the same for every tile): float4 index tid + 512 u = (image, y, x, channel group)
    int in_dst[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = tid + 512 * u, c4 = i % CQ, pix = i / CQ, im = pix / (H * W), y = (pix / W) % H, x = pix % W;
        in_dst[u] = im * g.IMG_B + (y + 1) * g.RS + ((x + 1) & 1) * g.PL + ((x + 1) >> 1) * g.PXB + (TERMS != 1 ? 8 : 16) * c4;
    }
    float4 v[NV];
    const int n_v_total = (p.M / (OH * OW)) * (H * W * CQ);              // float4 of all the batch's images (a last tile may be short of images)
#define AMPH_G4_FETCH(tile_)                                                               \
    {                                                                                      \
        const int base_ = (tile_) * (IM * H * W * CQ) + tid;                               \
        const float4* src_ = reinterpret_cast<const float4*>(p.x);                         \
        _Pragma("unroll") for (int u = 0; u < NV; ++u)                                     \
            v[u] = (IM == 1 || base_ + 512 * u < n_v_total) ? src_[base_ + 512 * u] : make_float4(0.f, 0.f, 0.f, 0.f); \
    }
    int tile = blockIdx.x;
    if (tile < g.n_tiles) AMPH_G4_FETCH(tile)
    // zero padding of the patch (row 0 and column 0 of every image): written once, never overwritten
    {
        constexpr int CH = TERMS * TERM_B / 16, NPAD = IM * (PC + PR - 1) * CH;
        for (int i = tid; i < NPAD; i += 512) {
            const int j = i / CH, ch = i - j * CH, im = j / (PC + PR - 1), jj = j - im * (PC + PR - 1);
            const int row = jj < PC ? 0 : jj - PC + 1, xi = jj < PC ? jj : 0;
            *reinterpret_cast<uint4*>(patch + im * g.IMG_B + row * g.RS + (xi & 1) * g.PL + (xi >> 1) * g.PXB + 16 * ch) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    float4* const inbox = reinterpret_cast<float4*>(patch);             // [8 waves][RT][KW - 1 senders][64 lanes] (RPO == 4) after the multiply
#ifdef AMPH_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMPH_F4_T(q_a)
    q_start = q_a - q_t0;
#endif
    bool bad = false;                                                    // an output the next layer's fp16 split cannot hold (amph_internal.h kH2Max)
    for (; tile < g.n_tiles; tile += gridDim.x) {
        // ---- (1) the tile's input into the patch: split, stage by stage over the 2 NV pairs of this thread (bf16 input: copied)
        AMPH_F4_T(q_a)
        if (TERMS == 1) {
#pragma unroll
            for (int u = 0; u < NV; ++u) *reinterpret_cast<float4*>(patch + in_dst[u]) = v[u];
        } else {
            // two fp16 terms, stage by stage over the 2 NV pairs (every instruction's inputs are 2 NV instructions old)
            float x_[4 * NV], r_[4 * NV];
            uint32_t h_[2 * NV], l_[2 * NV];
#pragma unroll
            for (int u = 0; u < NV; ++u) { x_[4 * u] = v[u].x; x_[4 * u + 1] = v[u].y; x_[4 * u + 2] = v[u].z; x_[4 * u + 3] = v[u].w; }
            amph_h2v hp_[2 * NV];
#pragma unroll
            for (int k = 0; k < 2 * NV; ++k) { hp_[k] = amph_h2v{(_Float16)x_[2 * k], (_Float16)x_[2 * k + 1]}; h_[k] = __builtin_bit_cast(uint32_t, hp_[k]); }
#pragma unroll
            for (int k = 0; k < 2 * NV; ++k) { r_[2 * k] = (x_[2 * k] - (float)hp_[k].x) * 2048.0f; r_[2 * k + 1] = (x_[2 * k + 1] - (float)hp_[k].y) * 2048.0f; }
#pragma unroll
            for (int k = 0; k < 2 * NV; ++k) { const amph_h2v lp = {(_Float16)r_[2 * k], (_Float16)r_[2 * k + 1]}; l_[k] = __builtin_bit_cast(uint32_t, lp); }
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                char* dst = patch + in_dst[u];
                *reinterpret_cast<uint2*>(dst) = make_uint2(h_[2 * u], h_[2 * u + 1]);
                *reinterpret_cast<uint2*>(dst + TERM_B) = make_uint2(l_[2 * u], l_[2 * u + 1]);
            }
        }
        AMPH_F4_T(q_b)
        __syncthreads();
        AMPH_F4_T(q_c)
        if (tile + (int)gridDim.x < g.n_tiles) AMPH_G4_FETCH(tile + gridDim.x)      // lands under the multiply
        // ---- (2) multiply: 9 k-steps, fragments of the next one read while this one is multiplied
        typedef typename G4Frag<TERMS>::type frag_t;
        f32x16 acc[RT], accl[RT];                                        // (accl: the cross terms of the two-term split)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; accl[t][r] = 0.0f; }
#define AMPH_G4_FR(ptr_) __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(ptr_))
        frag_t fc[RT][TERMS], fn[RT][TERMS];
        {
            const char* pa = patch + aoff[0];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int q = 0; q < TERMS; ++q) fc[t][q] = AMPH_G4_FR(pa + a_base[t] + q * TERM_B);
        }
#pragma unroll
        for (int i = 0; i < KPW; ++i) {
            if (i < KPW - 1) {
                const char* pa = patch + aoff[i + 1];
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int q = 0; q < TERMS; ++q) fn[t][q] = AMPH_G4_FR(pa + a_base[t] + q * TERM_B);
            }
            const frag_t wh = __builtin_bit_cast(frag_t, breg[TERMS * i]), wl = __builtin_bit_cast(frag_t, breg[TERMS * i + TERMS - 1]);
#define AMPH_G4_M(a_, w_, q_)                                                              \
            _Pragma("unroll") for (int t = 0; t < RT; ++t) a_[t] = g4_mfma(w_, fc[t][q_], a_[t]);
            if (TERMS == 2) { AMPH_G4_M(accl, wl, 0) AMPH_G4_M(accl, wh, TERMS - 1) }
            AMPH_G4_M(acc, wh, 0)
#undef AMPH_G4_M
            if (i < KPW - 1) {
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int q = 0; q < TERMS; ++q) fc[t][q] = fn[t][q];
            }
        }
#undef AMPH_G4_FR
        if (TERMS == 2) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(accl[t][r], kG4Lo, acc[t][r]);
        }
        AMPH_F4_T(q_d)
        __syncthreads();                                                 // every wave is done with the patch: it becomes the inbox
        AMPH_F4_T(q_e)
        // ---- (3) K parts -> their owners: wave kq owns accumulator registers RPO kq .. RPO kq + RPO - 1 (RPO consecutive channels x 32 pixels)
        // of its channel tile; it sends the others' registers to their inboxes: [owner wave][row tile][sender kq][lane] x RPO floats
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int o = 0; o < KW; ++o) {
                if (o == kq) continue;
                const int ow_ = CT == 2 ? (2 * o + ct) : o;                   // the owner's wave index
                float* dst = reinterpret_cast<float*>(inbox) + ((((size_t)ow_ * RT + t) * KW + kq) * 64 + lane) * RPO;
                if (RPO == 4) *reinterpret_cast<float4*>(dst) = make_float4(acc[t][RPO * o], acc[t][RPO * o + 1], acc[t][RPO * o + 2 % RPO], acc[t][RPO * o + 3 % RPO]);
                else *reinterpret_cast<float2*>(dst) = make_float2(acc[t][RPO * o], acc[t][RPO * o + 1]);
            }
        __syncthreads();
        {
            const int m_img = tile * IM;
#pragma unroll
            for (int t = 0; t < RT; ++t) {
                float sum[RPO];
#pragma unroll
                for (int r = 0; r < RPO; ++r) sum[r] = 0.0f;
#pragma unroll
                for (int sdr = 0; sdr < KW; ++sdr) {                      // part 0 first: a fixed order
                    if (sdr == kq) {
#pragma unroll
                        for (int r = 0; r < RPO; ++r) sum[r] += acc[t][RPO * kq + r];
                    } else {
                        const float* src = reinterpret_cast<const float*>(inbox) + ((((size_t)wave * RT + t) * KW + sdr) * 64 + lane) * RPO;
                        float in_[4];
                        if (RPO == 4) { const float4 q4 = *reinterpret_cast<const float4*>(src); in_[0] = q4.x; in_[1] = q4.y; in_[2] = q4.z; in_[3] = q4.w; }
                        else { const float2 q2 = *reinterpret_cast<const float2*>(src); in_[0] = q2.x; in_[1] = q2.y; in_[2] = 0.f; in_[3] = 0.f; }
#pragma unroll
                        for (int r = 0; r < RPO; ++r) sum[r] += in_[r];
                    }
                }
                const int n0 = n_own;
                const int m = m_img * (OH * OW) + 32 * t + px;          // output pixel index: tiles are whole images
                if (n0 < p.N && m < p.M) {
#pragma unroll
                    for (int r = 0; r < RPO; ++r) {
                        const float t_ = fmaf(sum[r], e_sc[r], e_sh[r]);
                        if (TERMS != 1) bad |= h2_unsplittable(t_);
                        sum[r] = fmaxf(t_, 0.0f);
                    }
                    if (TERMS == 1) {                                  // bf16 NHWC
                        uint16_t* yo = reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.N + n0;
                        const amph_bf16x2 lo_ = {(__bf16)sum[0], (__bf16)sum[1]}, hi_ = {(__bf16)sum[2 % RPO], (__bf16)sum[3 % RPO]};
                        if (RPO == 4) *reinterpret_cast<uint2*>(yo) = make_uint2(__builtin_bit_cast(uint32_t, lo_), __builtin_bit_cast(uint32_t, hi_));
                        else *reinterpret_cast<uint32_t*>(yo) = __builtin_bit_cast(uint32_t, lo_);
                    } else if (RPO == 4) *reinterpret_cast<float4*>(p.y + (long long)m * p.N + n0) = make_float4(sum[0], sum[1], sum[2 % RPO], sum[3 % RPO]);
                    else *reinterpret_cast<float2*>(p.y + (long long)m * p.N + n0) = make_float2(sum[0], sum[1]);
                }
            }
        }
        AMPH_F4_T(q_f)
#ifdef AMPH_STAMPS
        q_s0 += q_b - q_a; q_s1 += q_c - q_b; q_s2 += q_d - q_c; q_s3 += q_e - q_d; q_s4 += q_f - q_e;
#endif
        __syncthreads();                                                 // the inbox is read: the next tile's patch may be written
        // (the zero padding of the patch was overwritten by the inbox: restore it)
        {
            constexpr int CH = TERMS * TERM_B / 16, NPAD = IM * (PC + PR - 1) * CH;
            for (int i = tid; i < NPAD; i += 512) {
                const int j = i / CH, ch = i - j * CH, im = j / (PC + PR - 1), jj = j - im * (PC + PR - 1);
                const int row = jj < PC ? 0 : jj - PC + 1, xi = jj < PC ? jj : 0;
                *reinterpret_cast<uint4*>(patch + im * g.IMG_B + row * g.RS + (xi & 1) * g.PL + (xi >> 1) * g.PXB + 16 * ch) = make_uint4(0u, 0u, 0u, 0u);
            }
        }
    }
    if (bad && p.range_flag) *p.range_flag = 1;
#ifdef AMPH_STAMPS
    if (tid == 0 && p.stamps) {       // wave 0: start-up (weights + first image landed) | patch build | barrier | multiply | barrier | exchange + epilogue
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x + gridDim.x * blockIdx.y) * 8;
        o[0] = q_start; o[1] = q_s0; o[2] = q_s1; o[3] = q_s2; o[4] = q_s3;
        o[5] = __builtin_amdgcn_s_memrealtime() - q_rt0; o[6] = q_s4; o[7] = q_rt0;
    }
#endif
#undef AMPH_G4_FETCH
}

template <int C_IN, int IM, int OWL, int TERMS, int OH = (1 << OWL)>
static G4Geo g4_geo(const ConvGemmParams& p) {
    constexpr int OW = 1 << OWL, W = 2 * OW, H = 2 * OH;
    G4Geo g{};
    g.PXB = TERMS * 2 * C_IN + 16;
    g.PL = (OW + 1) * g.PXB;
    // row stride / image stride: the 16 lanes of every ds_read_b128 group (lane -> image, output row, output column) on 16 different
    // 16-byte slots: searched, not derived (OW = 8: 2 RS = 128 mod 256; OW = 4: 2 RS = 64 mod 256 and whole 256-byte images)
    static const int groups[4][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31},
                                      {0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int rs0 = (W + 1) * g.PXB;
    for (int rs = rs0; rs < rs0 + 512; rs += 16)
        for (int ib = (H + 1) * rs; ib < (H + 1) * rs + 512; ib += 16) {
            bool ok = true;
            for (int gi = 0; gi < 2 && ok; ++gi) {
                unsigned used = 0;
                for (int k = 0; k < 16; ++k) {
                    const int px = groups[gi][k], im = px / (OH * OW), oh = (px - im * (OH * OW)) >> OWL, ow = px & (OW - 1);      // (row tile 0)
                    const int slot = ((im * ib + oh * 2 * rs + ow * g.PXB) >> 4) & 15;
                    if (used & (1u << slot)) { ok = false; break; }
                    used |= 1u << slot;
                }
            }
            if (ok) { g.RS = rs; g.IMG_B = ib; goto found; }
        }
    g.RS = rs0; g.IMG_B = (H + 1) * rs0;                                 // (no conflict-free stride in range: correct, slower)
found:
    g.patch_bytes = IM * g.IMG_B;
    g.n_tiles = (p.M + IM * OH * OW - 1) / (IM * OH * OW);
    return g;
}

// ----------------------------------------------------------------------------
// k_conv_gemm_g4's conv3 form for images that are NOT 16 x 16 ("g5", round 3): the tile is TH = 64 / OW whole output rows of an
// image (OW = 8: 8 rows, OW = 16: 4 rows), several tiles per image, and the patch holds the 2 TH + 1 input rows the tile's taps
// touch: a HALO row on top (the image's zero padding for its first tile, the previous tile's last input row otherwise) and the
// left zero column.  Everything else is g4's kind 3: 64 input channels, 64 output channels per workgroup, wave = (channel tile,
// K quarter) with its 9 k-steps x TERMS of the weights in registers, fragments read straight from the patch, K quarters added
// through LDS in part order.  Covers the 128-band geometry of BASELINE configs[3] (24 x 32 x 64 -> 12 x 16 x 128: three tiles of
// 9 x 33 pixels per image) and the reference's default (48 x 16 x 64 -> 24 x 8 x 128: three tiles of 17 x 17).
// ----------------------------------------------------------------------------
struct G5Geo { int PXB, PL, RS, patch_bytes, n_tiles, tpi, H; };

template <int OWL, int TERMS>
__global__ __launch_bounds__(512, 2) void k_conv_gemm_g5(ConvGemmParams p, G5Geo g, const uint4* __restrict__ wfrag) {
    constexpr int C_IN = 64, KW = 4, RT = 2;
    constexpr int OW = 1 << OWL, TH = 64 >> OWL, WW = 2 * OW, PR = 2 * TH + 1;
    constexpr int KSTEPS = 9 * C_IN / 16, KPW = KSTEPS / KW, SPT = C_IN / 16, TERM_B = 2 * C_IN;
    constexpr int CQ = TERMS != 1 ? C_IN / 4 : C_IN / 8;                 // 16-byte units per input pixel
    constexpr int NE = PR * WW * CQ, NV = (NE + 511) / 512;              // units of a tile's input (halo row included), per thread
    constexpr int UB = TERMS != 1 ? 8 : 16;                              // patch bytes per unit and term
    constexpr int RPO = 16 / KW;
    static_assert(KPW == 9 && RPO == 4 && TH * OW == 64, "shape");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* const patch = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ct = wave & 1, kq = wave >> 1;
    const int px = lane & 31, kh2 = lane >> 5;
    uint4 breg[9 * TERMS];
    {
        const uint4* wf = wfrag + (((size_t)blockIdx.y * 2 + ct) * KSTEPS + KPW * kq) * (TERMS * 64) + lane;
#pragma unroll
        for (int f = 0; f < 9 * TERMS; ++f) breg[f] = wf[f * 64];
    }
    int aoff[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int s = KPW * kq + i, tap = s / SPT, kh = tap / 3, kw = tap - 3 * kh;
        aoff[i] = kh * g.RS + (kw & 1) * g.PL + (kw >> 1) * g.PXB + 32 * (s % SPT);
    }
    const int oh0 = px >> OWL, ow0 = px & (OW - 1);
    const int a_base = oh0 * 2 * g.RS + ow0 * g.PXB + 16 * kh2;
    const int rt1_off = (32 >> OWL) * 2 * g.RS;
    const int n_base = blockIdx.y * 64 + ct * 32;
    const int n_own = n_base + 8 * ((RPO * kq) >> 2) + 4 * kh2 + ((RPO * kq) & 3);
    float e_sc[RPO], e_sh[RPO];
#pragma unroll
    for (int r = 0; r < RPO; ++r) { e_sc[r] = n_own < p.N ? p.scale[n_own + r] : 0.0f; e_sh[r] = n_own < p.N ? p.shift[n_own + r] : 0.0f; }
    // this thread's input units (the same for every tile): unit tid + 512 u = (patch row y incl. the halo row 0, column x, channel group)
    int in_dst[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = min(tid + 512 * u, NE - 1), c4 = i % CQ, pix = i / CQ, y = pix / WW, x = pix % WW;
        in_dst[u] = y * g.RS + ((x + 1) & 1) * g.PL + ((x + 1) >> 1) * g.PXB + UB * c4;
    }
    float4 v[NV];
    // a tile's input rows 2 TH ti - 1 .. 2 TH ti + 2 TH - 1 are contiguous in the NHWC tensor; row -1 of an image's first tile is padding
#define AMPH_G5_FETCH(tile_)                                                               \
    {                                                                                      \
        const int img_ = (tile_) / g.tpi, ti_ = (tile_) - img_ * g.tpi;                    \
        const long long base_ = ((long long)img_ * g.H + 2 * TH * ti_ - 1) * (WW * CQ);    \
        const float4* src_ = reinterpret_cast<const float4*>(p.x);                         \
        _Pragma("unroll") for (int u = 0; u < NV; ++u) {                                   \
            const int i_ = tid + 512 * u;                                                  \
            const bool ok_ = i_ < NE && (ti_ > 0 || i_ >= WW * CQ);                        \
            v[u] = ok_ ? src_[base_ + i_] : make_float4(0.f, 0.f, 0.f, 0.f);               \
        }                                                                                  \
    }
    int tile = blockIdx.x;
    if (tile < g.n_tiles) AMPH_G5_FETCH(tile)
    // the left zero column (column 0 of every patch row): written once, restored behind the inbox
#define AMPH_G5_ZERO_COL()                                                                 \
    {                                                                                      \
        constexpr int CH = TERMS * TERM_B / 16;                                            \
        for (int i = tid; i < PR * CH; i += 512) {                                         \
            const int row = i / CH, ch = i - row * CH;                                     \
            *reinterpret_cast<uint4*>(patch + row * g.RS + 16 * ch) = make_uint4(0u, 0u, 0u, 0u); \
        }                                                                                  \
    }
    AMPH_G5_ZERO_COL()
    float4* const inbox = reinterpret_cast<float4*>(patch);
    bool bad = false;                                                    // an output the next layer's fp16 split cannot hold (amph_internal.h kH2Max)
    for (; tile < g.n_tiles; tile += gridDim.x) {
        // ---- (1) the tile's input into the patch (fp32: split in two fp16 terms; bf16: copied)
        if (TERMS == 1) {
#pragma unroll
            for (int u = 0; u < NV; ++u) if (tid + 512 * u < NE) *reinterpret_cast<float4*>(patch + in_dst[u]) = v[u];
        } else {
#pragma unroll
            for (int u = 0; u < NV; ++u) {
                uint32_t h0, l0, h1, l1;
                g4_split2(v[u].x, v[u].y, h0, l0);
                g4_split2(v[u].z, v[u].w, h1, l1);
                if (tid + 512 * u < NE) {
                    char* dst = patch + in_dst[u];
                    *reinterpret_cast<uint2*>(dst) = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(dst + TERM_B) = make_uint2(l0, l1);
                }
            }
        }
        __syncthreads();
        if (tile + (int)gridDim.x < g.n_tiles) AMPH_G5_FETCH(tile + gridDim.x)      // lands under the multiply
        // ---- (2) multiply: 9 k-steps, fragments of the next one read while this one is multiplied
        typedef typename G4Frag<TERMS>::type frag_t;
        f32x16 acc[RT], accl[RT];
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; accl[t][r] = 0.0f; }
#define AMPH_G5_FR(ptr_) __builtin_bit_cast(frag_t, *reinterpret_cast<const uint4*>(ptr_))
        frag_t fc[RT][TERMS], fn[RT][TERMS];
        {
            const char* pa = patch + a_base + aoff[0];
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int q = 0; q < TERMS; ++q) fc[t][q] = AMPH_G5_FR(pa + t * rt1_off + q * TERM_B);
        }
#pragma unroll
        for (int i = 0; i < 9; ++i) {
            if (i < 8) {
                const char* pa = patch + a_base + aoff[i + 1];
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int q = 0; q < TERMS; ++q) fn[t][q] = AMPH_G5_FR(pa + t * rt1_off + q * TERM_B);
            }
            const frag_t wh = __builtin_bit_cast(frag_t, breg[TERMS * i]), wl = __builtin_bit_cast(frag_t, breg[TERMS * i + TERMS - 1]);
#define AMPH_G5_M(a_, w_, q_)                                                              \
            _Pragma("unroll") for (int t = 0; t < RT; ++t) a_[t] = g4_mfma(w_, fc[t][q_], a_[t]);
            if (TERMS == 2) { AMPH_G5_M(accl, wl, 0) AMPH_G5_M(accl, wh, TERMS - 1) }
            AMPH_G5_M(acc, wh, 0)
#undef AMPH_G5_M
            if (i < 8) {
#pragma unroll
                for (int t = 0; t < RT; ++t)
#pragma unroll
                    for (int q = 0; q < TERMS; ++q) fc[t][q] = fn[t][q];
            }
        }
#undef AMPH_G5_FR
        if (TERMS == 2) {
#pragma unroll
            for (int t = 0; t < RT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = fmaf(accl[t][r], kG4Lo, acc[t][r]);
        }
        __syncthreads();                                                 // every wave is done with the patch: it becomes the inbox
        // ---- (3) K quarters -> their owners (k_conv_gemm_g4's exchange: part 0 first, a fixed order)
#pragma unroll
        for (int t = 0; t < RT; ++t)
#pragma unroll
            for (int o = 0; o < KW; ++o) {
                if (o == kq) continue;
                const int ow_ = 2 * o + ct;
                float* dst = reinterpret_cast<float*>(inbox) + ((((size_t)ow_ * RT + t) * KW + kq) * 64 + lane) * RPO;
                *reinterpret_cast<float4*>(dst) = make_float4(acc[t][RPO * o], acc[t][RPO * o + 1], acc[t][RPO * o + 2], acc[t][RPO * o + 3]);
            }
        __syncthreads();
#pragma unroll
        for (int t = 0; t < RT; ++t) {
            float sum[RPO];
#pragma unroll
            for (int r = 0; r < RPO; ++r) sum[r] = 0.0f;
#pragma unroll
            for (int sdr = 0; sdr < KW; ++sdr) {
                if (sdr == kq) {
#pragma unroll
                    for (int r = 0; r < RPO; ++r) sum[r] += acc[t][RPO * kq + r];
                } else {
                    const float4 q4 = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(inbox) + ((((size_t)wave * RT + t) * KW + sdr) * 64 + lane) * RPO);
                    sum[0] += q4.x; sum[1] += q4.y; sum[2] += q4.z; sum[3] += q4.w;
                }
            }
            const int m = tile * 64 + 32 * t + px;                       // a tile = 64 consecutive output pixels (TH whole rows)
            if (n_own < p.N && m < p.M) {
#pragma unroll
                for (int r = 0; r < RPO; ++r) {
                    const float t_ = fmaf(sum[r], e_sc[r], e_sh[r]);
                    if (TERMS != 1) bad |= h2_unsplittable(t_);
                    sum[r] = fmaxf(t_, 0.0f);
                }
                if (TERMS == 1) {
                    uint16_t* yo = reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.N + n_own;
                    const amph_bf16x2 lo_ = {(__bf16)sum[0], (__bf16)sum[1]}, hi_ = {(__bf16)sum[2], (__bf16)sum[3]};
                    *reinterpret_cast<uint2*>(yo) = make_uint2(__builtin_bit_cast(uint32_t, lo_), __builtin_bit_cast(uint32_t, hi_));
                } else *reinterpret_cast<float4*>(p.y + (long long)m * p.N + n_own) = make_float4(sum[0], sum[1], sum[2], sum[3]);
            }
        }
        __syncthreads();                                                 // the inbox is read: the next tile's patch may be written
        AMPH_G5_ZERO_COL()                                               // (overwritten by the inbox)
    }
    if (bad && p.range_flag) *p.range_flag = 1;
#undef AMPH_G5_FETCH
#undef AMPH_G5_ZERO_COL
}

template <int OWL, int TERMS>
static G5Geo g5_geo(const ConvGemmParams& p) {
    constexpr int OW = 1 << OWL, TH = 64 >> OWL, PR = 2 * TH + 1;
    G5Geo g{};
    g.PXB = TERMS * 128 + 16;
    g.PL = (OW + 1) * g.PXB;
    // row stride: the 16 lanes of every ds_read_b128 group (lane -> output row, column of its 32-pixel row tile) on 16 different 16-byte
    // slots - searched as in g4_geo
    static const int groups[2][16] = {{0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27}, {4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31}};
    const int rs0 = (2 * OW + 1) * g.PXB;
    g.RS = rs0;
    for (int rs = rs0; rs < rs0 + 512; rs += 16) {
        bool ok = true;
        for (int gi = 0; gi < 2 && ok; ++gi) {
            unsigned used = 0;
            for (int k = 0; k < 16; ++k) {
                const int px = groups[gi][k], oh = px >> OWL, ow = px & (OW - 1);
                const int slot = ((oh * 2 * rs + ow * g.PXB) >> 4) & 15;
                if (used & (1u << slot)) { ok = false; break; }
                used |= 1u << slot;
            }
        }
        if (ok) { g.RS = rs; break; }
    }
    g.patch_bytes = PR * g.RS;
    g.n_tiles = p.M / 64;
    g.tpi = p.OH / TH;
    g.H = p.H;
    return g;
}

// 0: not covered; 3 / 4: k_conv_gemm_g4's two shapes; 5 / 6: k_conv_gemm_g5 with 8- / 16-wide output rows (64 input channels, whole
// rows per 64-pixel tile, several tiles per image)
static int conv_gemm_g5_owl(const ConvGemmParams& p) {
    if (!(p.k == 3 && p.stride == 2 && p.pad == 1 && p.C_in == 64 && p.K == 576 && p.K_pad == 576 && p.splits == 1 && p.epilogue == 0 && !p.img && p.M > 0))
        return 0;
    if (p.W != 2 * p.OW || p.H != 2 * p.OH || (p.N % 64) || (p.M % 64)) return 0;
    if (p.OW == 8 && (p.OH % 8) == 0) return 3;
    if (p.OW == 16 && (p.OH % 4) == 0) return 4;
    return 0;
}

// 0: not covered; 3: conv3's shape (16 x 16 x 64 -> 8 x 8 x 128k); 4: conv4's (8 x 8 x 128 -> 4 x 4 x 64k); 7: conv4 of the 128-band geometry
// (12 x 16 x 128 -> 6 x 8 x 64k), single-term (bf16) operands only: two images of two terms would not fit LDS
int conv_gemm_g4_kind(const ConvGemmParams& p) {
    if (!(p.k == 3 && p.stride == 2 && p.pad == 1 && p.K == 9 * p.C_in && p.K_pad == p.K && p.splits == 1 && p.epilogue == 0 && !p.img && p.M > 0))
        return 0;
    if (p.C_in == 64 && p.OW == 8 && p.OH == 8 && p.W == 16 && p.H == 16 && (p.N % 64) == 0 && (p.M % 64) == 0) return 3;
    if (p.C_in == 128 && p.OW == 4 && p.OH == 4 && p.W == 8 && p.H == 8 && (p.N % 64) == 0 && (p.M % 16) == 0) return 4;
    if (p.C_in == 128 && p.OW == 8 && p.OH == 6 && p.W == 16 && p.H == 12 && (p.N % 64) == 0 && (p.M % 48) == 0) return 7;      // bf16 operands only (LDS)
    const int owl = conv_gemm_g5_owl(p);
    if (owl) return owl + 2;                                             // 5: OW 8, 6: OW 16 (k_conv_gemm_g5)
    return 0;
}

// wfrag = conv_gemm_h2_fragment_weights' (TERMS 2: fp32 in / out) or conv_gemm_f4b_fragment_weights' (TERMS 1: bf16 in / out) output of the layer
template <int TERMS>
static void launch_g4_t(const ConvGemmParams& p, const void* wfrag, int wgs, hipStream_t s) {
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_g4<64, 4, 2, 2, 1, 3, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_g4<128, 4, 2, 1, 2, 2, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        raised[dev] = true;
    }
    const int kind = conv_gemm_g4_kind(p);
    // workgroups: four tiles each amortise the 221 KB of weights a workgroup loads (one per CU: 57 MB through L2 per launch), half the
    // CUs at most - the other lanes' kernels fill the rest (tools/sweep.py patch_gemm_wgs=...: 64..128 workgroups 2.48 M chunks/s, 256: 2.47 M)
    if (kind == 3) {
        const G4Geo g = g4_geo<64, 1, 3, TERMS>(p);
        const int ny = p.N / 64;
        // (the two-term fp32 form: eight tiles each on a quarter of the CUs - see launch_conv_gemm_f4s)
        if (wgs <= 0) wgs = TERMS == 2 ? std::min(g_num_cus / 4, (g.n_tiles * ny + 7) / 8) : std::min(g_num_cus / 2, (g.n_tiles * ny + 3) / 4);
        const int per = std::max(1, wgs / ny);
        hipLaunchKernelGGL((k_conv_gemm_g4<64, 4, 2, 2, 1, 3, TERMS>), dim3((unsigned)std::min(g.n_tiles, per), (unsigned)ny), dim3(512), (size_t)std::max(g.patch_bytes, 64 * 1024), s, p, g,
                           reinterpret_cast<const uint4*>(wfrag));
    } else if (kind == 4) {
        const G4Geo g = g4_geo<128, 2, 2, TERMS>(p);
        // 64 channels per workgroup, K in four parts of 18 k-steps (round 4; the 32-channel / eight-part form it replaces: 28.9 against 24.8 us per
        // 256 chunks on 128 workgroups - seven senders per owner in the K-part exchange and 8-byte output stores for half the multiply)
        const int ny = p.N / 64;
        if (wgs <= 0) wgs = std::min(g_num_cus / 2, (g.n_tiles * ny + 3) / 4);
        const int per = std::max(1, wgs / ny);
        hipLaunchKernelGGL((k_conv_gemm_g4<128, 4, 2, 1, 2, 2, TERMS>), dim3((unsigned)std::min(g.n_tiles, per), (unsigned)ny), dim3(512), (size_t)std::max(g.patch_bytes, 32 * 1024), s, p, g,
                           reinterpret_cast<const uint4*>(wfrag));
    } else if (kind == 7) {
        if (TERMS != 1) return;                                          // (the caller checks: conv_gemm_g4_kind's comment)
        static bool raised7[64] = {false};
        if (dev >= 0 && dev < 64 && !raised7[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_g4<128, 4, 2, 3, 2, 3, 1, 6>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised7[dev] = true;
        }
        const G4Geo g = g4_geo<128, 2, 3, 1, 6>(p);
        const int ny = p.N / 64;
        if (wgs <= 0) wgs = std::min(g_num_cus / 2, (g.n_tiles * ny + 3) / 4);
        const int per = std::max(1, wgs / ny);
        hipLaunchKernelGGL((k_conv_gemm_g4<128, 4, 2, 3, 2, 3, 1, 6>), dim3((unsigned)std::min(g.n_tiles, per), (unsigned)ny), dim3(512), (size_t)std::max(g.patch_bytes, 96 * 1024), s, p, g,
                           reinterpret_cast<const uint4*>(wfrag));
    } else if (kind == 5 || kind == 6) {
        static bool raised5[64] = {false};
        if (dev >= 0 && dev < 64 && !raised5[dev]) {
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_g5<3, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_g5<4, TERMS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
            raised5[dev] = true;
        }
        const int ny = p.N / 64;
        const G5Geo g = kind == 5 ? g5_geo<3, TERMS>(p) : g5_geo<4, TERMS>(p);
        if (wgs <= 0) wgs = std::min(g_num_cus, (g.n_tiles * ny + 3) / 4);
        const int per = std::max(1, wgs / ny);
        const dim3 grid((unsigned)std::min(g.n_tiles, per), (unsigned)ny);
        const size_t lds = (size_t)std::max(g.patch_bytes, 64 * 1024);
        if (kind == 5) hipLaunchKernelGGL((k_conv_gemm_g5<3, TERMS>), grid, dim3(512), lds, s, p, g, reinterpret_cast<const uint4*>(wfrag));
        else hipLaunchKernelGGL((k_conv_gemm_g5<4, TERMS>), grid, dim3(512), lds, s, p, g, reinterpret_cast<const uint4*>(wfrag));
    }
}
// terms 1: bf16 encoder (wfrag = conv_gemm_f4b_fragment_weights' output); otherwise fp32 layers on the two-term fp16 split
// (wfrag = conv_gemm_h2_fragment_weights' output)
void launch_conv_gemm_g4(const ConvGemmParams& p, const void* wfrag, int wgs, int terms, hipStream_t s) {
    if (terms == 1) launch_g4_t<1>(p, wfrag, wgs, s);
    else launch_g4_t<2>(p, wfrag, wgs, s);
}

// two-term fp16 weights in MFMA fragment order: [n tile][channel tile][k-step (K_pad / 16)][term][lane][8 fp16];
// lane l holds row (l & 31), k = 16 kstep + 8 (l >> 5) + j
void conv_gemm_h2_fragment_weights(const float* w, int n_pad, int K_pad, uint16_t* out) {
    const int ksteps = K_pad / 16;
    for (int nt = 0; nt < n_pad / 64; ++nt)
        for (int ct = 0; ct < 2; ++ct)
            for (int ks = 0; ks < ksteps; ++ks)
                for (int lane = 0; lane < 64; ++lane)
                    for (int j = 0; j < 8; ++j) {
                        const int n = nt * 64 + ct * 32 + (lane & 31), k = 16 * ks + 8 * (lane >> 5) + j;
                        const float x = w[(size_t)n * K_pad + k];
                        const _Float16 h = (_Float16)x;
                        const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
                        uint16_t hb, lb;
                        std::memcpy(&hb, &h, 2);
                        std::memcpy(&lb, &l, 2);
                        const size_t base = (((size_t)(nt * 2 + ct) * ksteps + ks) * 2) * 64 * 8;
                        out[base + (0 * 64 + lane) * 8 + j] = hb;
                        out[base + (1 * 64 + lane) * 8 + j] = lb;
                    }
}

}  // namespace amph
