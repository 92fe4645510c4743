// fp32-accurate implicit-GEMM convolution on the 16-bit matrix cores ("s3": split operands).
//
// Replaces, like k_conv_gemm_p3, the conv2d + BatchNorm(eval) + ReLU layers 3.. and the dense head of
// `encoder(x)` at latent_space_exploration/map_detector_core.py:270 (module built at :175-179).
//
// Why: v_mfma_f32_32x32x2_f32 runs at the fp32 VECTOR rate (64 cycles per 32x32x2 = 157 TF chip-wide) and time-shares
// the SIMD with every VALU instruction beside it (tools/probe/mfma_valu_*.hip): the fp32 GEMM layers sit at 0.55-0.65
// of a peak that is itself 1/16 of the 16-bit one.  An fp32 operand is split into 16-bit terms whose products are exact
// in fp32 and the terms' products are accumulated in fp32 by 16-bit MFMAs:
//   round 2: three bf16 terms, x = h + m + l (8 + 8 + 8 significand bits), x w = hh + hm + mh + mm + hl + lh: SIX
//            v_mfma_f32_32x32x16_bf16 per K = 16 (ml, lm, ll below 2^-23 |x w| dropped).  Still the form of k_conv_gemm_f3s below.
//   round 3: two fp16 terms, x = h + 2^-11 l with h = fp16(x), l = fp16((x - h) 2^11) (11 + 11 bits; x - h is exact in fp32),
//            x w = h h' + 2^-11 (h l' + l h'): THREE v_mfma_f32_32x32x16_f16 per K = 16, h h' into one accumulator, the
//            cross terms into a second that is scaled once in the epilogue (2^-22 l l' dropped).  k_conv_gemm_s3 / _s3w here,
//            k_conv_gemm_f4s / _g4 / _g5 in k_encoder_patch.hip.  What the split itself loses against the float64 product of the
//            same operands: 1.8e-7 relative, three bf16 terms 1.3e-8, a plain fp32 matmul's rounding 5.9e-7
//            (tools/split_schemes_model.py); against the torch-CPU fp32 encoder both forms measure 6e-7 .. 1e-6 (tests).
//            Half the MFMAs, two thirds of the LDS bytes and of the staging stores: conv4 25.8 -> 21.4 us, conv3 (g4) 28.4 -> 20.2.
//            fp16's range (65 504) holds the encoder's operands - z-scored images, BN + ReLU activations, weights - with room;
//            below 6e-5 h is subnormal and l keeps the value to an absolute 3e-11.
// The result differs from an fp32 fma chain by the dropped terms and by the accumulation order inside a 16-deep MFMA.
// Order is fixed (slice, k-step, term pair): results do not depend on batch size, lanes or tile placement.
//
// Data: weights are split ONCE on the host into [n tile][slice][64 n][term][32 k] (64 LDS row images, contiguous);
// activations stay fp32 NHWC in HBM and the A tile is split on the way from the staging registers into LDS.  What bounds the
// kernel at a 256-chunk batch is data movement, not the matrix pipe (in-kernel stamps of the three-term form,
// profiles/r02_s3_stamps.txt: ~1 400 cycles per slice for 768 cycles of MFMA time on a SIMD's two waves); larger tiles halve the
// bytes per output (k_conv_gemm_s3w below) at the price of fewer workgroups.  Tried and dropped: activations handed over
// pre-split (written by the producer's epilogue: A staging a pure copy) - more A bytes and epilogue stores cost more than the
// VALU work they removed; four register stages instead of three (no change: the loop is not latency-bound).
// LDS rows are packed (128 B: [h | l] x 32 fp16) with XOR-swizzled 16-byte slots: fragment reads and staging stores are
// conflict-free.  Geometry, addressing (buffer loads, zero guard for padded taps, scalar tap state) and the
// transposed-accumulator epilogue are k_conv_gemm_p3's.
#include "amph_internal.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace amph {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 amph_h2 __attribute__((ext_vector_type(2)));
constexpr float kH2Lo = 1.0f / 2048.0f;          // weight of the cross-term accumulator
// x -> (h, l) pairs of two values: h = fp16(x) (RNE), l = fp16((x - h) 2^11) - x - h is exact in fp32
// The low terms of a pair, l = fp16((x - h) 2^11), with one v_fma_mix per value: fma(h, -2^11, 2^11 x) is exact in fp32 (x - h is), so
// the conversion to fp16 is the only rounding - the same bits as subtract, multiply, convert (tools/probe/h2_split_mix.hip: 2^24
// pairs), five instructions per pair instead of ten (the compiler converts h back to fp32 per value for the subtraction).
__device__ __forceinline__ uint32_t h2_lo_terms(uint32_t h, float t0, float t1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "s"(-2048.0f), "v"(t0), "v"(t1));
    return l;
}
__device__ __forceinline__ void h2_split2(float x0, float x1, uint32_t& h, uint32_t& l) {
    const amph_h2 hp = {(_Float16)x0, (_Float16)x1};
    h = __builtin_bit_cast(uint32_t, hp);
    l = h2_lo_terms(h, x0 * 2048.0f, x1 * 2048.0f);
}

constexpr int kSBK = 32;                 // K elements per slice (one tap's 32-channel group)
constexpr int kSWRow = 192;              // bytes per (n, slice) in the split weight matrix
// Two-term fp16 split (k_conv_gemm_s3 / _s3w since round 3): x = h + 2^-11 l with h = fp16(x), l = fp16((x - h) 2^11) - 22 significand
// bits in TWO terms (bf16 needs three for 24), both products of 11-bit values exact in fp32.  x w = h h' + 2^-11 (h l' + l h') (+ 2^-22 l l':
// dropped) is THREE v_mfma_f32_32x32x16_f16 per K = 16 instead of six: h h' goes to one accumulator, the cross terms to a second that is
// scaled once in the epilogue.  Measured against the float64 product of the same operands: 1.8e-7 relative (bf16 x 3: 1.3e-8, a plain fp32
// matmul 5.9e-7 - tools/split_schemes_model.py), i.e. still below the rounding noise of fp32 accumulation.  fp16's range holds the encoder's
// operands with room (|x| < 65 504; below 6e-5 the split keeps an absolute 3e-11).
constexpr int kHRow = 128;               // LDS bytes per row: [h | l] x 32 fp16; 16-byte slot j of row r at 16 (j ^ h_sw(r))
constexpr int kHBuf = 64 * kHRow;        // bytes per 64-row operand tile per slice buffer
constexpr int kHWRow = 128;              // bytes per (n, slice) in the two-term weight matrix

// Slot swizzle of the slice buffers (tools/lds_swizzle_model.py): a ds_read_b128 lane group holds 8 even and 8 odd rows (row parity =
// half of the 256-byte bank row), so the 8 slots of the rows of one parity must differ - three bits from r >> 1; a ds_write_b64 group
// (16 lanes = rows r, r + 1 x 64 bytes of one term, banks mod 32) needs the two rows' terms in opposite halves - bit 2 from r & 1.
// (Until r03d: (r >> 2) & 3, two bits - every fragment read and A store 2-way conflicted, 43 % of the kernel's LDS cycles.)
__device__ __forceinline__ int h_sw(int r) { return ((r >> 1) & 7) ^ ((r & 1) << 2); }

__device__ __forceinline__ int s3_div_small(int n, int d, float rcp) {
    int q = (int)((float)n * rcp);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) ++q;
    return q;
}

#define AMPH_S3_SB() __builtin_amdgcn_sched_barrier(0)

#ifdef AMPH_STAMPS
#define AMPH_S3_T(v_)                                                                      \
    {                                                                                      \
        __builtin_amdgcn_sched_barrier(0);                                                 \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");         \
        __builtin_amdgcn_sched_barrier(0);                                                 \
    }
#else
#define AMPH_S3_T(v_)
#endif

__global__ __launch_bounds__(256, 3) void k_conv_gemm_s3(ConvGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
#ifdef AMPH_STAMPS
    unsigned long long q_t0 = 0, q_t1 = 0, q_t2 = 0, q_t3 = 0, q_ta = 0, q_tb = 0;
    const unsigned long long q_rt0 = __builtin_amdgcn_s_memrealtime();
    AMPH_S3_T(q_t0)
#endif
    char* As = smem;                                   // [3][64][128]
    char* Bs = smem + 3 * kHBuf;                       // [3][64][128]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;             // A staging: rows r0, r0 + 32; k = 4 c4 .. 4 c4 + 3 of the slice
    // B staging: the (n tile, slice) block of the split weight matrix is 64 rows x 192 B CONTIGUOUS in global memory; load i of
    // thread t takes its 16 bytes at (256 i + t) * 16: every wave instruction reads 1 KB of whole cache lines.  (Per-row
    // addressing - 4 threads x 3 loads per 192-byte row - made every instruction touch 16 rows x 2-3 lines for a third of
    // their bytes: the layers were bound by line processing in the vector cache, 22.8 us for 6.6 us of MFMA time.)
    const int m_blk = blockIdx.x * 64, n_blk = blockIdx.y * 64;
    const int n_slices = p.K_pad / kSBK;
    const int sps = n_slices / p.splits;
    const int ks_begin = blockIdx.z * sps, ks_end = ks_begin + sps;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) - p.x_guard), 0, -1, 0x00020000);
#define AMPH_S3_BUFLOAD(rs_, voff_, soff_) __builtin_amdgcn_raw_buffer_load_b128(rs_, voff_, soff_, 0)
    // register stages (plain named variables, see k_conv_gemm_p3)
    // FOUR stages: the loads of slice s + 4 are issued in iteration s and consumed in iteration s + 2.  The layers are bound by
    // operand delivery from L2 (20 KB per workgroup and slice against 384 cycles of MFMA time per wave), i.e. by the bytes a
    // CU keeps in flight: with three stages (one slice in flight per workgroup) conv3 took 22.8 us against 6.6 us of MFMA time
    float4 sa0x, sa0y, sa1x, sa1y, sa2x, sa2y, sa3x, sa3y;             // rows r0, r0 + 32
    uint4 sb0a, sb0b, sb1a, sb1b, sb2a, sb2b, sb3a, sb3b;
    int w_soff = (blockIdx.y * n_slices + ks_begin) * (64 * kHWRow);
    const unsigned woff = 16u * (unsigned)tid;
#define AMPH_S3_LOADB(q)                                                                   \
    {                                                                                      \
        sb##q##a = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff, w_soff));          \
        sb##q##b = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff + 4096u, w_soff));  \
    }

    // the B halves of the four prologue slices are requested before any setup arithmetic (slice min(s, sps - 1), as ADV clamps)
    {
        const int blk = 64 * kHWRow;
        const int o1 = sps > 1 ? blk : 0, o2 = sps > 2 ? 2 * blk : o1, o3 = sps > 3 ? 3 * blk : o2;
        AMPH_S3_LOADB(0)
        w_soff += o1; AMPH_S3_LOADB(1)
        w_soff += o2 - o1; AMPH_S3_LOADB(2)
        w_soff += o3 - o2; AMPH_S3_LOADB(3)
        w_soff -= o3;
    }
    const int ohw = p.OH * p.OW;
    const float rcp_ohw = 1.0f / (float)ohw, rcp_ow = 1.0f / (float)p.OW;
    constexpr int kEl = 4;                              // bytes per activation element in HBM
    unsigned aoff0, aoff1, vmask0, vmask1;
#define AMPH_S3_ROW(i)                                                                     \
    {                                                                                      \
        const int m = m_blk + r0 + 32 * i;                                                 \
        const bool mv = m < p.M;                                                           \
        const int mm = mv ? m : 0;                                                         \
        const int b = s3_div_small(mm, ohw, rcp_ohw), rem = mm - b * ohw;                  \
        const int oh = s3_div_small(rem, p.OW, rcp_ow), ow = rem - oh * p.OW;              \
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;                \
        aoff##i = (unsigned)p.x_guard + (unsigned)kEl * (unsigned)(((b * p.H + ih0) * p.W + iw0) * p.C_in) + 16u * (unsigned)c4; \
        unsigned vm = 0;                                                                   \
        if (p.k == 3) {                                                                    \
            const unsigned hb = ((unsigned)iw0 < (unsigned)p.W ? 0x049u : 0u) | ((unsigned)(iw0 + 1) < (unsigned)p.W ? 0x092u : 0u) | \
                                ((unsigned)(iw0 + 2) < (unsigned)p.W ? 0x124u : 0u);       \
            const unsigned rb_ = ((unsigned)ih0 < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ih0 + 1) < (unsigned)p.H ? 0x038u : 0u) | \
                                 ((unsigned)(ih0 + 2) < (unsigned)p.H ? 0x1c0u : 0u);      \
            vm = hb & rb_;                                                                 \
        } else {                                                                           \
            for (int kh = 0; kh < p.k; ++kh)                                               \
                for (int kw = 0; kw < p.k; ++kw)                                           \
                    if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) vm |= 1u << (kh * p.k + kw); \
        }                                                                                  \
        vmask##i = mv ? vm : 0u;                                                           \
    }
    AMPH_S3_ROW(0)
    AMPH_S3_ROW(1)
#undef AMPH_S3_ROW

    int l_left = sps;
    int l_ci0, l_kw, x_soff;
    unsigned l_bit;
    {
        const int kpos = (ks_begin * kSBK) / p.C_in;
        l_ci0 = ks_begin * kSBK - kpos * p.C_in;
        const int kh = kpos / p.k;
        l_kw = kpos - kh * p.k;
        x_soff = kEl * ((kh * p.W + l_kw) * p.C_in + l_ci0);
        l_bit = 1u << kpos;
    }
    const int row_skip = kEl * (p.W - p.k) * p.C_in;
#define AMPH_S3_LOADA(q)                                                                   \
    {                                                                                      \
        sa##q##x = __builtin_bit_cast(float4, AMPH_S3_BUFLOAD(rs_x, (vmask0 & l_bit) ? aoff0 : 0u, x_soff)); \
        sa##q##y = __builtin_bit_cast(float4, AMPH_S3_BUFLOAD(rs_x, (vmask1 & l_bit) ? aoff1 : 0u, x_soff)); \
    }
#define AMPH_S3_ADV()                                                                      \
    {                                                                                      \
        const int adv_ = l_left > 1 ? 1 : 0;                                               \
        l_left -= adv_;                                                                    \
        w_soff = __builtin_amdgcn_readfirstlane(w_soff + 64 * kHWRow * adv_);              \
        x_soff += kSBK * kEl * adv_;                                                       \
        l_ci0 += kSBK * adv_;                                                              \
        const int wrap_ = l_ci0 >= p.C_in ? 1 : 0;                                         \
        l_ci0 -= wrap_ * p.C_in;                                                           \
        l_bit = __builtin_amdgcn_readfirstlane(l_bit << wrap_);                            \
        l_kw += wrap_;                                                                     \
        const int wrap2_ = l_kw >= p.k ? 1 : 0;                                            \
        l_kw -= wrap2_ * p.k;                                                              \
        x_soff = __builtin_amdgcn_readfirstlane(x_soff + wrap2_ * row_skip);               \
    }
#define AMPH_S3_LOAD(q) { AMPH_S3_LOADB(q) AMPH_S3_LOADA(q) AMPH_S3_ADV() }
#define AMPH_S3_LOADA_ADV(q) { AMPH_S3_LOADA(q) AMPH_S3_ADV() }

    // stage q -> LDS buffer buf_: B rows are the LDS image already; A rows are split here
    const int st_a = r0 * kHRow + 16 * ((c4 >> 1) ^ h_sw(r0)) + 8 * (c4 & 1);             // (row r0 + 32 has the same swizzle; the l term at st_a ^ 64)
    int st_b0, st_b1;                                  // byte o of the block -> LDS row o / 128, swizzled slot of column o % 128
    {
        const int o0 = 16 * tid, o1 = o0 + 4096;
#define AMPH_S3_SW(o_) (((o_) / kHWRow) * kHRow + 16 * ((((o_) % kHWRow) >> 4) ^ h_sw((o_) / kHWRow)))
        st_b0 = AMPH_S3_SW(o0); st_b1 = AMPH_S3_SW(o1);
#undef AMPH_S3_SW
    }
#define AMPH_S3_SPLIT_ROW(v_, base_, off_)                                                 \
    {                                                                                      \
        uint32_t h0, l0, h1, l1;                                                           \
        h2_split2(v_.x, v_.y, h0, l0);                                                     \
        h2_split2(v_.z, v_.w, h1, l1);                                                     \
        *reinterpret_cast<uint2*>((base_) + (off_)) = make_uint2(h0, h1);                  \
        *reinterpret_cast<uint2*>((base_) + ((off_) ^ 64)) = make_uint2(l0, l1);           \
    }
#define AMPH_S3_STORE(q, buf_)                                                             \
    {                                                                                      \
        char* a_ = As + (buf_) * kHBuf;                                                    \
        char* b_ = Bs + (buf_) * kHBuf;                                                    \
        *reinterpret_cast<uint4*>(b_ + st_b0) = sb##q##a;                                  \
        *reinterpret_cast<uint4*>(b_ + st_b1) = sb##q##b;                                  \
        AMPH_S3_SPLIT_ROW(sa##q##x, a_, st_a)                                              \
        AMPH_S3_SPLIT_ROW(sa##q##y, a_ + 32 * kHRow, st_a)                                 \
    }

    f32x16 acc, accl;                                  // h h' | h l' + l h' (x 2^-11 in the epilogue)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[r] = 0.0f; accl[r] = 0.0f; }
    const int row_tile = wave >> 1, col_tile = wave & 1;
    // BN constants of the tile's 64 channels wait in LDS for the epilogue (32 registers through the loop otherwise: with them the kernel
    // needed 198 and two workgroups shared a CU - conv4's 768 workgroups at a 256-chunk batch ran as one and a half rounds of 512)
    float* const bn_lds = reinterpret_cast<float*>(smem + 6 * kHBuf);   // [2][64]
    if (tid < 128) {
        const int n = n_blk + (tid & 63);
        bn_lds[tid] = (p.epilogue == 0 && n < p.N) ? (tid < 64 ? p.scale : p.shift)[n] : 0.0f;
    }
    // fragment addresses: lane l holds row (l & 31), k = 16 kstep + 8 (l >> 5) + j of each term: slot (4 t + 2 kstep + (l >> 5)) ^ h_sw(row)
    // of the row - one offset per (kstep, term), the same for both operands (32-row tiles: row & 15 is the lane's)
    const int swl = h_sw(lane & 31);
    const char* a_rdb = As + (row_tile * 32 + (lane & 31)) * kHRow;
    const char* b_rdb = Bs + (col_tile * 32 + (lane & 31)) * kHRow;
    const int rd_o00 = 16 * ((lane >> 5) ^ swl), rd_o01 = rd_o00 ^ 64, rd_o10 = 16 * ((2 + (lane >> 5)) ^ swl), rd_o11 = rd_o10 ^ 64;
#define AMPH_S3_FRAG(base_, buf_, ks_, t_) __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>((base_##b) + (buf_) * kHBuf + rd_o##ks_##t_))
    // weights as the FIRST operand: transposed accumulator (lane = output row m, registers 4g .. 4g + 3 = channels), see p3
#define AMPH_S3_MFMA(a_, w_, x_) a_ = __builtin_amdgcn_mfma_f32_32x32x16_f16(w_, x_, a_, 0, 0, 0);
#define AMPH_S3_READ6(pre_, buf_, ks_)                                                     \
    pre_##xh = AMPH_S3_FRAG(a_rd, buf_, ks_, 0); pre_##xl = AMPH_S3_FRAG(a_rd, buf_, ks_, 1); \
    pre_##wh = AMPH_S3_FRAG(b_rd, buf_, ks_, 0); pre_##wl = AMPH_S3_FRAG(b_rd, buf_, ks_, 1);
    // pieces of the staging work of one slice (stage q -> LDS buffer buf_), one per MFMA gap
#define AMPH_S3_ST_B(q, buf_)                                                              \
    {                                                                                      \
        char* b_ = Bs + (buf_) * kHBuf;                                                    \
        *reinterpret_cast<uint4*>(b_ + st_b0) = sb##q##a;                                  \
        *reinterpret_cast<uint4*>(b_ + st_b1) = sb##q##b;                                  \
    }
#define AMPH_S3_ST_AX(q, buf_) { char* a_ = As + (buf_) * kHBuf; AMPH_S3_SPLIT_ROW(sa##q##x, a_, st_a) }
#define AMPH_S3_ST_AY(q, buf_) { char* a_ = As + (buf_) * kHBuf + 32 * kHRow; AMPH_S3_SPLIT_ROW(sa##q##y, a_, st_a) }
#if defined(AMPH_S3_ABL_NOLOAD)            /* timing-only ablations (make ablations_s3; never shipped) */
#define AMPH_S3_LOOPLOAD(QL) AMPH_S3_ADV()
#else
#define AMPH_S3_LOOPLOAD(QL) AMPH_S3_LOAD(QL)
#endif
#if defined(AMPH_S3_ABL_NOMFMA)
#define AMPH_S3_M(a_, w_, x_) a_[0] += (float)w_[0] + (float)x_[1];
#else
#define AMPH_S3_M(a_, w_, x_) AMPH_S3_MFMA(a_, w_, x_)
#endif
#if defined(AMPH_S3_ABL_NOSTORE)
#define AMPH_S3_SB_(q, buf_) acc[1] += __uint_as_float(sb##q##a.x ^ sb##q##b.y);
#define AMPH_S3_SX_(q, buf_) acc[2] += sa##q##x.x;
#define AMPH_S3_SY_(q, buf_) acc[3] += sa##q##y.y;
#else
#define AMPH_S3_SB_(q, buf_) AMPH_S3_ST_B(q, buf_)
#define AMPH_S3_SX_(q, buf_) AMPH_S3_ST_AX(q, buf_)
#define AMPH_S3_SY_(q, buf_) AMPH_S3_ST_AY(q, buf_)
#endif
    // One slice from LDS buffer R.  On entry f0* holds its k-step-0 fragments (read before the previous barrier).  Stage QL
    // receives slice s + 4, stage QS (slice s + 2) goes to buffer (R + 2) % 3 in the gaps of the first six MFMAs; the k-step-1
    // fragments are read under those, the next slice's k-step-0 fragments (buffer (R + 1) % 3, published by the previous
    // barrier) under the last six: no MFMA waits for an LDS round trip it could not overlap.
#define AMPH_S3_STEP(R, QL, QS)                                                            \
    {                                                                                      \
        f16x8 f1xh, f1xl, f1wh, f1wl, n0xh, n0xl, n0wh, n0wl;                              \
        AMPH_S3_LOOPLOAD(QL)                                                               \
        AMPH_S3_READ6(f1, R, 1)                                                            \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_M(accl, f0wl, f0xh)                                                        \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_SB_(QS, (R + 2) % 3)                                                       \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_M(accl, f0wh, f0xl)                                                        \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_SX_(QS, (R + 2) % 3)                                                       \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_M(acc, f0wh, f0xh)                                                         \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_SY_(QS, (R + 2) % 3)                                                       \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_READ6(n0, (R + 1) % 3, 0)                                                  \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_M(accl, f1wl, f1xh)                                                        \
        AMPH_S3_M(accl, f1wh, f1xl)                                                        \
        AMPH_S3_M(acc, f1wh, f1xh)                                                         \
        AMPH_S3_SB();                                                                      \
        __syncthreads();                                                                   \
        f0xh = n0xh; f0xl = n0xl; f0wh = n0wh; f0wl = n0wl;                                \
    }

    // prologue: slices 0 .. 3 into the register stages, 0 and 1 on to LDS
    AMPH_S3_LOADA_ADV(0)
    AMPH_S3_LOADA_ADV(1)
    AMPH_S3_LOADA_ADV(2)
    AMPH_S3_LOADA_ADV(3)
    AMPH_S3_T(q_ta)
    AMPH_S3_STORE(0, 0)
    AMPH_S3_T(q_tb)
    AMPH_S3_STORE(1, 1)
    __syncthreads();
    f16x8 f0xh, f0xl, f0wh, f0wl;
    AMPH_S3_READ6(f0, 0, 0)
    AMPH_S3_T(q_t1)
    // step s uses LDS buffer s % 3 and register stages s % 4 (load) / (s + 2) % 4 (store): twelve steps per trip keep every
    // index a compile-time constant
#define AMPH_S3_STEPS_0_11(X)                                                              \
    X(0, 0, 0, 2) X(1, 1, 1, 3) X(2, 2, 2, 0) X(3, 0, 3, 1) X(4, 1, 0, 2) X(5, 2, 1, 3)   \
    X(6, 0, 2, 0) X(7, 1, 3, 1) X(8, 2, 0, 2) X(9, 0, 1, 3) X(10, 1, 2, 0) X(11, 2, 3, 1)
    int ks = ks_begin;
    for (; ks + 12 <= ks_end; ks += 12) {
#define AMPH_S3_X(i, R, QL, QS) AMPH_S3_STEP(R, QL, QS)
        AMPH_S3_STEPS_0_11(AMPH_S3_X)
#undef AMPH_S3_X
    }
    {
        const int rem = ks_end - ks;
#define AMPH_S3_X(i, R, QL, QS) if (rem > i) AMPH_S3_STEP(R, QL, QS)
        AMPH_S3_STEPS_0_11(AMPH_S3_X)
#undef AMPH_S3_X
    }

    AMPH_S3_T(q_t2)
    {
        const int m = m_blk + row_tile * 32 + (lane & 31);
        float* ybase = p.y + (p.epilogue == 1 ? (long long)blockIdx.z * p.M * p.N : 0ll);
        bool bad = false;                               // an output the next layer's fp16 split cannot hold (amph_internal.h kH2Max)
        if (m < p.M) {
#define AMPH_S3_OUT(g)                                                                     \
            {                                                                              \
                const int n0 = n_blk + col_tile * 32 + 8 * g + 4 * (lane >> 5);            \
                if (n0 < p.N) {                                                            \
                    float4 v = make_float4(fmaf(accl[4 * g], kH2Lo, acc[4 * g]), fmaf(accl[4 * g + 1], kH2Lo, acc[4 * g + 1]),       \
                                           fmaf(accl[4 * g + 2], kH2Lo, acc[4 * g + 2]), fmaf(accl[4 * g + 3], kH2Lo, acc[4 * g + 3])); \
                    if (p.epilogue == 0) {                                                 \
                        const float4 e_sc = *reinterpret_cast<const float4*>(bn_lds + n0 - n_blk), e_sh = *reinterpret_cast<const float4*>(bn_lds + 64 + n0 - n_blk); \
                        v.x = fmaf(v.x, e_sc.x, e_sh.x); v.y = fmaf(v.y, e_sc.y, e_sh.y); v.z = fmaf(v.z, e_sc.z, e_sh.z); v.w = fmaf(v.w, e_sc.w, e_sh.w); \
                        bad |= h2_unsplittable(v.x) | h2_unsplittable(v.y) | h2_unsplittable(v.z) | h2_unsplittable(v.w); \
                        v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f); \
                    }                                                                      \
                    *reinterpret_cast<float4*>(ybase + (long long)m * p.N + n0) = v;       \
                }                                                                          \
            }
            AMPH_S3_OUT(0) AMPH_S3_OUT(1) AMPH_S3_OUT(2) AMPH_S3_OUT(3)
#undef AMPH_S3_OUT
        }
        if (bad && p.range_flag) *p.range_flag = 1;
    }
#ifdef AMPH_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMPH_S3_T(q_t3)
    if (tid == 0 && p.stamps) {       // setup + issue of the prologue loads | wait for slice 0 (first global round trip) + split | rest of prologue | loop | epilogue
        unsigned long long* o = p.stamps + (size_t)(blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * 8;
        o[0] = q_ta - q_t0; o[1] = q_tb - q_ta; o[2] = q_t1 - q_tb; o[3] = q_t2 - q_t1; o[4] = q_t3 - q_t0;
        o[5] = __builtin_amdgcn_s_memrealtime() - q_rt0;
        o[6] = q_t3 - q_t2; o[7] = q_rt0;
    }
#endif
}

// ----------------------------------------------------------------------------
// k_conv_gemm_s3 on 128 x 128 tiles ("s3w", round 3): 512 threads, wave = (64-row half, 32-channel tile) with TWO accumulators.
//
// Why: with four free-running lanes a batch costs the SUM of what its kernels cost in CU-time (profiles/r03_stage_skip_sweep.txt:
// leaving a stage out shortens the batch period by its duration x the share of the CUs it holds), so a GEMM layer is worth what
// it does per CU and second, not how soon it ends alone.  The 64 x 64 tile of s3 stages 20 KB and issues ~170 instructions per
// wave for twelve MFMAs; here the same per-thread staging work (two A float4 + their split, three B uint4) feeds 24 MFMAs per wave,
// a fragment set serves two row tiles, and the two waves of a SIMD overlap each other's staging: conv4 at a 256-chunk batch is 64
// workgroups (a quarter of the CUs) instead of 256.  Per accumulator the MFMA order is s3's (slice, k-step, term pair), so the
// results are bit-identical to s3's.  Weights: s3's split matrix; threads 0-255 stage 64-channel block 2 by, 256-511 block 2 by + 1.
// ----------------------------------------------------------------------------
constexpr int kWBuf = 128 * kHRow;       // bytes per operand per slice buffer (128 rows)
constexpr size_t kS3wLds = (size_t)6 * kWBuf + 1024;   // 99 328 B: slice buffers + the tile's BN constants

__global__ __launch_bounds__(512, 1) void k_conv_gemm_s3w(ConvGemmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                   // [3][128][128]
    char* Bs = smem + 3 * kWBuf;                       // [3][128][128]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c4 = tid & 7, r0 = tid >> 3;             // A staging: rows r0, r0 + 64; k = 4 c4 .. 4 c4 + 3 of the slice
    const int bh = wave >> 2, bt = tid & 255;          // B staging: 64-channel block 2 by + bh, 16-byte piece bt + 256 i of its 8 KB
    const int m_blk = blockIdx.x * 128, n_blk = blockIdx.y * 128;
    const int n_slices = p.K_pad / kSBK;

    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.x) - p.x_guard), 0, -1, 0x00020000);
    float4 sa[4][2];                                   // register stages (indices are literals everywhere: registers, never scratch)
    uint4 sb[4][2];
    int w_soff = __builtin_amdgcn_readfirstlane(((2 * (int)blockIdx.y + bh) * n_slices) * (64 * kHWRow));
    const unsigned woff = 16u * (unsigned)bt;
#define AMPH_W_LOADB(q)                                                                    \
    {                                                                                      \
        sb[q][0] = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff, w_soff));          \
        sb[q][1] = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff + 4096u, w_soff));  \
    }
    {
        const int blk = 64 * kHWRow;
        const int o1 = n_slices > 1 ? blk : 0, o2 = n_slices > 2 ? 2 * blk : o1, o3 = n_slices > 3 ? 3 * blk : o2;
        AMPH_W_LOADB(0)
        w_soff += o1; AMPH_W_LOADB(1)
        w_soff += o2 - o1; AMPH_W_LOADB(2)
        w_soff += o3 - o2; AMPH_W_LOADB(3)
        w_soff -= o3;
    }
    const int ohw = p.OH * p.OW;
    const float rcp_ohw = 1.0f / (float)ohw, rcp_ow = 1.0f / (float)p.OW;
    constexpr int kEl = 4;
    unsigned aoff[2], vmask[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int m = m_blk + r0 + 64 * i;
        const bool mv = m < p.M;
        const int mm = mv ? m : 0;
        const int b = s3_div_small(mm, ohw, rcp_ohw), rem = mm - b * ohw;
        const int oh = s3_div_small(rem, p.OW, rcp_ow), ow = rem - oh * p.OW;
        const int ih0 = oh * p.stride - p.pad, iw0 = ow * p.stride - p.pad;
        aoff[i] = (unsigned)p.x_guard + (unsigned)kEl * (unsigned)(((b * p.H + ih0) * p.W + iw0) * p.C_in) + 16u * (unsigned)c4;
        unsigned vm = 0;
        if (p.k == 3) {
            const unsigned hb = ((unsigned)iw0 < (unsigned)p.W ? 0x049u : 0u) | ((unsigned)(iw0 + 1) < (unsigned)p.W ? 0x092u : 0u) |
                                ((unsigned)(iw0 + 2) < (unsigned)p.W ? 0x124u : 0u);
            const unsigned rb_ = ((unsigned)ih0 < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ih0 + 1) < (unsigned)p.H ? 0x038u : 0u) |
                                 ((unsigned)(ih0 + 2) < (unsigned)p.H ? 0x1c0u : 0u);
            vm = hb & rb_;
        } else {
            for (int kh = 0; kh < p.k; ++kh)
                for (int kw = 0; kw < p.k; ++kw)
                    if ((unsigned)(ih0 + kh) < (unsigned)p.H && (unsigned)(iw0 + kw) < (unsigned)p.W) vm |= 1u << (kh * p.k + kw);
        }
        vmask[i] = mv ? vm : 0u;
    }
    int l_left = n_slices;
    int l_ci0 = 0, l_kw = 0, x_soff = 0;
    unsigned l_bit = 1u;
    const int row_skip = kEl * (p.W - p.k) * p.C_in;
#define AMPH_W_LOADA(q)                                                                    \
    {                                                                                      \
        sa[q][0] = __builtin_bit_cast(float4, AMPH_S3_BUFLOAD(rs_x, (vmask[0] & l_bit) ? aoff[0] : 0u, x_soff)); \
        sa[q][1] = __builtin_bit_cast(float4, AMPH_S3_BUFLOAD(rs_x, (vmask[1] & l_bit) ? aoff[1] : 0u, x_soff)); \
    }
#define AMPH_W_LOAD(q) { AMPH_W_LOADB(q) AMPH_W_LOADA(q) AMPH_S3_ADV() }
#define AMPH_W_LOADA_ADV(q) { AMPH_W_LOADA(q) AMPH_S3_ADV() }
    const int st_a = r0 * kHRow + 16 * ((c4 >> 1) ^ h_sw(r0)) + 8 * (c4 & 1);             // (row r0 + 64 has the same swizzle; the l term at st_a ^ 64)
    int st_b[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int o = 16 * bt + 4096 * i, row = o / kHWRow, col = o % kHWRow;
        st_b[i] = (64 * bh + row) * kHRow + 16 * ((col >> 4) ^ h_sw(row));
    }
#define AMPH_W_ST_B(q, buf_)                                                               \
    {                                                                                      \
        char* b_ = Bs + (buf_) * kWBuf;                                                    \
        *reinterpret_cast<uint4*>(b_ + st_b[0]) = sb[q][0];                                \
        *reinterpret_cast<uint4*>(b_ + st_b[1]) = sb[q][1];                                \
    }
#define AMPH_W_ST_A(q, buf_, i_) { char* a_ = As + (buf_) * kWBuf + (i_) * 64 * kHRow; AMPH_S3_SPLIT_ROW(sa[q][i_], a_, st_a) }

    f32x16 acc[2], accl[2];                            // per row tile: h h' | h l' + l h' (x 2^-11 in the epilogue)
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.0f; accl[t][r] = 0.0f; }
    const int rg = wave >> 2, ct = wave & 3;
    // BN constants of the tile's 128 channels wait in LDS for the epilogue (32 registers through the loop otherwise)
    float* const bn = reinterpret_cast<float*>(smem + 6 * kWBuf);       // [2][128]
    if (tid < 256) bn[tid] = (tid < 128 ? p.scale : p.shift)[n_blk + (tid & 127)];
    // fragment addresses (s3's): lane l holds row (l & 31), k = 16 kstep + 8 (l >> 5) + j of each term
    const int swl = h_sw(lane & 31);
    const char* a_rdb = As + (rg * 64 + (lane & 31)) * kHRow;
    const char* b_rdb = Bs + (ct * 32 + (lane & 31)) * kHRow;
    const int rd_o[2][2] = {{16 * ((lane >> 5) ^ swl), (16 * ((lane >> 5) ^ swl)) ^ 64},       // [kstep][term]
                            {16 * ((2 + (lane >> 5)) ^ swl), (16 * ((2 + (lane >> 5)) ^ swl)) ^ 64}};
    struct Frag { f16x8 x[2][2], w[2]; };                // [row tile][term], [term]
#define AMPH_W_FR(ptr_) __builtin_bit_cast(f16x8, *reinterpret_cast<const uint4*>(ptr_))
#define AMPH_W_READ9(f_, buf_, ks_)                                                        \
    {                                                                                      \
        _Pragma("unroll") for (int q_ = 0; q_ < 2; ++q_) {                                 \
            f_.x[0][q_] = AMPH_W_FR(a_rdb + (buf_) * kWBuf + rd_o[ks_][q_]);               \
            f_.x[1][q_] = AMPH_W_FR(a_rdb + (buf_) * kWBuf + 32 * kHRow + rd_o[ks_][q_]);  \
            f_.w[q_] = AMPH_W_FR(b_rdb + (buf_) * kWBuf + rd_o[ks_][q_]);                  \
        }                                                                                  \
    }
    // term pair (w term, x term) on both row tiles; weights first: transposed accumulator (lane = output row, registers = channels)
#define AMPH_W_M(a_, f_, wq_, xq_)                                                         \
    a_[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_.w[wq_], f_.x[0][xq_], a_[0], 0, 0, 0); \
    a_[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(f_.w[wq_], f_.x[1][xq_], a_[1], 0, 0, 0);
    // one slice from LDS buffer R (s3's step with two row tiles): stage QL receives slice s + 4, stage QS (slice s + 2) goes to buffer
    // (R + 2) % 3 in the MFMA gaps of k-step 0, the k-step-1 fragments are read under those, the next slice's k-step-0 fragments under k-step 1
#define AMPH_W_STEP(R, QL, QS)                                                             \
    {                                                                                      \
        Frag f1;                                                                           \
        AMPH_W_LOAD(QL)                                                                    \
        AMPH_W_READ9(f1, R, 1)                                                             \
        AMPH_S3_SB();                                                                      \
        AMPH_W_M(accl, f0, 1, 0)                                                           \
        AMPH_S3_SB();                                                                      \
        AMPH_W_ST_B(QS, (R + 2) % 3)                                                       \
        AMPH_S3_SB();                                                                      \
        AMPH_W_M(accl, f0, 0, 1)                                                           \
        AMPH_S3_SB();                                                                      \
        AMPH_W_ST_A(QS, (R + 2) % 3, 0)                                                    \
        AMPH_S3_SB();                                                                      \
        AMPH_W_M(acc, f0, 0, 0)                                                            \
        AMPH_S3_SB();                                                                      \
        AMPH_W_ST_A(QS, (R + 2) % 3, 1)                                                    \
        AMPH_S3_SB();                                                                      \
        AMPH_W_READ9(f0, (R + 1) % 3, 0)                                                   \
        AMPH_S3_SB();                                                                      \
        AMPH_W_M(accl, f1, 1, 0)                                                           \
        AMPH_W_M(accl, f1, 0, 1)                                                           \
        AMPH_W_M(acc, f1, 0, 0)                                                            \
        AMPH_S3_SB();                                                                      \
        __syncthreads();                                                                   \
    }
    // prologue: slices 0 .. 3 into the register stages, 0 and 1 on to LDS
    AMPH_W_LOADA_ADV(0)
    AMPH_W_LOADA_ADV(1)
    AMPH_W_LOADA_ADV(2)
    AMPH_W_LOADA_ADV(3)
    AMPH_W_ST_B(0, 0) AMPH_W_ST_A(0, 0, 0) AMPH_W_ST_A(0, 0, 1)
    AMPH_W_ST_B(1, 1) AMPH_W_ST_A(1, 1, 0) AMPH_W_ST_A(1, 1, 1)
    __syncthreads();
    Frag f0;                                           // (the next slice's k-step-0 fragments are read into the same registers once its last MFMA has issued)
    AMPH_W_READ9(f0, 0, 0)
    int ks = 0;
    for (; ks + 12 <= n_slices; ks += 12) {
#define AMPH_S3_X(i, R, QL, QS) AMPH_W_STEP(R, QL, QS)
        AMPH_S3_STEPS_0_11(AMPH_S3_X)
#undef AMPH_S3_X
    }
    {
        const int rem = n_slices - ks;
#define AMPH_S3_X(i, R, QL, QS) if (rem > i) AMPH_W_STEP(R, QL, QS)
        AMPH_S3_STEPS_0_11(AMPH_S3_X)
#undef AMPH_S3_X
    }
    bool bad = false;                                   // an output the next layer's fp16 split cannot hold (amph_internal.h kH2Max)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int m = m_blk + rg * 64 + t * 32 + (lane & 31);
        if (m < p.M) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int n0 = n_blk + ct * 32 + 8 * g + 4 * (lane >> 5);
                if (n0 < p.N) {
                    const float4 e_sc = *reinterpret_cast<const float4*>(bn + n0 - n_blk), e_sh = *reinterpret_cast<const float4*>(bn + 128 + n0 - n_blk);
                    float4 v = make_float4(fmaf(accl[t][4 * g], kH2Lo, acc[t][4 * g]), fmaf(accl[t][4 * g + 1], kH2Lo, acc[t][4 * g + 1]),
                                           fmaf(accl[t][4 * g + 2], kH2Lo, acc[t][4 * g + 2]), fmaf(accl[t][4 * g + 3], kH2Lo, acc[t][4 * g + 3]));
                    v.x = fmaf(v.x, e_sc.x, e_sh.x); v.y = fmaf(v.y, e_sc.y, e_sh.y); v.z = fmaf(v.z, e_sc.z, e_sh.z); v.w = fmaf(v.w, e_sc.w, e_sh.w);
                    bad |= h2_unsplittable(v.x) | h2_unsplittable(v.y) | h2_unsplittable(v.z) | h2_unsplittable(v.w);
                    v.x = fmaxf(v.x, 0.0f); v.y = fmaxf(v.y, 0.0f); v.z = fmaxf(v.z, 0.0f); v.w = fmaxf(v.w, 0.0f);
                    *reinterpret_cast<float4*>(p.y + (long long)m * p.N + n0) = v;
                }
            }
        }
    }
    if (bad && p.range_flag) *p.range_flag = 1;
#undef AMPH_W_LOADB
#undef AMPH_W_LOADA
#undef AMPH_W_LOAD
#undef AMPH_W_LOADA_ADV
#undef AMPH_W_ST_B
#undef AMPH_W_ST_A
#undef AMPH_W_FR
#undef AMPH_W_READ9
#undef AMPH_W_M
#undef AMPH_W_STEP
}

// ----------------------------------------------------------------------------
// Fused conv1 -> conv2 on the bf16 matrix cores ("f3s"): k_conv_gemm_f3's structure - persistent over the M tiles, layer 1
// (C_in == 1 direct 3x3 conv + BN + ReLU, k_conv_first's FMA order in fp32) recomputed from an image patch in LDS for the
// 2 x 4 A-tile values a thread stages per slice, continuous slice stream across tiles, cyclic B pointer - with the
// multiply of k_conv_gemm_s3: the recomputed fp32 A values are split into three bf16 terms on their way into LDS, the
// weights come pre-split, twelve v_mfma_f32_32x32x16_bf16 per slice instead of sixteen v_mfma_f32_32x32x2_f32
// (384 instead of 1 024 matrix-pipe cycles) and, unlike the f32 MFMAs, they leave the SIMD's VALU issue to the recompute.
// LDS rows are packed (192 B, no pad) with the 16-byte slots XOR-swizzled by (row >> 2) & 3 - conflict-free like the padded
// form - so that three A + three B slice buffers and the patch stay under 80 KB: two workgroups per CU.
// ----------------------------------------------------------------------------
constexpr int kFRow = 192;               // packed LDS row: [h | m | l] x 32 bf16, slot j at 16 (j ^ ((row >> 2) & 3))
constexpr int kFBuf = 64 * kFRow;        // 12 288 B per operand per slice buffer

template <int PU>                        // patch elements per thread (n_patch <= 256 PU)
__global__ __launch_bounds__(256, 2) void k_conv_gemm_f3s(ConvGemmParams p) {
    constexpr int BM = 64;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                                                    // [3][64][192]
    char* Bs = smem + 3 * kFBuf;                                        // [3][64][192]
    float* patch = reinterpret_cast<float*>(smem + 6 * kFBuf);          // [rows][W0 + 2 p1]: ONE buffer (see the tile loop)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c4 = tid & 7, r0 = tid >> 3;
    const int n_blk = blockIdx.y * 64;
    const int ohw = p.OH * p.OW;
    const int PW = p.W0 + 2 * p.p1;
    const float rcp_pw = 1.0f / (float)PW, rcp_ow = 1.0f / (float)p.OW, rcp_ohw = 1.0f / (float)ohw;
    const int n_tiles = (p.M + BM - 1) / BM;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    // this thread's 4 first-layer channels: 9 tap weights + BN constants, as channel pairs (v_pk_fma_f32), for the whole kernel
    f32x2 wl_0, wl_1, wl_2, wl_3, wl_4, wl_5, wl_6, wl_7, wl_8, wh_0, wh_1, wh_2, wh_3, wh_4, wh_5, wh_6, wh_7, wh_8, sc1l, sc1h, sh1l, sh1h;
    {
        const float* wt = p.w1 + 4 * c4;
#define AMPH_FS_W(t) { const float4 v_ = *reinterpret_cast<const float4*>(wt + t * p.C_in); wl_##t = f32x2{v_.x, v_.y}; wh_##t = f32x2{v_.z, v_.w}; }
        AMPH_FS_W(0) AMPH_FS_W(1) AMPH_FS_W(2) AMPH_FS_W(3) AMPH_FS_W(4) AMPH_FS_W(5) AMPH_FS_W(6) AMPH_FS_W(7) AMPH_FS_W(8)
#undef AMPH_FS_W
        const float4 sc1 = *reinterpret_cast<const float4*>(p.scale1 + 4 * c4), sh1 = *reinterpret_cast<const float4*>(p.shift1 + 4 * c4);
        sc1l = f32x2{sc1.x, sc1.y}; sc1h = f32x2{sc1.z, sc1.w}; sh1l = f32x2{sh1.x, sh1.y}; sh1h = f32x2{sh1.z, sh1.w};
    }
    // patch slots of this thread (the same for every tile): element i = tid + 256 u of the [rows][PW] patch
    const int n_oh = s3_div_small(BM, p.OW, rcp_ow);
    const int r1_cnt = (n_oh - 1) * p.stride + p.k;
    const int n_patch = ((r1_cnt - 1) * p.s1 + 3) * PW;            // <= 256 * PU (host check)
    float pf[PU];                                                    // the NEXT tile's patch values, prefetched
    // (row / column of a slot are recomputed per fetch - twice per workgroup - rather than kept in 16 registers: the kernel
    // sits at the 256-register limit of two waves per SIMD)
#define AMPH_FS_FETCH_PATCH(tile_)                                                         \
    {                                                                                      \
        const int mb_ = (tile_) * BM;                                                      \
        const int b_ = s3_div_small(mb_, ohw, rcp_ohw), ohf_ = s3_div_small(mb_ - b_ * ohw, p.OW, rcp_ow); \
        const int plo_ = (ohf_ * p.stride - p.pad) * p.s1 - p.p1;                          \
        const float* img_ = p.img + (long long)b_ * p.H0 * p.W0;                           \
        _Pragma("unroll") for (int u = 0; u < PU; ++u) {                                   \
            const int i_ = tid + 256 * u;                                                  \
            const int r_ = s3_div_small(i_, PW, rcp_pw), ic_ = i_ - r_ * PW - p.p1;        \
            const int ir_ = plo_ + r_;                                                     \
            pf[u] = *((i_ < n_patch && ir_ >= 0 && ir_ < p.H0 && ic_ >= 0 && ic_ < p.W0) ? img_ + ir_ * p.W0 + ic_ : p.zeros); \
        }                                                                                  \
    }
    if ((int)blockIdx.x < n_tiles) AMPH_FS_FETCH_PATCH(blockIdx.x)
    const int ohl0 = s3_div_small(r0, p.OW, rcp_ow), owl0 = r0 - ohl0 * p.OW, ohl1 = s3_div_small(r0 + 32, p.OW, rcp_ow), owl1 = (r0 + 32) - ohl1 * p.OW;
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
    // B: the split weight block of (n tile, slice) is 64 x 192 B contiguous; the pointer runs CYCLICALLY over the K slices
    const int n_slices = p.K_pad / kSBK;                 // == k * k taps (C_in == 32), splits == 1
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.w), 0, -1, 0x00020000);
    const unsigned woff = 16u * (unsigned)tid;
    const int w_base = blockIdx.y * n_slices * (64 * kSWRow);
    int w_soff = w_base;
    int l_left = n_slices;                               // slices until the offset wraps
    uint4 sb0a, sb0b, sb0c, sb1a, sb1b, sb1c, sb2a, sb2b, sb2c;
    float* const patch_c = patch;                        // patch of the tile being COMPUTED
    unsigned vmc0 = 0, vmc1 = 0;
#define AMPH_FS_MASKS(t_)                                                                  \
    {                                                                                      \
        const int mb_ = (t_) * BM;                                                         \
        const int b_ = s3_div_small(mb_, ohw, rcp_ohw), ohf_ = s3_div_small(mb_ - b_ * ohw, p.OW, rcp_ow); \
        const int ihA = (ohf_ + ohl0) * p.stride - p.pad, ihB = (ohf_ + ohl1) * p.stride - p.pad; \
        const unsigned rbA = ((unsigned)(ihA) < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ihA + 1) < (unsigned)p.H ? 0x038u : 0u) | \
                             ((unsigned)(ihA + 2) < (unsigned)p.H ? 0x1c0u : 0u);          \
        const unsigned rbB = ((unsigned)(ihB) < (unsigned)p.H ? 0x007u : 0u) | ((unsigned)(ihB + 1) < (unsigned)p.H ? 0x038u : 0u) | \
                             ((unsigned)(ihB + 2) < (unsigned)p.H ? 0x1c0u : 0u);          \
        vmc0 = mb_ + r0 < p.M ? hm0 & rbA : 0u;                                            \
        vmc1 = mb_ + r0 + 32 < p.M ? hm1 & rbB : 0u;                                       \
    }
#define AMPH_FS_WRITE_PATCH(dst_)                                                          \
    {                                                                                      \
        _Pragma("unroll") for (int u = 0; u < PU; ++u)                                     \
            if (tid + 256 * u < n_patch) (dst_)[tid + 256 * u] = pf[u];                    \
    }
    int c_kpos = 0, c_kh = 0, c_kw = 0;                  // tap of the next slice to COMPUTE (wave-uniform)
    f32x2 a0l, a0h, a1l, a1h;                            // the A stage being computed (rows r0, r0 + 32; channel pairs)
    float pv0_0, pv0_1, pv0_2, pv0_3, pv0_4, pv0_5, pv0_6, pv0_7, pv0_8;
    float pv1_0, pv1_1, pv1_2, pv1_3, pv1_4, pv1_5, pv1_6, pv1_7, pv1_8;
    float cap0 = 0.0f, cap1 = 0.0f;                      // +inf for a tap inside the image, 0 for a padded one
#define AMPH_FS_LOADB(q)                                                                   \
    {                                                                                      \
        sb##q##a = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff, w_soff));          \
        sb##q##b = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff + 4096u, w_soff));  \
        sb##q##c = __builtin_bit_cast(uint4, AMPH_S3_BUFLOAD(rs_w, woff + 8192u, w_soff));  \
        --l_left;                                                                          \
        w_soff = __builtin_amdgcn_readfirstlane(l_left == 0 ? w_base : w_soff + 64 * kSWRow);   /* wrap to slice 0 */ \
        l_left = l_left == 0 ? n_slices : l_left;                                          \
    }
#define AMPH_FS_WINDOW(i)                                                                  \
    {                                                                                      \
        const bool ok_ = (vmc##i >> c_kpos) & 1u;                                          \
        cap##i = ok_ ? __builtin_inff() : 0.0f;                                            \
        const float* pp_ = patch_c + (ok_ ? rowbase##i + (c_kh * p.s1 * PW + c_kw * p.s1) : 0); \
        pv##i##_0 = pp_[0]; pv##i##_1 = pp_[1]; pv##i##_2 = pp_[2];                        \
        pv##i##_3 = pp_[PW]; pv##i##_4 = pp_[PW + 1]; pv##i##_5 = pp_[PW + 2];             \
        pv##i##_6 = pp_[2 * PW]; pv##i##_7 = pp_[2 * PW + 1]; pv##i##_8 = pp_[2 * PW + 2]; \
    }
#define AMPH_FS_TAP(i, t)                                                                  \
    a##i##l.x = fmaf(pv##i##_##t, wl_##t.x, a##i##l.x); a##i##l.y = fmaf(pv##i##_##t, wl_##t.y, a##i##l.y); \
    a##i##h.x = fmaf(pv##i##_##t, wh_##t.x, a##i##h.x); a##i##h.y = fmaf(pv##i##_##t, wh_##t.y, a##i##h.y);
#define AMPH_FS_ZERO(i) a##i##l = f32x2{0.f, 0.f}; a##i##h = f32x2{0.f, 0.f};
#define AMPH_FS_FIN(i)                                                                     \
    a##i##l.x = fmaf(a##i##l.x, sc1l.x, sh1l.x); a##i##l.y = fmaf(a##i##l.y, sc1l.y, sh1l.y); \
    a##i##h.x = fmaf(a##i##h.x, sc1h.x, sh1h.x); a##i##h.y = fmaf(a##i##h.y, sc1h.y, sh1h.y); \
    a##i##l.x = __builtin_amdgcn_fmed3f(a##i##l.x, 0.0f, cap##i); a##i##l.y = __builtin_amdgcn_fmed3f(a##i##l.y, 0.0f, cap##i); \
    a##i##h.x = __builtin_amdgcn_fmed3f(a##i##h.x, 0.0f, cap##i); a##i##h.y = __builtin_amdgcn_fmed3f(a##i##h.y, 0.0f, cap##i);
#define AMPH_FS_NEXT_TAP()                                                                 \
    {                                                                                      \
        ++c_kpos; ++c_kw;                                                                  \
        const int wrap_ = c_kw >= p.k ? 1 : 0;                                             \
        c_kw -= wrap_ * p.k;                                                               \
        c_kh += wrap_;                                                                     \
    }
    // LDS offsets.  A: row r0 (+32), the thread's 4 k = 8 bytes of slot (c4 >> 1) of each term; B: byte o of the block
    const int sw0 = (r0 >> 2) & 3;                       // == ((r0 + 32) >> 2) & 3
    const int st_a = r0 * kFRow + 16 * ((c4 >> 1) ^ sw0) + 8 * (c4 & 1);
    int st_b0, st_b1, st_b2;
    {
        const int o0 = 16 * tid, o1 = o0 + 4096, o2 = o0 + 8192;
#define AMPH_FS_SW(o_) (((o_) / kFRow) * kFRow + 16 * ((((o_) % kFRow) >> 4) ^ ((((o_) / kFRow) >> 2) & 3)))
        st_b0 = AMPH_FS_SW(o0); st_b1 = AMPH_FS_SW(o1); st_b2 = AMPH_FS_SW(o2);
#undef AMPH_FS_SW
    }
    // (slot 4 t + u of term t: the swizzle only touches u, so the three terms of a split row are 64 bytes apart as in s3)
#define AMPH_FS_SPLIT_STORE(lo_, hi_, dst_)                                                \
    {                                                                                      \
        uint32_t h0, m0, l0, h1, m1, l1;                                                   \
        s3_split2(lo_.x, lo_.y, h0, m0, l0);                                               \
        s3_split2(hi_.x, hi_.y, h1, m1, l1);                                               \
        *reinterpret_cast<uint2*>(dst_) = make_uint2(h0, h1);                              \
        *reinterpret_cast<uint2*>(dst_ + 64) = make_uint2(m0, m1);                         \
        *reinterpret_cast<uint2*>(dst_ + 128) = make_uint2(l0, l1);                        \
    }
#define AMPH_FS_STORE_A0(buf_) { char* a_ = As + (buf_) * kFBuf + st_a; AMPH_FS_SPLIT_STORE(a0l, a0h, a_) }
#define AMPH_FS_STORE_A1(buf_) { char* a_ = As + (buf_) * kFBuf + st_a + 32 * kFRow; AMPH_FS_SPLIT_STORE(a1l, a1h, a_) }
#define AMPH_FS_STORE_B(q, buf_)                                                           \
    {                                                                                      \
        char* b_ = Bs + (buf_) * kFBuf;                                                    \
        *reinterpret_cast<uint4*>(b_ + st_b0) = sb##q##a;                                  \
        *reinterpret_cast<uint4*>(b_ + st_b1) = sb##q##b;                                  \
        *reinterpret_cast<uint4*>(b_ + st_b2) = sb##q##c;                                  \
    }
#define AMPH_FS_COMPUTE_ALL()                                                              \
    {                                                                                      \
        AMPH_FS_WINDOW(0) AMPH_FS_WINDOW(1)                                                \
        AMPH_FS_ZERO(0) AMPH_FS_ZERO(1)                                                    \
        AMPH_FS_TAP(0, 0) AMPH_FS_TAP(0, 1) AMPH_FS_TAP(0, 2) AMPH_FS_TAP(0, 3) AMPH_FS_TAP(0, 4) \
        AMPH_FS_TAP(0, 5) AMPH_FS_TAP(0, 6) AMPH_FS_TAP(0, 7) AMPH_FS_TAP(0, 8) AMPH_FS_FIN(0) \
        AMPH_FS_TAP(1, 0) AMPH_FS_TAP(1, 1) AMPH_FS_TAP(1, 2) AMPH_FS_TAP(1, 3) AMPH_FS_TAP(1, 4) \
        AMPH_FS_TAP(1, 5) AMPH_FS_TAP(1, 6) AMPH_FS_TAP(1, 7) AMPH_FS_TAP(1, 8) AMPH_FS_FIN(1) \
        AMPH_FS_NEXT_TAP()                                                                 \
    }

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int row_tile = wave >> 1, col_tile = wave & 1;
    // fragment addresses: lane l holds row (l & 31), k = 16 kstep + 8 (l >> 5) + j of each term = slot 4 t + 2 kstep + (l >> 5)
    const int swl = ((lane & 31) >> 2) & 3;              // (the row tile offset, a multiple of 32, does not change the swizzle)
    const char* a_rd0 = As + (row_tile * 32 + (lane & 31)) * kFRow + 16 * ((lane >> 5) ^ swl);
    const char* a_rd1 = As + (row_tile * 32 + (lane & 31)) * kFRow + 16 * ((2 + (lane >> 5)) ^ swl);
    const char* b_rd0 = Bs + (col_tile * 32 + (lane & 31)) * kFRow + 16 * ((lane >> 5) ^ swl);
    const char* b_rd1 = Bs + (col_tile * 32 + (lane & 31)) * kFRow + 16 * ((2 + (lane >> 5)) ^ swl);
#define AMPH_FS_FRAG(base_, buf_, t_) __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>((base_) + (buf_) * kFBuf + 64 * (t_)))
#define AMPH_FS_READ6(pre_, buf_, ks_)                                                     \
    pre_##xh = AMPH_FS_FRAG(a_rd##ks_, buf_, 0); pre_##xm = AMPH_FS_FRAG(a_rd##ks_, buf_, 1); pre_##xl = AMPH_FS_FRAG(a_rd##ks_, buf_, 2); \
    pre_##wh = AMPH_FS_FRAG(b_rd##ks_, buf_, 0); pre_##wm = AMPH_FS_FRAG(b_rd##ks_, buf_, 1); pre_##wl = AMPH_FS_FRAG(b_rd##ks_, buf_, 2);
#define AMPH_FS_M(w_, x_) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(w_, x_, acc, 0, 0, 0);
    // One slice from LDS buffer R.  B stage QL receives slice s + 3; slice s + 2 (A recomputed here, B stage QS) goes to buffer
    // (R + 2) % 3.  bf16 MFMAs leave the VALU to the same wave (an MFMA holds the vector issue for 8 of its 32 cycles), so the
    // recompute is spread over the MFMA gaps instead of being packed into two of them as in the f32 kernel.  The kernel sits at
    // the 256-register limit of two waves per SIMD: each k-step's six fragments are read when the previous six are dead (no
    // read-ahead across the barrier as in k_conv_gemm_s3); the recompute's VALU work covers the two LDS round trips.
#define AMPH_FS_STEP(R, QL, QS)                                                            \
    {                                                                                      \
        bf16x8 f0xh, f0xm, f0xl, f0wh, f0wm, f0wl, f1xh, f1xm, f1xl, f1wh, f1wm, f1wl;     \
        AMPH_FS_READ6(f0, R, 0)                                                            \
        AMPH_FS_LOADB(QL)                                                                  \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_ZERO(0) AMPH_FS_ZERO(1)                                                    \
        AMPH_FS_TAP(0, 0) AMPH_FS_TAP(1, 0) AMPH_FS_TAP(0, 1) AMPH_FS_TAP(1, 1) AMPH_FS_TAP(0, 2) AMPH_FS_TAP(1, 2) \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_T(fq_a)                                                                    \
        AMPH_FS_M(f0wl, f0xh)                                                              \
        AMPH_FS_M(f0wh, f0xl)                                                              \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_TAP(0, 3) AMPH_FS_TAP(1, 3) AMPH_FS_TAP(0, 4) AMPH_FS_TAP(1, 4) AMPH_FS_TAP(0, 5) AMPH_FS_TAP(1, 5) \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_M(f0wm, f0xm)                                                              \
        AMPH_FS_M(f0wm, f0xh)                                                              \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_TAP(0, 6) AMPH_FS_TAP(1, 6) AMPH_FS_TAP(0, 7) AMPH_FS_TAP(1, 7) AMPH_FS_TAP(0, 8) AMPH_FS_TAP(1, 8) \
        AMPH_FS_FIN(0) AMPH_FS_FIN(1)                                                      \
        AMPH_FS_NEXT_TAP()                                                                 \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_M(f0wh, f0xm)                                                              \
        AMPH_FS_M(f0wh, f0xh)                                                              \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_T(fq_m)                                                                    \
        AMPH_FS_READ6(f1, R, 1)                                                            \
        AMPH_FS_WINDOW(0)           /* the NEXT step's image windows: the window registers are dead, the patch is read-only */ \
        AMPH_FS_WINDOW(1)           /* between two patch writes, and an LDS round trip right behind a barrier is the slowest  */ \
        AMPH_FS_STORE_A0((R + 2) % 3)                                                      \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_M(f1wl, f1xh)                                                              \
        AMPH_FS_M(f1wh, f1xl)                                                              \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_STORE_A1((R + 2) % 3)                                                      \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_M(f1wm, f1xm)                                                              \
        AMPH_FS_M(f1wm, f1xh)                                                              \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_STORE_B(QS, (R + 2) % 3)                                                   \
        AMPH_S3_SB();                                                                      \
        AMPH_FS_M(f1wh, f1xm)                                                              \
        AMPH_FS_M(f1wh, f1xh)                                                              \
        AMPH_S3_SB();                                                                      \
        AMPH_S3_T(fq_b0)                                                                   \
        __syncthreads();                                                                   \
        AMPH_S3_T(fq_b1)                                                                   \
        AMPH_FS_ACC()                                                                      \
    }
#ifdef AMPH_STAMPS
    unsigned long long fq_b0 = 0, fq_b1 = 0, fq_prev = 0, fq_bar = 0, fq_work = 0, fq_t0 = 0, fq_t1 = 0, fq_a = 0, fq_m = 0, fq_p1 = 0, fq_p2 = 0;
    const unsigned long long fq_rt0 = __builtin_amdgcn_s_memrealtime();
    AMPH_S3_T(fq_t0)
#define AMPH_FS_ACC() { fq_bar += fq_b1 - fq_b0; if (fq_prev) { fq_work += fq_b0 - fq_prev; fq_p1 += fq_a - fq_prev; fq_p2 += fq_m - fq_a; } fq_prev = fq_b1; }
#else
#define AMPH_FS_ACC()
#endif

    // ---- first tile: patch, masks, B slices 0..2, A of slices 0 and 1, on to LDS buffers 0 and 1
    AMPH_FS_LOADB(0)
    AMPH_FS_LOADB(1)
    AMPH_FS_LOADB(2)
    if ((int)blockIdx.x < n_tiles) {
        AMPH_FS_WRITE_PATCH(patch)
        AMPH_FS_MASKS(blockIdx.x)
    }
    __syncthreads();                                     // the patch is staged
    if ((int)(blockIdx.x + gridDim.x) < n_tiles) AMPH_FS_FETCH_PATCH(blockIdx.x + gridDim.x)
    AMPH_FS_COMPUTE_ALL()
    AMPH_FS_STORE_A0(0) AMPH_FS_STORE_A1(0) AMPH_FS_STORE_B(0, 0)
    AMPH_FS_COMPUTE_ALL()
    AMPH_FS_STORE_A0(1) AMPH_FS_STORE_A1(1) AMPH_FS_STORE_B(1, 1)
    AMPH_FS_WINDOW(0) AMPH_FS_WINDOW(1)                  // the windows of the first step (every step reads the next one's)
    __syncthreads();

    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int m_blk = tile * BM;
        const int nt = tile + (int)gridDim.x;
        const bool has_next = nt < n_tiles;
        AMPH_FS_STEP(0, 0, 2)                            // slices 0..2 multiplied, 2..4 recomputed
        AMPH_FS_STEP(1, 1, 0)
        AMPH_FS_STEP(2, 2, 1)
        AMPH_FS_STEP(0, 0, 2)
        AMPH_FS_STEP(1, 1, 0)
        AMPH_FS_STEP(2, 2, 1)
        AMPH_FS_STEP(0, 0, 2)                            // slice 6 multiplied, 8 recomputed: the last read of this tile's patch
        // The next tile's patch (in registers since the previous tile) replaces it - the barrier that ended the step above is
        // behind every wave's last read, one more barrier publishes the new patch - and the patch after it is requested.
        // (One patch buffer, not two as in the f32 kernel: with the 19-row patch of a 64 x 64 image two of them put the
        // workgroup over 80 KB, i.e. one workgroup per CU; the extra barrier per tile is ~0.1 % of it.)
        if (has_next) AMPH_FS_WRITE_PATCH(patch)
        __syncthreads();
        if (nt + (int)gridDim.x < n_tiles) AMPH_FS_FETCH_PATCH(nt + gridDim.x)
        // the compute side moves on to the next tile (no tile: empty masks -> all-zero stages, never multiplied)
        c_kpos = 0; c_kh = 0; c_kw = 0;
        if (has_next) { AMPH_FS_MASKS(nt) }
        else { vmc0 = 0; vmc1 = 0; }
        AMPH_FS_WINDOW(0) AMPH_FS_WINDOW(1)              // (what the step above read ahead belonged to the old patch and masks)
        AMPH_FS_STEP(1, 1, 0)                            // slices 7, 8 multiplied; the NEXT tile's 0, 1 recomputed
        AMPH_FS_STEP(2, 2, 1)
        // BN constants of the epilogue: loaded here (L1 / L2 hits), not held in 32 registers for the whole kernel
        float4 e_sc0, e_sc1, e_sc2, e_sc3, e_sh0, e_sh1, e_sh2, e_sh3;
        {
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
        {
            const int m = m_blk + row_tile * 32 + (lane & 31);
            if (m < p.M) {
#define AMPH_FS_OUT(g)                                                                     \
                {                                                                          \
                    const int n0 = n_blk + col_tile * 32 + 8 * g + 4 * (lane >> 5);        \
                    if (n0 < p.N) {                                                        \
                        float4 v;                                                          \
                        v.x = fmaxf(fmaf(acc[4 * g], e_sc##g.x, e_sh##g.x), 0.0f);         \
                        v.y = fmaxf(fmaf(acc[4 * g + 1], e_sc##g.y, e_sh##g.y), 0.0f);     \
                        v.z = fmaxf(fmaf(acc[4 * g + 2], e_sc##g.z, e_sh##g.z), 0.0f);     \
                        v.w = fmaxf(fmaf(acc[4 * g + 3], e_sc##g.w, e_sh##g.w), 0.0f);     \
                        if (p.y_bf16) {                                                    \
                            const amph_bf16x2 lo_ = {(__bf16)v.x, (__bf16)v.y}, hi_ = {(__bf16)v.z, (__bf16)v.w};   /* RNE */ \
                            *reinterpret_cast<uint2*>(reinterpret_cast<uint16_t*>(p.y) + (long long)m * p.N + n0) =       \
                                make_uint2(__builtin_bit_cast(uint32_t, lo_), __builtin_bit_cast(uint32_t, hi_));          \
                        } else *reinterpret_cast<float4*>(p.y + (long long)m * p.N + n0) = v; \
                    }                                                                      \
                }
                AMPH_FS_OUT(0) AMPH_FS_OUT(1) AMPH_FS_OUT(2) AMPH_FS_OUT(3)
#undef AMPH_FS_OUT
            }
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    }
#ifdef AMPH_STAMPS
    AMPH_S3_T(fq_t1)
    if (tid == 0 && p.stamps) {       // cycles of wave 0: between barriers (work) | at the step barriers | - | - | total | lifetime ticks
        unsigned long long* o = p.stamps + (size_t)blockIdx.x * 8;
        o[0] = fq_work; o[1] = fq_p1; o[2] = fq_bar; o[3] = fq_p2; o[4] = fq_t1 - fq_t0;   // o[1]: step start -> 6 taps done; o[3]: -> k-step 0 done
        o[5] = __builtin_amdgcn_s_memrealtime() - fq_rt0;
        o[6] = 0; o[7] = fq_rt0 & 0xffffffffull;
    }
#endif
}

static size_t conv_gemm_f3s_lds(const ConvGemmParams& p) {
    const int n_oh = 64 / p.OW;
    const int rows = ((n_oh - 1) * p.stride + p.k - 1) * p.s1 + p.k1;
    const size_t patch = ((size_t)rows * (p.W0 + 2 * p.p1) + 3) & ~(size_t)3;
    return (size_t)6 * kFBuf + patch * sizeof(float);
}
// the shapes k_conv_gemm_f3 covers (host: conv_gemm_f3_ok) with split weights at hand; both kernels need n_slices % 3 == 0
bool conv_gemm_f3s_ok(const ConvGemmParams& p) {
    return conv_gemm_f3_ok(p, 64) && p.N <= 64 * 65535 && conv_gemm_f3s_lds(p) <= 80 * 1024 &&
           (unsigned long long)((p.N + 63) / 64 * 64) * (p.K_pad / kSBK) * kSWRow < (1ull << 31);
}
// p.w = the split weight matrix (conv_gemm_s3_split_weights)
void launch_conv_gemm_f3s(const ConvGemmParams& p, hipStream_t s) {
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_f3s<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_f3s<8>), hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
        raised[dev] = true;
    }
    const unsigned n_tiles = (unsigned)((p.M + 63) / 64);
    const dim3 grid(std::min(n_tiles, (unsigned)(2 * g_num_cus)), (unsigned)((p.N + 63) / 64), 1);      // persistent: one resident round
    const int n_oh = 64 / p.OW;
    const int n_patch = (((n_oh - 1) * p.stride + p.k - 1) * p.s1 + p.k1) * (p.W0 + 2 * p.p1);
    if (getenv("AMPH_DEBUG_OCC")) {
        static bool said = false;
        if (!said) {
            said = true;
            int nb8 = -1, nb4 = -1, nbs = -1;
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb8, k_conv_gemm_f3s<8>, 256, conv_gemm_f3s_lds(p));
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb4, k_conv_gemm_f3s<4>, 256, conv_gemm_f3s_lds(p));
            (void)hipOccupancyMaxActiveBlocksPerMultiprocessor(&nbs, k_conv_gemm_s3, 256, (size_t)6 * kHBuf + 512);
            fprintf(stderr, "[amph] occupancy (workgroups per CU): f3s<8> %d, f3s<4> %d (lds %zu, n_patch %d), s3 %d (lds %zu)\n", nb8, nb4,
                    conv_gemm_f3s_lds(p), n_patch, nbs, (size_t)6 * kHBuf + 512);
        }
    }
    if (n_patch <= 256 * 4) hipLaunchKernelGGL(k_conv_gemm_f3s<4>, grid, dim3(256), conv_gemm_f3s_lds(p), s, p);
    else hipLaunchKernelGGL(k_conv_gemm_f3s<8>, grid, dim3(256), conv_gemm_f3s_lds(p), s, p);
}

constexpr size_t kS3Lds = (size_t)6 * kHBuf + 512;     // 49 664 B (slice buffers + the tile's BN constants): three workgroups per CU

bool conv_gemm_s3_ok(const ConvGemmParams& p) {
    const unsigned long long span = (unsigned long long)p.M / (p.OH * p.OW) * p.H * p.W * p.C_in * 4ull + p.x_guard;     // 32-bit lane offsets
    return (p.C_in % kSBK) == 0 && p.K == p.K_pad && p.k * p.k <= 32 && (p.N % 4) == 0 && p.pad < p.k && p.zeros != nullptr &&
           (size_t)p.x_guard >= conv_gemm_x_guard(p.k, p.W, p.C_in) && span < (1ull << 32) &&
           (unsigned long long)((p.N + 63) / 64 * 64) * (p.K_pad / kSBK) * kHWRow < (1ull << 31) && p.M < (1 << 22);
}

// p.w = the two-term weight matrix [n_pad / 64][K_pad / 32][64][2][32] fp16 (conv_gemm_h2_split_weights)
void launch_conv_gemm_s3(const ConvGemmParams& p, hipStream_t s) {
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_s3), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS3Lds);
        raised[dev] = true;
    }
    const dim3 grid((unsigned)((p.M + 63) / 64), (unsigned)((p.N + 63) / 64), (unsigned)p.splits);
    hipLaunchKernelGGL(k_conv_gemm_s3, grid, dim3(256), kS3Lds, s, p);
}

// the 128 x 128 tile form: BN + ReLU layers whose channel count is a multiple of 128 (both 64-channel weight blocks of a tile exist)
bool conv_gemm_s3w_ok(const ConvGemmParams& p) {
    return conv_gemm_s3_ok(p) && p.splits == 1 && p.epilogue == 0 && (p.N % 128) == 0 && p.scale && p.shift;
}
void launch_conv_gemm_s3w(const ConvGemmParams& p, hipStream_t s) {
    static bool raised[64] = {false};
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (dev >= 0 && dev < 64 && !raised[dev]) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_conv_gemm_s3w), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kS3wLds);
        raised[dev] = true;
    }
    const dim3 grid((unsigned)((p.M + 127) / 128), (unsigned)(p.N / 128), 1);
    hipLaunchKernelGGL(k_conv_gemm_s3w, grid, dim3(512), kS3wLds, s, p);
}

// host: x -> (h, m, l) exactly as s3_split2 does on the device (RNE; every subtraction exact)
static uint16_t s3_bf16_rne(float f) {
    uint32_t u;
    std::memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (uint16_t)((u >> 16) | 0x40u);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float s3_bf16_f32(uint16_t h) {
    const uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}
// host: x -> (h, l) as h2_split2 does on the device; [n_pad / 64][K_pad / 32][64][2][32] fp16
void conv_gemm_h2_split_weights(const float* w, int n_pad, int K_pad, uint16_t* out) {
    const int n_slices = K_pad / kSBK;
    for (int n = 0; n < n_pad; ++n)
        for (int sl = 0; sl < n_slices; ++sl)
            for (int k = 0; k < kSBK; ++k) {
                const float x = w[(size_t)n * K_pad + (size_t)sl * kSBK + k];
                const _Float16 h = (_Float16)x;
                const _Float16 l = (_Float16)((x - (float)h) * 2048.0f);
                uint16_t hb, lb;
                std::memcpy(&hb, &h, 2);
                std::memcpy(&lb, &l, 2);
                uint16_t* o = out + (((size_t)(n / 64) * n_slices + sl) * 64 + (size_t)(n % 64)) * 2 * kSBK;
                o[k] = hb; o[kSBK + k] = lb;
            }
}
void conv_gemm_s3_split_weights(const float* w, int n_pad, int K_pad, uint16_t* out) {
    const int n_slices = K_pad / kSBK;
    for (int n = 0; n < n_pad; ++n)
        for (int sl = 0; sl < n_slices; ++sl)
            for (int k = 0; k < kSBK; ++k) {
                const float x = w[(size_t)n * K_pad + (size_t)sl * kSBK + k];
                const uint16_t h = s3_bf16_rne(x);
                const float r = x - s3_bf16_f32(h);
                const uint16_t m = s3_bf16_rne(r);
                const uint16_t l = s3_bf16_rne(r - s3_bf16_f32(m));
                uint16_t* o = out + (((size_t)(n / 64) * n_slices + sl) * 64 + (size_t)(n % 64)) * 3 * kSBK;
                o[k] = h; o[kSBK + k] = m; o[2 * kSBK + k] = l;
            }
}

}  // namespace amph
