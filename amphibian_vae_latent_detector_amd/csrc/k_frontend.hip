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

#include <algorithm>

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

// Wave-wide reductions on the DPP network - no LDS round trip per step (a __shfl_down is a ds_bpermute: six of them, twelve for a double,
// ~100 cycles apiece behind one another): butterflies inside a quad, half-row and row mirrors (every lane of a row then holds the row's
// value), then the row broadcasts.  The result is valid in LANE 63 only.  Fixed order, like the shuffle form (a different one).
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) { return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), CTRL, 0xF, 0xF, true)); }
template <int CTRL, int ROWS>
__device__ __forceinline__ float dpp_f_rows(float v, float other) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(other), __float_as_int(v), CTRL, ROWS, 0xF, false));
}
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_mov_dpp((int)b, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int CTRL, int ROWS>
__device__ __forceinline__ double dpp_d_rows(double v) {            // 0.0 in the rows outside ROWS
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, ROWS, 0xF, false), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, ROWS, 0xF, false);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = fmaxf(v, dpp_f<0xB1>(v));                        // quad_perm [1, 0, 3, 2]
    v = fmaxf(v, dpp_f<0x4E>(v));                        // quad_perm [2, 3, 0, 1]
    v = fmaxf(v, dpp_f<0x141>(v));                       // row_half_mirror
    v = fmaxf(v, dpp_f<0x140>(v));                       // row_mirror
    v = fmaxf(v, dpp_f_rows<0x142, 0xA>(v, v));          // row_bcast:15 into rows 1, 3
    v = fmaxf(v, dpp_f_rows<0x143, 0xC>(v, v));          // row_bcast:31 into rows 2, 3
    return v;
}
__device__ __forceinline__ double wave_sum_dpp(double v) {
    v += dpp_d<0xB1>(v);
    v += dpp_d<0x4E>(v);
    v += dpp_d<0x141>(v);
    v += dpp_d<0x140>(v);
    v += dpp_d_rows<0x142, 0xA>(v);
    v += dpp_d_rows<0x143, 0xC>(v);
    return v;
}
// Block-wide reductions for blocks of NW waves (4 or 16); result valid in every thread.
template <int NW = 4>
__device__ __forceinline__ double block_sum(double v, double* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum_dpp(v);
    __syncthreads();
    if (lane == 63) sh[w] = v;
    __syncthreads();
    if (NW == 4) return (sh[0] + sh[1]) + (sh[2] + sh[3]);
    double q[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) q[g] = (sh[4 * g] + sh[4 * g + 1]) + (sh[4 * g + 2] + sh[4 * g + 3]);
    return (q[0] + q[1]) + (q[2] + q[3]);
}
template <int NW = 4>
__device__ __forceinline__ float block_max(float v, float* sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_max_dpp(v);
    __syncthreads();
    if (lane == 63) sh[w] = v;
    __syncthreads();
    float m = sh[0];
#pragma unroll
    for (int g = 1; g < NW; ++g) m = fmaxf(m, sh[g]);
    return m;
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
// chunks staged in LDS per pass: 4 for PCM16 (64 KB: 256 blocks, every thread owns one; a 1 s / 22.05 kHz
// row is a single pass), 2 for float32 (the same 64 KB)
template <typename T> struct NpPass { static constexpr int kChunks = sizeof(T) == 2 ? 4 : 2; };

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

// LDS image of a pass: element e of the pass at byte e*sizeof(T) + (e >> 7) * 16: one 16-byte pad
// per 128-element block makes the per-block ds_read_b128 of 64 lanes conflict-free (odd slot stride)
template <typename T>
__device__ __forceinline__ int np_lds_off(int e) { return e * (int)sizeof(T) + ((e >> 7) << 4); }

// NOTE: `__fmul_rn` / `__fadd_rn` are device-library functions whose fmul / fadd carry the
// library's own `contract` flag: hipcc fused them into v_fmac (seen in the ISA; 1-ulp
// differences from numpy).  Code that must round each operation separately is written
// with plain operators under `#pragma clang fp contract(off)` and checked in the ISA.
__device__ __forceinline__ float sq_of(int16_t q) {
#pragma clang fp contract(off)
    const float y = (float)q * (1.0f / 32768.0f);
    return y * y;
}
__device__ __forceinline__ float sq_of(float y) {
#pragma clang fp contract(off)
    return y * y;
}

// 1024 threads: the reduction tree of a pass is written for 256 and leaves the rest idle, but staging and the
// normalised-row write pass - latency- and issue-bound at one wave per SIMD - run on four times the waves
// (rms stage 18.4 -> 16.1 us with 512, 14.8 us with 1024 threads; same bits)
constexpr int kRmsThreads = 1024;

// TH threads per workgroup, CH numpy chunks staged per pass: <T, 1024, 4 | 2> for whole rows (one workgroup per row), <int16_t, 256, 1>
// for the chunk-sums mode on long rows (one workgroup per chunk: 8 x the workgroups, each a fraction of the latency chain)
// The body of K0 for ONE row on a workgroup of TH threads (smem: rms_scale_lds<T, CH>() bytes, 16-byte aligned): k_rms_scale below, and - since
// round 4 - the head of k_stft_pair's row loop ("stft_k0_fuse": the row's workgroup normalises it itself, two launches less per batch).
// Ends without a barrier: the caller synchronises before smem is reused or norm_out is read.
template <typename T, int TH, int CH>
__device__ __forceinline__ void rms_scale_row(char* smem, const int tid, const int row, const int pass_y, const T* __restrict__ in, long long row_stride,
                                              const int32_t* __restrict__ lengths, int n_in,
                                              float* __restrict__ scale, int32_t* __restrict__ ok, int16_t* __restrict__ norm_out,
                                              const NormPad& pad, int scale_f64, float* __restrict__ chunk_sums, int chunk_stride) {
#pragma clang fp contract(off)
    // chunk_sums != nullptr (rows longer than one pass): workgroup (row, y) sums ONE pass - numpy chunks CH y .. CH y + CH - 1 -
    // and writes their sums; k_rms_fold_apply folds them in order.  Same leaves, same tree.
    constexpr int kChunksPerPass = CH, kRmsThreads = TH;
    constexpr int kPassElems = kChunksPerPass * kNpChunk;
    constexpr int kVec = 16 / (int)sizeof(T);            // elements per 16-byte LDS read
    float* heap = reinterpret_cast<float*>(smem);        // [kChunksPerPass][256]: nodes 1 .. 255 of a tree of depth <= 7
    char* img = smem + kChunksPerPass * 256 * 4;
    const T* x = in + (long long)row * row_stride;
    const int n = lengths ? min(lengths[row], n_in) : n_in;
    const int n_chunks = (n + kNpChunk - 1) / kNpChunk;
    float total = 0.0f;                                  // thread 0 only
    const int c_first = chunk_sums ? pass_y * kChunksPerPass : 0;
    const int c_last = chunk_sums ? min(n_chunks, c_first + kChunksPerPass) : n_chunks;
    for (int c0 = c_first; c0 < c_last; c0 += kChunksPerPass) {
        // ---- stage this pass's samples, coalesced (dword granules when the row is 4-byte aligned)
        const int e0 = c0 * kNpChunk, ne = min(kPassElems, n - e0);
        // every load of a batch is issued before the first LDS write: a 1 s / 22.05 kHz row (11 025 dwords) is ONE batch,
        // i.e. one global round trip (batches of 8 loads per thread were six serialised round trips: 16 -> 10 us)
        // (at most 16 per batch: the 640- / 704- / 768-thread workgroups of k_stft_pair stage a one-second row - 11 025 dwords - in one batch too,
        //  a full 32 768-sample pass in two)
        constexpr int U0 = sizeof(T) == 2 ? (kPassElems / 2 + kRmsThreads - 1) / kRmsThreads : (kPassElems + kRmsThreads - 1) / kRmsThreads;   // one batch per pass
        constexpr int U = U0 < 16 ? U0 : 16;
        if (sizeof(T) == 2 && (((uintptr_t)(x + e0)) & 3u) == 0) {
            const uint32_t* src = reinterpret_cast<const uint32_t*>(x + e0);
            const int nd = ne >> 1;
            for (int base = 0; base < nd; base += kRmsThreads * U) {
                uint32_t v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int d = base + tid + kRmsThreads * u; v[u] = d < nd ? src[d] : 0u; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int d = base + tid + kRmsThreads * u;
                    if (d < nd) *reinterpret_cast<uint32_t*>(img + np_lds_off<T>(2 * d)) = v[u];
                }
            }
            if ((ne & 1) && tid == 0) *reinterpret_cast<T*>(img + np_lds_off<T>(ne - 1)) = x[e0 + ne - 1];
        } else {
            for (int base = 0; base < ne; base += kRmsThreads * U) {
                T v[U];
#pragma unroll
                for (int u = 0; u < U; ++u) { const int e = base + tid + kRmsThreads * u; v[u] = e < ne ? x[e0 + e] : (T)0; }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = base + tid + kRmsThreads * u;
                    if (e < ne) *reinterpret_cast<T*>(img + np_lds_off<T>(e)) = v[u];
                }
            }
        }
        __syncthreads();
        // ---- leaves: thread = (chunk of the pass, leaf slot); 8 accumulators in registers.  A full chunk is 64 blocks of 128 at
        //      depth 6; a partial chunk whose right-most path still holds more than 128 elements at depth 6 (441 of the 8 191
        //      sizes, e.g. 7 922 -> ... -> 130 -> 64 + 66) has leaves at depth 7: 128 slots, heap nodes up to 255
        if (tid < kChunksPerPass * 128) {
            const int cl = tid >> 7, slot = tid & 127, chunk = c0 + cl;
            if (chunk < n_chunks) {
                const int cn = min(kNpChunk, n - chunk * kNpChunk);
                int off, size, lvl, rem = 0;
                bool owner;
                if (cn == kNpChunk) { off = (slot & 63) * kNpBlock; size = kNpBlock; lvl = 6; owner = slot < 64; }
                else {
                    const int D = np_depth(cn);
                    owner = slot < (1 << D);
                    np_walk(cn, owner ? slot : 0, D, off, size, lvl);
                    rem = D - lvl;
                    owner = owner && (slot & ((1 << rem) - 1)) == 0;   // an early leaf belongs to the slot with zero low bits
                }
                if (owner) {
                    const int eb = cl * kNpChunk + off;                 // element index inside the pass
                    float res;
                    if (size < 8) {                                     // numpy: res = -0.0; res += a[i]
                        res = -0.0f;
                        for (int i = 0; i < size; ++i) res = res + sq_of(*reinterpret_cast<const T*>(img + np_lds_off<T>(eb + i)));
                    } else {
                        float r[8];
                        const int body = size - (size & 7);
                        for (int i = 0; i < body; i += 8) {
                            T v[8];
#pragma unroll
                            for (int h = 0; h < 8 / kVec; ++h)
                                *reinterpret_cast<int4*>(&v[h * kVec]) = *reinterpret_cast<const int4*>(img + np_lds_off<T>(eb + i + h * kVec));
#pragma unroll
                            for (int j = 0; j < 8; ++j) r[j] = (i == 0) ? sq_of(v[j]) : r[j] + sq_of(v[j]);
                        }
                        res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                        for (int i = body; i < size; ++i) res = res + sq_of(*reinterpret_cast<const T*>(img + np_lds_off<T>(eb + i)));
                    }
                    heap[cl * 256 + (1 << lvl) + (slot >> rem)] = res;
                }
            }
        }
        __syncthreads();
        // ---- internal nodes, deepest level first: value = left + right
        for (int lvl = 6; lvl >= 0; --lvl) {
            const int t = tid;
            if (t < (kChunksPerPass << lvl)) {
                const int cl = t >> lvl, path = t & ((1 << lvl) - 1), chunk = c0 + cl;
                if (chunk < n_chunks) {
                    const int cn = min(kNpChunk, n - chunk * kNpChunk);
                    bool internal = cn == kNpChunk && lvl < 6;         // a full chunk's level 6 is its 64 leaves
                    if (cn != kNpChunk) {
                        int off, size, reached;
                        np_walk(cn, path, lvl, off, size, reached);
                        internal = reached == lvl && size > kNpBlock;
                    }
                    if (internal) {
                        const int idx = cl * 256 + (1 << lvl) + path;
                        heap[idx] = heap[cl * 256 + 2 * ((1 << lvl) + path)] + heap[cl * 256 + 2 * ((1 << lvl) + path) + 1];
                    }
                }
            }
            __syncthreads();
        }
        if (chunk_sums) {
            if (tid < kChunksPerPass && c0 + tid < n_chunks) chunk_sums[(long long)row * chunk_stride + c0 + tid] = heap[tid * 256 + 1];
            return;
        }
        if (tid == 0) {
            for (int cl = 0; cl < kChunksPerPass && c0 + cl < n_chunks; ++cl)
                total = (c0 + cl == 0) ? heap[cl * 256 + 1] : total + heap[cl * 256 + 1];
        }
        __syncthreads();
    }
    if (chunk_sums) return;                              // (a pass past the row's end: nothing to sum)
    if (tid == 0) {
        float sc = 1.0f;
        int good = 0;
        if (n > 0) {
            const float rms = sqrtf(total / (float)n);           // np.mean = sum / n in float32; IEEE sqrt
            if (scale_f64) {
                // numpy 1.26.4 (the reference's pin, requirements-thesis-baseline-macos-arm64.txt:46): np.float32 scalar <op> Python
                // float promotes to float64, so the gate compares in float64 and `target_rms / (rms + eps)` is a float64 that the
                // array multiply casts to float32 ONCE (00_normalize_dataset_rms.py:32-36)
                if (!((double)rms < 1e-4)) { sc = (float)(0.05 / ((double)rms + 1e-8)); good = 1; }
            } else if (!(rms < 1e-4f)) { sc = 0.05f / (rms + 1e-8f); good = 1; }   // numpy 2.x (NEP 50): float32 throughout
        }
        scale[row] = sc;
        if (ok) ok[row] = good;
        heap[0] = sc;
        heap[1] = good ? 1.0f : 0.0f;
    }
    // PCM16 rows, optional: also write the normalised, re-quantised row - the samples 00_normalize_dataset_rms.py:51-57 puts on
    // disk (decode q / 32768, scale, clip, rint(x * 32767)): k_rms_apply_pcm16's arithmetic, in this launch.  The STFT kernel
    // then decodes plain PCM16 instead of normalising every sample of every (81 % overlapping) frame.
    if constexpr (sizeof(T) == 2) {
        if (norm_out) {
            __syncthreads();
            const float sc = heap[0];
            const bool scaled = heap[1] != 0.0f;
            int16_t* o = norm_out + (long long)row * (pad.lead > 0 ? pad.out_stride : row_stride);
            auto conv = [&](int q) {
                float v = (float)q * (1.0f / 32768.0f);
                if (scaled) v = fminf(fmaxf(v * sc, -1.0f), 1.0f);
                return (int)rintf(v * 32767.0f);
            };
            if (pad.lead > 0) {
                // the row as the STFT sees it (map_detector_core.py:212-228): truncated / zero padded to n_fix samples, then
                // pad.lead samples of librosa's centre padding on each side (reflect, or zeros) - every frame of the row then
                // is an interior frame of this buffer: no index reflection, no per-sample path in the STFT kernel
                const bool in_lds = n_chunks <= kChunksPerPass;
                const int nv = min(n, pad.n_fix), total = pad.n_fix + 2 * pad.lead;
                auto sample = [&](int idx) {
                    int sidx = idx - pad.lead;
                    if (pad.mode == 0) {                      // reflect (np.pad mode="reflect")
                        if (sidx < 0) sidx = -sidx;
                        if (sidx >= pad.n_fix) sidx = 2 * (pad.n_fix - 1) - sidx;
                    }
                    if (sidx < 0 || sidx >= nv) return 0;     // constant padding / the zeros of fix_length
                    const int q = in_lds ? (int)*reinterpret_cast<const T*>(img + np_lds_off<T>(sidx)) : (int)x[sidx];
                    return conv(q);
                };
                if ((reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
                    uint32_t* o2 = reinterpret_cast<uint32_t*>(o);
                    // three loops without a branch inside: [0, i_lo) padding, [i_lo, i_hi) the row itself - an output dword there is
                    // one dword of the LDS image -, [i_hi, total / 2) zeros of fix_length + padding
                    const bool x_al = (reinterpret_cast<uintptr_t>(x) & 3u) == 0;
                    const bool pairs = (in_lds || x_al) && (pad.lead & 1) == 0;
                    const int i_lo = pairs ? pad.lead >> 1 : 0, i_hi = pairs ? i_lo + (nv >> 1) : 0;
                    for (int i = tid; i < i_lo; i += kRmsThreads)
                        o2[i] = ((uint32_t)sample(2 * i) & 0xFFFFu) | ((uint32_t)sample(2 * i + 1) << 16);
                    if (in_lds) {
                        for (int i = i_lo + tid; i < i_hi; i += kRmsThreads) {
                            const uint32_t v = *reinterpret_cast<const uint32_t*>(img + np_lds_off<T>(2 * i - pad.lead));
                            const int lo = conv((int)(short)(v & 0xFFFFu)), hi = conv((int)v >> 16);
                            o2[i] = ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
                        }
                    } else {                              // rows longer than one pass: from global memory again, 32 loads in flight
                        const uint32_t* x2 = reinterpret_cast<const uint32_t*>(x);
                        constexpr int UW = 32;
                        for (int base = i_lo; base < i_hi; base += kRmsThreads * UW) {
                            uint32_t v[UW];
#pragma unroll
                            for (int u = 0; u < UW; ++u) { const int i = base + tid + kRmsThreads * u; v[u] = i < i_hi ? x2[i - i_lo] : 0u; }
#pragma unroll
                            for (int u = 0; u < UW; ++u) {
                                const int i = base + tid + kRmsThreads * u;
                                if (i < i_hi) {
                                    const int lo = conv((int)(short)(v[u] & 0xFFFFu)), hi = conv((int)v[u] >> 16);
                                    o2[i] = ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
                                }
                            }
                        }
                    }
                    for (int i = i_hi + tid; i < (total >> 1); i += kRmsThreads)
                        o2[i] = ((uint32_t)sample(2 * i) & 0xFFFFu) | ((uint32_t)sample(2 * i + 1) << 16);
                    if ((total & 1) && tid == 0) o[total - 1] = (int16_t)sample(total - 1);
                } else {
                    for (int i = tid; i < total; i += kRmsThreads) o[i] = (int16_t)sample(i);
                }
            } else if (n_chunks <= kChunksPerPass && (reinterpret_cast<uintptr_t>(o) & 3u) == 0) {
                // the whole valid row is still in the LDS image of its single pass (samples past the valid length are never
                // read by the STFT kernel): no second global round trip
                const int nd = n >> 1;
                uint32_t* o2 = reinterpret_cast<uint32_t*>(o);
                for (int i = tid; i < nd; i += kRmsThreads) {
                    const uint32_t v = *reinterpret_cast<const uint32_t*>(img + np_lds_off<T>(2 * i));
                    const int lo = conv((int)(short)(v & 0xFFFFu)), hi = conv((int)v >> 16);
                    o2[i] = ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
                }
                if ((n & 1) && tid == 0) o[n - 1] = (int16_t)conv((int)*reinterpret_cast<const T*>(img + np_lds_off<T>(n - 1)));
            } else if (((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(o)) & 3u) == 0) {
                const uint32_t* x2 = reinterpret_cast<const uint32_t*>(x);
                uint32_t* o2 = reinterpret_cast<uint32_t*>(o);
                const int nd = n_in >> 1;
                for (int i = tid; i < nd; i += kRmsThreads) {
                    const uint32_t v = x2[i];
                    const int lo = conv((int)(short)(v & 0xFFFFu)), hi = conv((int)v >> 16);
                    o2[i] = ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
                }
                if ((n_in & 1) && tid == 0) o[n_in - 1] = (int16_t)conv((int)x[n_in - 1]);
            } else {
                for (int i = tid; i < n_in; i += kRmsThreads) o[i] = (int16_t)conv((int)x[i]);
            }
        }
    }
}

template <typename T, int TH, int CH>
__global__ __launch_bounds__(TH) void k_rms_scale(const T* __restrict__ in, long long row_stride,
                                                   const int32_t* __restrict__ lengths, int n_in,
                                                   float* __restrict__ scale, int32_t* __restrict__ ok, int16_t* __restrict__ norm_out,
                                                   NormPad pad, int scale_f64, float* __restrict__ chunk_sums, int chunk_stride) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    rms_scale_row<T, TH, CH>(smem, threadIdx.x, blockIdx.x, blockIdx.y, in, row_stride, lengths, n_in, scale, ok, norm_out, pad, scale_f64, chunk_sums, chunk_stride);
}

// ----------------------------------------------------------------------------
// K0's chunk sums, second form (round 4; PCM16 rows, one 256-thread workgroup per (row, numpy chunk)): k_rms_scale<int16_t, 256, 1> gives a
// leaf - a <= 128-sample block, 8 accumulators - to ONE thread (~600 dependent vector instructions on 64 of its 256 threads) and walks the
// tree in seven barrier-separated levels.  Here a leaf sits on a lane QUAD: lane q holds numpy's accumulators r[2q], r[2q + 1], fed by one
// aligned dword per 8 elements (all 16 requested at once); (r0 + r1) + (r2 + r3) and the rest of numpy's combine are two DPP butterflies;
// 64 leaves = 256 lanes, one round.  Squares of the INTEGER samples: fl(q^2) 2^-30 == fl(y^2), y = q / 32768, and the power of two commutes
// with every rounding of the sum - scaled once at the end.  A COMPLETE tree (every leaf at the same depth: all full chunks, and e.g. the
// 5 666-sample tail of a one-second row) is a butterfly of one wave on the DPP network (sums are commutative: either operand order gives
// the node's bits); anything else takes the level-by-level walk on that wave.  Same leaves, same tree, same bits as the first form
// (tests/test_gpu_parity.py::test_rms_normalize_pcm16_bit_exact, ..._inside_the_pair_kernel...).
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_rms_chunk_sums_q(const int16_t* __restrict__ in, long long row_stride, const int32_t* __restrict__ lengths, int n_in,
                                                          float* __restrict__ chunk_sums, int chunk_stride) {
#pragma clang fp contract(off)
    __shared__ __attribute__((aligned(16))) char img[kNpChunk * 2 + 64 * 16];
    __shared__ float heap[256];
    const int row = blockIdx.x, chunk = blockIdx.y, tid = threadIdx.x, lane = tid & 63, q = tid & 3;
    const int n = lengths ? min(lengths[row], n_in) : n_in;
    const int cn = min(kNpChunk, n - chunk * kNpChunk);
    if (cn <= 0) return;                                          // (uniform: a chunk past the row's end)
    const int16_t* x = in + (long long)row * row_stride + (long long)chunk * kNpChunk;
    // ---- stage the chunk: 16 bytes per lane and load where the chunk start is 4-byte aligned (every load requested before the first LDS write)
    if ((reinterpret_cast<uintptr_t>(x) & 3u) == 0) {
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(x), 0, (cn * 2 + 3) & ~3, 0x00020000);   // (past the chunk: zeros)
        uint4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = __builtin_bit_cast(uint4, __builtin_amdgcn_raw_buffer_load_b128(rs, 16 * (tid + 256 * u), 0, 0));
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int d = tid + 256 * u; if (8 * d < cn) *reinterpret_cast<uint4*>(img + np_lds_off<int16_t>(8 * d)) = v[u]; }
    } else {
        for (int e = tid; e < cn; e += 256) *reinterpret_cast<int16_t*>(img + np_lds_off<int16_t>(e)) = x[e];
    }
    __syncthreads();
    // ---- leaves
    int D = 6;                                                     // depth of the deepest leaf; the leftmost path is the shallowest
    bool complete = true;
    if (cn != kNpChunk) {
        D = np_depth(cn);
        int m = cn, d = 0;
        while (m > kNpBlock) { m = np_split(m); ++d; }
        complete = d == D;
    }
    for (int slot = tid >> 2; slot < (1 << D); slot += 64) {       // (one round unless the tree is 7 deep)
        int off, size, lvl, rem = 0;
        if (cn == kNpChunk) { off = slot * kNpBlock; size = kNpBlock; lvl = 6; }
        else {
            np_walk(cn, slot, D, off, size, lvl);
            rem = D - lvl;
            if (slot & ((1 << rem) - 1)) continue;                 // an early leaf belongs to the slot with zero low bits (quad-uniform)
        }
        auto sq = [&](int e) { const float y = (float)*reinterpret_cast<const int16_t*>(img + np_lds_off<int16_t>(e)); return y * y; };
        float res;
        if (size < 8) {
            res = -0.0f;
            for (int i = 0; i < size; ++i) res = res + sq(off + i);
        } else {
            const int body = size - (size & 7);
            uint32_t d[16];
#pragma unroll
            for (int i = 0; i < 16; ++i)                           // (off: a multiple of 8 - every group of 8 lies inside one 128-element LDS row)
                d[i] = 8 * i < body ? *reinterpret_cast<const uint32_t*>(img + np_lds_off<int16_t>(off + 8 * i + 2 * q)) : 0u;
            float r0 = 0.0f, r1 = 0.0f;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float a = (float)(short)(d[i] & 0xFFFFu), c = (float)((int)d[i] >> 16);
                const float sa = a * a, sb = c * c;
                if (i == 0) { r0 = sa; r1 = sb; }
                else if (8 * i < body) { r0 = r0 + sa; r1 = r1 + sb; }
            }
            res = r0 + r1;
            res = res + dpp_xor1(res);
            res = res + dpp_xor2(res);
            for (int i = body; i < size; ++i) res = res + sq(off + i);
        }
        if (q == 0) heap[(1 << lvl) + (slot >> rem)] = res;
    }
    __syncthreads();
    if (tid >= 64) return;
    // ---- the tree, on wave 0
    float root;
    if (complete) {
        float v = lane < (1 << D) ? heap[(1 << D) + lane] : 0.0f;
        int src = 0;
        if (D >= 1) v = v + dpp_f<0xB1>(v);                       // quad_perm [1, 0, 3, 2]
        if (D >= 2) v = v + dpp_f<0x4E>(v);                       // quad_perm [2, 3, 0, 1]
        if (D >= 3) v = v + dpp_f<0x141>(v);                      // row_half_mirror
        if (D >= 4) v = v + dpp_f<0x140>(v);                      // row_mirror
        if (D >= 5) { v = v + dpp_f_rows<0x142, 0xA>(v, 0.0f); src = 31; }      // row_bcast:15 into rows 1, 3
        if (D >= 6) { v = v + dpp_f_rows<0x143, 0xC>(v, 0.0f); src = 63; }      // row_bcast:31 into rows 2, 3
        root = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(v), src));
    } else {
        for (int lvl = 6; lvl >= 0; --lvl) {                       // deepest level first: value = left + right
            if (lane < (1 << lvl)) {
                int off, size, reached;
                np_walk(cn, lane, lvl, off, size, reached);
                if (reached == lvl && size > kNpBlock) heap[(1 << lvl) + lane] = heap[2 * ((1 << lvl) + lane)] + heap[2 * ((1 << lvl) + lane) + 1];
            }
            wave_lds_sync();
        }
        root = heap[1];
    }
    if (lane == 0) chunk_sums[(long long)row * chunk_stride + chunk] = root * (1.0f / 1073741824.0f);     // x 2^-30, exact
}

template <typename T, int CH = NpPass<T>::kChunks>
static constexpr size_t rms_scale_lds() {
    constexpr int kChunksPerPass = CH;
    return (size_t)kChunksPerPass * 256 * 4 + (size_t)kChunksPerPass * kNpChunk * sizeof(T) + (size_t)kChunksPerPass * 64 * 16;
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

// Long rows (more than one LDS pass of k_rms_scale: the reference's 5 s / 48 kHz default, 240 000 samples = 30 numpy
// chunks): the single-workgroup-per-row kernel walked its 8 passes one after the other and then read the row from global
// memory again for the normalised copy - 181 us per 256 rows, 1.4 TB/s.  Here the chunk sums come from (row, pass)
// workgroups of k_rms_scale (chunk_sums mode) and THIS kernel, on (row, tile) workgroups, folds them in numpy's order
// (sequentially, chunk 0 first: the same float32 additions thread 0 of the one-kernel form does), derives the scale with the
// same expressions, and writes its tile of the normalised / padded row (the arithmetic of k_rms_scale's write pass).
// out == nullptr (float32 input): blockIdx.y == 0 only writes scale / ok.
__global__ __launch_bounds__(256) void k_rms_fold_apply(const int16_t* __restrict__ in, long long row_stride, const int32_t* __restrict__ lengths,
                                                        int n_in, const float* __restrict__ chunk_sums, int chunk_stride, int scale_f64,
                                                        float* __restrict__ scale, int32_t* __restrict__ ok, int16_t* __restrict__ norm_out,
                                                        NormPad pad) {
#pragma clang fp contract(off)
    __shared__ float sh[2];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = lengths ? min(lengths[row], n_in) : n_in;
    if (tid == 0) {
        const int n_chunks = (n + kNpChunk - 1) / kNpChunk;
        float total = 0.0f, sc = 1.0f;
        int good = 0;
        for (int c = 0; c < n_chunks; ++c) total = c == 0 ? chunk_sums[(long long)row * chunk_stride] : total + chunk_sums[(long long)row * chunk_stride + c];
        if (n > 0) {
            const float rms = sqrtf(total / (float)n);
            if (scale_f64) { if (!((double)rms < 1e-4)) { sc = (float)(0.05 / ((double)rms + 1e-8)); good = 1; } }
            else if (!(rms < 1e-4f)) { sc = 0.05f / (rms + 1e-8f); good = 1; }
        }
        if (blockIdx.y == 0) { scale[row] = sc; if (ok) ok[row] = good; }
        sh[0] = sc; sh[1] = good ? 1.0f : 0.0f;
    }
    if (!norm_out) return;
    __syncthreads();
    const float sc = sh[0];
    const bool scaled = sh[1] != 0.0f;
    const int16_t* x = in + (long long)row * row_stride;
    int16_t* o = norm_out + (long long)row * (pad.lead > 0 ? pad.out_stride : row_stride);
    auto conv = [&](int q) {
        float v = (float)q * (1.0f / 32768.0f);
        if (scaled) v = fminf(fmaxf(v * sc, -1.0f), 1.0f);
        return (int)rintf(v * 32767.0f);
    };
    const int lead = pad.lead, n_fix = pad.lead > 0 ? pad.n_fix : n_in, nv = min(n, n_fix), total_out = n_fix + 2 * lead;
    auto sample = [&](int idx) {                         // output sample idx of the (padded) row
        int sidx = idx - lead;
        if (lead > 0 && pad.mode == 0) {                 // reflect (np.pad mode="reflect")
            if (sidx < 0) sidx = -sidx;
            if (sidx >= n_fix) sidx = 2 * (n_fix - 1) - sidx;
        }
        if (sidx < 0 || sidx >= nv) return lead > 0 ? 0 : (sidx >= 0 && sidx < n_in ? conv((int)x[sidx]) : 0);
        return conv((int)x[sidx]);
    };
    // this workgroup's share of the output dwords; the body - output dword i = input dword i - lead / 2 - is the fast path
    const int nd = total_out >> 1;
    const int per = (nd + (int)gridDim.y - 1) / (int)gridDim.y;
    const int d0 = blockIdx.y * per, d1 = min(nd, d0 + per);
    const bool al = ((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(x)) & 3u) == 0 && (lead & 1) == 0;
    const int i_lo = lead >> 1, i_hi = i_lo + (nv >> 1);
    if (al) {
        uint32_t* o2 = reinterpret_cast<uint32_t*>(o);
        const uint32_t* x2 = reinterpret_cast<const uint32_t*>(x);
        constexpr int UW = 8;
        for (int base = d0; base < d1; base += 256 * UW) {
            uint32_t v[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) { const int i = base + tid + 256 * u; v[u] = (i < d1 && i >= i_lo && i < i_hi) ? x2[i - i_lo] : 0u; }
#pragma unroll
            for (int u = 0; u < UW; ++u) {
                const int i = base + tid + 256 * u;
                if (i < d1) {
                    int lo, hi;
                    if (i >= i_lo && i < i_hi) { lo = conv((int)(short)(v[u] & 0xFFFFu)); hi = conv((int)v[u] >> 16); }
                    else { lo = sample(2 * i); hi = sample(2 * i + 1); }
                    o2[i] = ((uint32_t)lo & 0xFFFFu) | ((uint32_t)hi << 16);
                }
            }
        }
        if ((total_out & 1) && blockIdx.y == gridDim.y - 1 && tid == 0) o[total_out - 1] = (int16_t)sample(total_out - 1);
    } else {
        for (int i = 2 * d0 + tid; i < min(total_out, 2 * d1 + ((blockIdx.y == gridDim.y - 1) ? 1 : 0)); i += 256) o[i] = (int16_t)sample(i);
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

// The same 16-point DFT on PACKED PAIRS (round 3): T = float2 holding element j of TWO sequences a lane transforms side by side
// (the pair kernels: sequences e = 0 / 1 of stage 1, p = 0 / 1 of stage 2).  Every operation of the scalar form becomes one
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 with the constants broadcast from SGPRs - half the instructions.  What bounds the STFT
// kernels is the instruction stream of ONE wave (tools/probe/valu_rate.hip: a wave issues a vector instruction every ~5 cycles
// whether it is alone on its SIMD or one of four; v_pk_fma_f32 5.5), so instructions per frame are time.
typedef float amph_f2 __attribute__((ext_vector_type(2)));
#define AMPH_BFLY4_T(r0, i0, r1, i1, r2, i2, r3, i3)                   \
    {                                                                  \
        const auto ar = r0 + r2, ai = i0 + i2;                         \
        const auto br = r0 - r2, bi = i0 - i2;                         \
        const auto cr = r1 + r3, ci = i1 + i3;                         \
        const auto dr = i1 - i3, di = r3 - r1;                         \
        r0 = ar + cr; i0 = ai + ci;                                    \
        r1 = br + dr; i1 = bi + di;                                    \
        r2 = ar - cr; i2 = ai - ci;                                    \
        r3 = br - dr; i3 = bi - di;                                    \
    }
#define AMPH_CMUL_T(r, i, wr, wi)                                      \
    {                                                                  \
        const auto tr_ = r * (wr) - i * (wi);                          \
        i = r * (wi) + i * (wr);                                       \
        r = tr_;                                                       \
    }
__device__ __forceinline__ void dft16_pk(amph_f2 (&re)[16], amph_f2 (&im)[16]) {
    constexpr float C1 = 0.92387953251128675613f;   // cos(pi/8)
    constexpr float S1 = 0.38268343236508977173f;   // sin(pi/8)
    constexpr float R2 = 0.70710678118654752440f;   // sqrt(1/2)
#pragma unroll
    for (int j0 = 0; j0 < 4; ++j0)
        AMPH_BFLY4_T(re[j0], im[j0], re[j0 + 4], im[j0 + 4], re[j0 + 8], im[j0 + 8], re[j0 + 12], im[j0 + 12]);
    AMPH_CMUL_T(re[1 + 4], im[1 + 4], C1, -S1);
    AMPH_CMUL_T(re[2 + 4], im[2 + 4], R2, -R2);
    AMPH_CMUL_T(re[3 + 4], im[3 + 4], S1, -C1);
    AMPH_CMUL_T(re[1 + 8], im[1 + 8], R2, -R2);
    { const amph_f2 t = re[2 + 8]; re[2 + 8] = im[2 + 8]; im[2 + 8] = -t; }
    AMPH_CMUL_T(re[3 + 8], im[3 + 8], -R2, -R2);
    AMPH_CMUL_T(re[1 + 12], im[1 + 12], S1, -C1);
    AMPH_CMUL_T(re[2 + 12], im[2 + 12], -R2, -R2);
    AMPH_CMUL_T(re[3 + 12], im[3 + 12], -C1, S1);
#pragma unroll
    for (int ka = 0; ka < 4; ++ka)
        AMPH_BFLY4_T(re[4 * ka], im[4 * ka], re[4 * ka + 1], im[4 * ka + 1], re[4 * ka + 2], im[4 * ka + 2], re[4 * ka + 3], im[4 * ka + 3]);
}
// Y[k + 16 p] = E[k] +- W32^k O[k] with E = the .x and O = the .y sequence of every pair, in place (.x <- Y[k], .y <- Y[k + 16]).
// The operands of one butterfly sit in the two HALVES of a register pair, so plain packed arithmetic cannot reach them; the VOP3P
// operand-half selectors can (op_sel / op_sel_hi: which half of each source feeds the low / the high result; neg_lo / neg_hi):
//   tmp = (or c, -or s)                v_pk_mul_f32  re[.y, .y] * P[c | s]
//   T   = (oi s + or c, oi c - or s)   v_pk_fma_f32  im[.y, .y] * P[s | c] + tmp         T = W32^k O = (tr, ti)
//   re  = (er + tr, er - tr)           v_pk_add_f32  re[.x, .x] + T[.x, -.x]
//   im  = (ei + ti, ei - ti)           v_pk_add_f32  im[.x, .x] + T[.y, -.y]
// four instructions per k instead of eight scalar ones (1 228 -> 1 031 vector instructions per pair; a SIMD's first wave runs its
// pairs 4 % sooner, the aggregate is set by the LDS pipe: profiles/r03_stft_pair_timeline.txt).  The constant
// pair P = (cos, sin)(pi j / 16), j = 1..4, serves all k through the selectors: c, s = P.x, P.y (k = j), P.y, P.x (k = 8 - j),
// -P.y, P.x (k = 8 + j), -P.x, P.y (k = 16 - j) - four SGPR pairs.  tools/probe/pk_opsel.hip checks the selector semantics.
template <int K>
__device__ __forceinline__ void dft32_comb_k(amph_f2& re, amph_f2& im) {
    amph_f2 r2, i2;
    if constexpr (K == 0) {                              // T = O
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r2) : "v"(re));
        asm("v_pk_add_f32 %0, %1, %1 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(i2) : "v"(im));
    } else if constexpr (K == 8) {                       // T = -i O = (oi, -or)
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r2) : "v"(re), "v"(im));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(i2) : "v"(im), "v"(re));
    } else {
        constexpr int J = K < 5 ? K : K < 8 ? 8 - K : K < 13 ? K - 8 : 16 - K;
        constexpr bool SWAP = (K >= 5 && K <= 7) || (K >= 9 && K <= 12), CNEG = K > 8;
        constexpr float CJ[5] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f, 0.70710678118654752440f};
        constexpr float SJ[5] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f, 0.70710678118654752440f};
        const amph_f2 P = {CJ[J], SJ[J]};
        amph_f2 tmp, T;
        if constexpr (!SWAP && !CNEG) {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(tmp) : "v"(re), "s"(P));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,0] neg_hi:[0,0,0]" : "=v"(T) : "v"(im), "s"(P), "v"(tmp));
        } else if constexpr (SWAP && !CNEG) {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(tmp) : "v"(re), "s"(P));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,0,0] neg_hi:[0,0,0]" : "=v"(T) : "v"(im), "s"(P), "v"(tmp));
        } else if constexpr (SWAP && CNEG) {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(tmp) : "v"(re), "s"(P));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,1,1] neg_lo:[0,0,0] neg_hi:[0,1,0]" : "=v"(T) : "v"(im), "s"(P), "v"(tmp));
        } else {
            asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(tmp) : "v"(re), "s"(P));
            asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,0] neg_hi:[0,1,0]" : "=v"(T) : "v"(im), "s"(P), "v"(tmp));
        }
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r2) : "v"(re), "v"(T));
        asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(i2) : "v"(im), "v"(T));
    }
    re = r2; im = i2;
}
__device__ __forceinline__ void dft32_combine_pk(amph_f2 (&re)[16], amph_f2 (&im)[16]) {
#define AMPH_COMB(k_) dft32_comb_k<k_>(re[DFT16_POS(k_)], im[DFT16_POS(k_)]);
    AMPH_COMB(0) AMPH_COMB(1) AMPH_COMB(2) AMPH_COMB(3) AMPH_COMB(4) AMPH_COMB(5) AMPH_COMB(6) AMPH_COMB(7)
    AMPH_COMB(8) AMPH_COMB(9) AMPH_COMB(10) AMPH_COMB(11) AMPH_COMB(12) AMPH_COMB(13) AMPH_COMB(14) AMPH_COMB(15)
#undef AMPH_COMB
}
// LDS stores with ONE base register and a 16-bit byte offset.  Left to the compiler, neighbouring float stores are paired into
// ds_write2_b32, whose 8-bit offsets reach 1 KB: the 32 rows of an exchange tile then need sixteen loop-invariant base registers
// (~35 over the pair kernel), which at three waves per SIMD are spilled and reloaded in front of every store.  `base` is the
// LDS byte address (low 32 bits of the generic pointer).  The "memory" clobber keeps the compiler's own LDS accesses on their side
// of the store; its s_waitcnt counts only over-wait for these (LDS operations of a wave complete in order).
#define AMPH_DS_ST32(base_, off_, val_) asm volatile("ds_write_b32 %0, %1 offset:%2" : : "v"(base_), "v"(val_), "n"(off_) : "memory")
#define AMPH_DS_ST64(base_, off_, val_) asm volatile("ds_write_b64 %0, %1 offset:%2" : : "v"(base_), "v"(val_), "n"(off_) : "memory")

// sample s of a row as float32 before normalisation (0 beyond the valid length: the zero
// pad of fix_length, map_detector_core.py:214-215)
template <bool PCM16>
__device__ __forceinline__ float load_sample(const void* rowp, int s, int len) {
    if (s < 0 || s >= len) return 0.0f;
    if (PCM16) return (float)reinterpret_cast<const int16_t*>(rowp)[s] * (1.0f / 32768.0f);
    return reinterpret_cast<const float*>(rowp)[s];
}
// RMS scale + clip + the PCM16 write/read of 00_normalize (x*32767, rint, /32768)
__device__ __forceinline__ float normalise_sample(const WaveSrc& src, float x, float sc) {
    if (src.scale) {
        x = fminf(fmaxf(x * sc, -1.0f), 1.0f);
        if (src.requant) x = rintf(x * 32767.0f) * (1.0f / 32768.0f);
    }
    return x;
}

// Diagnostic build only (-DAMPH_STAMPS, `make stamps`): wave 0 of every workgroup stamps s_memtime at
// the phase boundaries of its first frame; the stamps go to a buffer nothing else reads.  Never timed.
#ifdef AMPH_STAMPS
#define AMPH_STAMP(k)                                                                       \
    if (wave == 0 && fl == 0) {                                                             \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (lane == 0) p.stamps[(size_t)blockIdx.x * 16 + (k)] = t_;                        \
    }
#else
#define AMPH_STAMP(k)
#endif

// Geometry of the wave FFT for n_fft = 512 * T_LO (tools/fft_lane_model.py): NC = 256 * T_LO complex
// points per frame on L = 16 * T_LO lanes, so a wave transforms 64 / L frames at once (T_LO = 4: one
// frame per wave; T_LO = 2: one frame per half-wave).  The last stage is a radix-T_LO across adjacent lanes.
template <int T_LO>
struct FftGeo {
    static constexpr int L = 16 * T_LO;
    static constexpr int NC = 256 * T_LO;
    static constexpr int SLOTS = 64 / L;                       // frames per wave per iteration
    static constexpr int EXS = T_LO == 4 ? 68 : 34;            // exchange row stride (float2): conflict-free both ways
    static constexpr int SLOT_BYTES = 16 * EXS * 8;            // exchange tile, then Z[NC + pads], then P[NC + 1 + tail]
    static constexpr int WIN_BYTES = ((NC + 1) * 4 + 15) & ~15;        // w[0 .. NC] of the periodic Hann window (w[n] == w[N - n])
    static constexpr int TWB_BYTES = 16 * T_LO * 8;                    // W_L^(t_lo*k2a)
    static constexpr int TWP_BYTES = ((NC / 2 + 1) * 8 + 15) & ~15;    // W_{2NC}^k, k = 0 .. NC/2
    static constexpr int TABLE_BYTES = WIN_BYTES + TWB_BYTES + TWP_BYTES;
};
constexpr int kWaveScratch = 8704;                    // = SLOTS * SLOT_BYTES for both geometries
// W_NC^(t*k1), k1 = 1..15, [15][L] float2 stays in global memory (L1/L2 resident): its 15 loads per
// frame are independent of everything and are issued at the top of the frame, under the first DFT.
// The mel weights, whose loads sat on the critical path of every frame, live in LDS instead.

// Persistent: gridDim.x workgroups (2 per CU) walk the (row, frame tile) list with stride gridDim.x.
// The tables and mel rows are staged into LDS once per workgroup (staged per tile they were 37 MB of
// L2 traffic per launch, fetched by every workgroup at the same moment); the next tile's samples are
// prefetched into registers while the current tile's FFTs run.
template <bool PCM16, int T_LO, int NW>
__global__ __launch_bounds__(64 * NW) void k_stft_mel(StftParams p, int n_tiles) {
    constexpr int kThreads = 64 * NW;                  // NW waves per workgroup
    using G = FftGeo<T_LO>;
    constexpr int kNc = G::NC, L = G::L, SLOTS = G::SLOTS, EXS = G::EXS;
    constexpr int kWinHalfBytes = G::WIN_BYTES, kTwBBytes = G::TWB_BYTES, kTableBytes = G::TABLE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane_f = lane & (L - 1), slot = lane / L;       // lane inside its frame, frame slot of the wave
    const int FT = p.frames_per_block;
#ifdef AMPH_STAMPS
    { const int fl = 0; AMPH_STAMP(0) }
    unsigned long long w_t0 = 0, w_ta = 0, w_t1 = 0, w_t2 = 0, w_t3 = 0, w_stage = 0, w_stagework = 0, w_frames = 0, w_endbar = 0, w_begin = 0;
    const unsigned long long w_rt0 = __builtin_amdgcn_s_memrealtime();
#define AMPH_WSTAMP(v_)                                                                     \
    {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    AMPH_WSTAMP(w_begin)
#else
#define AMPH_WSTAMP(v_)
#endif
    constexpr int n_fft = 2 * kNc;
    const int span_len = (FT - 1) * p.hop + n_fft;
    const int span_bytes = (span_len * 4 + 15) & ~15;
    float* win_h = reinterpret_cast<float*>(smem);
    float2* tw_b = reinterpret_cast<float2*>(smem + kWinHalfBytes);
    float2* tw_p = reinterpret_cast<float2*>(smem + kWinHalfBytes + kTwBBytes);
    const float2* tw_a = p.tw_a;
    const int meta_bytes = (p.n_mels * 12 + 15) & ~15;
    const int melw_bytes = p.mel_w_floats * 4;                             // multiple of 64
    int3* mel_meta = reinterpret_cast<int3*>(smem + kTableBytes);          // (start4, steps, wofs) per filter
    float* mel_w = reinterpret_cast<float*>(smem + kTableBytes + meta_bytes);
    float* span = reinterpret_cast<float*>(smem + kTableBytes + meta_bytes + melw_bytes);
    char* wbase = smem + kTableBytes + meta_bytes + melw_bytes + span_bytes + wave * kWaveScratch;
    float2* ex = reinterpret_cast<float2*>(wbase + slot * G::SLOT_BYTES);   // this frame slot's scratch
    float* P = reinterpret_cast<float*>(wbase + slot * G::SLOT_BYTES);

    // ---- once per workgroup: tables (one packed global array = the LDS image) and the mel rows
    {
        const float4* src = reinterpret_cast<const float4*>(p.tables);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < kTableBytes / 16; i += kThreads) dst[i] = src[i];
        for (int m = tid; m < p.n_mels; m += kThreads) mel_meta[m] = make_int3(p.mel_start[m], p.mel_len[m], p.mel_wofs[m]);
        const float4* wsrc = reinterpret_cast<const float4*>(p.mel_w);
        float4* wdst = reinterpret_cast<float4*>(mel_w);
        for (int i = tid; i < (p.mel_w_floats >> 2); i += kThreads) wdst[i] = wsrc[i];
    }

    // per-tile source description
    struct TileSrc { const char* rowp; int len, s0, row, f_base; float sc; bool fast; };
    auto describe = [&](int t) {
        TileSrc d;
        const int r = t / p.tiles_per_row, tl = t - r * p.tiles_per_row;
        d.rowp = reinterpret_cast<const char*>(p.src.wave) + (long long)r * p.src.row_stride * (PCM16 ? 2 : 4);
        int len = p.src.lengths ? min(p.src.lengths[r], p.src.n_in) : p.src.n_in;
        d.len = min(len, p.n_samples);               // truncation of fix_length (:217)
        d.sc = p.src.scale ? p.src.scale[r] : 1.0f;
        d.s0 = tl * FT * p.hop - (n_fft >> 1);
        d.row = r;
        d.f_base = tl * FT;
        // interior tile of a PCM16 row: one contiguous, 4-byte aligned run -> dword loads
        d.fast = PCM16 && d.s0 >= 0 && d.s0 + span_len <= d.len && (span_len & 1) == 0 &&
                 ((reinterpret_cast<uintptr_t>(d.rowp) + 2 * (size_t)d.s0) & 3u) == 0;
        return d;
    };
    constexpr int U = NW == 4 ? 10 : 5;                // dwords per thread per batch (one batch for FT = 8 / 12)
    constexpr int US = 2 * U;                          // samples per thread per tile on the per-sample path
    // next tile's samples, prefetched while this tile's frames are transformed.  pre_kind 1: dwords (two PCM16
    // samples) of an interior PCM16 tile; 2: one decoded, not yet normalised sample per register - the edge tiles of
    // a row (reflect / zero padding: two of the five tiles of a 1 s row) and every tile of a float32 row, whose
    // synchronous loads were ~2.5 k exposed cycles per tile in front of the first barrier
    uint32_t pre[US];
#pragma unroll
    for (int u = 0; u < US; ++u) pre[u] = 0;
    int pre_kind = 0;
    const int nd = span_len >> 1;
    const bool pre_fits = nd <= kThreads * U;

    TileSrc d = describe(min((int)blockIdx.x, n_tiles - 1)), dn = d;      // the next tile's description is computed once
    for (int t = blockIdx.x; t < n_tiles; t += gridDim.x) {
        AMPH_WSTAMP(w_t0)
        const int row = d.row, f_base = d.f_base;
        // ---- stage this tile's sample span (decode / normalise on the way into LDS)
        if (d.fast) {
            const uint32_t* src32 = reinterpret_cast<const uint32_t*>(d.rowp + 2 * (size_t)d.s0);
            for (int base = 0; base < nd; base += kThreads * U) {
                uint32_t v[U];
                if (pre_kind == 1) {
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = pre[u];
                } else {
#pragma unroll
                    for (int u = 0; u < U; ++u) v[u] = src32[min(base + tid + kThreads * u, nd - 1)];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = base + tid + kThreads * u;
                    if (e < nd) {
                        const float lo = (float)(short)(v[u] & 0xFFFFu) * (1.0f / 32768.0f);
                        const float hi = (float)((int)v[u] >> 16) * (1.0f / 32768.0f);
                        *reinterpret_cast<float2*>(span + 2 * e) =
                            make_float2(normalise_sample(p.src, lo, d.sc), normalise_sample(p.src, hi, d.sc));
                    }
                }
            }
        } else if (pre_kind == 2) {
#pragma unroll
            for (int u = 0; u < US; ++u) {
                const int i = tid + kThreads * u;
                if (i < span_len) span[i] = normalise_sample(p.src, __uint_as_float(pre[u]), d.sc);
            }
        } else {
            const int N = p.n_samples;
            for (int base = 0; base < span_len; base += kThreads * U) {
                float x[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    int sidx = d.s0 + base + tid + kThreads * u;
                    if (p.pad_mode == 0) {                // reflect (np.pad mode="reflect")
                        if (sidx < 0) sidx = -sidx;
                        if (sidx >= N) sidx = 2 * (N - 1) - sidx;
                    } else if (sidx >= N) sidx = -1;      // constant (zero) padding
                    x[u] = load_sample<PCM16>(d.rowp, sidx, d.len);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int i = base + tid + kThreads * u;
                    if (i < span_len) span[i] = normalise_sample(p.src, x[u], d.sc);
                }
            }
        }
        AMPH_WSTAMP(w_ta)
        __syncthreads();
        AMPH_WSTAMP(w_t1)
        // ---- prefetch the next tile's samples; they land while this tile's frames are transformed
        pre_kind = 0;
        if (t + (int)gridDim.x < n_tiles) dn = describe(t + gridDim.x);
        if (t + (int)gridDim.x < n_tiles && pre_fits) {
            if (dn.fast) {
                const uint32_t* src32 = reinterpret_cast<const uint32_t*>(dn.rowp + 2 * (size_t)dn.s0);
#pragma unroll
                for (int u = 0; u < U; ++u) pre[u] = src32[min(tid + kThreads * u, nd - 1)];
                pre_kind = 1;
            } else {
                const int N = p.n_samples;
#pragma unroll
                for (int u = 0; u < US; ++u) {
                    int sidx = dn.s0 + tid + kThreads * u;
                    if (p.pad_mode == 0) {                // reflect (np.pad mode="reflect")
                        if (sidx < 0) sidx = -sidx;
                        if (sidx >= N) sidx = 2 * (N - 1) - sidx;
                    } else if (sidx >= N) sidx = -1;      // constant (zero) padding
                    pre[u] = __float_as_uint(load_sample<PCM16>(dn.rowp, sidx, dn.len));
                }
                pre_kind = 2;
            }
        }

    const int k1b = lane_f / T_LO, t_lo = lane_f % T_LO;
    // stage C leaves k2b bit-reversed across the quad (radix-4); the radix-2 form is in order
    const int k2b = T_LO == 4 ? (((t_lo & 1) << 1) | (t_lo >> 1)) : t_lo;
    const float sgn_a = t_lo < 2 ? 1.0f : -1.0f;          // first butterfly (lanes 2 apart): x + p or p - x
    const float sgn_b = (t_lo & 1) == 0 ? 1.0f : -1.0f;   // second butterfly (adjacent lanes)
    const bool rot3 = T_LO == 4 && t_lo == 3;
    for (int fl0 = wave * SLOTS; fl0 < FT; fl0 += NW * SLOTS) {
        if (f_base + fl0 >= p.n_frames) break;        // wave-uniform: the wave's first frame slot is past the end
        const int fl = fl0 + slot;
        const int frame = f_base + fl;
        const bool fvalid = fl < FT && frame < p.n_frames;    // the second half-wave may hold no frame
        AMPH_STAMP(1)
        float re[16], im[16];
        // ---- window + even/odd packing: z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1], n = lane_f + L j
        {
            const float2* xs = reinterpret_cast<const float2*>(span + min(fl, FT - 1) * p.hop);
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = lane_f + L * j;
                const float2 x = xs[n];
                float w0, w1;
                if (j < 8) { const float2 w = reinterpret_cast<const float2*>(win_h)[n]; w0 = w.x; w1 = w.y; }
                else { const int m = n_fft - 2 * n; w0 = win_h[m]; w1 = win_h[m - 1]; }   // w[2n] = w[N-2n], w[2n+1] = w[N-2n-1]
                re[j] = x.x * w0;
                im[j] = x.y * w1;
            }
        }
        AMPH_STAMP(2)
        // ---- stage A
        dft16(re, im);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            float r = re[DFT16_POS(k1)], i = im[DFT16_POS(k1)];
            if (k1 > 0) {
                const float2 w = tw_a[(k1 - 1) * L + lane_f];
                AMPH_CMUL(r, i, w.x, w.y);
            }
            ex[k1 * EXS + lane_f] = make_float2(r, i);
        }
        wave_lds_sync();
        AMPH_STAMP(3)
        // ---- stage B: lane (k1b, t_lo) gathers t = t_lo + T_LO m
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 v = ex[k1b * EXS + t_lo + T_LO * m];
            re[m] = v.x;
            im[m] = v.y;
        }
        AMPH_STAMP(4)
        dft16(re, im);
        AMPH_STAMP(5)
        wave_lds_sync();                              // the Z image below overwrites `ex`
        float2* Z = ex;
#pragma unroll
        for (int k2a = 0; k2a < 16; ++k2a) {
            float r = re[DFT16_POS(k2a)], i = im[DFT16_POS(k2a)];
            if (k2a > 0) {
                const float2 w = tw_b[k2a * T_LO + t_lo];
                AMPH_CMUL(r, i, w.x, w.y);
            }
            // ---- stage C: radix-T_LO over adjacent lanes (DPP xor butterflies).  The lane-dependent add / subtract
            //      is one fma with a per-lane sign (x * (+-1) + partner rounds exactly as x + partner / partner - x),
            //      the -i of lane 3 two selects
            if (T_LO == 4) {
                const float nr = fmaf(r, sgn_a, dpp_xor2(r)), ni = fmaf(i, sgn_a, dpp_xor2(i));
                const float tr = rot3 ? ni : nr, ti = rot3 ? -nr : ni;          // lane 3: * (-i)
                r = fmaf(tr, sgn_b, dpp_xor1(tr));
                i = fmaf(ti, sgn_b, dpp_xor1(ti));
            } else {
                r = fmaf(r, sgn_b, dpp_xor1(r));
                i = fmaf(i, sgn_b, dpp_xor1(i));
            }
            // lane holds X[k1b + 16 k2a + 256 k2b]
            Z[k1b + 16 * k2a + (256 + kZPad) * k2b] = make_float2(r, i);
        }
        wave_lds_sync();
        AMPH_STAMP(6)
        // ---- real-FFT split: bins k and NC - k from Z[k], Z[NC - k]; the power spectrum
        //      is written over the Z image, so every pair is read into registers first
        {
            float2 za[9], zb[9];
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int k = min(lane_f + L * it, kNc / 2);
                const int k2 = (kNc - k) & (kNc - 1);
                za[it] = Z[k + kZPad * (k >> 8)];
                zb[it] = Z[k2 + kZPad * (k2 >> 8)];
            }
            wave_lds_sync();
            AMPH_STAMP(7)
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int k = lane_f + L * it;
                if (k <= kNc / 2) {
                    const float2 a = za[it], b = zb[it];
                    const float2 w = tw_p[k];
                    const float xer = 0.5f * (a.x + b.x), xei = 0.5f * (a.y - b.y);      // (Z[k] + conj Z[N-k]) / 2
                    const float xor_ = 0.5f * (a.y + b.y), xoi = -0.5f * (a.x - b.x);    // -i (Z[k] - conj Z[N-k]) / 2
                    const float tr = xor_ * w.x - xoi * w.y, ti = xor_ * w.y + xoi * w.x;
                    const float pr = xer + tr, pi = xei + ti, mr = xer - tr, mi = xei - ti;
                    P[k] = pr * pr + pi * pi;
                    if (k != kNc - k) P[kNc - k] = mr * mr + mi * mi;
                }
            }
            for (int i = lane_f; i < p.p_tail; i += L) P[kNc + 1 + i] = 0.0f;   // zero the tail the padded mel rows touch
        }
        wave_lds_sync();
        AMPH_STAMP(8)
        // ---- mel rows: 4 lanes per filter, each lane 4 consecutive bins per step (aligned float4).
        //      The host pads every filter of a 16-filter round to the round's longest row (zero
        //      weights), so the step count is wave-uniform: no per-lane predication, no branches.
        for (int sl = 0; sl < SLOTS; ++sl) {            // all 64 lanes work on one frame slot at a time
            const int fr = f_base + fl0 + sl;
            if (fl0 + sl >= FT || fr >= p.n_frames) break;    // wave-uniform
            float* out = p.melpow + ((long long)row * p.n_frames + fr) * p.n_mels;
            const float* Ps = reinterpret_cast<const float*>(wbase + sl * G::SLOT_BYTES);
            const int q = lane & 3;
            for (int m0 = 0; m0 < p.n_mels; m0 += 16) {
                const int m = min(m0 + (lane >> 2), p.n_mels - 1);
                const int3 meta = mel_meta[m];                                // (start4, steps of the round, wofs)
                const int steps = __builtin_amdgcn_readfirstlane(meta.y);
                const float4* w = reinterpret_cast<const float4*>(mel_w + meta.z) + q;
                const float4* pp = reinterpret_cast<const float4*>(Ps + meta.x) + q;
                float acc0 = 0.0f, acc1 = 0.0f;
                int i = 0;
                for (; i + 1 < steps; i += 2) {
                    const float4 a0 = w[4 * i], b0 = pp[4 * i], a1 = w[4 * i + 4], b1 = pp[4 * i + 4];
                    acc0 = fmaf(a0.x, b0.x, acc0); acc1 = fmaf(a1.x, b1.x, acc1);
                    acc0 = fmaf(a0.y, b0.y, acc0); acc1 = fmaf(a1.y, b1.y, acc1);
                    acc0 = fmaf(a0.z, b0.z, acc0); acc1 = fmaf(a1.z, b1.z, acc1);
                    acc0 = fmaf(a0.w, b0.w, acc0); acc1 = fmaf(a1.w, b1.w, acc1);
                }
                if (i < steps) {
                    const float4 a0 = w[4 * i], b0 = pp[4 * i];
                    acc0 = fmaf(a0.x, b0.x, acc0);
                    acc0 = fmaf(a0.y, b0.y, acc0);
                    acc0 = fmaf(a0.z, b0.z, acc0);
                    acc0 = fmaf(a0.w, b0.w, acc0);
                }
                float acc = acc0 + acc1;
                acc += dpp_xor1(acc);
                acc += dpp_xor2(acc);
                if (q == 0 && m0 + (lane >> 2) < p.n_mels) out[m] = acc;
            }
        }
        (void)fvalid;
        wave_lds_sync();                              // the scratch is rewritten by the next frame
        AMPH_STAMP(9)
    }
        AMPH_WSTAMP(w_t2)
        __syncthreads();                              // every wave is done with `span` before it is restaged
        AMPH_WSTAMP(w_t3)
#ifdef AMPH_STAMPS
        w_stage += w_t1 - w_t0; w_stagework += w_ta - w_t0; w_frames += w_t2 - w_t1; w_endbar += w_t3 - w_t2;
#endif
        d = dn;
    }
#ifdef AMPH_STAMPS
    if (tid == 0) {       // per-workgroup timeline (wave 0): where the tiles' time goes, and the in-kernel clock
        unsigned long long w_end;
        AMPH_WSTAMP(w_end)
        p.stamps[(size_t)blockIdx.x * 16 + 10] = w_stage;
        p.stamps[(size_t)blockIdx.x * 16 + 11] = w_frames;
        p.stamps[(size_t)blockIdx.x * 16 + 12] = w_endbar;
        p.stamps[(size_t)blockIdx.x * 16 + 13] = w_end - w_begin;
        p.stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime() - w_rt0;
        p.stamps[(size_t)blockIdx.x * 16 + 15] = w_stagework;      // staging before its barrier (wave 0)
    }
#endif
}

// K3 inside the STFT kernel: power_to_db(ref=max, amin=1e-10, top_db=80) -> z-score over the whole [M, T_raw] array
// (population std) -> centre crop / zero pad -> [T, M] or [M, T] (map_detector_core.py:229-237, 185-195) for ONE row whose
// mel power [n_frames][n_mels] the workgroup holds in LDS.  Same arithmetic as k_dbnorm (separately rounded dB terms, sums
// in double, two-pass std); the mel power never reaches HBM (-3.7 MB written, -1.9 MB read and one launch less per 256 rows).
__device__ __forceinline__ float db10(float x);
__device__ __forceinline__ float to_db(float s, float ref_db);
// HAVE_MAX: the caller's waves have left the maxima of the values they wrote in redf[0 .. NT / 64) (and passed a barrier): no max pass
template <int NT, int C = 8, bool HAVE_MAX = false>
__device__ __forceinline__ void db_row_from_lds(const float* mp, int T, int M, int TT, int layout, float* out, double* red, unsigned long long* dbg = nullptr) {
#ifdef AMPH_STAMPS
#define AMPH_DB_T(k_) if (dbg && threadIdx.x == 0) { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); dbg[k_] = t_; }
#else
#define AMPH_DB_T(k_)
#endif
    constexpr int NWV = NT / 64;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = T * M;
    float* redf = reinterpret_cast<float*>(red + 16);
    // C values kept in registers per thread; beyond that recomputed per pass (8: a 58 x 64 row on 768 threads; the 87 x 128 rows of
    // the 128-band preset on 512 threads take 24)
    float v[C];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < C; ++j) { const int i = tid + NT * j; v[j] = i < n ? mp[i] : -INFINITY; mx = fmaxf(mx, v[j]); }
    if (!HAVE_MAX) {
        for (int i = tid + NT * C; i < n; i += NT) mx = fmaxf(mx, mp[i]);
        mx = wave_max_dpp(mx);
        if (lane == 63) redf[w] = mx;
        __syncthreads();
    }
    AMPH_DB_T(0)
    mx = redf[0];
#pragma unroll
    for (int g = 1; g < NWV; ++g) mx = fmaxf(mx, redf[g]);
    const float ref_db = db10(fmaxf(1e-10f, mx));             // same two operations as to_db (no FMA)
    double acc = 0.0;
#pragma unroll
    for (int j = 0; j < C; ++j) if (tid + NT * j < n) { v[j] = to_db(v[j], ref_db); acc += (double)v[j]; }
    for (int i = tid + NT * C; i < n; i += NT) acc += (double)to_db(mp[i], ref_db);
    AMPH_DB_T(1)
    acc = wave_sum_dpp(acc);
    if (lane == 63) red[w] = acc;
    __syncthreads();
    AMPH_DB_T(2)
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < NWV; ++g) tot += red[g];
    const float mean = (float)(tot / (double)n);
    acc = 0.0;
#pragma unroll
    for (int j = 0; j < C; ++j) if (tid + NT * j < n) { const float dlt = v[j] - mean; acc += (double)(dlt * dlt); }
    for (int i = tid + NT * C; i < n; i += NT) { const float dlt = to_db(mp[i], ref_db) - mean; acc += (double)(dlt * dlt); }
    acc = wave_sum_dpp(acc);
    // the second sums go BESIDE the first ones when there are at most 12 waves (doubles 12 .. 23 of the scratch: the unused tail of red and
    // redf, whose last read - the row maximum - lies behind the barrier above): no barrier to wait for the readers of the first sums
    double* const red2 = NWV <= 12 ? red + 12 : red;
    if (NWV > 12) __syncthreads();                       // every wave has read the first sums
    if (lane == 63) red2[w] = acc;
    __syncthreads();
    AMPH_DB_T(3)
    tot = 0.0;
#pragma unroll
    for (int g = 0; g < NWV; ++g) tot += red2[g];
    const float sd = sqrtf((float)(tot / (double)n));
    const float den = sd + 1e-8f;
    const float rden = 1.0f / den;                       // (x / den as x * (1 / den): one IEEE division per row, 1 ulp on z)
    AMPH_DB_T(4)
    // crop_or_pad_time (map_detector_core.py:185-195): destination frame = source frame - shift
    const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);
    const int m_sh = (M & (M - 1)) == 0 ? __builtin_ctz(M) : -1;        // band counts are powers of two in every preset: e / M as a shift
    // source-major, like k_dbnorm's cached form: the dB values of passes 2 and 3 are still in registers (no third log10)
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int e = tid + NT * j;
        if (e < n) {
            const int fr = m_sh >= 0 ? e >> m_sh : e / M, m = e - fr * M, tt = fr - shift;
            if (tt >= 0 && tt < TT) {
                const float z = (v[j] - mean) * rden;
                if (layout == 1) out[tt * M + m] = z; else out[m * TT + tt] = z;
            }
        }
    }
    for (int e = tid + NT * C; e < n; e += NT) {         // rows larger than the register cache
        const int fr = e / M, m = e - fr * M, tt = fr - shift;
        if (tt >= 0 && tt < TT) {
            const float z = (to_db(mp[e], ref_db) - mean) * rden;
            if (layout == 1) out[tt * M + m] = z; else out[m * TT + tt] = z;
        }
    }
    if (T < TT) {                                        // literal zero padding, left then right
        const int left = -shift, right0 = left + T;
        for (int i = tid; i < (TT - T) * M; i += NT) {
            const int fr = i / M, m = i - fr * M;
            const int tt = fr < left ? fr : right0 + (fr - left);
            if (layout == 1) out[tt * M + m] = 0.0f; else out[m * TT + tt] = 0.0f;
        }
    }
}

template <int NT, int C = 8, bool HAVE_MAX = false>
__device__ __attribute__((noinline)) void db_row_call(const float* mp, int T, int M, int TT, int layout, float* out, double* red) {
    db_row_from_lds<NT, C, HAVE_MAX>(mp, T, M, TT, layout, out, red);
}

// TWO rows at once (k_stft_pair's G2 form: both rows' mel power in LDS, their maxima already in each row's redf): every pass handles the
// two rows side by side, so the pair costs one row's barriers and latencies.  Per row the thread -> element map, the per-thread sums,
// the wave sums and the order of the cross-wave totals are db_row_from_lds<NT, C, true>'s: the same bits.  red: row r's scratch at red + 24 r.
template <int NT, int C = 8>
__device__ __attribute__((noinline)) void db_rows2_call(const float* mp0, const float* mp1, int T, int M, int TT, int layout, float* out0, float* out1, double* red) {
    constexpr int NWV = NT / 64;
    static_assert(NWV <= 12, "the second sums sit beside the first ones");
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int n = T * M;
    const float* mp[2] = {mp0, mp1};
    float* out[2] = {out0, out1};
    float v[2][C], ref_db[2], mean[2], den[2];
    double acc[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
#pragma unroll
        for (int j = 0; j < C; ++j) { const int i = tid + NT * j; v[r][j] = i < n ? mp[r][i] : -INFINITY; }
        const float* redf = reinterpret_cast<const float*>(red + 24 * r + 16);
        float mx = redf[0];
#pragma unroll
        for (int g = 1; g < NWV; ++g) mx = fmaxf(mx, redf[g]);
        ref_db[r] = db10(fmaxf(1e-10f, mx));
    }
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < C; ++j) if (tid + NT * j < n) { v[r][j] = to_db(v[r][j], ref_db[r]); a += (double)v[r][j]; }
        for (int i = tid + NT * C; i < n; i += NT) a += (double)to_db(mp[r][i], ref_db[r]);
        acc[r] = wave_sum_dpp(a);
    }
    if (lane == 63) { red[w] = acc[0]; red[24 + w] = acc[1]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < NWV; ++g) tot += red[24 * r + g];
        mean[r] = (float)(tot / (double)n);
        double a = 0.0;
#pragma unroll
        for (int j = 0; j < C; ++j) if (tid + NT * j < n) { const float dlt = v[r][j] - mean[r]; a += (double)(dlt * dlt); }
        for (int i = tid + NT * C; i < n; i += NT) { const float dlt = to_db(mp[r][i], ref_db[r]) - mean[r]; a += (double)(dlt * dlt); }
        acc[r] = wave_sum_dpp(a);
    }
    if (lane == 63) { red[12 + w] = acc[0]; red[24 + 12 + w] = acc[1]; }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        double tot = 0.0;
#pragma unroll
        for (int g = 0; g < NWV; ++g) tot += red[24 * r + 12 + g];
        den[r] = 1.0f / (sqrtf((float)(tot / (double)n)) + 1e-8f);       // (the reciprocal: see db_row_from_lds)
    }
    const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);
    const int m_sh = (M & (M - 1)) == 0 ? __builtin_ctz(M) : -1;
#pragma unroll
    for (int j = 0; j < C; ++j) {
        const int e = tid + NT * j;
        if (e < n) {
            const int fr = m_sh >= 0 ? e >> m_sh : e / M, m = e - fr * M, tt = fr - shift;
            if (tt >= 0 && tt < TT) {
                const int o = layout == 1 ? tt * M + m : m * TT + tt;
                out[0][o] = (v[0][j] - mean[0]) * den[0];
                out[1][o] = (v[1][j] - mean[1]) * den[1];
            }
        }
    }
    for (int e = tid + NT * C; e < n; e += NT) {         // rows larger than the register cache
        const int fr = e / M, m = e - fr * M, tt = fr - shift;
        if (tt >= 0 && tt < TT) {
            const int o = layout == 1 ? tt * M + m : m * TT + tt;
            out[0][o] = (to_db(mp[0][e], ref_db[0]) - mean[0]) * den[0];
            out[1][o] = (to_db(mp[1][e], ref_db[1]) - mean[1]) * den[1];
        }
    }
    if (T < TT) {                                        // literal zero padding, left then right
        const int left = -shift, right0 = left + T;
        for (int i = tid; i < (TT - T) * M; i += NT) {
            const int fr = i / M, m = i - fr * M;
            const int tt = fr < left ? fr : right0 + (fr - left);
            const int o = layout == 1 ? tt * M + m : m * TT + tt;
            out[0][o] = 0.0f; out[1][o] = 0.0f;
        }
    }
}

// The same stage run by ONE wave on a row image it owns (k_stft_pair's continuous-queue form: the wave that completed the row's last pair,
// while the workgroup's other waves are on the next row's pairs).  The wave plays db_row_from_lds<NT, C, true>'s NT / 64 waves one after
// the other: virtual thread 64 w + lane takes elements e = 64 w + lane + NT j in the order of j, its double sum goes through the same DPP
// tree, the wave totals are added in the order of w - the same bits as the NT-thread forms.  The dB values replace the mel power IN PLACE
// (every element is read and written by the same lane), so the log10 runs once per element; nothing but the image is touched in LDS.
// mx: the row's largest mel power.  Rolled loops on purpose: the stage is off the critical path here, what it costs is the issue slots it takes.
__device__ __forceinline__ double readlane63_d(double a) {
    const long long b = __double_as_longlong(a);
    return __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)b, 63));
}
typedef __attribute__((address_space(3))) float lds_float;
template <int NT, int C = 5>
__device__ __attribute__((noinline)) void db_row_wave_call(float* mp_g, int T, int M, int TT, int layout, float* out, float mx) {
    constexpr int NWV = NT / 64;
    const int lane = threadIdx.x & 63;
    // (the arguments of a function that is not inlined arrive in vector registers and count as divergent: every `frame < T` below would be
    // an exec-mask branch, every address a 64-bit per-lane sum)
    T = __builtin_amdgcn_readfirstlane(T); M = __builtin_amdgcn_readfirstlane(M); TT = __builtin_amdgcn_readfirstlane(TT);
    layout = __builtin_amdgcn_readfirstlane(layout);
    mx = __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(mx)));
    {
        const unsigned long long ob = (unsigned long long)out;
        out = reinterpret_cast<float*>(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(ob >> 32)) << 32) | (unsigned)__builtin_amdgcn_readfirstlane((int)ob));
    }
    lds_float* const mp = (lds_float*)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(unsigned long long)(lds_float*)mp_g);
    const int n = T * M;
    const float ref_db = db10(fmaxf(1e-10f, mx));
    if (M == 64 && n <= NT * C && NWV % 4 == 0) {
        // 64 bands: element e = 64 w + lane + NT j is band `lane` of frame w + NWV j - no index arithmetic per element.  FOUR virtual waves
        // at a time (four independent chains of C elements, 4 C LDS reads in flight): a lone wave beside two busy waves of its SIMD runs
        // DEPENDENT instructions at ~20 cycles apiece - the one-chain form at the end of this function took ~75 000 cycles per row, longer
        // than the row's pairs take the other eleven waves - and the fully unrolled form (NWV chains, the row in registers) is 30 KB of
        // code beside a 30 KB pair loop in a 64 KB instruction cache.
        lds_float* const src = mp + lane;
        double tot = 0.0;
#pragma unroll 1
        for (int w0 = 0; w0 < NWV; w0 += 4) {
            float v[4][C];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < C; ++j) v[u][j] = w0 + u + NWV * j < T ? src[64 * (w0 + u + NWV * j)] : 0.0f;
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = 0.0;
#pragma unroll
                for (int j = 0; j < C; ++j) {
                    v[u][j] = to_db(v[u][j], ref_db);
                    if (w0 + u + NWV * j < T) { src[64 * (w0 + u + NWV * j)] = v[u][j]; a[u] += (double)v[u][j]; }
                }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = wave_sum_dpp(a[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) tot += readlane63_d(a[u]);
        }
        const float mean = (float)(tot / (double)n);
        wave_lds_sync();
        tot = 0.0;
#pragma unroll 1
        for (int w0 = 0; w0 < NWV; w0 += 4) {
            float v[4][C];
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < C; ++j) v[u][j] = w0 + u + NWV * j < T ? src[64 * (w0 + u + NWV * j)] : 0.0f;
            double a[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                a[u] = 0.0;
#pragma unroll
                for (int j = 0; j < C; ++j) if (w0 + u + NWV * j < T) { const float dlt = v[u][j] - mean; a[u] += (double)(dlt * dlt); }
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) a[u] = wave_sum_dpp(a[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u) tot += readlane63_d(a[u]);
        }
        const float den = 1.0f / (sqrtf((float)(tot / (double)n)) + 1e-8f);
        const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);
        const int f0 = shift > 0 ? shift : 0, f1 = T < TT + shift ? T : TT + shift;       // source frames f0 .. f1 - 1 land in the image
        float* const o1 = layout == 1 ? out + lane - 64 * shift : out + lane * TT - shift;  // element (frame fr, band lane) at o1[fr * os]
        const int os = layout == 1 ? 64 : 1;
#pragma unroll 8
        for (int fr = f0; fr < f1; ++fr) o1[fr * os] = (src[64 * fr] - mean) * den;
        if (T < TT) {                                    // literal zero padding, left then right
            const int left = -shift, right0 = left + T;
            for (int fr = 0; fr < TT - T; ++fr) {
                const int tt = fr < left ? fr : right0 + (fr - left);
                out[layout == 1 ? tt * 64 + lane : lane * TT + tt] = 0.0f;
            }
        }
        wave_lds_sync();
        return;
    }
    double tot = 0.0;
    for (int w = 0; w < NWV; ++w) {
        double a = 0.0;
        for (int e0 = 64 * w + lane; e0 < n; e0 += 8 * NT) {       // eight elements in flight (the same order of additions as one by one)
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = e0 + NT * j < n ? mp[e0 + NT * j] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = to_db(v[j], ref_db);
#pragma unroll
            for (int j = 0; j < 8; ++j) if (e0 + NT * j < n) { mp[e0 + NT * j] = v[j]; a += (double)v[j]; }
        }
        a = wave_sum_dpp(a);
        const long long b = __double_as_longlong(a);
        tot += __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)b, 63));
    }
    const float mean = (float)(tot / (double)n);
    wave_lds_sync();
    tot = 0.0;
    for (int w = 0; w < NWV; ++w) {
        double a = 0.0;
        for (int e0 = 64 * w + lane; e0 < n; e0 += 8 * NT) {
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = e0 + NT * j < n ? mp[e0 + NT * j] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j) if (e0 + NT * j < n) { const float dlt = v[j] - mean; a += (double)(dlt * dlt); }
        }
        a = wave_sum_dpp(a);
        const long long b = __double_as_longlong(a);
        tot += __longlong_as_double(((long long)__builtin_amdgcn_readlane((int)(b >> 32), 63) << 32) | (unsigned int)__builtin_amdgcn_readlane((int)b, 63));
    }
    const float den = 1.0f / (sqrtf((float)(tot / (double)n)) + 1e-8f);       // (the reciprocal: see db_row_from_lds)
    const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);
    const int m_sh = (M & (M - 1)) == 0 ? __builtin_ctz(M) : -1;
#pragma unroll 4
    for (int e = lane; e < n; e += 64) {
        const int fr = m_sh >= 0 ? e >> m_sh : e / M, m = e - fr * M, tt = fr - shift;
        if (tt >= 0 && tt < TT) out[layout == 1 ? tt * M + m : m * TT + tt] = (mp[e] - mean) * den;
    }
    if (T < TT) {                                        // literal zero padding, left then right
        const int left = -shift, right0 = left + T;
        for (int i = lane; i < (TT - T) * M; i += 64) {
            const int fr = i / M, m = i - fr * M;
            const int tt = fr < left ? fr : right0 + (fr - left);
            out[layout == 1 ? tt * M + m : m * TT + tt] = 0.0f;
        }
    }
    wave_lds_sync();                                     // the image's last read lies in front of whatever the caller publishes
}

// The dB stage on mel power that never had a ROW image in LDS (k_stft_pair2's RDB form, round 4): lane (quad qd, band column mq = lane >> 2)
// holds, for slot si (= the wave's si-th group of four frames, q = wave + NW si) and mel round r, the value of frame 4 q + qd and band
// 16 r + mq in v[8 si + r].  Same arithmetic as db_row_from_lds / k_dbnorm (separately rounded dB terms, per-thread sums in double, wave
// sums, cross-wave totals in wave order, two-pass std).  44.5 KB less LDS for an 87 x 128 row - what lets the 12-wave workgroup of the
// 128-band preset run the stage at all (176 KB with the image).
// Each wave transposes ITS eight frames through its own FFT scratch (free behind its last slot) into image order [frame][band] and takes
// them back as four float4 per lane - four consecutive bands of one frame: the output is four 16-byte stores per lane on whole 512-byte
// frame rows (a first form stored each register where it was: sixteen 4-byte stores per lane, 6 k cycles of address processing per row;
// profiles/r04_pair2_regdb.txt).  A float4 that does not exist (slot past the row's last, frame past its last frame, bands >= M) is skipped.
// red: 24 doubles (first sums | second sums) + NW floats (maxima); wscr: >= 8 x 144 floats of the wave's scratch; M % 4 == 0, M <= 128
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
#define AMPH_RDB_T(i_) { unsigned long long t_; asm volatile("s_waitcnt lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); q_t[i_] = (unsigned)t_; }
#else
#define AMPH_RDB_T(i_)
#endif
template <int NW>
__device__ __forceinline__ void db_row_from_regs(const float (&v)[16], float* wscr, int wave, int n_slots, int T, int M, int TT, int layout, float* out, double* red, unsigned q_t0 = 0) {
    constexpr int FS = 144;                              // floats between two frames of the wave's image (= 16 mod 32: two lanes per bank, the minimum, in the writes)
    const int tid = threadIdx.x, lane = tid & 63, qd = lane & 3, mq = lane >> 2;
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
    unsigned q_t[8];
    q_t[0] = q_t0;
    AMPH_RDB_T(1)
#endif
    float* redf = reinterpret_cast<float*>(red + 24);
    const int n = T * M;
    // (1) registers -> image order; the row's maximum on the way
    {
        const bool ok0 = wave < n_slots && 4 * wave + qd < T, ok1 = wave + NW < n_slots && 4 * (wave + NW) + qd < T;
        float* wr = wscr + qd * FS + mq;
        float mx = -INFINITY;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            wr[(k >> 3) * 4 * FS + 16 * (k & 7)] = v[k];
            mx = fmaxf(mx, ((k < 8 ? ok0 : ok1) && 16 * (k & 7) + mq < M) ? v[k] : -INFINITY);
        }
        mx = wave_max_dpp(mx);
        if (lane == 63) redf[wave] = mx;
    }
    AMPH_RDB_T(2)
    __syncthreads();                                     // (also orders the wave's own image writes before the reads below)
    AMPH_RDB_T(3)
    // (2) this lane's float4 i: frame slot f8 = 2 i + (lane >> 5) of the wave (slot f8 >> 2, frame f8 & 3 of it), bands 4 c4 .. 4 c4 + 3
    const int c4 = lane & 31, fh = lane >> 5;
    float4 x[4];
    bool ok[4];
    int tt[4];
    const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);      // crop_or_pad_time (map_detector_core.py:185-195): destination frame = source frame - shift
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int f8 = 2 * i + fh, q = wave + NW * (f8 >> 2), fr = 4 * q + (f8 & 3);
        ok[i] = q < n_slots && fr < T && 4 * c4 < M;
        tt[i] = fr - shift;
        x[i] = *reinterpret_cast<const float4*>(wscr + f8 * FS + 4 * c4);
    }
    float mx = redf[0];
#pragma unroll
    for (int g = 1; g < NW; ++g) mx = fmaxf(mx, redf[g]);
    const float ref_db = db10(fmaxf(1e-10f, mx));
    double acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (ok[i]) {
            x[i].x = to_db(x[i].x, ref_db); x[i].y = to_db(x[i].y, ref_db); x[i].z = to_db(x[i].z, ref_db); x[i].w = to_db(x[i].w, ref_db);
            acc += (double)x[i].x; acc += (double)x[i].y; acc += (double)x[i].z; acc += (double)x[i].w;
        }
    }
    acc = wave_sum_dpp(acc);
    if (lane == 63) red[wave] = acc;
    __syncthreads();
    double tot = 0.0;
#pragma unroll
    for (int g = 0; g < NW; ++g) tot += red[g];
    const float mean = (float)(tot / (double)n);
    acc = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (ok[i]) {
            const float d0 = x[i].x - mean, d1 = x[i].y - mean, d2 = x[i].z - mean, d3 = x[i].w - mean;
            acc += (double)(d0 * d0); acc += (double)(d1 * d1); acc += (double)(d2 * d2); acc += (double)(d3 * d3);
        }
    }
    acc = wave_sum_dpp(acc);
    if (lane == 63) red[12 + wave] = acc;
    AMPH_RDB_T(4)
    __syncthreads();
    AMPH_RDB_T(5)
    tot = 0.0;
#pragma unroll
    for (int g = 0; g < NW; ++g) tot += red[12 + g];
    const float rden = 1.0f / (sqrtf((float)(tot / (double)n)) + 1e-8f);
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
    { float r_ = rden; asm volatile("v_mov_b32 %0, %0" : "+v"(r_)); }
    AMPH_RDB_T(4)
#endif
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (ok[i] && tt[i] >= 0 && tt[i] < TT) {
            const float4 z = make_float4((x[i].x - mean) * rden, (x[i].y - mean) * rden, (x[i].z - mean) * rden, (x[i].w - mean) * rden);
            if (layout == 1) *reinterpret_cast<float4*>(out + tt[i] * M + 4 * c4) = z;
            else { float* o = out + (4 * c4) * TT + tt[i]; o[0] = z.x; o[TT] = z.y; o[2 * TT] = z.z; o[3 * TT] = z.w; }
        }
    }
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
    AMPH_RDB_T(2)
#endif
    if (T < TT) {                                        // literal zero padding, left then right
        const int left = -shift, right0 = left + T;
        for (int i = tid; i < (TT - T) * M; i += 64 * NW) {
            const int fr = i / M, m = i - fr * M;
            const int t2 = fr < left ? fr : right0 + (fr - left);
            if (layout == 1) out[t2 * M + m] = 0.0f; else out[m * TT + t2] = 0.0f;
        }
    }
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
    AMPH_RDB_T(6)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    AMPH_RDB_T(7)
    __syncthreads();
    if (lane == 0) for (int i = 0; i < 8; ++i) out[8 * wave + i] = __uint_as_float(q_t[i]);      // diagnostic build: the row's first 96 values are stamps
#endif
}

// ----------------------------------------------------------------------------
// The same transform with INDEPENDENT waves ("w"): no shared sample span, no workgroup barrier after the tables are
// staged.  See the comment at the frame list below.  Same arithmetic on the same values as k_stft_mel: bit-identical.
// ----------------------------------------------------------------------------
template <bool PCM16, int T_LO, int NW>
__global__ __launch_bounds__(64 * NW) void k_stft_mel_w(StftParams p, int n_tiles) {
    constexpr int kThreads = 64 * NW;                  // NW waves per workgroup
    using G = FftGeo<T_LO>;
    constexpr int kNc = G::NC, L = G::L, SLOTS = G::SLOTS, EXS = G::EXS;
    constexpr int kWinHalfBytes = G::WIN_BYTES, kTwBBytes = G::TWB_BYTES, kTableBytes = G::TABLE_BYTES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lane_f = lane & (L - 1), slot = lane / L;       // lane inside its frame, frame slot of the wave
    const int FT = p.frames_per_block;
#ifdef AMPH_STAMPS
    { const int fl = 0; AMPH_STAMP(0) }
    unsigned long long w_begin = 0;
    const unsigned long long w_rt0 = __builtin_amdgcn_s_memrealtime();
#define AMPH_WSTAMP(v_)                                                                     \
    {                                                                                       \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(v_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                  \
    }
    AMPH_WSTAMP(w_begin)
#else
#define AMPH_WSTAMP(v_)
#endif
    constexpr int n_fft = 2 * kNc;
    float* win_h = reinterpret_cast<float*>(smem);
    float2* tw_b = reinterpret_cast<float2*>(smem + kWinHalfBytes);
    float2* tw_p = reinterpret_cast<float2*>(smem + kWinHalfBytes + kTwBBytes);
    const float2* tw_a = p.tw_a;
    const int meta_bytes = (p.n_mels * 12 + 15) & ~15;
    const int melw_bytes = p.mel_w_floats * 4;                             // multiple of 64
    int3* mel_meta = reinterpret_cast<int3*>(smem + kTableBytes);          // (start4, steps, wofs) per filter
    float* mel_w = reinterpret_cast<float*>(smem + kTableBytes + meta_bytes);
    char* wbase = smem + kTableBytes + meta_bytes + melw_bytes + wave * kWaveScratch;   // no shared sample span here
    float2* ex = reinterpret_cast<float2*>(wbase + slot * G::SLOT_BYTES);   // this frame slot's scratch
    float* P = reinterpret_cast<float*>(wbase + slot * G::SLOT_BYTES);
    // fused dB stage (p.db_out): the row's mel power [n_frames][n_mels] stays here, behind the wave scratches
    float* mp_row = reinterpret_cast<float*>(smem + kTableBytes + meta_bytes + melw_bytes + NW * kWaveScratch);
    const bool fuse_db = p.db_out != nullptr;

    // ---- once per workgroup: tables (one packed global array = the LDS image) and the mel rows
    {
        const float4* src = reinterpret_cast<const float4*>(p.tables);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < kTableBytes / 16; i += kThreads) dst[i] = src[i];
        for (int m = tid; m < p.n_mels; m += kThreads) mel_meta[m] = make_int3(p.mel_start[m], p.mel_len[m], p.mel_wofs[m]);
        const float4* wsrc = reinterpret_cast<const float4*>(p.mel_w);
        float4* wdst = reinterpret_cast<float4*>(mel_w);
        for (int i = tid; i < (p.mel_w_floats >> 2); i += kThreads) wdst[i] = wsrc[i];
    }

    __syncthreads();                                  // tables and mel rows staged: the only workgroup barrier of the unfused kernel

    // ---- per-wave frame list: wave w of the workgroup takes frame slots w * SLOTS (+ NW * SLOTS ...) of the tiles
    //      blockIdx.x, + gridDim.x, ... - the tile order keeps the overlapping frames of a row on one CU (L1 hits) - but
    //      nothing is shared between the waves: every wave fetches its frame's n_fft samples itself (81 % overlap with
    //      its neighbours: 5 x the global reads of a shared span, all from L1 / L2, and 5 x the normalise arithmetic,
    //      on a VALU that has room) and no wave ever waits for another.  In the tile-synchronous kernel wave 0 spent 19 %
    //      of its time staging + at the staging barrier and 24 % at the end-of-tile barrier.
    struct Fr { const char* rowp; int len, s_begin, row, fr0; float sc; bool fast; };
    int it_t = blockIdx.x, it_fl0 = wave * SLOTS;
    auto seek = [&]() {                              // first valid (tile, slot) at or after the iterator; false at the end
        while (it_t < n_tiles) {
            const int r = it_t / p.tiles_per_row, tl = it_t - r * p.tiles_per_row;
            if (it_fl0 < FT && tl * FT + it_fl0 < p.n_frames) return true;
            it_t += gridDim.x;
            it_fl0 = wave * SLOTS;
        }
        return false;
    };
    auto describe = [&]() {
        Fr d;
        const int r = it_t / p.tiles_per_row, tl = it_t - r * p.tiles_per_row;
        d.row = r;
        d.fr0 = tl * FT + it_fl0;
        d.rowp = reinterpret_cast<const char*>(p.src.wave) + (long long)r * p.src.row_stride * (PCM16 ? 2 : 4);
        const int len = p.src.lengths ? min(p.src.lengths[r], p.src.n_in) : p.src.n_in;
        d.len = p.src.lead > 0 ? p.src.n_in : min(len, p.n_samples);   // truncation of fix_length (:217); a padded row is complete
        d.sc = p.src.scale ? p.src.scale[r] : 1.0f;
        d.s_begin = d.fr0 * p.hop - (n_fft >> 1) + p.src.lead;         // src.lead: centre padding already materialised (K0)
        // every frame slot of the wave inside the row, even start, 4-byte aligned: one dword = the (even, odd) sample pair
        const int s_last = d.s_begin + (SLOTS - 1) * p.hop;
        d.fast = d.s_begin >= 0 && s_last + n_fft <= d.len && d.fr0 + SLOTS <= p.n_frames && ((p.hop | d.s_begin) & 1) == 0 &&
                 ((reinterpret_cast<uintptr_t>(d.rowp) + (PCM16 ? 2 : 4) * (size_t)d.s_begin) & (PCM16 ? 3u : 7u)) == 0;
        return d;
    };
    // the NEXT frame's raw dwords (PCM16 fast path): requested before the mel rows of the current frame, when the 16
    // complex registers of the transform are dead
    uint32_t nx[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) nx[j] = 0;
    bool have_nx = false;
    auto fetch_fast = [&](const Fr& d) {
        const uint32_t* src32 = reinterpret_cast<const uint32_t*>(d.rowp + 2 * (size_t)(d.s_begin + slot * p.hop));
#pragma unroll
        for (int j = 0; j < 16; ++j) nx[j] = src32[lane_f + L * j];
    };
    bool more = seek();
    Fr d = {};
    if (more) {
        d = describe();
        if (PCM16 && d.fast) { fetch_fast(d); have_nx = true; }
    }

    const int k1b = lane_f / T_LO, t_lo = lane_f % T_LO;
    // stage C leaves k2b bit-reversed across the quad (radix-4); the radix-2 form is in order
    const int k2b = T_LO == 4 ? (((t_lo & 1) << 1) | (t_lo >> 1)) : t_lo;
    const float sgn_a = t_lo < 2 ? 1.0f : -1.0f;          // first butterfly (lanes 2 apart): x + p or p - x
    const float sgn_b = (t_lo & 1) == 0 ? 1.0f : -1.0f;   // second butterfly (adjacent lanes)
    const bool rot3 = T_LO == 4 && t_lo == 3;
    int last_row = -1;
    while (more) {
        const int row = d.row;
        const int fl = 0; (void)fl;                   // (stamps: every frame of wave 0)
        const int frame = d.fr0 + slot;
        const bool fvalid = frame < p.n_frames;       // the second half-wave may hold no frame
        AMPH_STAMP(1)
        float re[16], im[16];
        // ---- this lane's 16 (even, odd) sample pairs, normalised, then window + even/odd packing:
        //      z[n] = x[2n] w[2n] + i x[2n+1] w[2n+1], n = lane_f + L j
        {
            float x0[16], x1[16];
            if (PCM16 && d.fast) {
                if (!have_nx) fetch_fast(d);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    x0[j] = (float)(short)(nx[j] & 0xFFFFu) * (1.0f / 32768.0f);
                    x1[j] = (float)((int)nx[j] >> 16) * (1.0f / 32768.0f);
                }
            } else if (!PCM16 && d.fast) {            // float32 rows: the pair is one 8-byte load (not prefetched: 32 registers)
                const float2* src2 = reinterpret_cast<const float2*>(d.rowp + 4 * (size_t)(d.s_begin + slot * p.hop));
#pragma unroll
                for (int j = 0; j < 16; ++j) { const float2 v = src2[lane_f + L * j]; x0[j] = v.x; x1[j] = v.y; }
            } else {
                const int N = p.n_samples;
                const int sb = d.s_begin + slot * p.hop;
#pragma unroll
                for (int j = 0; j < 16; ++j) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        int sidx = sb + 2 * (lane_f + L * j) + h;
                        if (p.src.lead > 0) {                 // padded rows: plain indexing (only reached for unaligned rows)
                        } else if (p.pad_mode == 0) {         // reflect (np.pad mode="reflect")
                            if (sidx < 0) sidx = -sidx;
                            if (sidx >= N) sidx = 2 * (N - 1) - sidx;
                        } else if (sidx >= N) sidx = -1;      // constant (zero) padding
                        const float v = fvalid ? load_sample<PCM16>(d.rowp, sidx, d.len) : 0.0f;
                        if (h == 0) x0[j] = v; else x1[j] = v;
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int n = lane_f + L * j;
                float w0, w1;
                if (j < 8) { const float2 w = reinterpret_cast<const float2*>(win_h)[n]; w0 = w.x; w1 = w.y; }
                else { const int m = n_fft - 2 * n; w0 = win_h[m]; w1 = win_h[m - 1]; }   // w[2n] = w[N-2n], w[2n+1] = w[N-2n-1]
                re[j] = normalise_sample(p.src, x0[j], d.sc) * w0;
                im[j] = normalise_sample(p.src, x1[j], d.sc) * w1;
            }
        }
        AMPH_STAMP(2)
        // ---- stage A
        dft16(re, im);
#pragma unroll
        for (int k1 = 0; k1 < 16; ++k1) {
            float r = re[DFT16_POS(k1)], i = im[DFT16_POS(k1)];
            if (k1 > 0) {
                const float2 w = tw_a[(k1 - 1) * L + lane_f];
                AMPH_CMUL(r, i, w.x, w.y);
            }
            ex[k1 * EXS + lane_f] = make_float2(r, i);
        }
        wave_lds_sync();
        AMPH_STAMP(3)
        // ---- stage B: lane (k1b, t_lo) gathers t = t_lo + T_LO m
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            const float2 v = ex[k1b * EXS + t_lo + T_LO * m];
            re[m] = v.x;
            im[m] = v.y;
        }
        AMPH_STAMP(4)
        dft16(re, im);
        AMPH_STAMP(5)
        wave_lds_sync();                              // the Z image below overwrites `ex`
        float2* Z = ex;
#pragma unroll
        for (int k2a = 0; k2a < 16; ++k2a) {
            float r = re[DFT16_POS(k2a)], i = im[DFT16_POS(k2a)];
            if (k2a > 0) {
                const float2 w = tw_b[k2a * T_LO + t_lo];
                AMPH_CMUL(r, i, w.x, w.y);
            }
            // ---- stage C: radix-T_LO over adjacent lanes (DPP xor butterflies).  The lane-dependent add / subtract
            //      is one fma with a per-lane sign (x * (+-1) + partner rounds exactly as x + partner / partner - x),
            //      the -i of lane 3 two selects
            if (T_LO == 4) {
                const float nr = fmaf(r, sgn_a, dpp_xor2(r)), ni = fmaf(i, sgn_a, dpp_xor2(i));
                const float tr = rot3 ? ni : nr, ti = rot3 ? -nr : ni;          // lane 3: * (-i)
                r = fmaf(tr, sgn_b, dpp_xor1(tr));
                i = fmaf(ti, sgn_b, dpp_xor1(ti));
            } else {
                r = fmaf(r, sgn_b, dpp_xor1(r));
                i = fmaf(i, sgn_b, dpp_xor1(i));
            }
            // lane holds X[k1b + 16 k2a + 256 k2b]
            Z[k1b + 16 * k2a + (256 + kZPad) * k2b] = make_float2(r, i);
        }
        wave_lds_sync();
        AMPH_STAMP(6)
        // ---- real-FFT split: bins k and NC - k from Z[k], Z[NC - k]; the power spectrum
        //      is written over the Z image, so every pair is read into registers first
        {
            float2 za[9], zb[9];
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int k = min(lane_f + L * it, kNc / 2);
                const int k2 = (kNc - k) & (kNc - 1);
                za[it] = Z[k + kZPad * (k >> 8)];
                zb[it] = Z[k2 + kZPad * (k2 >> 8)];
            }
            wave_lds_sync();
            AMPH_STAMP(7)
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int k = lane_f + L * it;
                if (k <= kNc / 2) {
                    const float2 a = za[it], b = zb[it];
                    const float2 w = tw_p[k];
                    const float xer = 0.5f * (a.x + b.x), xei = 0.5f * (a.y - b.y);      // (Z[k] + conj Z[N-k]) / 2
                    const float xor_ = 0.5f * (a.y + b.y), xoi = -0.5f * (a.x - b.x);    // -i (Z[k] - conj Z[N-k]) / 2
                    const float tr = xor_ * w.x - xoi * w.y, ti = xor_ * w.y + xoi * w.x;
                    const float pr = xer + tr, pi = xei + ti, mr = xer - tr, mi = xei - ti;
                    P[k] = pr * pr + pi * pi;
                    if (k != kNc - k) P[kNc - k] = mr * mr + mi * mi;
                }
            }
            for (int i = lane_f; i < p.p_tail; i += L) P[kNc + 1 + i] = 0.0f;   // zero the tail the padded mel rows touch
        }
        wave_lds_sync();
        AMPH_STAMP(8)
        // ---- the wave's next frame: its samples are requested now and land under the mel rows
        const int cur_fr0 = d.fr0;
        it_fl0 += NW * SLOTS;
        more = seek();
        have_nx = false;
        if (more) {
            d = describe();
            if (PCM16 && d.fast) { fetch_fast(d); have_nx = true; }
        }
        // ---- mel rows: 4 lanes per filter, each lane 4 consecutive bins per step (aligned float4).
        //      The host pads every filter of a 16-filter round to the round's longest row (zero
        //      weights), so the step count is wave-uniform: no per-lane predication.
        //      The stage was a chain of LDS round trips (descriptor -> first operands -> ... per round, 2.7 k of a frame's
        //      ~10 k cycles): a round's descriptor is requested one round ahead.  Same accumulation order as before (even steps -> acc0, odd -> acc1).
        {
            const int q = lane & 3;
            const int n_rounds = (p.n_mels + 15) >> 4;
            // (hoisting the descriptors of all rounds into registers spilled: the frame's FFT state leaves no room at three waves
            // per SIMD; the next round's descriptor is requested before the current round's operands instead)
            for (int sl = 0; sl < SLOTS; ++sl) {            // all 64 lanes work on one frame slot at a time
                const int fr = cur_fr0 + sl;
                if (fr >= p.n_frames) break;                      // wave-uniform
                float* out = fuse_db ? mp_row + fr * p.n_mels : p.melpow + ((long long)row * p.n_frames + fr) * p.n_mels;
                const float* Ps = reinterpret_cast<const float*>(wbase + sl * G::SLOT_BYTES);
                int3 mt = mel_meta[min(lane >> 2, p.n_mels - 1)];                             // (start4, steps of the round, wofs)
                for (int r = 0; r < n_rounds; ++r) {
                    const int steps = __builtin_amdgcn_readfirstlane(mt.y);
                    const float4* w = reinterpret_cast<const float4*>(mel_w + mt.z) + q;    // (from global / L1 instead: +1 us)
                    const float4* pp = reinterpret_cast<const float4*>(Ps + mt.x) + q;
                    if (r + 1 < n_rounds) mt = mel_meta[min(16 * (r + 1) + (lane >> 2), p.n_mels - 1)];
                    float acc0 = 0.0f, acc1 = 0.0f;
#define AMPH_MEL_FMA(acc_, a_, b_)                                                         \
    acc_ = fmaf(a_.x, b_.x, acc_); acc_ = fmaf(a_.y, b_.y, acc_); acc_ = fmaf(a_.z, b_.z, acc_); acc_ = fmaf(a_.w, b_.w, acc_);
                    for (int i0 = 0; i0 < steps; i0 += 2) {         // two steps (four 16-byte requests) per trip: more spilled
                        const bool two = i0 + 1 < steps;            // wave-uniform
                        const float4 a0 = w[4 * i0], b0 = pp[4 * i0];
                        float4 a1 = a0, b1 = b0;
                        if (two) { a1 = w[4 * i0 + 4]; b1 = pp[4 * i0 + 4]; }
                        AMPH_MEL_FMA(acc0, a0, b0)
                        if (two) { AMPH_MEL_FMA(acc1, a1, b1) }
                    }
#undef AMPH_MEL_FMA
                    float acc = acc0 + acc1;
                    acc += dpp_xor1(acc);
                    acc += dpp_xor2(acc);
                    const int m = 16 * r + (lane >> 2);
                    if (q == 0 && m < p.n_mels) out[m] = acc;
                }
            }
        }
        (void)fvalid;
        wave_lds_sync();                              // the scratch is rewritten by the next frame
        AMPH_STAMP(9)
        // ---- fused dB stage: this wave's last frame of the row (tile == row here).  Every wave owns a frame of every row
        //      (host check), so all of them arrive: power_to_db + z-score + crop / pad on the row held in LDS, by the whole
        //      workgroup, while the samples of the waves' next frames are already in flight.
        //      The LAST row of the workgroup (with a 256-row batch: its only one) is finished behind the loop, where nothing of
        //      the frame loop is live any more: as a call from inside the loop the stage cost ~5 us per row in register saves
        //      (scratch traffic of 12 waves x ~170 registers), inlined here it spilled the frame loop.
        if (fuse_db && more && d.row != row) {
            __syncthreads();
            db_row_call<kThreads>(mp_row, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                  p.db_out + (long long)row * p.db_target_frames * p.n_mels,
                                  reinterpret_cast<double*>(mp_row + ((p.n_frames * p.n_mels + 3) & ~3)));
            __syncthreads();                          // the next row's frames overwrite mp_row
        }
        last_row = row;
    }
    if (fuse_db && last_row >= 0) {
        __syncthreads();
        db_row_from_lds<kThreads>(mp_row, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                  p.db_out + (long long)last_row * p.db_target_frames * p.n_mels,
                                  reinterpret_cast<double*>(mp_row + ((p.n_frames * p.n_mels + 3) & ~3)));
    }
#ifdef AMPH_STAMPS
    if (tid == 0) {       // per-workgroup timeline (wave 0): where the tiles' time goes, and the in-kernel clock
        unsigned long long w_end;
        AMPH_WSTAMP(w_end)
        p.stamps[(size_t)blockIdx.x * 16 + 10] = 0;
        p.stamps[(size_t)blockIdx.x * 16 + 11] = w_end - w_begin;
        p.stamps[(size_t)blockIdx.x * 16 + 12] = 0;
        p.stamps[(size_t)blockIdx.x * 16 + 13] = w_end - w_begin;
        p.stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime() - w_rt0;
        p.stamps[(size_t)blockIdx.x * 16 + 15] = 0;
    }
#endif
}

// ----------------------------------------------------------------------------
// K1/K2, third form ("pair", round 3): TWO consecutive frames a, b of a row as ONE complex 2048-point FFT of
// z = w (a + i b) per wave, 32 complex points per lane.  tools/fft_pair_model.py is the numpy model of exactly this data
// movement (written first).  Against k_stft_mel_w: half the table reads and all of the per-frame loop overhead per frame,
// no twiddle in the real-FFT split (|A|^2, |B|^2 from Z[k], Z[N - k] are 8 operations per bin pair), a radix-2 instead of
// a radix-4 cross-lane stage, twice the independent work per wave between two LDS round trips, ~900 instead of ~1300 VALU
// instructions per frame.
//   load     lane l, dword j' (0 .. 15 + HJ): samples 128 j' + 2 d + e, d = dw(l) = (l & 15) + 16 (l >> 5) + 32 ((l >> 4) & 1), of the
//            K0-padded PCM16 row (whole 256-byte rows per wave load; hop = 128 HJ, so frame b's dword j' is frame a's dword j' + HJ);
//            e = 0, 1 are two 16-point sequences: with p = (l & 15) + 16 (l >> 5), lanes with bit 4 clear hold the even-j halves of
//            columns 2 p + e, their neighbours l + 16 the odd-j halves
//   stage 1  dft16 per sequence, v_permlane16_swap so that every lane has both halves of ONE column col(l) = 2 p + ((l >> 4) & 1) -
//            lanes 0..31 own columns 0..31: conflict-free exchange stores -, radix-2 combine -> Y[k1], k1 = 0..31; twiddle W_2048^(col k1)
//   exchange T[k1][col] floats (real parts, then imaginary parts: the tile is 8.5 KB per wave, not 17);
//            lane 2 k1' + h reads T[k1'][32 h .. 32 h + 31] as eight 16-byte reads per pass
//   stage 2  radix-2 DIF across the lane pair (DPP), twiddle W_64^mm on h = 1, DFT32 in registers:
//            X[k1' + 32 h + 64 k2'], k2' = 0..31
//   split    lane' = k1' + 32 h owns bins lane' + 64 i (i < 16, in its registers); the partners N - k are the registers
//            k2' >= 16 of lane' -> (64 - lane') & 63: only those go through LDS (17 x 80 floats per pass)
//   power    4 |A|^2, 4 |B|^2 (the 1/4 is folded into the mel weights: exact) into two LDS rows; mel rows for both frames
//            from one weight read
// Only the hot configuration: PCM16 rows that K0 (or the pad kernel) has fix_length'd and centre padded, n_fft = 2048,
// hop = 384; everything else takes k_stft_mel_w / k_stft_mel.
// ----------------------------------------------------------------------------
constexpr int kPairExS = 68;                        // exchange row stride in floats (= 4 mod 64: conflict-free 16-byte reads)
constexpr int kPairScratch = 32 * kPairExS * 4;     // per-wave scratch: exchange tile, then partner tile, then the two power spectra
constexpr int kPairPStride = kPairScratch / 8;      // floats between the two power spectra (1088 >= 1025 + tail)
constexpr int kPairZS = 96;                         // partner tile row stride in floats (columns >= 32 shifted by 16 banks; lane' 0 reads one row further: same bank as column 0)
// (stage-2 twiddles: [2][17] 16-byte pairs - the h = 0 and h = 1 lanes of a ds_read_b128 group read entries m and 17 + m: with 16 + m
//  the two addresses shared their banks, every one of the 16 reads per pair 2-way conflicted)
constexpr int kPairTw2S = 17;
constexpr int kPairWinBytes = 16 * 64 * 8, kPairTw2Bytes = 2 * kPairTw2S * 16, kPairTw1Bytes = 16 * 64 * 16;

// Y[k + 16 p] = E[k] +- W32^k O[k]: e = DFT16 of the even inputs (X[k] at DFT16_POS(k)), o = of the odd ones, in place
__device__ __forceinline__ void dft32_combine(float (&er)[16], float (&ei)[16], float (&orr)[16], float (&oi)[16]) {
    constexpr float C[16] = {1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f,
                             0.0f, -0.19509032201612826785f, -0.38268343236508977173f, -0.55557023301960222474f,
                             -0.70710678118654752440f, -0.83146961230254523708f, -0.92387953251128675613f, -0.98078528040323044913f};
    constexpr float S[16] = {0.0f, 0.19509032201612826785f, 0.38268343236508977173f, 0.55557023301960222474f,
                             0.70710678118654752440f, 0.83146961230254523708f, 0.92387953251128675613f, 0.98078528040323044913f,
                             1.0f, 0.98078528040323044913f, 0.92387953251128675613f, 0.83146961230254523708f,
                             0.70710678118654752440f, 0.55557023301960222474f, 0.38268343236508977173f, 0.19509032201612826785f};
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        constexpr int dummy = 0; (void)dummy;
        const int q = DFT16_POS(k);
        float tr, ti;                               // W32^k o = (c - i s)(or + i oi)
        if (k == 0) { tr = orr[q]; ti = oi[q]; }
        else if (k == 8) { tr = oi[q]; ti = -orr[q]; }
        else { tr = C[k] * orr[q] + S[k] * oi[q]; ti = C[k] * oi[q] - S[k] * orr[q]; }
        const float ar = er[q], ai = ei[q];
        er[q] = ar + tr; ei[q] = ai + ti;
        orr[q] = ar - tr; oi[q] = ai - ti;
    }
}

// TWL: the stage-1 twiddles (15.5 KB) live in LDS and are read where they are used; otherwise they are buffer loads issued a phase
// ahead (62 registers in flight across the first DFTs: at three waves per SIMD the allocator spills)
// K0F (round 4, "stft_k0_fuse"): 00_normalize_dataset_rms.py's rms_normalize + PCM16 write / read of the row inside this kernel, except the
// per-chunk sums of squares, which stay K0's first kernel (k_rms_scale<int16_t, 256, 1>, StftParams::k0_chunk_sums); K0's second kernel
// (fold + apply) and its 48 KB padded copy of every row are gone.  (1) Every wave folds the row's <= 4 chunk sums into the scale itself, in
// numpy's order and the pinned numpy's scalar arithmetic (no LDS, no barrier).  (2) The pairs read the RAW rows: a pair whose samples all lie
// inside the row (25 of a one-second row's 29) is the padded-row form's aligned dword loads; librosa's centre padding (reflect / zeros),
// fix_length's zero fill and truncation are index arithmetic on the pairs that touch them - and those are drawn from the MIDDLE of the
// row's queue (pair_of), neither in front of the tables' barrier nor in the row's tail.  (3) The decode re-quantises: q' = rint(clip(q sc,
// +-32768) 32767 / 32768) - bit for bit the sample 00 writes to disk (the powers of two commute with the roundings) - once per dword (a
// sliding window shares it between frame a and frame b), three vector instructions per sample.  Measured forms, incl. the two that put
// the summation itself into this kernel's prologue and lost (cold one-shot code, the row's first touch from HBM with the whole CU waiting,
// three barriers): profiles/r04_k0_fuse.txt.
// G2 (round 4; with K0F and the fused dB stage): the workgroup takes its rows TWO at a time - one queue over the 2 x 29 pairs of both rows, a
// mel-power image per row in LDS (the stage-1 twiddles then come from global memory: measured free), both dB stages at the end.  A one-second
// row is 29 pairs on 12 waves = 12 + 12 + 5: its third round keeps 5 of 12 waves busy for ~15 000 of the row's ~64 000 cycles; two rows are
// 58 pairs = four full rounds + 10, and the tables are staged once for both.
// K0F 2: the raw rows as they are (apply_rms == 0: files 00_normalize_dataset_rms.py has already processed) - the same loads and index
// arithmetic, a plain decode; replaces the pad kernel's fix_length'd, centre-padded copy of every row.
template <int NW, int HJ, bool TWL, int K0F = 0, bool G2 = false, bool CQ = false>
__global__ __launch_bounds__(64 * NW) void k_stft_pair(StftParams p, int n_rows) {
    static_assert(!CQ || (G2 && K0F != 0 && NW <= 12), "the continuous queue runs on the two-image (G2) layout of the K0F forms");
    constexpr int kThreads = 64 * NW;
    constexpr int ND = 16 + HJ;                         // dwords per lane and pair
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float2* win = reinterpret_cast<const float2*>(smem);                      // [16][64] (w[n], w[n + 1]) 2^-15
    const float4* tw2 = reinterpret_cast<const float4*>(smem + kPairWinBytes);      // [2][17] twiddle pairs (16 used)
    const int n_meta = ((p.n_mels + 15) >> 4) * 16;       // (round, lane quad) entries of pair_mel_meta
    const int meta_bytes = n_meta * 16;
    const int melw_bytes = p.pair_mel_w_floats * 4;
    constexpr int kTab = kPairWinBytes + kPairTw2Bytes + (TWL ? kPairTw1Bytes : 0);
    const float4* tw1_l = reinterpret_cast<const float4*>(smem + kPairWinBytes + kPairTw2Bytes) + lane;      // [16][64] (wr_k, wr_k+16, wi_k, wi_k+16) (TWL)
    int4* mel_meta = reinterpret_cast<int4*>(smem + kTab);
    float* mel_w = reinterpret_cast<float*>(smem + kTab + meta_bytes);
    char* wbase = smem + kTab + meta_bytes + melw_bytes + wave * kPairScratch;
    float* T = reinterpret_cast<float*>(wbase);
    amph_f2* P2 = reinterpret_cast<amph_f2*>(wbase);        // the pair's power spectra, interleaved: P2[k] = (4 |A[k]|^2, 4 |B[k]|^2)
    int* const next_pair = reinterpret_cast<int*>(smem + kTab + meta_bytes + melw_bytes + NW * kPairScratch);      // the row's next unassigned pair
    float* mp_row = reinterpret_cast<float*>(smem + kTab + meta_bytes + melw_bytes + NW * kPairScratch + 16);
    const bool fuse_db = p.db_out != nullptr;
#ifdef AMPH_STAMPS
    unsigned long long pw_begin = 0;
    const unsigned long long pw_rt0 = __builtin_amdgcn_s_memrealtime();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pw_begin)::"memory");
#ifdef AMPH_STAMPS_PHASES
#define AMPH_PSTAMP(k_)                                                                     \
    if (wave == 0 && pr == 0) {                                                             \
        unsigned long long t_;                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); \
        __builtin_amdgcn_sched_barrier(0);                                                  \
        if (lane == 0) p.stamps[(size_t)blockIdx.x * 16 + (k_)] = t_;                       \
    }
#else
#define AMPH_PSTAMP(k_)
#endif
#else
#define AMPH_PSTAMP(k_)
#endif

    // the next pair's raw dwords: requested before the mel rows of the current pair
    uint32_t nx[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) nx[j] = 0;
    // buffer loads (resource in SGPRs + a per-lane byte offset that never changes + a scalar offset): written as pointer arithmetic
    // the 31 table rows and 19 sample rows each got their own 64-bit per-lane address, i.e. ~100 live VGPRs of addresses
    const __amdgpu_buffer_rsrc_t rs_tw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.pair_tw1), 0, -1, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src.wave), 0, -1, 0x00020000);
    // lane -> dword of a 64-dword sample row: lanes l and l + 16 hold the even-j / odd-j halves of ONE column pair (dwords d, d + 32),
    // so that v_permlane16_swap completes a column per lane and each 32-lane half ends up with 32 CONSECUTIVE columns - its exchange
    // stores hit 32 different banks (with lanes l, l + 32 paired a half owned the even columns only: every store 2-way conflicted)
    const int lane_dw = (lane & 15) + 16 * (lane >> 5) + 32 * ((lane >> 4) & 1);
    const int lane4 = lane_dw * 4, lane16 = lane * 16;
    const __amdgpu_buffer_rsrc_t rs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(p.k0_in), 0, -1, 0x00020000);
    auto k0_row_len = [&](int row_) {                        // samples of the row that count (wave-uniform)
        const int n_ = p.k0_lengths ? max(0, min(p.k0_lengths[row_], p.k0_n_in)) : p.k0_n_in;
        return __builtin_amdgcn_readfirstlane(n_);
    };
    auto fetch = [&](int row_, int pr_) {
        const bool full = 2 * pr_ + 1 < p.n_frames;        // wave-uniform: the row's last pair may hold one frame only
        if constexpr (K0F) {
            // the RAW row: sample row j of the pair = padded samples pr 2 hop + 128 j .. + 127 = raw r0 .. r0 + 127, r0 = that - lead
            const int lead = p.n_fft >> 1, n_fix = p.n_samples;
            const int nv = min(k0_row_len(row_), n_fix), nv2 = nv & ~1;
            const int rbase = __builtin_amdgcn_readfirstlane((int)((long long)row_ * p.k0_row_stride * 2));
            const int s0 = pr_ * 2 * p.hop - lead;
            if (s0 >= 0 && s0 + 128 * (full ? ND : 16) <= nv2) {
                // every sample of the pair is a sample of the row (25 of a one-second row's 29 pairs): the aligned dword loads of the padded-row form
                const int base = rbase + 2 * s0;
#pragma unroll
                for (int j = 0; j < 16; ++j) nx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, lane4, base + 256 * j, 0);
#pragma unroll
                for (int j = 16; j < ND; ++j) nx[j] = full ? __builtin_amdgcn_raw_buffer_load_b32(rs_raw, lane4, base + 256 * j, 0) : 0u;
            } else {
                // the pair touches the centre padding (np.pad mode="reflect" of the fix_length'd row, map_detector_core.py:212-228; mode 1:
                // zeros), fix_length's zero fill or the row's last odd sample: per sample, through a resource that ends at the row's
                // last valid sample (a sample outside reads 0 from past its end).  All 2 ND loads first, ONE wait, then the halves are joined.
                const __amdgpu_buffer_rsrc_t rs_row = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(reinterpret_cast<const char*>(p.k0_in) + (long long)row_ * p.k0_row_stride * 2), 0, 2 * nv, 0x00020000);
                uint32_t lo[ND], hi[ND];
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    const int r0 = s0 + 128 * j;                 // (wave-uniform: the rows of the pair that lie inside the row stay dword loads)
                    lo[j] = 0u; hi[j] = 0u;
                    if (j >= 16 && !full) continue;
                    if (r0 >= 0 && r0 + 128 <= nv2) { lo[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, lane4, rbase + 2 * r0, 0); continue; }
                    uint32_t half[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        int sidx = r0 + 2 * lane_dw + e;
                        if (p.pad_mode == 0) {
                            sidx = sidx < 0 ? -sidx : sidx;
                            if (sidx >= n_fix) sidx = 2 * (n_fix - 1) - sidx;
                        }
                        const bool in = (unsigned)sidx < (unsigned)nv;
                        half[e] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_row, in ? 2 * sidx : 0x7FFFFFF0, 0, 0);
                    }
                    lo[j] = half[0]; hi[j] = half[1];
                }
#pragma unroll
                for (int j = 0; j < ND; ++j) nx[j] = lo[j] | (hi[j] << 16);
            }
        } else {
            const int base = __builtin_amdgcn_readfirstlane((int)(((long long)row_ * p.src.row_stride) * 2 + (long long)pr_ * p.hop * 4));
#pragma unroll
            for (int j = 0; j < 16; ++j) nx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, lane4, base + 256 * j, 0);
#pragma unroll
            for (int j = 16; j < ND; ++j) nx[j] = full ? __builtin_amdgcn_raw_buffer_load_b32(rs_x, lane4, base + 256 * j, 0) : 0u;
        }
    };
    float k0_sc = 1.0f;                                      // the row's scale as the decode applies it (1 for a gated row)
    // K0F: the order in which a row's pairs are drawn.  The two pairs at each end of a row touch the centre padding and take the per-sample
    // path (more loads, a wait, code that is cold at its first use): they are neither the waves' first pairs (in front of the tables' barrier)
    // nor the row's last (its tail), but the first four drawn from the queue - draw i -> pair i + 2 | 0, n - 2, 1, n - 1 | i - 2
    const int n_pairs_k = (p.n_frames + 1) >> 1;
    auto pair_of = [&](int i) {
        if (!K0F || n_pairs_k < NW + 4) return i;
        if (i < NW) return i + 2;
        if (i < NW + 4) return (i - NW) & 1 ? n_pairs_k - 2 + ((i - NW) >> 1) : (i - NW) >> 1;
        return i - 2;
    };
    if ((int)blockIdx.x < n_rows && wave < n_pairs_k) fetch(blockIdx.x, pair_of(wave));

    // ---- once per workgroup: window pairs + stage-2 twiddles (one packed array), stage-1 twiddles (TWL), mel rows (x 1/4: see above).  Every load is requested before the
    //      first store: one global round trip (beside the sample fetch above), not one per table
    {
        const float4* src = reinterpret_cast<const float4*>(p.pair_tables);
        const float4* s1 = reinterpret_cast<const float4*>(p.pair_tw1);
        float4* dst = reinterpret_cast<float4*>(smem);
        float4* d1 = reinterpret_cast<float4*>(smem + kPairWinBytes + kPairTw2Bytes);
        constexpr int NWIN = (kPairWinBytes + kPairTw2Bytes) / 16, NT1 = TWL ? kPairTw1Bytes / 16 : 0;
        constexpr int UW = (NWIN + kThreads - 1) / kThreads, UT = (NT1 + kThreads - 1) / kThreads;
        float4 vw[UW], vt[UT ? UT : 1];
#pragma unroll
        for (int u = 0; u < UW; ++u) { const int i = tid + kThreads * u; vw[u] = i < NWIN ? src[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
#pragma unroll
        for (int u = 0; u < UT; ++u) { const int i = tid + kThreads * u; vt[u] = i < NT1 ? s1[i] : make_float4(0.f, 0.f, 0.f, 0.f); }
        const int4* msrc = reinterpret_cast<const int4*>(p.pair_mel_meta);
        int4 mm_ = make_int4(0, 0, 0, -1);
        if (tid < n_meta) mm_ = msrc[tid];
        const float4* wsrc = reinterpret_cast<const float4*>(p.pair_mel_w);
        float4* wdst = reinterpret_cast<float4*>(mel_w);
        const int n4 = p.pair_mel_w_floats >> 2;
        for (int base = 0; base < n4; base += 2 * kThreads) {
            const int i0 = base + tid, i1 = i0 + kThreads;
            float4 v0 = i0 < n4 ? wsrc[i0] : make_float4(0.f, 0.f, 0.f, 0.f), v1 = i1 < n4 ? wsrc[i1] : make_float4(0.f, 0.f, 0.f, 0.f);
            v0.x *= 0.25f; v0.y *= 0.25f; v0.z *= 0.25f; v0.w *= 0.25f;
            v1.x *= 0.25f; v1.y *= 0.25f; v1.z *= 0.25f; v1.w *= 0.25f;
            if (i0 < n4) wdst[i0] = v0;
            if (i1 < n4) wdst[i1] = v1;
        }
        for (int m = tid + kThreads; m < n_meta; m += kThreads) mel_meta[m] = msrc[m];
        if (tid < n_meta) mel_meta[tid] = mm_;
        if (tid == 0) *next_pair = NW;                   // pairs 0 .. NW - 1 are the waves' first ones
        if (CQ && tid < 8) {                             // the continuous queue's words (see below; on the reduction scratch behind the two images)
            const int mpf_ = (p.n_frames * p.n_mels + 3) & ~3;
            reinterpret_cast<int*>(mp_row + 2 * mpf_)[tid] = 0;
        }
#pragma unroll
        for (int u = 0; u < UW; ++u) { const int i = tid + kThreads * u; if (i < NWIN) dst[i] = vw[u]; }
#pragma unroll
        for (int u = 0; u < UT; ++u) { const int i = tid + kThreads * u; if (i < NT1) d1[i] = vt[u]; }
    }
    __syncthreads();
#ifdef AMPH_STAMPS
    // workgroup timeline (tools/stft_stamps.py timeline): [0] start, [1] tables staged, [2 + i] the LAST wave's end of pair round i,
    // [6] every wave out of the pair loop, [7] dB stage done; second region of the stamp buffer
    unsigned long long* const tl = p.stamps + 16 * 4096 + (size_t)blockIdx.x * 16;
    int tl_round = 0;
    if (tid == 0) {
        unsigned long long t_;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
        tl[0] = pw_begin; tl[1] = t_;
        for (int k_ = 2; k_ < 16; ++k_) tl[k_] = 0;
    }
    __syncthreads();
#endif

    const int n_pairs = (p.n_frames + 1) >> 1;
    const int col = 2 * ((lane & 15) + 16 * (lane >> 5)) + ((lane >> 4) & 1);      // stage-1 column of this lane
    const int k1p = lane >> 1, h = lane & 1;                // stage-2 role
    const int lanep = k1p + 32 * h;                         // bins lanep + 64 k2'
    const int colp = (64 - lanep) & 63;                     // the lane' that holds the partners N - k
    const int zw = lanep + ((lanep >> 5) << 4);             // partner tile column (write), bank-shifted
    const int zr_ = colp + ((colp >> 5) << 4) + (lanep == 0 ? kPairZS : 0);   // (read; lane' 0: one row further, row 16 = Z[0])
    const float sgn = h ? -1.0f : 1.0f;
    // LDS byte addresses of this lane's stores (AMPH_DS_ST32 / 64): exchange column, partner column, power bin
    const uint32_t lds_w = (uint32_t)(uintptr_t)wbase;
    const uint32_t st_col = lds_w + 4u * (uint32_t)col, st_zw = lds_w + 4u * (uint32_t)zw, st_p2 = lds_w + 8u * (uint32_t)lanep;


    // The phases of a pair are fenced for the instruction scheduler (sched_barrier): left alone it builds ONE region of ~2 000
    // instructions, runs out of registers and then serialises every table load behind its own wait (31 dependent L2 round
    // trips for the stage-1 twiddles in the first build).  Inside a phase the loads are issued up front, by hand.
#define AMPH_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
    // Pairs are handed out DYNAMICALLY (an LDS counter per row; wave w starts with pair w): the SIMD issues oldest-wave-first, so its
    // first wave runs a pair in ~12.6 k cycles, its second in ~20 k, its third in ~35 k (profiles/r03_stft_pair_timeline.txt) - with
    // a fixed pr = wave + 12 i the first waves idle at the end while the SIMD that drew the row's 29th pair still works
    // LDS behind the waves' scratch: [RB rows of mel power][RB x (16 doubles + 16 floats) of reduction scratch]
    constexpr int RB = G2 ? 2 : 1;
    const int mp_floats = (p.n_frames * p.n_mels + 3) & ~3;
    double* const db_red0 = reinterpret_cast<double*>(mp_row + RB * mp_floats);
    float* const db_redf = reinterpret_cast<float*>(db_red0 + 16);   // db_row_from_lds's redf (row 1 of a group: 24 doubles further)
    // the chunk sums of K0's first kernel, folded in numpy's order (chunk 0 first) into the scale - k_rms_fold_apply's thread 0, on every wave
    auto k0_fold = [&](int row_, bool wr_) {                 // wr_: this thread publishes the row's scale / gate
#pragma clang fp contract(off)
        const int n_ = k0_row_len(row_);
        const int n_chunks = (n_ + kNpChunk - 1) / kNpChunk;
        const float* cs = p.k0_chunk_sums + (long long)row_ * p.k0_chunk_stride;
        float total = 0.0f, sc_ = 1.0f;
        int good_ = 0;
        for (int c = 0; c < n_chunks; ++c) total = c == 0 ? cs[0] : total + cs[c];
        if (n_ > 0) {
            const float rms = sqrtf(total / (float)n_);
            if (p.k0_scale_f64) { if (!((double)rms < 1e-4)) { sc_ = (float)(0.05 / ((double)rms + 1e-8)); good_ = 1; } }
            else if (!(rms < 1e-4f)) { sc_ = 0.05f / (rms + 1e-8f); good_ = 1; }
        }
        if (wr_) { p.k0_scale[row_] = sc_; if (p.k0_ok) p.k0_ok[row_] = good_; }
        return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(good_ ? sc_ : 1.0f)));
    };
    // CQ (round 4, "stft_pair_cq"): ONE queue over ALL the rows of the workgroup (rows blockIdx.x + k gridDim.x, k = 0 .. cq_rows - 1, row k's
    // mel power in image k & 1) and NO workgroup barrier behind the tables': the wave that completes a row's last pair runs the row's dB
    // stage ALONE (db_row_wave_call: db_row_from_lds's sums in the same order - twelve virtual waves one after the other - the same bits)
    // while the other eleven are already on the next row's pairs.  The barrier form pays, per two rows, the queue's tail (waves that find
    // it empty wait ~12 600 cycles for the last pair) and a dB stage every thread waits through (~14 500) out of ~131 000.
    // LDS words (cq[], on the reduction scratch the barrier forms use): [0, 1] pairs of the image's row completed, [2, 3] the row's largest
    // mel power so far (bits of a non-negative float: unsigned order), [4, 5] ordinal + 1 of the last row whose dB stage has left the image.
    int* const cq = reinterpret_cast<int*>(db_red0);
    for (int row0 = blockIdx.x; row0 < n_rows; row0 += CQ ? n_rows : RB * gridDim.x) {
        const int row1 = row0 + (int)gridDim.x;             // (G2: the group's second row, if the batch has it)
        const bool has1 = G2 && row1 < n_rows;
        const int cq_rows = CQ ? (n_rows - 1 - row0) / (int)gridDim.x + 1 : 0;
        const int n_q = CQ ? cq_rows * n_pairs : has1 ? 2 * n_pairs : n_pairs;       // the group's queue: row 0's pairs, then row 1's
        float sc_row0 = 1.0f, sc_row1 = 1.0f;
        if (K0F == 1 && !CQ) { sc_row0 = k0_fold(row0, tid == 0); if (has1) sc_row1 = k0_fold(row1, tid == 0); }
        int cq_k = 0, cq_base = 0;                          // CQ: ordinal of the current pair's row, queue index of that row's first pair
        int cq_nrow = 0, cq_npr = 0;                        // ... row and pair of the wave's next draw
        float cq_sc = 1.0f;
        if (CQ && K0F == 1) cq_sc = k0_fold(row0, false);
        float wmax = -INFINITY, wmax1 = -INFINITY;          // the largest mel power this wave has stored for the row(s) (fused dB stage: its max pass)
        for (int pi = wave, pr_next = 0; pi < n_q; pi = pr_next) {
            // (the queue index through readfirstlane: `wave` is uniform only in fact, and a row / pair index the compiler takes for
            // divergent puts the buffer loads of the fetch into waterfall loops)
            pi = __builtin_amdgcn_readfirstlane(pi);
            if constexpr (CQ) {
                bool moved = false;
                while (pi >= cq_base + n_pairs) { cq_base += n_pairs; ++cq_k; moved = true; }      // (wave-uniform; one step, rarely two)
                if (moved && K0F == 1) cq_sc = k0_fold(row0 + cq_k * (int)gridDim.x, false);
            }
            const bool sub = CQ ? (cq_k & 1) != 0 : G2 && pi >= n_pairs;           // (wave-uniform) which row of the group / which image
            const int row = CQ ? row0 + cq_k * (int)gridDim.x : sub ? row1 : row0;
            const int pr = pair_of(CQ ? pi - cq_base : sub ? pi - n_pairs : pi);    // (identity without K0F)
            if (K0F == 1) k0_sc = CQ ? cq_sc : sub ? sc_row1 : sc_row0;
            // packed pairs: element j of the lane's TWO sequences (sample parity 0 / 1) side by side - dft16_pk
            amph_f2 XR[16], XI[16];
            float4 tw[16];
#ifdef AMPH_STAMPS
            if (tl_round < 4) {                 // per wave and round: start (here) and end, third region of the stamp buffer
                unsigned long long t_;
                asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
                if (lane == 0) p.stamps[32 * 4096 + ((size_t)blockIdx.x * 16 + wave) * 8 + 2 * tl_round] = t_;
            }
#endif
            AMPH_PSTAMP(0)
            // ---- decode + window: z = w (a + i b); the pair is the two int16 of one dword times the window pair (w[n], w[n + 1])
            // K0F: every dword is decoded and re-quantised ONCE - q' = rint(clip(q sc, +-32768) (32767 / 32768)), the PCM16 sample
            // 00_normalize_dataset_rms.py:29-57 writes (x = q / 32768, x sc, clip(+-1), x 32767, rint: the powers of two commute with the
            // roundings; a gated row has sc = 1 and the clip is the identity) - and serves frame a at j and frame b at j - HJ
            auto requant = [&](uint32_t d) {
#pragma clang fp contract(off)
                if constexpr (K0F == 2) return amph_f2{(float)(short)(d & 0xFFFFu), (float)((int)d >> 16)};       // (no RMS stage: the sample as it is)
#ifdef AMPH_ABL_NOREQ                          /* timing-only ablation: what the re-quantisation costs */
                return amph_f2{(float)(short)(d & 0xFFFFu), (float)((int)d >> 16)} * k0_sc;
#endif
                amph_f2 t = amph_f2{(float)(short)(d & 0xFFFFu), (float)((int)d >> 16)} * k0_sc;
                t = amph_f2{__builtin_amdgcn_fmed3f(t.x, -32768.0f, 32768.0f), __builtin_amdgcn_fmed3f(t.y, -32768.0f, 32768.0f)} * (32767.0f / 32768.0f);
                return amph_f2{__builtin_rintf(t.x), __builtin_rintf(t.y)};
            };
            amph_f2 fq[K0F ? ND : 1];
            if (K0F) {
#pragma unroll
                for (int j = 0; j < HJ; ++j) fq[j] = requant(nx[j]);
            }
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float2 wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = win[(8 * g + j) * 64 + lane];
#pragma unroll
                for (int j8 = 0; j8 < 8; ++j8) {
                    const int j = 8 * g + j8;
                    const amph_f2 w2 = {wv[j8].x, wv[j8].y};
                    if constexpr (K0F) {
                        fq[j + HJ] = requant(nx[j + HJ]);
                        XR[j] = fq[j] * w2;
                        XI[j] = fq[j + HJ] * w2;
                    } else {
                        const uint32_t da = nx[j], db = nx[j + HJ];
                        XR[j] = amph_f2{(float)(short)(da & 0xFFFFu), (float)((int)da >> 16)} * w2;
                        XI[j] = amph_f2{(float)(short)(db & 0xFFFFu), (float)((int)db >> 16)} * w2;
                    }
                }
                if (g == 0) AMPH_PHASE_FENCE();
            }
            AMPH_PSTAMP(1)
            // ---- stage-1 twiddles from global memory (!TWL): the first half is requested here, the second behind the DFT16
#pragma unroll
            for (int k8 = 0; k8 < 8; ++k8) if (!TWL) tw[k8] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_tw, lane16, k8 * 1024, 0));
            AMPH_PHASE_FENCE();
            // ---- stage 1: DFT32 down the lane's column: both 16-point halves at once, the swap that brings the halves of ONE column
            //      together (high lanes' sequence 0 <-> low lanes' sequence 1), the radix-2 combine
            dft16_pk(XR, XI);
            AMPH_PHASE_FENCE();
#pragma unroll
            for (int k8 = 8; k8 < 16; ++k8) if (!TWL) tw[k8] = __builtin_bit_cast(float4, __builtin_amdgcn_raw_buffer_load_b128(rs_tw, lane16, k8 * 1024, 0));
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const auto sr = __builtin_amdgcn_permlane16_swap(__float_as_uint(XR[q].x), __float_as_uint(XR[q].y), false, false);
                const auto si = __builtin_amdgcn_permlane16_swap(__float_as_uint(XI[q].x), __float_as_uint(XI[q].y), false, false);
                XR[q] = amph_f2{__uint_as_float(sr[0]), __uint_as_float(sr[1])};
                XI[q] = amph_f2{__uint_as_float(si[0]), __uint_as_float(si[1])};
            }
            dft32_combine_pk(XR, XI);
            AMPH_PHASE_FENCE();
            AMPH_PSTAMP(2)
            // ---- twiddle W_2048^(col k1): the pair (Y[k], Y[k + 16]) times the pair of its twiddles (LDS-resident table: four
            //      pairs at a time - 16 registers, not 64)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                if (TWL) {
#pragma unroll
                    for (int k4 = 0; k4 < 4; ++k4) tw[4 * g + k4] = tw1_l[(4 * g + k4) * 64];
                }
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const int kk = 4 * g + k4, q = DFT16_POS(kk);
                    const amph_f2 wr = {tw[kk].x, tw[kk].y}, wi = {tw[kk].z, tw[kk].w};
                    AMPH_CMUL_T(XR[q], XI[q], wr, wi);
                }
                if (TWL && g < 3) AMPH_PHASE_FENCE();
            }
            // ---- exchange, real parts then imaginary parts.  LDS operations of a wave execute in order: the reads of the first
            //      pass are queued behind its writes and in front of the second pass's writes, nothing waits in between.  A 16-byte
            //      read returns mm = 4 c .. 4 c + 3 = the pairs (p = 0, 1) of m = 2 c and 2 c + 1: no register shuffling
            amph_f2 UR[16], UI[16];
#ifdef AMPH_ABL_PAIR_NOEX                  /* timing-only ablation: no exchange through LDS at all (64 stores + 16 reads less), same arithmetic */
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) { UR[kk] = XR[kk]; UI[kk] = XI[kk]; }
#else
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                AMPH_DS_ST32(st_col, kk * kPairExS * 4, XR[DFT16_POS(kk)].x);
                AMPH_DS_ST32(st_col, (kk + 16) * kPairExS * 4, XR[DFT16_POS(kk)].y);
            }
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(T + k1p * kPairExS + 32 * h + 4 * c);
                UR[2 * c] = amph_f2{v.x, v.y}; UR[2 * c + 1] = amph_f2{v.z, v.w};
            }
            wave_lds_sync();
#ifndef AMPH_ABL_PAIR_NOIMAG               /* timing-only ablation: 32 LDS stores less per pair, same arithmetic */
#pragma unroll
            for (int kk = 0; kk < 16; ++kk) {
                AMPH_DS_ST32(st_col, kk * kPairExS * 4, XI[DFT16_POS(kk)].x);
                AMPH_DS_ST32(st_col, (kk + 16) * kPairExS * 4, XI[DFT16_POS(kk)].y);
            }
#else
            asm volatile("" :: "v"(XI[0].x), "v"(XI[1].x), "v"(XI[2].x), "v"(XI[3].x), "v"(XI[4].x), "v"(XI[5].x), "v"(XI[6].x), "v"(XI[7].x));
            asm volatile("" :: "v"(XI[8].x), "v"(XI[9].x), "v"(XI[10].x), "v"(XI[11].x), "v"(XI[12].x), "v"(XI[13].x), "v"(XI[14].x), "v"(XI[15].x));
            asm volatile("" :: "v"(XI[0].y), "v"(XI[1].y), "v"(XI[2].y), "v"(XI[3].y), "v"(XI[4].y), "v"(XI[5].y), "v"(XI[6].y), "v"(XI[7].y));
            asm volatile("" :: "v"(XI[8].y), "v"(XI[9].y), "v"(XI[10].y), "v"(XI[11].y), "v"(XI[12].y), "v"(XI[13].y), "v"(XI[14].y), "v"(XI[15].y));
#endif
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(T + k1p * kPairExS + 32 * h + 4 * c);
                UI[2 * c] = amph_f2{v.x, v.y}; UI[2 * c + 1] = amph_f2{v.z, v.w};
            }
            wave_lds_sync();
#endif
            AMPH_PHASE_FENCE();
            AMPH_PSTAMP(3)
            // ---- stage 2: radix-2 DIF across the lane pair (DPP), twiddle (h = 1: W_64^mm; (1, 0) on h = 0), DFT32
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                float4 t2[4];
#pragma unroll
                for (int m = 0; m < 4; ++m) t2[m] = tw2[h * kPairTw2S + 4 * g + m];
#pragma unroll
                for (int m4 = 0; m4 < 4; ++m4) {
                    const int m = 4 * g + m4;
                    amph_f2 r = UR[m] * sgn + amph_f2{dpp_xor1(UR[m].x), dpp_xor1(UR[m].y)};
                    amph_f2 i = UI[m] * sgn + amph_f2{dpp_xor1(UI[m].x), dpp_xor1(UI[m].y)};
                    const amph_f2 wr = {t2[m4].x, t2[m4].y}, wi = {t2[m4].z, t2[m4].w};
                    AMPH_CMUL_T(r, i, wr, wi);
                    UR[m] = r; UI[m] = i;
                }
                AMPH_PHASE_FENCE();
            }
            AMPH_PSTAMP(4)
            dft16_pk(UR, UI);
            dft32_combine_pk(UR, UI);          // .x: X[lane' + 64 k] (own bins), .y: X[lane' + 64 (k + 16)] (partners of another lane)
            AMPH_PHASE_FENCE();
            AMPH_PSTAMP(5)
            // ---- split: the partner values Z[N - k] come from the upper half (k2' >= 16) of lane' colp; both passes are queued
            //      back to back like the exchange's
            float pre[16], pim[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) AMPH_DS_ST32(st_zw, r * kPairZS * 4, UR[DFT16_POS(r)].y);
            if (lanep == 0) T[16 * kPairZS] = UR[DFT16_POS(0)].x;
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 16; ++i) pre[i] = T[(15 - i) * kPairZS + zr_];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) AMPH_DS_ST32(st_zw, r * kPairZS * 4, UI[DFT16_POS(r)].y);
            if (lanep == 0) T[16 * kPairZS] = UI[DFT16_POS(0)].x;
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 16; ++i) pim[i] = T[(15 - i) * kPairZS + zr_];
            wave_lds_sync();
            AMPH_PHASE_FENCE();
            AMPH_PSTAMP(6)
            // ---- (4 |A[k]|^2, 4 |B[k]|^2) as one pair: A = (Z[k] + conj Z[N-k]) / 2, B = (Z[k] - conj Z[N-k]) / 2i:
            //      (ar, bi) = (pr + zr, pr - zr), (ai, br) = (zi - pi, zi + pi)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float zr = UR[DFT16_POS(i)].x, zi = UI[DFT16_POS(i)].x;
                const amph_f2 u = amph_f2{pre[i], pre[i]} + amph_f2{zr, -zr};
                const amph_f2 v = amph_f2{zi, zi} + amph_f2{-pim[i], pim[i]};
                const amph_f2 pw = u * u + v * v;
                AMPH_DS_ST64(st_p2, 512 * i, pw);
            }
            if (lanep == 0) {                               // bin 1024 is its own partner
                const float zr = UR[DFT16_POS(0)].y, zi = UI[DFT16_POS(0)].y;
                P2[1024] = amph_f2{4.0f * zr * zr, 4.0f * zi * zi};
            }
            for (int i = lane; i < p.p_tail; i += 64) P2[1025 + i] = amph_f2{0.0f, 0.0f};     // the padded mel rows read there
            wave_lds_sync();
            AMPH_PHASE_FENCE();
            AMPH_PSTAMP(7)
            // ---- this wave's next pair: its samples are requested now and land under the mel rows
            //      (UNCONDITIONALLY - behind the last pair the row index is clamped and the samples are never used: a conditional
            //      fetch makes the old dwords the value of the not-taken path, i.e. 19 registers live through every phase of the pair)
            // CQ: a wave that sees the row's other pairs all counted will complete the row and run its dB stage (tens of thousands of cycles):
            // it draws its next pair BEHIND that stage - a pair drawn here would sit undone through it (measured: the workgroup's last
            // barrier 24 000 cycles later) - and requests this pair's samples again instead (the registers must be written on every path)
            bool cq_defer = false;
            if constexpr (CQ)
                cq_defer = cq_k != cq_rows - 1 &&
                           __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cq[(int)sub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)) == n_pairs - 1;
            auto cq_draw = [&]() {
                int got = 0;
                if (lane == 0) got = atomicAdd(next_pair, 1);
                pr_next = __builtin_amdgcn_readfirstlane(got);
                int npr = min(pr_next, n_q - 1);             // (behind the queue's end: any valid pair, never used)
                int nk = cq_k, nb = cq_base;
                while (npr >= nb + n_pairs) { nb += n_pairs; ++nk; }
                cq_nrow = row0 + nk * (int)gridDim.x; cq_npr = pair_of(npr - nb);
            };
            if (CQ && cq_defer) fetch(row, pr);
            else if constexpr (CQ) { cq_draw(); fetch(cq_nrow, cq_npr); }
            else {
                int got = 0;
                if (lane == 0) got = atomicAdd(next_pair, 1);
                pr_next = __builtin_amdgcn_readfirstlane(got);
                int nrow = row0, npr = pr_next;
                if (npr >= n_q) { nrow = row0 + RB * (int)gridDim.x; npr = wave; }     // nothing left in this group: the wave's first pair of its next one
                else if (G2 && npr >= n_pairs) { nrow = row1; npr -= n_pairs; }
                fetch(min(nrow, n_rows - 1), pair_of(min(npr, n_pairs - 1)));
            }
            AMPH_PHASE_FENCE();
#ifdef AMPH_STAMPS
            if (wave == 0 && pr == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) p.stamps[(size_t)blockIdx.x * 16 + 8] = t_; }
#endif
            if constexpr (CQ) {
                if (cq_k >= 2) {
                    // image cq_k & 1 still holds row cq_k - 2 until that row's dB stage is through (normally long done).  Waited for HERE,
                    // not where the pair is drawn: only the mel rows below touch the image, the pair's FFT has run meanwhile.  The wait
                    // cannot deadlock - that stage waits for nothing, and every pair of the earlier rows is held by a wave whose own wait
                    // is for an earlier stage - and is bounded anyway (~1 ms: a GPU that is not hung can be debugged).
                    const int need = cq_k - 1;
                    for (int spin = 0; spin < 4096; ++spin) {
                        const int got_ = __builtin_amdgcn_readfirstlane(__hip_atomic_load(&cq[4 + (cq_k & 1)], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
                        if (got_ >= need) break;
                        __builtin_amdgcn_s_sleep(8);
                    }
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                }
            }
            // ---- mel rows of both frames at once: 4 lanes per filter, 4 bins per lane and step; the accumulators are (frame a, frame b)
            //      pairs, a bin's two powers one 8-byte LDS element.  Four steps are requested at a time; even steps -> first
            //      accumulator, odd -> second, as in k_stft_mel_w.  Lane q of a quad takes bins 2q, 2q + 1 and 8 + 2q, 8 + 2q + 1 of a
            //      step: a quad's read is 64 contiguous bytes, and the host has placed the filters on the quads (and moved their start
            //      bins) so that the four quads of a ds_read_b128 lane group tile the 256-byte bank row (build_pair_mel, amph_api.hip:
            //      302 -> 104 array cycles of power reads per pair; the band index comes from the table)
            {
                const int q = lane & 3;
                const int n_rounds = (p.n_mels + 15) >> 4;
                const int fa = 2 * pr;
                const bool has_b = fa + 1 < p.n_frames;     // wave-uniform
                float* out_a = fuse_db ? mp_row + (sub ? mp_floats : 0) + fa * p.n_mels : p.melpow + ((long long)row * p.n_frames + fa) * p.n_mels;
                float* out_b = out_a + p.n_mels;
                int4 mt = mel_meta[lane >> 2];
                for (int r = 0; r < n_rounds; ++r) {
                    const int steps = __builtin_amdgcn_readfirstlane(mt.y);
                    const float4* w = reinterpret_cast<const float4*>(mel_w + mt.z) + q;
                    const float4* pp = reinterpret_cast<const float4*>(P2 + mt.x) + q;         // bins 2 q, 2 q + 1 | (+ 4:) 8 + 2 q, 8 + 2 q + 1 of a step
                    const int m = mt.w;                         // the band this quad computes (-1: none)
                    if (r + 1 < n_rounds) mt = mel_meta[16 * (r + 1) + (lane >> 2)];
                    amph_f2 a0 = {0.0f, 0.0f}, a1 = {0.0f, 0.0f};
#define AMPH_MEL_FMA2(acc_, w_, x_, y_)                                                    \
    acc_ += w_.x * amph_f2{x_.x, x_.y}; acc_ += w_.y * amph_f2{x_.z, x_.w}; acc_ += w_.z * amph_f2{y_.x, y_.y}; acc_ += w_.w * amph_f2{y_.z, y_.w};
                    int i0 = 0;
                    for (; i0 + 4 <= steps; i0 += 4) {
                        float4 wq[4], xq[4], yq[4];
#pragma unroll
                        for (int u = 0; u < 4; ++u) { wq[u] = w[4 * (i0 + u)]; xq[u] = pp[8 * (i0 + u)]; yq[u] = pp[8 * (i0 + u) + 4]; }
                        AMPH_MEL_FMA2(a0, wq[0], xq[0], yq[0])
                        AMPH_MEL_FMA2(a1, wq[1], xq[1], yq[1])
                        AMPH_MEL_FMA2(a0, wq[2], xq[2], yq[2])
                        AMPH_MEL_FMA2(a1, wq[3], xq[3], yq[3])
                    }
                    if (i0 + 2 <= steps) {
                        float4 wq[2], xq[2], yq[2];
#pragma unroll
                        for (int u = 0; u < 2; ++u) { wq[u] = w[4 * (i0 + u)]; xq[u] = pp[8 * (i0 + u)]; yq[u] = pp[8 * (i0 + u) + 4]; }
                        AMPH_MEL_FMA2(a0, wq[0], xq[0], yq[0])
                        AMPH_MEL_FMA2(a1, wq[1], xq[1], yq[1])
                        i0 += 2;
                    }
                    if (i0 < steps) {
                        const float4 w0 = w[4 * i0], x0 = pp[8 * i0], y0 = pp[8 * i0 + 4];
                        AMPH_MEL_FMA2(a0, w0, x0, y0)
                    }
#undef AMPH_MEL_FMA2
                    amph_f2 sa = a0 + a1;
                    sa += amph_f2{dpp_xor1(sa.x), dpp_xor1(sa.y)};
                    sa += amph_f2{dpp_xor2(sa.x), dpp_xor2(sa.y)};
                    if (q == 0 && m >= 0) {
                        float wm_ = sa.x;
                        out_a[m] = sa.x;
                        if (has_b) { out_b[m] = sa.y; wm_ = fmaxf(wm_, sa.y); }
                        if (sub) wmax1 = fmaxf(wmax1, wm_); else wmax = fmaxf(wmax, wm_);
                    }
                }
            }
            wave_lds_sync();                                // the scratch is rewritten by the next pair
            AMPH_PHASE_FENCE();
            if constexpr (CQ) {
                // the pair's mel rows are in the image (LDS operations of a wave complete in order): publish the wave's maximum, count the pair;
                // whoever counts the row's last pair owns the image until its dB stage is through
                const float wm_ = wave_max_dpp(fmaxf(fmaxf(wmax, wmax1), 0.0f));
                wmax = -INFINITY; wmax1 = -INFINITY;
                int old_ = 0;
                if (lane == 63) {
                    __hip_atomic_fetch_max(reinterpret_cast<unsigned*>(&cq[2 + (int)sub]), __float_as_uint(wm_), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    old_ = __hip_atomic_fetch_add(&cq[(int)sub], 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
                }
                old_ = __builtin_amdgcn_readlane(old_, 63);
                if (__builtin_expect(old_ == n_pairs - 1 && cq_k != cq_rows - 1, 0)) {     // (the workgroup's LAST row: behind the queue, by every wave)
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
                    const float mx_ = __uint_as_float((unsigned)__builtin_amdgcn_readfirstlane(__hip_atomic_load(&cq[2 + (int)sub], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP)));
                    if (K0F == 1) (void)k0_fold(row, lane == 0);
#ifdef AMPH_STAMPS
                    if (cq_k == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) tl[9] = t_; }
#endif
                    __builtin_amdgcn_s_setprio(3);       // (the row's last pair is as a rule a SIMD's youngest wave's: without this the stage runs on what the two older waves leave)
                    db_row_wave_call<kThreads>(mp_row + (sub ? mp_floats : 0), p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                               p.db_out + (long long)row * p.db_target_frames * p.n_mels, mx_);
                    __builtin_amdgcn_s_setprio(0);
#ifdef AMPH_STAMPS
                    if (cq_k == 0) { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); if (lane == 0) tl[10] = t_; }
#endif
                    if (lane == 0) { cq[(int)sub] = 0; cq[2 + (int)sub] = 0; }
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
                    if (lane == 0) __hip_atomic_store(&cq[4 + (int)sub], cq_k + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    if (!cq_defer) fetch(cq_nrow, cq_npr);    // (again: the samples requested above need not live through the call)
                }
                if (cq_defer) { cq_draw(); fetch(cq_nrow, cq_npr); }
            }
            AMPH_PSTAMP(9)
#ifdef AMPH_STAMPS
            {
                unsigned long long t_;
                asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");
                if (lane == 0 && tl_round < 4) { atomicMax(&tl[2 + tl_round], t_); p.stamps[32 * 4096 + ((size_t)blockIdx.x * 16 + wave) * 8 + 2 * tl_round + 1] = t_; }
                ++tl_round;
            }
#endif
        }
#undef AMPH_PHASE_FENCE
        // ---- fused dB stage on the row the workgroup holds (see k_stft_mel_w).  The LAST row (at batch 256: the only one) is
        //      finished behind the loops: inlined here, the stage's per-thread index arithmetic is hoisted in front of the row loop
        //      and sits in ~90 registers across the pair loop
        if constexpr (CQ) {
            // the last row's stage by the whole workgroup (nothing else is left to do; the single-wave form would be ~28 000 cycles alone)
            const int lk = cq_rows - 1, lrow = row0 + lk * (int)gridDim.x;
            float* const limg = mp_row + ((lk & 1) ? mp_floats : 0);
            __syncthreads();                                // every pair of the row is in the image, every other row's stage is through
#ifdef AMPH_STAMPS
            if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tl[6] = t_; }
#endif
            const float mx_ = __uint_as_float((unsigned)cq[2 + (lk & 1)]);
            if (K0F == 1) (void)k0_fold(lrow, tid == 0);
            __syncthreads();                                // (the queue's words lie where the stage keeps its sums)
            if (tid < NW) db_redf[tid] = mx_;
            __syncthreads();
            db_row_call<kThreads, 8, true>(limg, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                           p.db_out + (long long)lrow * p.db_target_frames * p.n_mels, db_red0);
#ifdef AMPH_STAMPS
            __syncthreads();
            if (tid == 0) { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tl[7] = t_; tl[8] = __builtin_amdgcn_s_memrealtime() - pw_rt0; }
#endif
            break;
        }
        const bool more = row0 + RB * (int)gridDim.x < n_rows;   // another group follows
        if (!fuse_db && more) {                             // (every wave has drawn its last index of this group before the counter starts over)
            __syncthreads();
            if (tid == 0) *next_pair = NW;
            __syncthreads();
        }
        if (fuse_db) {                                      // (every wave, also one without pairs: -inf)
            const float wm = wave_max_dpp(wmax);
            if (lane == 63) db_redf[wave] = wm;
            if (G2) { const float wm1 = wave_max_dpp(wmax1); if (lane == 63) db_redf[48 + wave] = wm1; }      // (row 1's scratch: 24 doubles = 48 floats on)
        }
        if (fuse_db && (more || G2)) {                      // (G2: every group's stages here - two inlined copies behind the loop would hold ~90 registers each)
            __syncthreads();
            if (tid == 0) *next_pair = NW;
            if (G2 && has1) {                               // both rows side by side: one row's barriers and latencies for the two
                if constexpr (NW <= 12)
                    db_rows2_call<kThreads, 8>(mp_row, mp_row + mp_floats, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                               p.db_out + (long long)row0 * p.db_target_frames * p.n_mels,
                                               p.db_out + (long long)row1 * p.db_target_frames * p.n_mels, db_red0);
            } else
                db_row_call<kThreads, 8, true>(mp_row, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                      p.db_out + (long long)row0 * p.db_target_frames * p.n_mels, db_red0);
            __syncthreads();
        }
    }
#ifdef AMPH_STAMPS
    if (tid == 0) {
        unsigned long long pw_end;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(pw_end)::"memory");
        for (int k_ = 10; k_ < 16; ++k_) p.stamps[(size_t)blockIdx.x * 16 + k_] = 0;
        p.stamps[(size_t)blockIdx.x * 16 + 13] = pw_end - pw_begin;        // frame loops of wave 0 (without the dB tail)
        p.stamps[(size_t)blockIdx.x * 16 + 14] = __builtin_amdgcn_s_memrealtime() - pw_rt0;
    }
#endif
#undef AMPH_PSTAMP
    if (!G2 && fuse_db && (int)blockIdx.x < n_rows) {
        const int last_row = blockIdx.x + (n_rows - 1 - (int)blockIdx.x) / (int)gridDim.x * (int)gridDim.x;
        __syncthreads();
#ifdef AMPH_STAMPS
        if (tid == 0) { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tl[6] = t_; }
#endif
        db_row_from_lds<kThreads, 8, true>(mp_row, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                  p.db_out + (long long)last_row * p.db_target_frames * p.n_mels, db_red0
#ifdef AMPH_STAMPS
                                  , tl + 9
#endif
                                  );
#ifdef AMPH_STAMPS
        __syncthreads();
        if (tid == 0) { unsigned long long t_; asm volatile("s_waitcnt vmcnt(0)\n\ts_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); tl[7] = t_; tl[8] = __builtin_amdgcn_s_memrealtime() - pw_rt0; }
#endif
    }
}

// ----------------------------------------------------------------------------
// The same for n_fft = 1024 / hop 256 ("pair2"; BASELINE configs[3]'s 128-band preset): a pair of frames is one complex 1024-point
// FFT on a HALF wave (32 lanes x 32 points), a wave transforms two consecutive pairs = four consecutive frames per iteration
// (tools/fft_pair_model.py, half_wave_pair_power).  Differences to k_stft_pair: lane' l = lane & 31 reads samples 64 j' + 2 l + e
// (frame b = dword j' + 4), v_permlane16_swap pairs the column halves inside a half wave, the second stage needs NO cross-lane
// step and no twiddle (lane k1' reads the whole exchange row T[k1'][0..31]), the partners N - k sit in lane' (32 - l) & 31, and
// the mel rows of four frames share one weight read.
// ----------------------------------------------------------------------------
constexpr int kP2ExS = 36;                          // exchange row stride in floats (= 4 mod 64 over 16 rows: conflict-free 16-byte reads)
constexpr int kP2Half = 32 * kP2ExS * 4;            // one half wave's tile: exchange (4 608 B), then partners (17 x 32 floats), then two power spectra
constexpr int kP2Scratch = 2 * kP2Half;
constexpr int kP2PStride = kP2Half / 8;             // floats between two power spectra (576 >= 513 + tail)
constexpr int kP2WinBytes = 16 * 32 * 8, kP2Tw1Bytes = 31 * 32 * 8;

// K0F (round 4, as in k_stft_pair): 1 = the row's RMS scale folded from K0's chunk sums on every wave, the RAW rows read (centre padding,
// fix_length and truncation as index arithmetic on the two slots of a row that touch them), the samples re-quantised in the decode - K0's
// second kernel and its padded copy of every row are not launched; 2 = raw rows as they are (no RMS stage: no pad kernel either).
template <int NW, bool TWL, bool RDB = false, int K0F = 0>
__global__ __launch_bounds__(64 * NW) void k_stft_pair2(StftParams p, int n_rows) {
    constexpr int kThreads = 64 * NW;
    constexpr int HJ = 4, ND = 16 + HJ;
    constexpr int kDbC = NW <= 8 ? 24 : 16;              // dB values a thread keeps in registers (87 x 128 on 512 threads: 22)
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
    unsigned q_kt0;
    { unsigned long long t_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); q_kt0 = (unsigned)t_; }
#endif
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int hw = lane >> 5, lp = lane & 31;
    const float2* win = reinterpret_cast<const float2*>(smem) + lp;                      // [16][32] (w[n], w[n + 1]) 2^-15
    constexpr int kTab = kP2WinBytes + (TWL ? kP2Tw1Bytes : 0);
    const float2* tw1_l = reinterpret_cast<const float2*>(smem + kP2WinBytes) + lp;      // [31][32] (TWL)
    const int meta_bytes = (p.n_mels * 12 + 15) & ~15;
    const int melw_bytes = p.mel_w_floats * 4;
    int3* mel_meta = reinterpret_cast<int3*>(smem + kTab);
    float* mel_w = reinterpret_cast<float*>(smem + kTab + meta_bytes);
    char* wbase = smem + kTab + meta_bytes + melw_bytes + wave * kP2Scratch;
    float* Tw = reinterpret_cast<float*>(wbase);                                         // the wave's scratch (four power spectra)
    float* T = reinterpret_cast<float*>(wbase + hw * kP2Half);                           // this half wave's tile
    float* Pa = T;
    float* Pb = T + kP2PStride;
    float* mp_row = reinterpret_cast<float*>(smem + kTab + meta_bytes + melw_bytes + NW * kP2Scratch);
    const bool fuse_db = !RDB && p.db_out != nullptr;
    {
        const float4* src = reinterpret_cast<const float4*>(p.pair_tables);
        float4* dst = reinterpret_cast<float4*>(smem);
        for (int i = tid; i < kP2WinBytes / 16; i += kThreads) dst[i] = src[i];
        if (TWL) {
            const float4* s1 = reinterpret_cast<const float4*>(p.pair_tw1);
            float4* d1 = reinterpret_cast<float4*>(smem + kP2WinBytes);
            for (int i = tid; i < kP2Tw1Bytes / 16; i += kThreads) d1[i] = s1[i];
        }
        for (int m = tid; m < p.n_mels; m += kThreads) mel_meta[m] = make_int3(p.mel_start[m], p.mel_len[m], p.mel_wofs[m]);
        const float4* wsrc = reinterpret_cast<const float4*>(p.mel_w);
        float4* wdst = reinterpret_cast<float4*>(mel_w);
        for (int i = tid; i < (p.mel_w_floats >> 2); i += kThreads) {
            float4 v = wsrc[i];
            v.x *= 0.25f; v.y *= 0.25f; v.z *= 0.25f; v.w *= 0.25f;
            wdst[i] = v;
        }
    }
    __syncthreads();

    const int n_pairs = (p.n_frames + 1) >> 1, n_slots = (n_pairs + 1) >> 1;      // a slot = two pairs = four frames of one wave iteration
    const int col = 2 * (lp & 15) + (lp >> 4);
    const int zr_ = ((32 - lp) & 31) + (lp == 0 ? 32 : 0);
    const __amdgpu_buffer_rsrc_t rs_tw = __builtin_amdgcn_make_buffer_rsrc(const_cast<float2*>(p.pair_tw1), 0, -1, 0x00020000);
    // (num_records = the rows' bytes: the frames of a pair past the row's last one read zeros behind the last row, finite samples of
    // the next row elsewhere; their results are never stored)
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.src.wave), 0, (int)((long long)n_rows * p.src.row_stride * 2), 0x00020000);
    const int voff_x = lp * 4 + hw * p.hop * 4, lp8 = lp * 8;
    uint32_t nx[ND];
#pragma unroll
    for (int j = 0; j < ND; ++j) nx[j] = 0;
    const __amdgpu_buffer_rsrc_t rs_raw = __builtin_amdgcn_make_buffer_rsrc(const_cast<int16_t*>(p.k0_in), 0, K0F ? (int)((long long)n_rows * p.k0_row_stride * 2) : 0, 0x00020000);
    auto k0_row_len = [&](int row_) {                        // samples of the row that count (wave-uniform)
        const int n_ = p.k0_lengths ? max(0, min(p.k0_lengths[row_], p.k0_n_in)) : p.k0_n_in;
        return __builtin_amdgcn_readfirstlane(n_);
    };
    auto fetch = [&](int row_, int q_) {
        if constexpr (K0F) {
            // slot q = padded samples 4 q hop + [0, 2 hop + 64 ND) = raw s0 + ..., s0 = 4 q hop - lead; this half wave's pair starts 2 hop hw further
            const int lead = p.n_fft >> 1, n_fix = p.n_samples;
            const int nv = min(k0_row_len(row_), n_fix), nv2 = nv & ~1;
            const int rbase = __builtin_amdgcn_readfirstlane((int)((long long)row_ * p.k0_row_stride * 2));
            const int s0 = __builtin_amdgcn_readfirstlane(q_ * 4 * p.hop - lead);
            if (s0 >= 0 && s0 + 2 * p.hop + 64 * ND <= nv2) {    // (wave-uniform) every sample of the slot is a sample of the row: 20 of a one-second row's 22 slots
                const int base = rbase + 2 * s0;
#pragma unroll
                for (int j = 0; j < ND; ++j) nx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_raw, voff_x, base + 128 * j, 0);
            } else {
                // the slot touches the centre padding (np.pad mode="reflect" of the fix_length'd row, map_detector_core.py:212-228; mode 1: zeros),
                // fix_length's zero fill or the row's last odd sample: per sample, through a resource that ends at the row's last valid sample
                const __amdgpu_buffer_rsrc_t rs_row = __builtin_amdgcn_make_buffer_rsrc(
                    const_cast<char*>(reinterpret_cast<const char*>(p.k0_in) + (long long)row_ * p.k0_row_stride * 2), 0, 2 * nv, 0x00020000);
                uint32_t lo[ND], hi[ND];
#pragma unroll
                for (int j = 0; j < ND; ++j) {
                    uint32_t half[2];
#pragma unroll
                    for (int e = 0; e < 2; ++e) {
                        int sidx = s0 + 2 * p.hop * hw + 64 * j + 2 * lp + e;
                        if (p.pad_mode == 0) {
                            sidx = sidx < 0 ? -sidx : sidx;
                            if (sidx >= n_fix) sidx = 2 * (n_fix - 1) - sidx;
                        }
                        const bool in = (unsigned)sidx < (unsigned)nv;
                        half[e] = (uint32_t)(uint16_t)__builtin_amdgcn_raw_buffer_load_b16(rs_row, in ? 2 * sidx : 0x7FFFFFF0, 0, 0);
                    }
                    lo[j] = half[0]; hi[j] = half[1];
                }
#pragma unroll
                for (int j = 0; j < ND; ++j) nx[j] = lo[j] | (hi[j] << 16);
            }
        } else {
            const int base = __builtin_amdgcn_readfirstlane((int)(((long long)row_ * p.src.row_stride) * 2 + (long long)q_ * p.hop * 8));
#pragma unroll
            for (int j = 0; j < ND; ++j) nx[j] = __builtin_amdgcn_raw_buffer_load_b32(rs_x, voff_x, base + 128 * j, 0);
        }
    };
    if ((int)blockIdx.x < n_rows && wave < n_slots) fetch(blockIdx.x, wave);
    // the chunk sums of K0's first kernel, folded in numpy's order (chunk 0 first) into the scale - k_rms_fold_apply's thread 0, on every wave
    auto k0_fold = [&](int row_) {
#pragma clang fp contract(off)
        const int n_ = k0_row_len(row_);
        const int n_chunks = (n_ + kNpChunk - 1) / kNpChunk;
        const float* cs = p.k0_chunk_sums + (long long)row_ * p.k0_chunk_stride;
        float total = 0.0f, sc_ = 1.0f;
        int good_ = 0;
        for (int c = 0; c < n_chunks; ++c) total = c == 0 ? cs[0] : total + cs[c];
        if (n_ > 0) {
            const float rms = sqrtf(total / (float)n_);
            if (p.k0_scale_f64) { if (!((double)rms < 1e-4)) { sc_ = (float)(0.05 / ((double)rms + 1e-8)); good_ = 1; } }
            else if (!(rms < 1e-4f)) { sc_ = 0.05f / (rms + 1e-8f); good_ = 1; }
        }
        if (tid == 0) { p.k0_scale[row_] = sc_; if (p.k0_ok) p.k0_ok[row_] = good_; }
        return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(good_ ? sc_ : 1.0f)));
    };
    float k0_sc = 1.0f;

#define AMPH_PHASE_FENCE() __builtin_amdgcn_sched_barrier(0)
    float mpv[16];                                      // RDB: the row's mel power of this lane (db_row_from_regs)
#pragma unroll
    for (int k = 0; k < 16; ++k) mpv[k] = 0.0f;
    for (int row = blockIdx.x; row < n_rows; row += gridDim.x) {
        int si = 0;
        if (K0F == 1) k0_sc = k0_fold(row);
        for (int q = wave; q < n_slots; q += NW, ++si) {
            float xr[2][16], xi[2][16];
            float2 tw[31];
            {
                // (no contraction of the window products with the first additions of the DFT: the K0F forms must round as the plain one does,
                // and which of an addition's two products the compiler fuses is its own choice per instantiation)
#pragma clang fp contract(off)
                float2 wv[16];
#pragma unroll
                for (int j = 0; j < 16; ++j) wv[j] = win[j * 32];
                // a frame b past the row's last frame must be ZERO, not the next row's samples: the two frames of a pair leak into each
                // other at the rounding level (1e-7 of the LOUDER one)
                const bool has_b = 2 * (2 * q + hw) + 1 < p.n_frames;
                if constexpr (K0F == 1) {
                    // every dword decoded and re-quantised ONCE - q' = rint(clip(q sc, +-32768) (32767 / 32768)), the PCM16 sample
                    // 00_normalize_dataset_rms.py:29-57 writes (see k_stft_pair) - and used by frame a at j and by frame b at j - HJ
                    auto requant = [&](uint32_t d) {
#pragma clang fp contract(off)
                        float t0 = (float)(short)(d & 0xFFFFu) * k0_sc, t1 = (float)((int)d >> 16) * k0_sc;
                        t0 = __builtin_amdgcn_fmed3f(t0, -32768.0f, 32768.0f) * (32767.0f / 32768.0f);
                        t1 = __builtin_amdgcn_fmed3f(t1, -32768.0f, 32768.0f) * (32767.0f / 32768.0f);
                        return make_float2(__builtin_rintf(t0), __builtin_rintf(t1));
                    };
                    float2 fq[ND];
#pragma unroll
                    for (int j = 0; j < HJ; ++j) fq[j] = requant(nx[j]);
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        fq[j + HJ] = requant(nx[j + HJ]);
                        xr[0][j] = fq[j].x * wv[j].x;
                        xr[1][j] = fq[j].y * wv[j].y;
                        xi[0][j] = has_b ? fq[j + HJ].x * wv[j].x : 0.0f;
                        xi[1][j] = has_b ? fq[j + HJ].y * wv[j].y : 0.0f;
                    }
                } else {
#pragma unroll
                    for (int j = 0; j < 16; ++j) {
                        const uint32_t da = nx[j], db = has_b ? nx[j + HJ] : 0u;
                        xr[0][j] = (float)(short)(da & 0xFFFFu) * wv[j].x;
                        xr[1][j] = (float)((int)da >> 16) * wv[j].y;
                        xi[0][j] = (float)(short)(db & 0xFFFFu) * wv[j].x;
                        xi[1][j] = (float)((int)db >> 16) * wv[j].y;
                    }
                }
            }
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) if (!TWL) tw[k1 - 1] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_tw, lp8, (k1 - 1) * 256, 0));
            AMPH_PHASE_FENCE();
            dft16(xr[0], xi[0]);
            AMPH_PHASE_FENCE();
#pragma unroll
            for (int k1 = 16; k1 < 32; ++k1) if (!TWL) tw[k1 - 1] = __builtin_bit_cast(float2, __builtin_amdgcn_raw_buffer_load_b64(rs_tw, lp8, (k1 - 1) * 256, 0));
            AMPH_PHASE_FENCE();
            dft16(xr[1], xi[1]);
#pragma unroll
            for (int qq = 0; qq < 16; ++qq) {               // lanes' 16..31 sequence 0 <-> lanes' 0..15 sequence 1, in both half waves
                const auto sr = __builtin_amdgcn_permlane16_swap(__float_as_uint(xr[0][qq]), __float_as_uint(xr[1][qq]), false, false);
                const auto si = __builtin_amdgcn_permlane16_swap(__float_as_uint(xi[0][qq]), __float_as_uint(xi[1][qq]), false, false);
                xr[0][qq] = __uint_as_float(sr[0]); xr[1][qq] = __uint_as_float(sr[1]);
                xi[0][qq] = __uint_as_float(si[0]); xi[1][qq] = __uint_as_float(si[1]);
            }
            dft32_combine(xr[0], xi[0], xr[1], xi[1]);
            AMPH_PHASE_FENCE();
            if (TWL) {
#pragma unroll
                for (int k1 = 1; k1 < 16; ++k1) tw[k1 - 1] = tw1_l[(k1 - 1) * 32];
            }
#pragma unroll
            for (int k1 = 1; k1 < 16; ++k1) {
                const int qq = DFT16_POS(k1 & 15);
                AMPH_CMUL(xr[0][qq], xi[0][qq], tw[k1 - 1].x, tw[k1 - 1].y);
            }
            if (TWL) {
                AMPH_PHASE_FENCE();
#pragma unroll
                for (int k1 = 16; k1 < 32; ++k1) tw[k1 - 1] = tw1_l[(k1 - 1) * 32];
            }
#pragma unroll
            for (int k1 = 16; k1 < 32; ++k1) {
                const int qq = DFT16_POS(k1 & 15);
                AMPH_CMUL(xr[1][qq], xi[1][qq], tw[k1 - 1].x, tw[k1 - 1].y);
            }
            // ---- exchange inside the half wave: lane k1' reads its whole row
            float ur[2][16], ui[2][16];
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) T[k1 * kP2ExS + col] = xr[k1 >> 4][DFT16_POS(k1 & 15)];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(T + lp * kP2ExS + 4 * c);
                ur[0][2 * c] = v.x; ur[1][2 * c] = v.y; ur[0][2 * c + 1] = v.z; ur[1][2 * c + 1] = v.w;
            }
            wave_lds_sync();
#pragma unroll
            for (int k1 = 0; k1 < 32; ++k1) T[k1 * kP2ExS + col] = xi[k1 >> 4][DFT16_POS(k1 & 15)];
            wave_lds_sync();
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                const float4 v = *reinterpret_cast<const float4*>(T + lp * kP2ExS + 4 * c);
                ui[0][2 * c] = v.x; ui[1][2 * c] = v.y; ui[0][2 * c + 1] = v.z; ui[1][2 * c + 1] = v.w;
            }
            wave_lds_sync();
            AMPH_PHASE_FENCE();
            // ---- stage 2: DFT32 of the row: X[k1' + 32 k2']
            dft16(ur[0], ui[0]);
            dft16(ur[1], ui[1]);
            dft32_combine(ur[0], ui[0], ur[1], ui[1]);
            AMPH_PHASE_FENCE();
            // ---- split: partners N - k from the upper half (k2' >= 16) of lane' (32 - l) & 31
            float pre[16], pim[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r * 32 + lp] = ur[1][DFT16_POS(r)];
            if (lp == 0) T[16 * 32] = ur[0][DFT16_POS(0)];
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 16; ++i) pre[i] = T[(15 - i) * 32 + zr_];
            wave_lds_sync();
#pragma unroll
            for (int r = 0; r < 16; ++r) T[r * 32 + lp] = ui[1][DFT16_POS(r)];
            if (lp == 0) T[16 * 32] = ui[0][DFT16_POS(0)];
            wave_lds_sync();
#pragma unroll
            for (int i = 0; i < 16; ++i) pim[i] = T[(15 - i) * 32 + zr_];
            wave_lds_sync();
            AMPH_PHASE_FENCE();
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float zr = ur[0][DFT16_POS(i)], zi = ui[0][DFT16_POS(i)];
                const float ar = zr + pre[i], ai = zi - pim[i], br = zi + pim[i], bi = pre[i] - zr;
                Pa[lp + 32 * i] = ar * ar + ai * ai;
                Pb[lp + 32 * i] = br * br + bi * bi;
            }
            if (lp == 0) {                                  // bin 512 is its own partner
                const float zr = ur[1][DFT16_POS(0)], zi = ui[1][DFT16_POS(0)];
                Pa[512] = 4.0f * zr * zr;
                Pb[512] = 4.0f * zi * zi;
            }
            for (int i = lp; i < p.p_tail; i += 32) { Pa[513 + i] = 0.0f; Pb[513 + i] = 0.0f; }
            wave_lds_sync();
            AMPH_PHASE_FENCE();
            {                                               // (unconditionally: see k_stft_pair)
                int nrow = row, nq = q + NW;
                if (nq >= n_slots) { nrow = row + gridDim.x; nq = wave; }
                fetch(min(nrow, n_rows - 1), nq);
            }
            AMPH_PHASE_FENCE();
            // ---- mel rows of the slot's four consecutive frames 4 q .. 4 q + 3: one weight read serves all four
            {
                const int qd = lane & 3;
                const int n_rounds = (p.n_mels + 15) >> 4;
                const int f0 = 4 * q;
                const int nf = min(4, p.n_frames - f0);      // wave-uniform: frames of the slot that exist
                float* out0 = fuse_db ? mp_row + f0 * p.n_mels : p.melpow + ((long long)row * p.n_frames + f0) * p.n_mels;
                int3 mt = mel_meta[min(lane >> 2, p.n_mels - 1)];
                for (int r = 0; r < n_rounds; ++r) {
                    const int steps = __builtin_amdgcn_readfirstlane(mt.y);
                    const float4* w = reinterpret_cast<const float4*>(mel_w + mt.z) + qd;
                    const float4* pp = reinterpret_cast<const float4*>(Tw + mt.x) + qd;        // frame f of the slot: + f * kP2PStride floats
                    if (r + 1 < n_rounds) mt = mel_meta[min(16 * (r + 1) + (lane >> 2), p.n_mels - 1)];
                    float acc[4][2];
#pragma unroll
                    for (int f = 0; f < 4; ++f) acc[f][0] = acc[f][1] = 0.0f;
#define AMPH_MEL_FMA(acc_, a_, b_)                                                         \
    acc_ = fmaf(a_.x, b_.x, acc_); acc_ = fmaf(a_.y, b_.y, acc_); acc_ = fmaf(a_.z, b_.z, acc_); acc_ = fmaf(a_.w, b_.w, acc_);
                    int i0 = 0;
                    for (; i0 + 2 <= steps; i0 += 2) {
                        float4 wq[2], xq[2][4];
#pragma unroll
                        for (int u = 0; u < 2; ++u) {
                            wq[u] = w[4 * (i0 + u)];
#pragma unroll
                            for (int f = 0; f < 4; ++f) xq[u][f] = pp[4 * (i0 + u) + f * (kP2PStride / 4)];
                        }
#pragma unroll
                        for (int u = 0; u < 2; ++u)
#pragma unroll
                            for (int f = 0; f < 4; ++f) { AMPH_MEL_FMA(acc[f][u], wq[u], xq[u][f]) }
                    }
                    if (i0 < steps) {
                        const float4 w0 = w[4 * i0];
#pragma unroll
                        for (int f = 0; f < 4; ++f) { const float4 x0 = pp[4 * i0 + f * (kP2PStride / 4)]; AMPH_MEL_FMA(acc[f][0], w0, x0) }
                    }
#undef AMPH_MEL_FMA
                    const int m = 16 * r + (lane >> 2);
                    float keep = 0.0f;
#pragma unroll
                    for (int f = 0; f < 4; ++f) {
                        float sa = acc[f][0] + acc[f][1];
                        sa += dpp_xor1(sa);
                        sa += dpp_xor2(sa);                   // (the same bits in the quad's four lanes)
                        if (RDB) keep = qd == f ? sa : keep;
                        else if (qd == 0 && m < p.n_mels && f < nf) out0[f * p.n_mels + m] = sa;
                    }
                    if (RDB) {
                        const int k = 8 * si + r;             // wave-uniform
#pragma unroll
                        for (int kk = 0; kk < 16; ++kk) mpv[kk] = k == kk ? keep : mpv[kk];
                    }
                }
            }
            wave_lds_sync();
            AMPH_PHASE_FENCE();
        }
        if (RDB) {
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 1
            if (p.n_mels < 0)           // timing only: no dB stage (mpv stays live)
#endif
            db_row_from_regs<NW>(mpv, Tw, wave, n_slots, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                 p.db_out + (long long)row * p.db_target_frames * p.n_mels, reinterpret_cast<double*>(mp_row)
#if defined(AMPH_ABL_RDB) && AMPH_ABL_RDB == 3
                                 , q_kt0
#endif
                                 );
        } else if (fuse_db && row + (int)gridDim.x < n_rows) {
            __syncthreads();
            db_row_call<kThreads, kDbC>(mp_row, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                  p.db_out + (long long)row * p.db_target_frames * p.n_mels,
                                  reinterpret_cast<double*>(mp_row + ((p.n_frames * p.n_mels + 3) & ~3)));
            __syncthreads();
        }
    }
#undef AMPH_PHASE_FENCE
    if (!RDB && fuse_db && (int)blockIdx.x < n_rows) {
        const int last_row = blockIdx.x + (n_rows - 1 - (int)blockIdx.x) / (int)gridDim.x * (int)gridDim.x;
        __syncthreads();
        db_row_from_lds<kThreads, kDbC>(mp_row, p.n_frames, p.n_mels, p.db_target_frames, p.db_layout,
                                  p.db_out + (long long)last_row * p.db_target_frames * p.n_mels,
                                  reinterpret_cast<double*>(mp_row + ((p.n_frames * p.n_mels + 3) & ~3)));
    }
}

size_t stft_pair_lds_bytes(int n_frames, int n_mels, int mel_w_floats, int n_waves, bool fused, int rows) {
    const size_t meta = (size_t)((n_mels + 15) / 16) * 16 * 16;            // (round, lane quad) entries of 16 bytes
    const size_t mp = fused ? (size_t)rows * ((((size_t)n_frames * n_mels + 3) & ~(size_t)3) * 4 + 16 * 8 + 16 * 4) : 0;     // per row: mel power + reduction scratch
    return (size_t)kPairWinBytes + kPairTw2Bytes + meta + (size_t)mel_w_floats * 4 + (size_t)n_waves * kPairScratch + 16 + mp;      // (16: the pair counter)
}
size_t stft_pair_tw1_lds_bytes() { return kPairTw1Bytes; }
int stft_pair_table_bytes() { return kPairWinBytes + kPairTw2Bytes; }
// the dB stage on register-resident mel power (db_row_from_regs): at most two slots of four frames per wave, eight mel rounds
bool stft_pair2_regdb_ok(int n_frames, int n_mels, int n_waves) {
    const int n_slots = (((n_frames + 1) >> 1) + 1) >> 1;
    return n_waves == 12 && n_slots <= 2 * n_waves && n_mels <= 128 && (n_mels % 4) == 0 && n_frames > 0;
}
size_t stft_pair2_lds_bytes(int n_frames, int n_mels, int mel_w_floats, int n_waves, bool fused, bool regdb) {
    const size_t meta = ((size_t)n_mels * 12 + 15) & ~(size_t)15;
    const bool rdb = fused && regdb && stft_pair2_regdb_ok(n_frames, n_mels, n_waves);
    const size_t mp = rdb ? 256 : fused ? (((size_t)n_frames * n_mels + 3) & ~(size_t)3) * 4 + 16 * 8 + 16 * 4 : 0;
    return (size_t)kP2WinBytes + meta + (size_t)mel_w_floats * 4 + (size_t)n_waves * kP2Scratch + mp;
}
int stft_pair2_table_bytes() { return kP2WinBytes; }
size_t stft_pair2_tw1_lds_bytes() { return kP2Tw1Bytes; }
bool stft_pair2_ok(int n_fft, int hop, int p_tail) { return n_fft == 1024 && hop == 256 && 513 + p_tail <= kP2PStride; }
// the configurations the pair kernel covers (the caller also needs PCM16 rows padded by K0: src.lead == n_fft / 2)
bool stft_pair_ok(int n_fft, int hop, int p_tail) { return n_fft == 2048 && hop == 384 && 1025 + p_tail <= kPairPStride; }
bool stft_pair_can_fuse_k0(int n_in, int n_waves) { return n_in > 0 && n_waves == 12; }
// ... k_stft_pair2's: the 12-wave form with the dB stage on register-resident mel power
bool stft_pair2_can_fuse_k0(int n_in, int n_frames, int n_mels, int n_waves, bool fused_db, bool regdb) {
    return n_in > 0 && fused_db && regdb && stft_pair2_regdb_ok(n_frames, n_mels, n_waves == 8 ? 8 : 12);
}      // (any row length: the fold walks the row's chunk sums)

static int table_bytes_for(int n_fft) { return n_fft == 2048 ? FftGeo<4>::TABLE_BYTES : FftGeo<2>::TABLE_BYTES; }


size_t stft_mel_lds_bytes(int n_fft, int hop, int frames_per_block, int n_mels, int mel_w_floats, int n_waves) {
    // (sized for the tile-synchronous kernel, which also needs the span: either kernel may be launched)
    const size_t span = (((size_t)(frames_per_block - 1) * hop + n_fft) * 4 + 15) & ~(size_t)15;
    const size_t meta = ((size_t)n_mels * 12 + 15) & ~(size_t)15;
    return (size_t)table_bytes_for(n_fft) + meta + (size_t)mel_w_floats * 4 + span + (size_t)n_waves * kWaveScratch;
}
size_t stft_mel_fused_lds_bytes(int n_fft, int n_frames, int n_mels, int mel_w_floats, int n_waves) {
    const size_t meta = ((size_t)n_mels * 12 + 15) & ~(size_t)15;
    const size_t mp = (((size_t)n_frames * n_mels + 3) & ~(size_t)3) * 4 + 16 * 8 + 16 * 4;      // the row's mel power + reduction scratch
    return (size_t)table_bytes_for(n_fft) + meta + (size_t)mel_w_floats * 4 + (size_t)n_waves * kWaveScratch + mp;
}
bool stft_mel_can_fuse_db(int n_fft, int n_frames, int n_mels, int mel_w_floats, int n_waves) {
    const int slots = n_fft == 2048 ? 1 : 2;
    return n_frames >= n_waves * slots && stft_mel_fused_lds_bytes(n_fft, n_frames, n_mels, mel_w_floats, n_waves) <= 160 * 1024;
}
int stft_mel_table_bytes(int n_fft) { return table_bytes_for(n_fft); }
int stft_mel_scratch_floats(int n_fft) { return (n_fft == 2048 ? FftGeo<4>::SLOT_BYTES : FftGeo<2>::SLOT_BYTES) / 4; }

hipError_t stft_mel_configure(int lds_bytes) {
    // hipFuncAttributeMaxDynamicSharedMemorySize is a property of the kernel, i.e. PROCESS-wide, and a process routinely holds
    // contexts with different LDS images (n_fft 2048 and 1024 presets side by side): only ever raise it
    static int s_lds_max = 0;
    if (lds_bytes <= s_lds_max) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rms_scale<float, kRmsThreads, NpPass<float>::kChunks>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)rms_scale_lds<float>());
    if (e != hipSuccess) return e;
    e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_rms_scale<int16_t, kRmsThreads, NpPass<int16_t>::kChunks>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, (int)rms_scale_lds<int16_t>());
    if (e != hipSuccess) return e;
    const void* fns[16] = {reinterpret_cast<const void*>(&k_stft_mel_w<true, 4, 4>), reinterpret_cast<const void*>(&k_stft_mel_w<false, 4, 4>),
                          reinterpret_cast<const void*>(&k_stft_mel_w<true, 2, 4>), reinterpret_cast<const void*>(&k_stft_mel_w<false, 2, 4>),
                          reinterpret_cast<const void*>(&k_stft_mel_w<true, 4, 12>), reinterpret_cast<const void*>(&k_stft_mel_w<false, 4, 12>),
                          reinterpret_cast<const void*>(&k_stft_mel_w<true, 2, 12>), reinterpret_cast<const void*>(&k_stft_mel_w<false, 2, 12>),
                          reinterpret_cast<const void*>(&k_stft_mel<true, 4, 4>), reinterpret_cast<const void*>(&k_stft_mel<false, 4, 4>),
                          reinterpret_cast<const void*>(&k_stft_mel<true, 2, 4>), reinterpret_cast<const void*>(&k_stft_mel<false, 2, 4>),
                          reinterpret_cast<const void*>(&k_stft_mel<true, 4, 12>), reinterpret_cast<const void*>(&k_stft_mel<false, 4, 12>),
                          reinterpret_cast<const void*>(&k_stft_mel<true, 2, 12>), reinterpret_cast<const void*>(&k_stft_mel<false, 2, 12>)};
    for (const void* f : fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    const void* pfns[16] = {reinterpret_cast<const void*>(&k_stft_pair<8, 3, false>), reinterpret_cast<const void*>(&k_stft_pair<10, 3, false>),
                            reinterpret_cast<const void*>(&k_stft_pair<11, 3, false>), reinterpret_cast<const void*>(&k_stft_pair<12, 3, false>),
                            reinterpret_cast<const void*>(&k_stft_pair<8, 3, true>), reinterpret_cast<const void*>(&k_stft_pair<10, 3, true>),
                            reinterpret_cast<const void*>(&k_stft_pair<11, 3, true>), reinterpret_cast<const void*>(&k_stft_pair<12, 3, true>),
                            reinterpret_cast<const void*>(&k_stft_pair<12, 3, false, 1>), reinterpret_cast<const void*>(&k_stft_pair<12, 3, true, 1>),
                            reinterpret_cast<const void*>(&k_stft_pair<12, 3, false, 1, true>),
                            reinterpret_cast<const void*>(&k_stft_pair<12, 3, false, 2>), reinterpret_cast<const void*>(&k_stft_pair<12, 3, true, 2>),
                            reinterpret_cast<const void*>(&k_stft_pair<12, 3, false, 2, true>),
                            reinterpret_cast<const void*>(&k_stft_pair<12, 3, false, 1, true, true>), reinterpret_cast<const void*>(&k_stft_pair<12, 3, false, 2, true, true>)};
    for (const void* f : pfns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    const void* p2fns[10] = {reinterpret_cast<const void*>(&k_stft_pair2<8, false>), reinterpret_cast<const void*>(&k_stft_pair2<8, true>),
                             reinterpret_cast<const void*>(&k_stft_pair2<12, false>), reinterpret_cast<const void*>(&k_stft_pair2<12, true>),
                             reinterpret_cast<const void*>(&k_stft_pair2<12, false, true>), reinterpret_cast<const void*>(&k_stft_pair2<12, true, true>),
                             reinterpret_cast<const void*>(&k_stft_pair2<12, false, true, 1>), reinterpret_cast<const void*>(&k_stft_pair2<12, true, true, 1>),
                             reinterpret_cast<const void*>(&k_stft_pair2<12, false, true, 2>), reinterpret_cast<const void*>(&k_stft_pair2<12, true, true, 2>)};
    for (const void* f : p2fns) {
        e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
        if (e != hipSuccess) return e;
    }
    s_lds_max = lds_bytes;
    return hipSuccess;
}


template <int NW>
static void launch_stft_mel_nw(const StftParams& p, int n_tiles, int sample_is_pcm16, hipStream_t s) {
    const size_t lds = p.db_out ? stft_mel_fused_lds_bytes(p.n_fft, p.n_frames, p.n_mels, p.mel_w_floats, NW)
                                : stft_mel_lds_bytes(p.n_fft, p.hop, p.frames_per_block, p.n_mels, p.mel_w_floats, NW);
    const int fill = NW == 4 ? 512 : 256;
    const dim3 grid((unsigned)std::min(n_tiles, p.grid_wgs > 0 ? p.grid_wgs : fill)), block(64 * NW);      // 0: fill the chip (2 per CU with 4 waves, 1 with 12)
    if (p.indep) {
        if (p.n_fft == 2048) {
            if (sample_is_pcm16) hipLaunchKernelGGL((k_stft_mel_w<true, 4, NW>), grid, block, lds, s, p, n_tiles);
            else hipLaunchKernelGGL((k_stft_mel_w<false, 4, NW>), grid, block, lds, s, p, n_tiles);
        } else {
            if (sample_is_pcm16) hipLaunchKernelGGL((k_stft_mel_w<true, 2, NW>), grid, block, lds, s, p, n_tiles);
            else hipLaunchKernelGGL((k_stft_mel_w<false, 2, NW>), grid, block, lds, s, p, n_tiles);
        }
        return;
    }
    if (p.n_fft == 2048) {
        if (sample_is_pcm16) hipLaunchKernelGGL((k_stft_mel<true, 4, NW>), grid, block, lds, s, p, n_tiles);
        else hipLaunchKernelGGL((k_stft_mel<false, 4, NW>), grid, block, lds, s, p, n_tiles);
    } else {
        if (sample_is_pcm16) hipLaunchKernelGGL((k_stft_mel<true, 2, NW>), grid, block, lds, s, p, n_tiles);
        else hipLaunchKernelGGL((k_stft_mel<false, 2, NW>), grid, block, lds, s, p, n_tiles);
    }
}

// n_waves: 4 (256-thread workgroups, two per CU) or 12 (768 threads, one per CU: three waves per SIMD hide
// more of the LDS round trips of a frame; needs frames_per_block * ... to fit 160 KB, checked by the caller)
void launch_stft_mel(const StftParams& p, int n_rows, int sample_is_pcm16, int n_waves, hipStream_t s) {
    const int n_tiles = n_rows * p.tiles_per_row;
    if (n_tiles <= 0) return;
    if (p.pair_tables && sample_is_pcm16 && p.src.lead == p.n_fft / 2 && stft_pair_ok(p.n_fft, p.hop, p.p_tail)) {
        // one persistent workgroup per row (and CU): the row's pairs of frames go round-robin over its waves
        const int nw = p.pair_waves;
        size_t lds = stft_pair_lds_bytes(p.n_frames, p.n_mels, p.pair_mel_w_floats, nw, p.db_out != nullptr);
        const bool twl = p.pair_tw_lds && lds + kPairTw1Bytes <= 160 * 1024;      // stage-1 twiddles in LDS when they fit
        if (twl) lds += kPairTw1Bytes;
        const dim3 grid((unsigned)std::min(n_rows, p.grid_wgs > 0 ? p.grid_wgs : g_num_cus)), block(64 * nw);
#define AMPH_PAIR_LAUNCH(NW_)                                                                              \
    if (twl) hipLaunchKernelGGL((k_stft_pair<NW_, 3, true>), grid, dim3(64 * NW_), lds, s, p, n_rows);   \
    else hipLaunchKernelGGL((k_stft_pair<NW_, 3, false>), grid, dim3(64 * NW_), lds, s, p, n_rows);
        (void)block;
        if (p.k0_in) {                                  // K0F (12 waves: the one-second rows of the benchmark; the caller has checked)
            // two rows per workgroup and queue (G2) when the batch has more rows than half the CUs and both rows' mel power fits LDS
            const size_t lds2 = stft_pair_lds_bytes(p.n_frames, p.n_mels, p.pair_mel_w_floats, nw, true, 2);
            const bool rms = p.k0_chunk_sums != nullptr;          // K0F 1 (fold + re-quantise) or 2 (plain decode)
            const int n_pairs = (p.n_frames + 1) / 2;
            if (p.db_out && p.pair_cq > 1 && n_rows > g_num_cus / 2 && lds2 <= 160 * 1024 && n_pairs >= 12 + 4) {
                // the continuous queue: pair_cq rows per workgroup (rows b, b + grid, ...), no barrier behind the tables'
                const dim3 gridq((unsigned)std::min((n_rows + p.pair_cq - 1) / p.pair_cq, p.grid_wgs > 0 ? p.grid_wgs : g_num_cus));
                if (rms) hipLaunchKernelGGL((k_stft_pair<12, 3, false, 1, true, true>), gridq, dim3(64 * 12), lds2, s, p, n_rows);
                else hipLaunchKernelGGL((k_stft_pair<12, 3, false, 2, true, true>), gridq, dim3(64 * 12), lds2, s, p, n_rows);
            }
            else if (p.db_out && p.pair_rows2 && n_rows > g_num_cus / 2 && lds2 <= 160 * 1024) {
                const dim3 grid2((unsigned)std::min((n_rows + 1) / 2, p.grid_wgs > 0 ? p.grid_wgs : g_num_cus));
                if (rms) hipLaunchKernelGGL((k_stft_pair<12, 3, false, 1, true>), grid2, dim3(64 * 12), lds2, s, p, n_rows);
                else hipLaunchKernelGGL((k_stft_pair<12, 3, false, 2, true>), grid2, dim3(64 * 12), lds2, s, p, n_rows);
            }
            else if (twl && rms) hipLaunchKernelGGL((k_stft_pair<12, 3, true, 1>), grid, dim3(64 * 12), lds, s, p, n_rows);
            else if (twl) hipLaunchKernelGGL((k_stft_pair<12, 3, true, 2>), grid, dim3(64 * 12), lds, s, p, n_rows);
            else if (rms) hipLaunchKernelGGL((k_stft_pair<12, 3, false, 1>), grid, dim3(64 * 12), lds, s, p, n_rows);
            else hipLaunchKernelGGL((k_stft_pair<12, 3, false, 2>), grid, dim3(64 * 12), lds, s, p, n_rows);
        }
        else if (nw == 8) { AMPH_PAIR_LAUNCH(8) }
        else if (nw == 11) { AMPH_PAIR_LAUNCH(11) }
        else if (nw == 12) { AMPH_PAIR_LAUNCH(12) }
        else { AMPH_PAIR_LAUNCH(10) }
#undef AMPH_PAIR_LAUNCH
        return;
    }
    if (p.pair_tables && sample_is_pcm16 && p.src.lead == p.n_fft / 2 && stft_pair2_ok(p.n_fft, p.hop, p.p_tail)) {
        const int nw = p.pair_waves == 8 ? 8 : 12;
        const bool rdb = p.db_out && p.pair2_regdb && stft_pair2_regdb_ok(p.n_frames, p.n_mels, nw);
        size_t lds = stft_pair2_lds_bytes(p.n_frames, p.n_mels, p.mel_w_floats, nw, p.db_out != nullptr, rdb);
        const bool twl = p.pair_tw_lds && lds + kP2Tw1Bytes <= 160 * 1024;
        if (twl) lds += kP2Tw1Bytes;
        const dim3 grid((unsigned)std::min(n_rows, p.grid_wgs > 0 ? p.grid_wgs : g_num_cus));
        if (p.k0_in) {                                  // K0F: the raw rows (the caller has checked stft_pair2_can_fuse_k0: the 12-wave RDB form)
            const bool rms = p.k0_chunk_sums != nullptr;
            if (!rdb) return;
            if (rms && twl) hipLaunchKernelGGL((k_stft_pair2<12, true, true, 1>), grid, dim3(768), lds, s, p, n_rows);
            else if (rms) hipLaunchKernelGGL((k_stft_pair2<12, false, true, 1>), grid, dim3(768), lds, s, p, n_rows);
            else if (twl) hipLaunchKernelGGL((k_stft_pair2<12, true, true, 2>), grid, dim3(768), lds, s, p, n_rows);
            else hipLaunchKernelGGL((k_stft_pair2<12, false, true, 2>), grid, dim3(768), lds, s, p, n_rows);
        } else if (rdb) {
            if (twl) hipLaunchKernelGGL((k_stft_pair2<12, true, true>), grid, dim3(768), lds, s, p, n_rows);
            else hipLaunchKernelGGL((k_stft_pair2<12, false, true>), grid, dim3(768), lds, s, p, n_rows);
        } else if (nw == 8) {
            if (twl) hipLaunchKernelGGL((k_stft_pair2<8, true>), grid, dim3(512), lds, s, p, n_rows);
            else hipLaunchKernelGGL((k_stft_pair2<8, false>), grid, dim3(512), lds, s, p, n_rows);
        } else {
            if (twl) hipLaunchKernelGGL((k_stft_pair2<12, true>), grid, dim3(768), lds, s, p, n_rows);
            else hipLaunchKernelGGL((k_stft_pair2<12, false>), grid, dim3(768), lds, s, p, n_rows);
        }
        return;
    }
    if (n_waves == 12) launch_stft_mel_nw<12>(p, n_tiles, sample_is_pcm16, s);
    else launch_stft_mel_nw<4>(p, n_tiles, sample_is_pcm16, s);
}

// ----------------------------------------------------------------------------
// K1/K2, general form (round 4): any power-of-two n_fft from 64 to 8192 and ANY hop_length - every reference CLI exposes both
// (07_encode_wav_to_latent.py:429-430, 08_fit_radial_detector.py:353-354) and librosa.feature.melspectrogram takes any value
// (map_detector_core.py:219-228); the tuned kernels above cover n_fft 1024 / 2048 with an even hop.  One 256-thread workgroup per
// (frame, row): fix_length + centre padding (reflect / zeros) + RMS scale / PCM16 re-quantisation as index arithmetic and
// normalise_sample per sample, periodic Hann window, an in-place radix-2 FFT of the frame in LDS (bit-reversed load, one barrier per
// stage, twiddles from a host table), |X|^2, the mel rows from the same tables the tuned kernels read.  Not tuned: it exists so
// that a configuration the reference accepts is not refused.
// ----------------------------------------------------------------------------
template <bool PCM16>
__global__ __launch_bounds__(256) void k_stft_generic(StftParams p, const float* __restrict__ win, const float2* __restrict__ tw, int log2n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float2* x = reinterpret_cast<float2*>(smem);           // [n_fft] complex; afterwards the power spectrum (float [n_fft/2 + 1 + tail]) in front
    const int tid = threadIdx.x, t = blockIdx.x, row = blockIdx.y;
    const int n = p.n_fft, N = p.n_samples;
    const char* rowp = reinterpret_cast<const char*>(p.src.wave) + (long long)row * p.src.row_stride * (PCM16 ? 2 : 4);
    const int len = min(p.src.lengths ? min(p.src.lengths[row], p.src.n_in) : p.src.n_in, N);      // fix_length: truncate / zero pad to N
    const float sc = p.src.scale ? p.src.scale[row] : 1.0f;
    for (int i = tid; i < n; i += 256) {
        int sidx = t * p.hop + i - (n >> 1);                // centre = True: the frame starts n_fft / 2 before t * hop
        bool pad0 = false;
        if (p.pad_mode == 0) {                              // np.pad(mode="reflect")
            if (sidx < 0) sidx = -sidx;
            if (sidx >= N) sidx = 2 * (N - 1) - sidx;
        } else pad0 = sidx < 0 || sidx >= N;
        float v = pad0 ? 0.0f : load_sample<PCM16>(rowp, sidx, len);
        v = normalise_sample(p.src, v, sc) * win[i];
        x[__brev((unsigned)i) >> (32 - log2n)] = make_float2(v, 0.0f);
    }
    __syncthreads();
    for (int s = 1; s <= log2n; ++s) {
        const int half = 1 << (s - 1);
        for (int j = tid; j < (n >> 1); j += 256) {
            const int k = j & (half - 1);
            const int i0 = ((j >> (s - 1)) << s) + k, i1 = i0 + half;
            const float2 w = tw[k << (log2n - s)];          // W_n^(k n / 2^s)
            const float2 a = x[i0], b = x[i1];
            const float tr = b.x * w.x - b.y * w.y, ti = b.x * w.y + b.y * w.x;
            x[i0] = make_float2(a.x + tr, a.y + ti);
            x[i1] = make_float2(a.x - tr, a.y - ti);
        }
        __syncthreads();
    }
    const int F = (n >> 1) + 1;
    float pw[17];                                           // (n_fft <= 8192: at most 17 bins per thread)
    int cnt = 0;
    for (int k = tid; k < F; k += 256) { const float2 z = x[k]; pw[cnt++] = z.x * z.x + z.y * z.y; }
    __syncthreads();
    float* P = reinterpret_cast<float*>(smem);
    cnt = 0;
    for (int k = tid; k < F; k += 256) P[k] = pw[cnt++];
    for (int k = F + tid; k < F + p.p_tail; k += 256) P[k] = 0.0f;       // the padded mel rows read there
    __syncthreads();
    for (int m = tid; m < p.n_mels; m += 256) {
        const float* w = p.mel_w + p.mel_wofs[m];
        const float* pp = P + p.mel_start[m];
        const int ne = 16 * p.mel_len[m];
        float a0 = 0.0f, a1 = 0.0f;
        for (int e = 0; e + 1 < ne; e += 2) { a0 = fmaf(w[e], pp[e], a0); a1 = fmaf(w[e + 1], pp[e + 1], a1); }
        p.melpow[((long long)row * p.n_frames + t) * p.n_mels + m] = a0 + a1;
    }
}
void launch_stft_generic(const StftParams& p, int n_rows, int sample_is_pcm16, const float* win, const float2* tw, hipStream_t s) {
    if (n_rows <= 0 || p.n_frames <= 0) return;
    int log2n = 0;
    while ((1 << log2n) < p.n_fft) ++log2n;
    const size_t lds = std::max((size_t)p.n_fft * 8, (size_t)(p.n_fft / 2 + 1 + p.p_tail) * 4);
    const dim3 grid((unsigned)p.n_frames, (unsigned)n_rows);
    if (sample_is_pcm16) hipLaunchKernelGGL((k_stft_generic<true>), grid, dim3(256), lds, s, p, win, tw, log2n);
    else hipLaunchKernelGGL((k_stft_generic<false>), grid, dim3(256), lds, s, p, win, tw, log2n);
}

// ----------------------------------------------------------------------------
// K3: power_to_db(ref=max, amin=1e-10, top_db=80) -> z-score over the whole
// [M, T_raw] array (population std) -> centre crop / zero pad -> [T, M] or [M, T].
// One 256-thread block per chunk; dB values are recomputed per pass.
// ----------------------------------------------------------------------------
// 10 log10(x) as librosa.power_to_db evaluates it (map_detector_core.py:229), on the hardware's base-2 logarithm (v_log_f32, 1 ulp) times
// 10 log10(2): two roundings like numpy's log10 then x 10, ~2e-5 dB absolute at 100 dB (the library log10f: 6e-6) - 2e-6 on the z-scored
// image, far inside its 1e-4 bar - for 2 instead of ~14 vector instructions per element; the dB stage is bound by its arithmetic
// (profiles/r04_stft_g2_timeline.txt).  ref_db is computed by the SAME function: the row's maximum gives exactly 0.
__device__ __forceinline__ float db10(float x) {
#pragma clang fp contract(off)
    return 3.0102999566398120f * __log2f(x);
}
__device__ __forceinline__ float to_db(float s, float ref_db) {
#pragma clang fp contract(off)
    // separately rounded multiply and subtract (numpy order): the max element gives exactly 0
    const float db = db10(fmaxf(1e-10f, s));
    return fmaxf(db - ref_db, -80.0f);
}

// kCache > 0: every thread keeps its kCache dB values in registers (one log10f per element);
// kCache == 0: values are recomputed per pass (any size).
// NT threads per block: 256, or 1024 so that a 5 s chunk's 626 x 64 values (40 per thread) stay in registers too
// (recomputed per pass they cost three log10 sweeps of an L2-resident 160 KB array: 149 -> 40 us per 256 chunks).
template <int kCache, int NT>
__global__ __launch_bounds__(NT) void k_dbnorm(DbNormParams p) {
    constexpr int NW = NT / 64;
    __shared__ double shd[16];
    __shared__ float shf[16];
    const int row = blockIdx.x, tid = threadIdx.x;
    const int n = p.n_frames * p.n_mels;
    const float* S = p.melpow + (long long)row * n;
    constexpr int C = kCache > 0 ? kCache : 1;
    float v[C];
    float mx = -INFINITY;
    if (kCache > 0) {
#pragma unroll
        for (int j = 0; j < C; ++j) { const int i = tid + NT * j; v[j] = i < n ? S[i] : -INFINITY; mx = fmaxf(mx, v[j]); }
    } else {
        for (int i = tid; i < n; i += NT) mx = fmaxf(mx, S[i]);
    }
    mx = block_max<NW>(mx, shf);
    const float ref_db = db10(fmaxf(1e-10f, mx));             // same two operations as to_db (no FMA)
    double acc = 0.0;
    if (kCache > 0) {
#pragma unroll
        for (int j = 0; j < C; ++j) if (tid + NT * j < n) { v[j] = to_db(v[j], ref_db); acc += (double)v[j]; }
    } else {
        for (int i = tid; i < n; i += NT) acc += (double)to_db(S[i], ref_db);
    }
    const float mean = (float)(block_sum<NW>(acc, shd) / (double)n);
    acc = 0.0;
    if (kCache > 0) {
#pragma unroll
        for (int j = 0; j < C; ++j) if (tid + NT * j < n) { const float d = v[j] - mean; acc += (double)(d * d); }
    } else {
        for (int i = tid; i < n; i += NT) { const float d = to_db(S[i], ref_db) - mean; acc += (double)(d * d); }
    }
    const float sd = sqrtf((float)(block_sum<NW>(acc, shd) / (double)n));
    const float rden = 1.0f / (sd + 1e-8f);              // (x / den as x * (1 / den): as db_row_from_lds, bit for bit)
    // crop_or_pad_time (map_detector_core.py:185-195): destination frame = source frame - shift
    const int T = p.n_frames, TT = p.target_frames, M = p.n_mels;
    const int shift = (T >= TT) ? (T - TT) / 2 : -((TT - T) / 2);
    float* out = p.out + (long long)row * TT * M;
    if (kCache > 0) {
#pragma unroll
        for (int j = 0; j < C; ++j) {
            const int i = tid + NT * j;
            if (i < n) {
                const int fr = i / M, m = i - fr * M, tt = fr - shift;
                if (tt >= 0 && tt < TT) {
                    const float z = (v[j] - mean) * rden;
                    if (p.layout == 1) out[tt * M + m] = z;
                    else out[m * TT + tt] = z;
                }
            }
        }
        if (T < TT) {                                   // literal zero padding, left then right
            const int left = -shift, right0 = left + T;
            for (int i = tid; i < (TT - T) * M; i += NT) {
                const int fr = i / M, m = i - fr * M;
                const int tt = fr < left ? fr : right0 + (fr - left);
                if (p.layout == 1) out[tt * M + m] = 0.0f;
                else out[m * TT + tt] = 0.0f;
            }
        }
    } else {
        for (int i = tid; i < TT * M; i += NT) {
            const int tt = i / M, m = i - tt * M;
            const int src = tt + shift;
            float z = 0.0f;
            if (src >= 0 && src < T) z = (to_db(S[src * M + m], ref_db) - mean) * rden;
            if (p.layout == 1) out[i] = z;
            else out[m * TT + tt] = z;
        }
    }
}

void launch_dbnorm(const DbNormParams& p, int n_rows, hipStream_t s) {
    const int n = p.n_frames * p.n_mels;
    if (n <= 1024 * 4) hipLaunchKernelGGL((k_dbnorm<4, 1024>), dim3(n_rows), dim3(1024), 0, s, p);         // 1 s chunks, 64 mels
    else if (n <= 256 * 16) hipLaunchKernelGGL((k_dbnorm<16, 256>), dim3(n_rows), dim3(256), 0, s, p);
    else if (n <= 1024 * 12) hipLaunchKernelGGL((k_dbnorm<12, 1024>), dim3(n_rows), dim3(1024), 0, s, p);  // e.g. 87 frames x 128 mels
    else if (n <= 1024 * 24) hipLaunchKernelGGL((k_dbnorm<24, 1024>), dim3(n_rows), dim3(1024), 0, s, p);  // 3 s chunks (376 x 64)
    else if (n <= 1024 * 40) hipLaunchKernelGGL((k_dbnorm<40, 1024>), dim3(n_rows), dim3(1024), 0, s, p);  // 5 s chunks (626 x 64)
    else hipLaunchKernelGGL((k_dbnorm<0, 256>), dim3(n_rows), dim3(256), 0, s, p);                          // anything longer: recompute per pass
}

#ifdef AMPH_ABL_SKIP
int g_abl_skip_k0 = 0;       // timing-only diagnostic build: bit 0 / 1 leave out K0's first / second kernel (amph_abl_skip bits 7 / 8)
#endif
int rms_scale_chunks(int n_in) { return (n_in + kNpChunk - 1) / kNpChunk; }
// Rows of more numpy chunks (8 192 samples) than this take the (row, chunk) + (row, tile) kernels.  A one-pass row fits the single kernel
// up to 4 chunks, and alone on the GPU that form is faster (15 vs 20 us per 256 one-second rows), but its 1024-thread, 70 KB workgroups wait
// for a free CU behind the other lanes' persistent kernels (49 us in the four-lane trace), the 256-thread ones of the two-kernel form
// fit beside them: +0.8 % end to end at 3 chunks per row (profiles/r02_rms_chunk_path_sweep.txt)
// (the threshold is a per-context setting, "rms_chunk_path_over", default 2)
void launch_rms_scale_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows,
                            float* scale, int32_t* ok, int scale_f64, int16_t* norm_out, const NormPad& pad, float* chunk_sums,
                            int chunk_path_over, hipStream_t s) {
    const int n_chunks = rms_scale_chunks(n_in);
    if (chunk_sums && n_chunks > chunk_path_over) {
        // rows longer than one LDS pass: (row, chunk) workgroups sum the chunks, (row, tile) workgroups fold + normalise + pad
#ifdef AMPH_ABL_SKIP
        if (!(g_abl_skip_k0 & 1))
#endif
        hipLaunchKernelGGL(k_rms_chunk_sums_q, dim3(n_rows, n_chunks), dim3(256), 0, s, in, row_stride, lengths, n_in, chunk_sums, n_chunks);   // (round 4: leaves on lane quads)
        const int total_out = (pad.lead > 0 ? pad.n_fix : n_in) + 2 * pad.lead;
        const int tiles = norm_out ? std::max(1, std::min(64, total_out / (2 * 256 * 8))) : 1;       // ~8 dwords per thread (16: 21.1 us for K0 on 256 one-second rows, 8: 19.3, 4: 19.9)
#ifdef AMPH_ABL_SKIP
        if (!(g_abl_skip_k0 & 2))
#endif
        hipLaunchKernelGGL(k_rms_fold_apply, dim3(n_rows, tiles), dim3(256), 0, s, in, row_stride, lengths, n_in, chunk_sums, n_chunks, scale_f64,
                           scale, ok, norm_out, pad);
        return;
    }
    hipLaunchKernelGGL((k_rms_scale<int16_t, kRmsThreads, NpPass<int16_t>::kChunks>), dim3(n_rows), dim3(kRmsThreads), rms_scale_lds<int16_t>(), s, in, row_stride,
                       lengths, n_in, scale, ok, norm_out, pad, scale_f64, static_cast<float*>(nullptr), 0);
}
// K0's first kernel alone: the per-chunk sums of squares of every row (k_stft_pair<.., 2> folds them and normalises in its decode)
void launch_rms_chunk_sums_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows, float* chunk_sums, hipStream_t s) {
    const int n_chunks = rms_scale_chunks(n_in);
    hipLaunchKernelGGL(k_rms_chunk_sums_q, dim3(n_rows, n_chunks), dim3(256), 0, s, in, row_stride, lengths, n_in, chunk_sums, n_chunks);
}
// Rows that are NOT RMS-normalised here (apply_rms == 0: files 00_normalize_dataset_rms.py already processed), PCM16: the
// same fix_length'd, centre-padded copy K0 writes, so that the independent-wave STFT kernel has no edge frames in this mode
// either.  A few workgroups per row; three branch-free loops (padding, the row as dwords, zeros + padding).
__global__ __launch_bounds__(256) void k_pad_rows_pcm16(const int16_t* __restrict__ in, long long row_stride,
                                                        const int32_t* __restrict__ lengths, int n_in, int16_t* __restrict__ out, NormPad pad) {
    const int row = blockIdx.x, tid = blockIdx.y * 256 + threadIdx.x, nt = 256 * gridDim.y;   // gridDim.y workgroups share a row
    const int16_t* x = in + (long long)row * row_stride;
    int16_t* o = out + (long long)row * pad.out_stride;
    const int n = lengths ? min(lengths[row], n_in) : n_in;
    const int nv = min(n, pad.n_fix), total = pad.n_fix + 2 * pad.lead;
    auto sample = [&](int idx) {
        int sidx = idx - pad.lead;
        if (pad.mode == 0) {
            if (sidx < 0) sidx = -sidx;
            if (sidx >= pad.n_fix) sidx = 2 * (pad.n_fix - 1) - sidx;
        }
        return (sidx < 0 || sidx >= nv) ? 0 : (int)x[sidx];
    };
    if (((reinterpret_cast<uintptr_t>(o) | reinterpret_cast<uintptr_t>(x)) & 3u) == 0 && (pad.lead & 1) == 0) {
        uint32_t* o2 = reinterpret_cast<uint32_t*>(o);
        const uint32_t* x2 = reinterpret_cast<const uint32_t*>(x);
        const int i_lo = pad.lead >> 1, i_hi = i_lo + (nv >> 1);
        for (int i = tid; i < i_lo; i += nt) o2[i] = ((uint32_t)sample(2 * i) & 0xFFFFu) | ((uint32_t)sample(2 * i + 1) << 16);
        constexpr int UW = 8;
        for (int base = i_lo; base < i_hi; base += nt * UW) {
            uint32_t v[UW];
#pragma unroll
            for (int u = 0; u < UW; ++u) { const int i = base + tid + nt * u; v[u] = i < i_hi ? x2[i - i_lo] : 0u; }
#pragma unroll
            for (int u = 0; u < UW; ++u) { const int i = base + tid + nt * u; if (i < i_hi) o2[i] = v[u]; }
        }
        for (int i = i_hi + tid; i < (total >> 1); i += nt) o2[i] = ((uint32_t)sample(2 * i) & 0xFFFFu) | ((uint32_t)sample(2 * i + 1) << 16);
        if ((total & 1) && tid == 0) o[total - 1] = (int16_t)sample(total - 1);
    } else {
        for (int i = tid; i < total; i += nt) o[i] = (int16_t)sample(i);
    }
}

void launch_pad_rows_pcm16(const int16_t* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows, int16_t* out,
                           const NormPad& pad, hipStream_t s) {
    const int total = pad.n_fix + 2 * pad.lead;
    const int by = std::max(1, std::min(16, total / (2 * 256 * 8)));          // ~8 dwords per thread
    hipLaunchKernelGGL(k_pad_rows_pcm16, dim3(n_rows, by), dim3(256), 0, s, in, row_stride, lengths, n_in, out, pad);
}
void launch_rms_scale_f32(const float* in, long long row_stride, const int32_t* lengths, int n_in, int n_rows,
                          float* scale, int32_t* ok, int scale_f64, hipStream_t s) {
    hipLaunchKernelGGL((k_rms_scale<float, kRmsThreads, NpPass<float>::kChunks>), dim3(n_rows), dim3(kRmsThreads), rms_scale_lds<float>(), s, in, row_stride, lengths,
                       n_in, scale, ok, static_cast<int16_t*>(nullptr), NormPad{}, scale_f64, static_cast<float*>(nullptr), 0);
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

// ----------------------------------------------------------------------------
// K8: sample-rate conversion of librosa.load (map_detector_core.py:210) = resampy 0.4.3's resample_f with
// the kaiser_best windowed-sinc table (restated in oracle/resample.py; parity unpinned, see there).
// One thread per output sample: left wing then right wing of the table with linear interpolation between
// entries; products in float64, the running sum rounded to float32 after every tap exactly as resampy's
// `y[t] += weight * x[n - i]` on a float32 output array.  Not on the benchmark path (synthetic chunks are
// written at the preset rate); ~280 taps per output sample at 48 kHz -> 22.05 kHz.
// ----------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_resample(ResampleParams p) {
#pragma clang fp contract(off)
    const int t = blockIdx.x * 256 + threadIdx.x;
    const int row = blockIdx.y;
    if (t >= p.n_fix) return;
    float* y = p.y + (long long)row * p.out_stride;
    if (t >= p.n_out) { y[t] = 0.0f; return; }             // librosa.util.fix_length pads with zeros
    const float* x = p.x + (long long)row * p.in_stride;
    const double tr = (double)t * p.time_inc;
    const int n = (int)tr;
    double frac = p.scale * (tr - (double)n);
    double index_frac = frac * (double)p.num_table;
    int offset = (int)index_frac;
    double eta = index_frac - (double)offset;
    float acc = 0.0f;
    const int i_max = min(n + 1, (p.nwin - offset) / p.index_step);
    for (int i = 0; i < i_max; ++i) {
        const int idx = offset + i * p.index_step;
        const double w = p.win[idx] + eta * p.delta[idx];
        acc = (float)((double)acc + w * (double)x[n - i]);
    }
    frac = p.scale - frac;
    index_frac = frac * (double)p.num_table;
    offset = (int)index_frac;
    eta = index_frac - (double)offset;
    const int k_max = min(p.n_in - n - 1, (p.nwin - offset) / p.index_step);
    for (int k = 0; k < k_max; ++k) {
        const int idx = offset + k * p.index_step;
        const double w = p.win[idx] + eta * p.delta[idx];
        acc = (float)((double)acc + w * (double)x[n + k + 1]);
    }
    y[t] = acc;
}

void launch_resample(const ResampleParams& p, int n_rows, hipStream_t s) {
    if (n_rows <= 0 || p.n_fix <= 0) return;
    hipLaunchKernelGGL(k_resample, dim3((unsigned)((p.n_fix + 255) / 256), (unsigned)n_rows), dim3(256), 0, s, p);
}

}  // namespace amph
