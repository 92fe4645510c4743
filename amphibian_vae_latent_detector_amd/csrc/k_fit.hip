// K9: the radial detector's fit on latents that are already in HBM.
//
// Replaces the array work of fit_species_with_fp_control, latent_space_exploration/08_fit_radial_detector.py:310-333:
//   mu     = np.mean(Z_in, axis=0)                     k_fit_centroids   (numpy's order for an axis-0 reduction of a C-contiguous
//                                                                         float32 matrix: row after row, sequentially, then / N)
//   rho    = sqrt(sum((Z - mu)^2, axis=1))             k_fit_radii       (numpy's pairwise order inside a row: np_pairwise_sumsq)
//   np.quantile(rho_in, q_in), np.quantile(rho_out, q_out)
//                                                      k_fit_select      (the two order statistics each quantile interpolates
//                                                                         between, by a 4-pass radix select: exact, order-free)
// The interpolation itself (two scalars per quantile, in numpy 1.26's or numpy 2's arithmetic) is done by the caller.
// Result: the same centroids and the same thresholds, to the last bit, as the host fit on the same latents (tests).
#include "amph_internal.h"

namespace amph {

// one workgroup per species, one thread per latent dimension: the rows are added in order, in float32
__global__ __launch_bounds__(128) void k_fit_centroids(FitParams p) {
#pragma clang fp contract(off)
    const int k = blockIdx.x, j = threadIdx.x;
    if (j >= p.D) return;
    const long long lo = p.offsets[k], hi = p.offsets[k + 1];
    const float* z = p.z + lo * p.D + j;
    float acc = 0.0f;
    long long i = 0;
    const long long n = hi - lo;
    for (; i + 8 <= n; i += 8) {                         // eight loads in flight, added in row order
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = z[(i + u) * p.D];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc = acc + v[u];
    }
    for (; i < n; ++i) acc = acc + z[i * p.D];
    p.mu[(long long)k * p.D + j] = acc / (float)n;       // np.mean: the float32 sum divided by the count
}

__global__ __launch_bounds__(256) void k_fit_radii(FitParams p) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= p.n) return;
    const float* z = p.z + i * p.D;
    for (int k = 0; k < p.K; ++k)
        p.rho[(long long)k * p.n + i] = sqrtf(np_pairwise_sumsq(z, p.mu + (long long)k * p.D, p.D));
}

// workgroup (k, s): order statistic ranks[k][s] of the in-class (s < 8) or out-class (s >= 8) distances to centroid k
// (per class: the two of the threshold quantile, then min, the two of p50, the two of p90, max - summarize_dist, 08:115-123).
// Distances are >= 0, so their float32 bit patterns order like the values: radix select over the 32 bits, 8 per pass.
__global__ __launch_bounds__(1024) void k_fit_select(FitParams p) {
    __shared__ unsigned hist[256];
    __shared__ unsigned long long sh_rank;
    __shared__ unsigned sh_prefix;
    const int k = blockIdx.x, s = blockIdx.y, tid = threadIdx.x;
    const long long want = p.ranks[k * 16 + s];
    if (want < 0) { if (tid == 0) p.stats[k * 16 + s] = 0.0f; return; }
    const long long lo = p.offsets[k], hi = p.offsets[k + 1];
    const unsigned* rho = reinterpret_cast<const unsigned*>(p.rho + (long long)k * p.n);
    const bool in_class = s < 8;
    if (tid == 0) { sh_rank = (unsigned long long)want; sh_prefix = 0u; }
    unsigned mask = 0u;
    for (int shift = 24; shift >= 0; shift -= 8) {
        for (int b = tid; b < 256; b += 1024) hist[b] = 0u;
        __syncthreads();
        const unsigned prefix = sh_prefix;
        for (long long i = tid; i < p.n; i += 1024) {
            const bool mine = (i >= lo && i < hi) == in_class;
            if (mine) {
                const unsigned key = rho[i];
                if ((key & mask) == prefix) atomicAdd(&hist[(key >> shift) & 255u], 1u);
            }
        }
        __syncthreads();
        if (tid == 0) {
            unsigned long long r = sh_rank, cum = 0;
            int b = 0;
            for (; b < 256; ++b) { if (cum + hist[b] > r) break; cum += hist[b]; }
            sh_rank = r - cum;
            sh_prefix = prefix | ((unsigned)b << shift);
        }
        mask |= 255u << shift;
        __syncthreads();
    }
    if (tid == 0) p.stats[k * 16 + s] = __uint_as_float(sh_prefix);
}

void launch_fit_radial(const FitParams& p, hipStream_t s) {
    hipLaunchKernelGGL(k_fit_centroids, dim3(p.K), dim3(128), 0, s, p);
    hipLaunchKernelGGL(k_fit_radii, dim3((unsigned)((p.n + 255) / 256)), dim3(256), 0, s, p);
    hipLaunchKernelGGL(k_fit_select, dim3(p.K, 16), dim3(1024), 0, s, p);
}

}  // namespace amph
