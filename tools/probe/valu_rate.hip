// What a SIMD issues per cycle with 1, 2, 3, 4 waves on it: independent v_fma_f32 streams, v_pk_fma_f32 streams, and a mix with
// LDS reads / SALU.  hipcc --offload-arch=gfx950 -O3 -o valu_rate tools/probe/valu_rate.hip && ./valu_rate
// Output: cycles (s_memtime) per instruction and SIMD for each (kind, waves per SIMD).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

template <int KIND>
__global__ void k(float* out, unsigned long long* cyc, int iters) {
    __shared__ float sh[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) sh[i] = (float)i;
    __syncthreads();
    float a[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) a[j] = threadIdx.x * 0.001f + j;
    const float b = 1.0001f, c = 0.5f;
    const int lane = threadIdx.x & 63;
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {          // 64 independent-ish v_fma_f32 (16 chains)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
        } else if (KIND == 1) {   // 32 v_pk_fma_f32 (8 chains on register pairs)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; j += 2) {
                    typedef float f2 __attribute__((ext_vector_type(2)));
                    f2 v = {a[j], a[j + 1]};
                    f2 bb = {b, b}, cc = {c, c};
                    asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(v) : "v"(bb), "v"(cc));
                    a[j] = v.x; a[j + 1] = v.y;
                }
        } else if (KIND == 2) {   // 48 v_fma + 16 ds_read_b32 (LDS beside VALU)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float l[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) l[j] = sh[(lane + 64 * (4 * r + j) + it) & 4095];
#pragma unroll
                for (int j = 0; j < 12; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
#pragma unroll
                for (int j = 0; j < 4; ++j) a[12 + j] += l[j];
            }
        } else if (KIND == 3) {   // 64 v_fma + 32 s_add (SALU beside VALU)
            int sacc = it;
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[j]) : "v"(b), "v"(c));
                    if (j & 1) asm volatile("s_add_i32 %0, %0, 1" : "+s"(sacc));
                }
            if (sacc == 12345678) a[0] += 1.0f;
        } else if (KIND == 4) {   // 64 v_add_f32 via DPP (quad_perm) - DPP issue cost
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int j = 0; j < 16; ++j) asm volatile("v_add_f32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[j]));
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.0f;
#pragma unroll
    for (int j = 0; j < 16; ++j) s += a[j];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* out; unsigned long long* cyc;
    hipMalloc(&out, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    const char* names[5] = {"v_fma_f32 x64", "v_pk_fma_f32 x32", "48 v_fma + 16 ds_read_b32 + 16 v_add", "64 v_fma + 32 s_add", "v_add_f32_dpp x64"};
    const int per_iter[5] = {64, 32, 80, 64, 64};
    for (int kind = 0; kind < 5; ++kind)
        for (int wps = 1; wps <= 4; ++wps) {
            const int threads = 256 * wps;
            for (int rep = 0; rep < 2; ++rep) {
                if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (kind == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (kind == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (kind == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                if (kind == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, cyc, iters);
                hipDeviceSynchronize();
            }
            std::vector<unsigned long long> h(256);
            hipMemcpy(h.data(), cyc, 256 * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double c = (double)h[128];
            printf("%-40s waves/SIMD %d: %.2f cycles per VALU-class instruction and SIMD (wave lifetime %.0f cycles)\n", names[kind], wps,
                   c / ((double)iters * per_iter[kind] * wps), c);
        }
    return 0;
}
