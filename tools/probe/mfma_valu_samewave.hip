// Probe: do plain VALU instructions hide in the shadow of f32 MFMAs?  One or two waves per SIMD, each loops over
// 8 x v_mfma_f32_32x32x2_f32 with PER VALU instructions of one kind after every MFMA; cycles per loop trip by s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_valu_samewave.hip -o build_ab/mfma_valu_samewave && build_ab/mfma_valu_samewave
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// KIND 5: ds_read_b128, 6: ds_write_b128, 7: global_load_dwordx4 (waits at the end of a trip)
// KIND 0: v_fma_f32 (independent registers), 1: v_pk_fma_f32, 2: v_cndmask/v_lshl int ops, 3: s_add_u32, 4: v_fma_f32 with NO MFMAs at all
template <int PER, int KIND>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    __shared__ float sh[256 * 4 * 2];
    const int tid = threadIdx.x;
    const int lds_off = tid * 16;
    const f32x4* gp = reinterpret_cast<const f32x4*>(in) + tid + (blockIdx.x & 63) * 256;
    const unsigned goff = (tid + (blockIdx.x & 63) * 256) * 16;
    f32x4 g[8];
    for (int j = 0; j < 8; ++j) g[j] = f32x4{0, 0, 0, 0};
    sh[tid] = 0;
    __syncthreads();
    float a = in[tid], b = in[tid + 256];
    f32x16 c0 = {0};
    float v[8];
    f32x2 pk[8];
    int iv[8];
    for (int j = 0; j < 8; ++j) { v[j] = in[tid + j]; pk[j] = f32x2{in[tid + j], in[tid + j + 1]}; iv[j] = tid + j; }
    int sacc = iters;
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND != 4 && KIND < 10) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int r = (j * PER + q) & 7;
                if (KIND == 0 || KIND == 4) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(a));
                else if (KIND == 1 || KIND == 10) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(pk[r]) : "v"(pk[(r + 1) & 7]));
                else if (KIND == 11) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(pk[r]) : "v"(pk[(r + 1) & 7]));
                else if (KIND == 12) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(pk[r]) : "v"(pk[(r + 1) & 7]));
                else if (KIND == 2) asm volatile("v_lshl_add_u32 %0, %0, 1, %1" : "+v"(iv[r]) : "v"(tid));
                else if (KIND == 3) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
                else if (KIND == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(g[r]) : "v"(lds_off) : "memory");
                else if (KIND == 6) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_off), "v"(g[r]) : "memory");
                else if (KIND == 7) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(g[r]) : "v"(gp) : "memory");
                else if (KIND == 8) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(g[r]) : "v"(goff), "s"(in) : "memory");
                else if (KIND == 9) asm volatile("global_load_dword %0, %1, %2" : "=v"(v[r]) : "v"(goff), "s"(in) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND >= 5) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = (float)sacc;
    for (int r = 0; r < 16; ++r) s += c0[r];
    for (int j = 0; j < 8; ++j) s += v[j] + pk[j][0] + pk[j][1] + (float)iv[j] + g[j][0] + g[j][3];
    out[blockIdx.x * 256 + tid] = s;
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    if ((tid & 63) == 0) { clk[blockIdx.x * 4 + (tid >> 6)] = t1 - t0; clk[4096 + blockIdx.x * 4 + (tid >> 6)] = r1 - r0; }
}

template <int PER, int KIND>
double run(int wgs, const float* d_in, float* d_out, unsigned long long* d_clk) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL((k<PER, KIND>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<PER, KIND>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    (void)hipEventRecord(e1);
    (void)hipDeviceSynchronize();
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(8192);
    (void)hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, ticks = 0;
    for (int i = 0; i < wgs * 4; ++i) { cyc += h[i]; ticks += h[4096 + i]; }
    if (false) fprintf(stderr, "    [wgs %d kind %d per %d: wall %.3f ms, mean wave lifetime %.3f ms (s_memrealtime), s_memtime/s_memrealtime = %.2f]\n",
                                      wgs, KIND, PER, ms, ticks / (wgs * 4) / 1e5, cyc / ticks);
    return cyc / (wgs * 4) / iters;
}

int main() {
    float* d_in; float* d_out; unsigned long long* d_clk;
    (void)hipMalloc(&d_in, 1 << 20); (void)hipMalloc(&d_out, 1024 * 256 * 4); (void)hipMalloc(&d_clk, 8192 * 8);
    (void)hipMemset(d_in, 0, 1 << 20);
    const char* names[13] = {"v_fma_f32", "v_pk_fma_f32", "v_lshl_add_u32", "s_add_u32", "v_fma_f32, no MFMA", "ds_read_b128", "ds_write_b128", "global_load_dwordx4", "global_load_dwordx4 saddr", "global_load_dword saddr", "v_pk_fma_f32, no MFMA", "v_pk_add_f32, no MFMA", "v_pk_mul_f32, no MFMA"};
    for (int wgs : {256, 512, 1024}) {
        printf("%d waves per SIMD: cycles per trip of 8 MFMAs (512 = MFMA bound) with 0 / 1 / 2 / 4 / 8 instructions after each MFMA\n", wgs / 256);
#define ROW(K) printf("  %-20s %7.0f %7.0f %7.0f %7.0f %7.0f\n", names[K], run<0, K>(wgs, d_in, d_out, d_clk), run<1, K>(wgs, d_in, d_out, d_clk), \
                      run<2, K>(wgs, d_in, d_out, d_clk), run<4, K>(wgs, d_in, d_out, d_clk), run<8, K>(wgs, d_in, d_out, d_clk));
        ROW(4) ROW(10) ROW(11) ROW(12)
    }
    return 0;
}
