// Probe: how fast can a second wave of a SIMD issue plain VALU / SALU / LDS / VMEM instructions while the first
// streams f32 MFMAs?  One 512-thread workgroup per CU: waves 0-3 (one per SIMD) run an MFMA chain, waves 4-7 a loop
// of the instruction class under test, timed with s_memtime.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_valu_coissue.hip -o build_ab/mfma_valu_coissue && build_ab/mfma_valu_coissue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

// MODE: 0 no MFMA wave work (baseline), 1 one dependent accumulator, 2 two alternating accumulators
// KIND: 0 independent v_add_f32 x16, 1 dependent v_add chain x16, 2 s_add x16, 3 ds_write_b128 x4, 4 global_load_dwordx4 x4
template <int MODE, int KIND>
__global__ __launch_bounds__(512) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    __shared__ float sh[512 * 4 + 64];
    const int tid = threadIdx.x, wave = tid >> 6;
    if (wave < 4) {
        if (MODE == 0) return;
        float a = in[tid], b = in[tid + 512];
        f32x16 c0 = {0}, c1 = {0};
        const int n = iters * 4;                  // runs well past the other role's loop
        for (int it = 0; it < n; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                if (MODE == 1) c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
                else if (j & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c0, 0, 0, 0);
            }
            if (*(volatile int*)&sh[512 * 4] != 0) break;    // the timed waves raise this flag when they are done
        }
        float s = 0;
        for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
        out[blockIdx.x * 512 + tid] = s;
        return;
    }
    if (tid == 256) *(volatile int*)&sh[512 * 4] = 0;
    float v[16];
    for (int j = 0; j < 16; ++j) v[j] = in[tid + j];
    int sacc = iters;
    f32x4 g0 = {0, 0, 0, 0}, g1 = g0, g2 = g0, g3 = g0;
    const f32x4* gp = reinterpret_cast<const f32x4*>(in) + (tid & 255) * 4 + blockIdx.x * 64;
    const int lds_off = (tid & 255) * 16;
    // let the MFMA waves get going
    __builtin_amdgcn_s_sleep(100);
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (KIND == 0) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[j]) : "v"(1.0f));
        } else if (KIND == 1) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[0]) : "v"(1.0f));
        } else if (KIND == 2) {
#pragma unroll
            for (int j = 0; j < 16; ++j) asm volatile("s_add_u32 %0, %0, 3" : "+s"(sacc));
        } else if (KIND == 3) {
#pragma unroll
            for (int j = 0; j < 4; ++j) asm volatile("ds_write_b128 %0, %1" ::"v"(lds_off), "v"(g0) : "memory");
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        } else {
            asm volatile("global_load_dwordx4 %0, %4, off\n\tglobal_load_dwordx4 %1, %4, off offset:1024\n\t"
                         "global_load_dwordx4 %2, %4, off offset:2048\n\tglobal_load_dwordx4 %3, %4, off offset:3072\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(g0), "=&v"(g1), "=&v"(g2), "=&v"(g3) : "v"(gp) : "memory");
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (tid == 256) *(volatile int*)&sh[512 * 4] = 1;
    float s = (float)sacc + g0.x + g1.y + g2.z + g3.w;
    for (int j = 0; j < 16; ++j) s += v[j];
    out[blockIdx.x * 512 + tid] = s;
    if ((tid & 63) == 0) clk[blockIdx.x * 4 + wave - 4] = t1 - t0;
}

template <int MODE, int KIND>
double run(const float* d_in, float* d_out, unsigned long long* d_clk, int per_iter) {
    const int iters = 2000, wgs = 256;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<MODE, KIND>), dim3(wgs), dim3(512), 0, 0, d_in, d_out, iters, d_clk);
    hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 4);
    hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (auto v : h) cyc += v;
    return cyc / h.size() / iters / per_iter;
}

int main() {
    float* d_in; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_in, 1 << 20); hipMalloc(&d_out, 256 * 512 * 4); hipMalloc(&d_clk, 256 * 4 * 8);
    hipMemset(d_in, 0, 1 << 20);
    const char* names[5] = {"v_add_f32 independent", "v_add_f32 dependent", "s_add_u32", "ds_write_b128 (+wait per 4)", "global_load_dwordx4 (+wait per 4)"};
    double r[3][5];
#define ROW(K, N) r[0][K] = run<0, K>(d_in, d_out, d_clk, N); r[1][K] = run<1, K>(d_in, d_out, d_clk, N); r[2][K] = run<2, K>(d_in, d_out, d_clk, N);
    ROW(0, 16) ROW(1, 16) ROW(2, 16) ROW(3, 4) ROW(4, 4)
    printf("cycles per instruction of the second wave of a SIMD   alone | beside a 1-accumulator MFMA chain | beside 2 alternating accumulators\n");
    for (int kk = 0; kk < 5; ++kk) printf("  %-36s %8.1f %8.1f %8.1f\n", names[kk], r[0][kk], r[1][kk], r[2][kk]);
    return 0;
}
