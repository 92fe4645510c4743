// Probe: what do plain instructions cost in the gaps of a bf16 MFMA stream of the SAME wave (gfx950)?  One wave per SIMD loops
// over 8 x v_mfma_f32_32x32x16_bf16 (two accumulators alternating) with PER instructions of one kind after every MFMA; cycles per
// trip by s_memtime (256 = MFMA bound).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_bf16_valu.hip -o build_ab/mfma_bf16_valu && build_ab/mfma_bf16_valu
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// KIND 0: v_fma_f32 on 8 rotating registers; 1: v_fma_f32 dependent chain; 2: v_cvt_pk_bf16_f32; 3: the split chain (cvt, lshl, sub: dependent);
// 4: ds_write_b64; 5: ds_read_b128; 6: v_accvgpr_read_b32; 7: v_med3_f32; 8: ds_read_b32; 9: v_fma_f32 with NO MFMAs
template <int PER, int KIND>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    __shared__ float sh[256 * 4 * 2];
    const int tid = threadIdx.x;
    const int lds_off = tid * 16, lds_off8 = tid * 8;
    sh[tid] = 0;
    __syncthreads();
    f32x4 g[8];
    f32x2 g2[8];
    for (int j = 0; j < 8; ++j) { g[j] = f32x4{0, 0, 0, 0}; g2[j] = f32x2{0, 0}; }
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (__bf16)in[tid + j]; b[j] = (__bf16)in[tid + 8 + j]; }
    f32x16 c0 = {0}, c1 = {0};
    float v[8], x = in[tid];
    unsigned iv[8];
    for (int j = 0; j < 8; ++j) { v[j] = in[tid + j]; iv[j] = tid + j; }
    float ag = in[tid + 3];
    asm volatile("v_accvgpr_write_b32 a0, %0" ::"v"(ag));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (KIND != 9) {
                if (j & 1) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < PER; ++q) {
                const int r = (j * PER + q) & 7;
                if (KIND == 0 || KIND == 9) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(x));
                else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(v[0]) : "v"(x));
                else if (KIND == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(iv[r]) : "v"(v[r]), "v"(x));
                else if (KIND == 3) {
                    if (q % 3 == 0) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(iv[0]) : "v"(v[0]), "v"(x));
                    else if (q % 3 == 1) asm volatile("v_lshlrev_b32 %0, 16, %1" : "=v"(iv[1]) : "v"(iv[0]));
                    else asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[0]) : "v"(iv[1]));
                }
                else if (KIND == 4) asm volatile("ds_write_b64 %0, %1" ::"v"(lds_off8), "v"(g2[r]) : "memory");
                else if (KIND == 5) asm volatile("ds_read_b128 %0, %1" : "=v"(g[r]) : "v"(lds_off) : "memory");
                else if (KIND == 6) asm volatile("v_accvgpr_read_b32 %0, a0" : "=v"(v[r]));
                else if (KIND == 7) asm volatile("v_med3_f32 %0, %0, %1, %1" : "+v"(v[r]) : "v"(x));
                else if (KIND == 8) asm volatile("ds_read_b32 %0, %1" : "=v"(v[r]) : "v"(lds_off8) : "memory");
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (KIND == 4 || KIND == 5 || KIND == 8) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += c0[r] + c1[r];
    for (int j = 0; j < 8; ++j) s += v[j] + (float)iv[j] + g[j][0] + g[j][3] + g2[j][0];
    out[blockIdx.x * 256 + tid] = s;
    if ((tid & 63) == 0) clk[blockIdx.x * 4 + (tid >> 6)] = t1 - t0;
}

template <int PER, int KIND>
double run(int wgs, const float* d_in, float* d_out, unsigned long long* d_clk) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<PER, KIND>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    hipLaunchKernelGGL((k<PER, KIND>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
    (void)hipDeviceSynchronize();
    std::vector<unsigned long long> h(wgs * 4);
    (void)hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int i = 0; i < wgs * 4; ++i) cyc += h[i];
    return cyc / (wgs * 4) / iters;
}

template <int KIND>
void row(const char* name, int wgs, const float* d_in, float* d_out, unsigned long long* d_clk) {
    printf("  %-40s %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f %7.0f\n", name, run<0, KIND>(wgs, d_in, d_out, d_clk), run<1, KIND>(wgs, d_in, d_out, d_clk),
           run<2, KIND>(wgs, d_in, d_out, d_clk), run<3, KIND>(wgs, d_in, d_out, d_clk), run<4, KIND>(wgs, d_in, d_out, d_clk),
           run<6, KIND>(wgs, d_in, d_out, d_clk), run<8, KIND>(wgs, d_in, d_out, d_clk));
}

int main() {
    float *d_in, *d_out;
    unsigned long long* d_clk;
    (void)hipMalloc(&d_in, 1 << 20); (void)hipMalloc(&d_out, 1 << 22); (void)hipMalloc(&d_clk, 1 << 16);
    (void)hipMemset(d_in, 0, 1 << 20);
    for (int wps = 1; wps <= 2; ++wps) {
        const int wgs = 256 * wps;
        printf("%d wave(s) per SIMD: cycles per trip of 8 bf16 MFMAs (256 = MFMA bound) with 0 / 1 / 2 / 3 / 4 / 6 / 8 instructions after each MFMA\n", wps);
        row<0>("v_fma_f32 independent", wgs, d_in, d_out, d_clk);
        row<1>("v_fma_f32 dependent chain", wgs, d_in, d_out, d_clk);
        row<2>("v_cvt_pk_bf16_f32", wgs, d_in, d_out, d_clk);
        row<3>("cvt -> lshl -> sub chain", wgs, d_in, d_out, d_clk);
        row<7>("v_med3_f32", wgs, d_in, d_out, d_clk);
        row<6>("v_accvgpr_read_b32", wgs, d_in, d_out, d_clk);
        row<4>("ds_write_b64", wgs, d_in, d_out, d_clk);
        row<5>("ds_read_b128", wgs, d_in, d_out, d_clk);
        row<8>("ds_read_b32", wgs, d_in, d_out, d_clk);
        row<9>("v_fma_f32, no MFMA", wgs, d_in, d_out, d_clk);
    }
    return 0;
}
