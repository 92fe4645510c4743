// Probe: sustained FLOP/s and in-kernel clock of the two f32-input MFMA shapes on random operands, with and without
// LDS fragment reads between them (MI355X_MICROARCH "DVFS give-back" item 7 reports 16x16 vs 32x32 differences for bf16).
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_f32_shapes.hip -o build_ab/mfma_f32_shapes && build_ab/mfma_f32_shapes
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int SHAPE, bool LDS>
__global__ __launch_bounds__(256) void k(const float* in, float* out, int iters, unsigned long long* clk) {
    __shared__ float sh[2 * 64 * 36];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 2 * 64 * 36; i += 256) sh[i] = in[(blockIdx.x * 97 + i) & 65535];
    __syncthreads();
    float a[8], b[8];
    for (int j = 0; j < 8; ++j) { a[j] = in[(tid * 8 + j) & 65535]; b[j] = in[(tid * 8 + j + 4096) & 65535]; }
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const float* ap = sh + (lane & 31) * 36 + 4 * (lane >> 5);
    const float* bp = sh + 64 * 36 + (lane & 31) * 36 + 4 * (lane >> 5);
    if (SHAPE == 32) {
        f32x16 acc = {0};
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 fa = *reinterpret_cast<const float4*>(ap + 8 * ((it + g) & 3));
                    const float4 fb = *reinterpret_cast<const float4*>(bp + 8 * ((it + g) & 3));
                    a[4 * g] = fa.x; a[4 * g + 1] = fa.y; a[4 * g + 2] = fa.z; a[4 * g + 3] = fa.w;
                    b[4 * g] = fb.x; b[4 * g + 1] = fb.y; b[4 * g + 2] = fb.z; b[4 * g + 3] = fb.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);      // 8 x 4096 FLOP
        }
        float s = 0;
        for (int r = 0; r < 16; ++r) s += acc[r];
        out[blockIdx.x * 256 + tid] = s;
    } else {
        f32x4 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
        for (int it = 0; it < iters; ++it) {
            if (LDS) {
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const float4 fa = *reinterpret_cast<const float4*>(ap + 8 * ((it + g) & 3));
                    const float4 fb = *reinterpret_cast<const float4*>(bp + 8 * ((it + g) & 3));
                    a[4 * g] = fa.x; a[4 * g + 1] = fa.y; a[4 * g + 2] = fa.z; a[4 * g + 3] = fa.w;
                    b[4 * g] = fb.x; b[4 * g + 1] = fb.y; b[4 * g + 2] = fb.z; b[4 * g + 3] = fb.w;
                }
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {                                                                          // 16 x 2048 FLOP
                c0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j + 4], c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 4], b[j], c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 4], b[j + 4], c3, 0, 0, 0);
            }
        }
        out[blockIdx.x * 256 + tid] = c0[0] + c1[1] + c2[2] + c3[3];
    }
    if (tid == 0) { clk[2 * blockIdx.x] = __builtin_amdgcn_s_memtime() - t0; clk[2 * blockIdx.x + 1] = __builtin_amdgcn_s_memrealtime() - r0; }
}

template <int SHAPE, bool LDS>
void run(const char* name, int wgs, const float* d_in, float* d_out, unsigned long long* d_clk) {
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<SHAPE, LDS>), dim3(wgs), dim3(256), 0, 0, d_in, d_out, iters, d_clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
    }
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(2 * wgs);
    hipMemcpy(h.data(), d_clk, h.size() * 8, hipMemcpyDeviceToHost);
    double cyc = 0, tick = 0;
    for (int i = 0; i < wgs; ++i) { cyc += h[2 * i]; tick += h[2 * i + 1]; }
    const double flop = (double)wgs * 4 * iters * 32768.0;      // per wave and iteration: 8 x 4096 = 16 x 2048
    printf("%-26s wgs %4d  %7.2f ms  %6.1f TFLOP/s  clock %.2f GHz  cycles/iter %.0f\n", name, wgs, ms, flop / ms / 1e9,
           cyc / tick / 10.0, cyc / wgs / iters);
}

int main() {
    float* d_in; float* d_out; unsigned long long* d_clk;
    hipMalloc(&d_in, 65536 * 4); hipMalloc(&d_out, 1024 * 256 * 4); hipMalloc(&d_clk, 1024 * 16);
    std::vector<float> h(65536);
    srand(1);
    for (auto& v : h) v = (float)rand() / RAND_MAX * 2.f - 1.f;
    hipMemcpy(d_in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int wgs : {256, 512}) {
        run<32, false>("32x32x2 regs", wgs, d_in, d_out, d_clk);
        run<16, false>("16x16x4 regs", wgs, d_in, d_out, d_clk);
        run<32, true>("32x32x2 + LDS reads", wgs, d_in, d_out, d_clk);
        run<16, true>("16x16x4 + LDS reads", wgs, d_in, d_out, d_clk);
    }
    return 0;
}
