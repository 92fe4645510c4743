// Does v_mfma_f32_32x32x16_f16 keep fp16 SUBNORMAL inputs (the two-term split's h below 6.1e-5 and most of its l terms of small values)?
// A = 2^-20 (subnormal in fp16), B = 1: every accumulator must read 16 * 2^-20 = 1.52587890625e-05; a flushing pipe gives 0.
// hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/mfma_f16_denorm tools/probe/mfma_f16_denorm.hip && /tmp/mfma_f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val) {
    f16x8 a, b;
    for (int j = 0; j < 8; ++j) { a[j] = (_Float16)a_val; b[j] = (_Float16)1.0f; }
    f32x16 c;
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    out[threadIdx.x] = c[0];
}
int main() {
    float* d; hipMalloc(&d, 64 * 4);
    const float vals[3] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24: the smallest subnormal */, 6.103515625e-05f /* 2^-14: smallest normal */};
    int bad = 0;
    for (int i = 0; i < 3; ++i) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, vals[i]);
        float h[64]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        const float want = 16.0f * vals[i];
        printf("a = %.8e: accumulator %.8e, expected %.8e %s\n", vals[i], h[0], want, h[0] == want ? "ok" : "FLUSHED / WRONG");
        bad += h[0] != want;
    }
    return bad;
}
