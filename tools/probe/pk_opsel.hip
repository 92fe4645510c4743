// v_pk_*_f32 with op_sel / op_sel_hi / neg_lo / neg_hi: the radix-2 combine Y[k], Y[k + 16] = E[k] +- W^k O[k] on registers that hold (E, O) side by side,
// four packed instructions per k (dft32_combine_pk).  hipcc --offload-arch=gfx950 -O3 -Wno-unused-value -o /tmp/pk_opsel tools/probe/pk_opsel.hip && /tmp/pk_opsel
#include <hip/hip_runtime.h>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const f2* in, f2* out) {
    f2 re = in[threadIdx.x], im = in[64 + threadIdx.x];
    const f2 cs = {0.98078528f, 0.19509032f};
    f2 tmp, tt, r2, i2;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[1,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(tmp) : "v"(re), "s"(cs));
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[1,1,0] op_sel_hi:[1,0,1] neg_lo:[0,0,0] neg_hi:[0,0,0]" : "=v"(tt) : "v"(im), "s"(cs), "v"(tmp));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(r2) : "v"(re), "v"(tt));
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1] neg_lo:[0,0] neg_hi:[0,1]" : "=v"(i2) : "v"(im), "v"(tt));
    out[threadIdx.x] = r2; out[64 + threadIdx.x] = i2;
}
int main() {
    f2 h[128], o[128]; for (int i = 0; i < 128; ++i) h[i] = f2{(float)(i + 1), (float)(2 * i + 3) * 0.5f};
    f2 *d, *e; hipMalloc(&d, sizeof h); hipMalloc(&e, sizeof o); hipMemcpy(d, h, sizeof h, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e); hipMemcpy(o, e, sizeof o, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        const float C = 0.98078528f, S = 0.19509032f, ar = h[l].x, orr = h[l].y, ai = h[64 + l].x, oi = h[64 + l].y;
        const float tr = C * orr + S * oi, ti = C * oi - S * orr;
        const float e0 = ar + tr, e1 = ar - tr, e2 = ai + ti, e3 = ai - ti;
        if (fabsf(o[l].x - e0) > 1e-4f * fabsf(e0) + 1e-5f || fabsf(o[l].y - e1) > 1e-4f * fabsf(e1) + 1e-5f || fabsf(o[64 + l].x - e2) > 1e-4f * fabsf(e2) + 1e-5f || fabsf(o[64 + l].y - e3) > 1e-4f * fabsf(e3) + 1e-5f) { if (bad < 4) printf("lane %d: got %g %g %g %g want %g %g %g %g\n", l, o[l].x, o[l].y, o[64+l].x, o[64+l].y, e0, e1, e2, e3); ++bad; }
    }
    printf("op_sel combine: %d bad lanes\n", bad);
    return bad != 0;
}
