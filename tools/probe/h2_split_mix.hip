// Probe: the two-term fp16 split with one v_fma_mix per low term against the subtract / multiply / convert form, bit for bit, on 2^24 pairs
// (fp16-range values, exact fp16 values and near-ties, fp32 subnormals, values near the fp16 maximum).  hipcc -O3 --offload-arch=gfx950; GPU box.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <cmath>
typedef _Float16 h2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t h2_lo_terms(uint32_t h, float t0, float t1) {
    uint32_t l;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]\n\t"
        "v_fma_mixhi_f16 %0, %1, %2, %4 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=&v"(l) : "v"(h), "s"(-2048.0f), "v"(t0), "v"(t1));
    return l;
}
__device__ __forceinline__ void split_new(float x0, float x1, uint32_t& h, uint32_t& l) {
    const h2v hp = {(_Float16)x0, (_Float16)x1};
    h = __builtin_bit_cast(uint32_t, hp);
    l = h2_lo_terms(h, x0 * 2048.0f, x1 * 2048.0f);
}
__device__ __forceinline__ void split_old(float x0, float x1, uint32_t& h, uint32_t& l) {
    const h2v hp = {(_Float16)x0, (_Float16)x1};
    h = __builtin_bit_cast(uint32_t, hp);
    const float r0 = (x0 - (float)hp.x) * 2048.0f, r1 = (x1 - (float)hp.y) * 2048.0f;
    const h2v lp = {(_Float16)r0, (_Float16)r1};
    l = __builtin_bit_cast(uint32_t, lp);
}
__global__ void k_cmp(const float2* x, unsigned long long* bad, uint4* first, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    uint32_t h0, l0, h1, l1;
    split_new(x[i].x, x[i].y, h0, l0); split_old(x[i].x, x[i].y, h1, l1);
    if (h0 != h1 || l0 != l1) { if (atomicAdd(bad, 1ull) == 0) *first = make_uint4(__float_as_uint(x[i].x), __float_as_uint(x[i].y), l0, l1); }
}
int main() {
    const int n = 1 << 24;
    std::vector<float2> h(n);
    uint64_t s = 88172645463325252ull;
    auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return s; };
    for (int i = 0; i < n; ++i) {
        float v[2];
        for (int j = 0; j < 2; ++j) {
            const uint64_t r = rnd();
            const int kind = (int)(r & 7);
            uint32_t bits;
            if (kind < 5) { // random fp32 with exponent in the fp16 range and around its subnormals: 2^-30 .. 2^15
                const uint32_t e = 127 - 30 + (uint32_t)((r >> 8) % 46);
                bits = (uint32_t)((r >> 63) << 31) | (e << 23) | (uint32_t)((r >> 20) & 0x7fffff);
            } else if (kind == 5) { // exactly representable in fp16, or half-way cases
                const _Float16 q = (_Float16)(float)((int)((r >> 8) & 0xffff) - 32768) * (_Float16)0.03125f;
                float f = (float)q; std::memcpy(&bits, &f, 4); if (r & (1ull << 40)) bits += (r >> 41) & 1 ? 0x1000 : 0xfff;
            } else if (kind == 6) { bits = (uint32_t)(r >> 32) & 0x807fffffu; /* fp32 subnormals / zeros */ }
            else { float f = 60000.0f + (float)((r >> 8) & 0xffff) * 0.08f; if (r & (1ull << 50)) f = -f; std::memcpy(&bits, &f, 4); }
            std::memcpy(&v[j], &bits, 4);
        }
        h[i] = make_float2(v[0], v[1]);
    }
    float2* d; unsigned long long* bad; uint4* first;
    (void)hipMalloc(&d, n * sizeof(float2)); (void)hipMalloc(&bad, 8); (void)hipMalloc(&first, 16);
    (void)hipMemcpy(d, h.data(), n * sizeof(float2), hipMemcpyHostToDevice); (void)hipMemset(bad, 0, 8); (void)hipMemset(first, 0, 16);
    hipLaunchKernelGGL(k_cmp, dim3(n / 256), dim3(256), 0, 0, d, bad, first, n);
    unsigned long long hb = 0; uint4 hf;
    (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost); (void)hipMemcpy(&hf, first, 16, hipMemcpyDeviceToHost);
    printf("h2 split: v_fma_mix form against subtract / multiply / convert on %d pairs: %llu differ", n, hb);
    if (hb) printf(" (first: x = %08x %08x, l new %08x old %08x)", hf.x, hf.y, hf.z, hf.w);
    printf("\n");
    return hb ? 1 : 0;
}
