// Probe: are sqrtf and '/' IEEE-correctly rounded in device code as hipcc builds them?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <random>
__global__ void k(const float* a, const float* b, float* s, float* d, float* m, int n) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) { s[i] = sqrtf(a[i]); d[i] = a[i] / b[i]; m[i] = 0.05f / (sqrtf(a[i] / b[i]) + 1e-8f); }
}
int main() {
    const int n = 1 << 20;
    std::vector<float> a(n), b(n), s(n), d(n), m(n);
    std::mt19937 g(1); std::uniform_real_distribution<float> u(1e-8f, 1.0f);
    for (int i = 0; i < n; ++i) { a[i] = u(g) * u(g); b[i] = 1.0f + (float)(g() % 100000); }
    float *da, *db, *ds, *dd, *dm;
    hipMalloc(&da, n * 4); hipMalloc(&db, n * 4); hipMalloc(&ds, n * 4); hipMalloc(&dd, n * 4); hipMalloc(&dm, n * 4);
    hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, da, db, ds, dd, dm, n);
    hipMemcpy(s.data(), ds, n * 4, hipMemcpyDeviceToHost); hipMemcpy(d.data(), dd, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(m.data(), dm, n * 4, hipMemcpyDeviceToHost);
    int bs = 0, bd = 0, bm = 0;
    for (int i = 0; i < n; ++i) {
        volatile float hs = sqrtf(a[i]); volatile float hd = a[i] / b[i];
        volatile float q = a[i] / b[i]; volatile float r = sqrtf(q); volatile float t = r + 1e-8f; volatile float hm = 0.05f / t;
        bs += hs != s[i]; bd += hd != d[i]; bm += hm != m[i];
    }
    printf("mismatches of %d: sqrtf %d, div %d, combined %d\n", n, bs, bd, bm);
    return 0;
}
