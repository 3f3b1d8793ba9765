// probe: does v_cvt_pknorm_u16_f32 equal round-to-nearest-even(clamp(v, 0, 1) * 65535)?   hipcc --offload-arch=gfx950 pknorm.hip -o /tmp/pknorm && /tmp/pknorm
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <math.h>
__global__ void k(const float* x, uint32_t* a, uint32_t* b, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    typedef unsigned short us2 __attribute__((ext_vector_type(2)));
    us2 p = __builtin_amdgcn_cvt_pknorm_u16(x[i], x[i]);
    a[i] = p.x;
    b[i] = __float2uint_rn(__builtin_amdgcn_fmed3f(x[i], 0.f, 1.f) * 65535.f);
}
int main() {
    const int n = 1 << 24;
    float* hx = (float*)malloc(n * 4);
    for (int i = 0; i < n; ++i) {
        if (i < (1 << 23)) hx[i] = (float)i / (float)(1 << 23) * 1.0f;             // dense in [0, 1)
        else hx[i] = -0.5f + 2.0f * (float)rand() / RAND_MAX;                       // incl. negatives and > 1
    }
    // exact half-way points k + 0.5
    for (int k2 = 0; k2 < 65535; ++k2) hx[(1 << 23) + k2] = ((float)k2 + 0.5f) / 65535.f;
    float* dx; uint32_t *da, *db;
    hipMalloc(&dx, n * 4); hipMalloc(&da, n * 4); hipMalloc(&db, n * 4);
    hipMemcpy(dx, hx, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, dx, da, db, n);
    uint32_t* ha = (uint32_t*)malloc(n * 4); uint32_t* hb = (uint32_t*)malloc(n * 4);
    hipMemcpy(ha, da, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hb, db, n * 4, hipMemcpyDeviceToHost);
    long diff = 0, big = 0; double maxerr = 0;
    for (int i = 0; i < n; ++i) {
        if (ha[i] != hb[i]) { ++diff; if (abs((int)ha[i] - (int)hb[i]) > 1) ++big; if (diff < 8) printf("x=%.9g pknorm=%u rn=%u\n", hx[i], ha[i], hb[i]); }
        double c = hx[i] < 0 ? 0 : hx[i] > 1 ? 1 : hx[i];
        double e = fabs((double)ha[i] / 65535.0 - c); if (e > maxerr) maxerr = e;
    }
    printf("n=%d differing=%ld (more than 1 apart: %ld) max |pknorm/65535 - clamp(x)| = %.3e (half a step = %.3e)\n", n, diff, big, maxerr, 0.5 / 65535);
    return 0;
}
