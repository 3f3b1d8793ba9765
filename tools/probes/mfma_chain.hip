// Probe (round 6): (1) rate of v_mfma_f32_32x32x2_f32 in ONE dependent chain per wave vs v_mfma_f32_16x16x4_f32 in FOUR independent
// chains per wave (the four 16 x 16 tiles of the same 32 x 32 block), one and two waves per SIMD; (2) the order in which
// v_mfma_f32_16x16x4_f32 sums its four k-slots: against fmaf chains in every slot permutation.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/mfma_chain tools/probes/mfma_chain.hip && /tmp/mfma_chain
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>      // 0: one 32x32x2 chain; 1: four 16x16x4 chains; 2: two 32x32x2 chains (independent)
__global__ void rate_kernel(const float* __restrict__ in, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = in[(lane * 8 + i) & 1023]; b[i] = in[(lane * 8 + i + 512) & 1023]; }
    if (MODE == 0) {
        f32x16 acc;
        for (int i = 0; i < 16; ++i) acc[i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 8; ++j) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc, 0, 0, 0);
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += acc[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else if (MODE == 2) {
        f32x16 acc0, acc1;
        for (int i = 0; i < 16; ++i) acc0[i] = acc1[i] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j], b[j], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a[j + 4], b[j + 4], acc1, 0, 0, 0);
            }
        }
        float s = 0; for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    } else {
        f32x4 t[4];
        for (int i = 0; i < 4; ++i) t[i] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {       // (same flops per iteration as MODE 0: 16 x 2048 = 8 x 4096)
                t[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j], t[0], 0, 0, 0);
                t[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j], b[j + 4], t[1], 0, 0, 0);
                t[2] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 4], b[j], t[2], 0, 0, 0);
                t[3] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j + 4], b[j + 4], t[3], 0, 0, 0);
            }
        }
        float s = 0; for (int i = 0; i < 4; ++i) s += t[i][0] + t[i][1] + t[i][2] + t[i][3];
        out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    }
}

// one 16x16x4 MFMA on given operands: A[16][4], B[4][16] -> D[16][16]
__global__ void order_kernel(const float* __restrict__ A, const float* __restrict__ B, const float* __restrict__ Cin, float* __restrict__ D) {
    const int lane = threadIdx.x, i = lane & 15, k = lane >> 4;
    f32x4 acc;
    for (int r = 0; r < 4; ++r) acc[r] = Cin[(4 * k + r) * 16 + i];          // D row = 4 (lane / 16) + r, column = lane % 16
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(A[i * 4 + k], B[k * 16 + i], acc, 0, 0, 0);
    for (int r = 0; r < 4; ++r) D[(4 * k + r) * 16 + i] = acc[r];
}

int main() {
    float *in, *out;
    std::vector<float> h(1024);
    srand(1);
    for (auto& x : h) x = (float)rand() / RAND_MAX;
    hipMalloc(&in, 4096); hipMalloc(&out, 4 << 20);
    hipMemcpy(in, h.data(), 4096, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int wps = 1; wps <= 2; ++wps)
        for (int mode = 0; mode < 3; ++mode) {
            dim3 grid(256), block(256 * wps);
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, grid, block, 0, 0, in, out, iters);
                else if (mode == 1) hipLaunchKernelGGL(rate_kernel<1>, grid, block, 0, 0, in, out, iters);
                else hipLaunchKernelGGL(rate_kernel<2>, grid, block, 0, 0, in, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                const double flops = 256.0 * 4 * wps * iters * 8 * 4096.0;
                if (rep == 2) printf("waves/SIMD %d  %s: %.2f ms  %.1f TFLOP/s\n", wps,
                                     mode == 0 ? "32x32x2 one dependent chain   " : mode == 1 ? "16x16x4 four independent tiles" : "32x32x2 two independent chains",
                                     ms, flops / ms / 1e9);
            }
        }
    // ---- summation order of the four k-slots of 16x16x4 ----
    std::vector<float> A(64), B(64), C(256), D(256);
    for (auto& x : A) x = (float)rand() / RAND_MAX * ((rand() & 1) ? 1.f : 1e-4f);
    for (auto& x : B) x = (float)rand() / RAND_MAX * ((rand() & 1) ? 1.f : 1e-3f);
    for (auto& x : C) x = (float)rand() / RAND_MAX;
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice);
    hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(order_kernel, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD);
    hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
    int perm[4] = {0, 1, 2, 3};
    do {
        int equal = 0;
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float acc = C[i * 16 + j];
                for (int s = 0; s < 4; ++s) acc = fmaf(A[i * 4 + perm[s]], B[perm[s] * 16 + j], acc);
                equal += acc == D[i * 16 + j];
            }
        printf("fmaf chain in slot order %d %d %d %d: %d / 256 bit-equal\n", perm[0], perm[1], perm[2], perm[3], equal);
    } while (std::next_permutation(perm, perm + 4));
    return 0;
}
