// shapebench -- development harness: bare bf16 MFMA loops on random operands, same 64 x 96 output tile per wave,
// v_mfma_f32_32x32x16_bf16 (12 per k32 step) against v_mfma_f32_16x16x32_bf16 (24 per k32 step); operands in
// registers (A/B re-used), so what differs is only the MFMA shape -> which one the chip clocks higher on.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/shapebench tools/shapebench.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const bf16x8* src, float* out, int iters, long long* clk)
{
    const int tid = threadIdx.x + blockIdx.x * 256;
    bf16x8 a[4], b[6];
    for (int i = 0; i < 4; ++i) a[i] = src[(tid * 10 + i) & 65535];
    for (int i = 0; i < 6; ++i) b[i] = src[(tid * 10 + 4 + i) & 65535];
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    if (SHAPE == 32) {
        f32x16 acc[2][3] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int m = 0; m < 2; ++m)
#pragma unroll
                    for (int n = 0; n < 3; ++n)
                        acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[n + 3 * ks], a[m + 2 * ks], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 2; ++m) for (int n = 0; n < 3; ++n) for (int q = 0; q < 16; ++q) sum += acc[m][n][q];
    } else {
        f32x4 acc[4][6] = {};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int m = 0; m < 4; ++m)
#pragma unroll
                for (int n = 0; n < 6; ++n)
                    acc[m][n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[n], a[m], acc[m][n], 0, 0, 0);
        }
        for (int m = 0; m < 4; ++m) for (int n = 0; n < 6; ++n) for (int q = 0; q < 4; ++q) sum += acc[m][n][q];
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    out[tid] = sum;
    if (threadIdx.x == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; __builtin_memcpy(&u, &f, 4); v = u >> 16; }
    if (getenv("ZERO")) std::fill(h.begin(), h.end(), 0);
    bf16x8* d; float* o; long long* c;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 2048 * 256 * 4)); CK(hipMalloc(&c, 2048 * 16));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; ++rep)
        for (int wgs : {256, 512})
            for (int shape : {32, 16}) {
                CK(hipEventRecord(e0));
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                else hipLaunchKernelGGL(k<16>, dim3(wgs), dim3(256), 0, 0, d, o, iters, c);
                CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                float ms; CK(hipEventElapsedTime(&ms, e0, e1));
                std::vector<long long> hc(wgs * 2); CK(hipMemcpy(hc.data(), c, wgs * 16, hipMemcpyDeviceToHost));
                const double flop = (double)wgs * 4 * iters * 64.0 * 96 * 32 * 2;
                printf("shape %2d  %4d WGs: %.3f ms  %.0f TFLOP/s  cycles/iter %.1f  clock %.2f GHz\n", shape, wgs, ms,
                       flop / ms * 1e-9, (double)hc[0] / iters, (double)hc[0] / hc[1] * 0.1);
            }
    return 0;
}
