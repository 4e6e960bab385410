// mfmabench -- development microbenchmark: cycles per v_mfma_f32_32x32x16_bf16 / 16x16x32 for
// 1 or 2 waves per SIMD, independent accumulators, operands in registers (s_memtime ticks).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k16(const bf16x8* in, float* out, long long* ticks, int iters)
{
    const int tid = threadIdx.x;
    bf16x8 a[4], b[6];
    for (int i = 0; i < 4; ++i) a[i] = in[tid + 256 * i];
    for (int i = 0; i < 6; ++i) b[i] = in[tid + 256 * (i % 5)];
    f32x4 c[24];
    for (int i = 0; i < 24; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 6; ++m)
#pragma unroll
            for (int n = 0; n < 4; ++n)
                c[m * 4 + n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(b[m], a[n], c[m * 4 + n], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 24; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void k(const bf16x8* in, float* out, long long* ticks, int iters)
{
    const int tid = threadIdx.x;
    bf16x8 a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768], b2 = in[tid + 1024];
    f32x16 c[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (SHAPE == 32) {
            c[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a0, c[0], 0, 0, 0);
            c[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b0, a1, c[1], 0, 0, 0);
            c[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a0, c[2], 0, 0, 0);
            c[3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b1, a1, c[3], 0, 0, 0);
            c[4] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a0, c[4], 0, 0, 0);
            c[5] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b2, a1, c[5], 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 6; ++i)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    f32x4 t = {c[i][4 * q], c[i][4 * q + 1], c[i][4 * q + 2], c[i][4 * q + 3]};
                    t = __builtin_amdgcn_mfma_f32_16x16x32_bf16(i & 1 ? b0 : b1, q & 1 ? a0 : a1, t, 0, 0, 0);
                    c[i][4 * q] = t[0]; c[i][4 * q + 1] = t[1]; c[i][4 * q + 2] = t[2]; c[i][4 * q + 3] = t[3];
                }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

__global__ __launch_bounds__(256) void kf32(const float* in, float* out, long long* ticks, int iters)
{
    const int tid = threadIdx.x;
    float a0 = in[tid], a1 = in[tid + 256], b0 = in[tid + 512], b1 = in[tid + 768], b2 = in[tid + 1024];
    f32x16 c[6];
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) c[i][r] = 0.f;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        c[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a0, c[0], 0, 0, 0);
        c[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(b0, a1, c[1], 0, 0, 0);
        c[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a0, c[2], 0, 0, 0);
        c[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(b1, a1, c[3], 0, 0, 0);
        c[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2, a0, c[4], 0, 0, 0);
        c[5] = __builtin_amdgcn_mfma_f32_32x32x2f32(b2, a1, c[5], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 6; ++i) for (int r = 0; r < 16; ++r) s += c[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

// v_mfma_f32_16x16x4_f32 with NACC independent accumulators (the Winograd kernel's shape: 24)
template <int NACC>
__global__ __launch_bounds__(256) void kf32_16(const float* in, float* out, long long* ticks, int iters)
{
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    const int tid = threadIdx.x;
    float a[6], b[4];
    for (int i = 0; i < 6; ++i) a[i] = in[tid + 256 * i];
    for (int i = 0; i < 4; ++i) b[i] = in[tid + 17 * i];
    f32x4 c[NACC];
    for (int i = 0; i < NACC; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < NACC; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i % 6], b[i % 4], c[i], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

int main()
{
    bf16x8* in; float* out; long long* ticks;
    hipMalloc(&in, 1280 * 16); hipMalloc(&out, 4096 * 256 * 4); hipMalloc(&ticks, 4096 * 8);
    std::vector<unsigned short> h(1280 * 8);
    for (auto& v : h) v = 0x3c00 + (rand() & 0x3ff) + ((rand() & 1) << 15);
    hipMemcpy(in, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    const int iters = 2000;
    for (int shape : {32, 16})
        for (int wgs : {256, 512, 1024}) {
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            for (int rep = 0; rep < 2; ++rep) {
                hipEventRecord(e0);
                if (shape == 32) hipLaunchKernelGGL(k<32>, dim3(wgs), dim3(256), 0, 0, in, out, ticks, iters);
                else hipLaunchKernelGGL(k16, dim3(wgs), dim3(256), 0, 0, in, out, ticks, iters / 2);
                hipEventRecord(e1); hipEventSynchronize(e1);
            }
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<long long> t(wgs);
            hipMemcpy(t.data(), ticks, wgs * 8, hipMemcpyDeviceToHost);
            std::sort(t.begin(), t.end());
            const double nm = shape == 32 ? 6.0 * iters : 24.0 * (iters / 2);
            const double flop = (double)wgs * 4 * nm * (shape == 32 ? 32768.0 : 16384.0);
            printf("shape %dx%d  %4d WGs (%.0f waves/SIMD): median %.1f ticks/MFMA per wave, %.3f ms, %.0f TFLOP/s, clock~%.2f GHz\n",
                   shape, shape, wgs, wgs / 256.0, t[wgs / 2] / nm, ms, flop / ms * 1e-9, t[wgs/2] / (ms * 1e6) * (wgs > 512 ? 0.5 : 1.0));
        }
    for (int wgs : {256, 512}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(kf32, dim3(wgs), dim3(256), 0, 0, (const float*)in, out, ticks, iters);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> t(wgs);
        hipMemcpy(t.data(), ticks, wgs * 8, hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double nm = 6.0 * iters, flop = (double)wgs * 4 * nm * 4096.0;
        printf("f32 32x32x2  %4d WGs: median %.1f ticks/MFMA per wave, %.3f ms, %.1f TFLOP/s, clock~%.2f GHz\n", wgs,
               t[wgs / 2] / nm, ms, flop / ms * 1e-9, t[wgs / 2] / (ms * 1e6));
    }
    for (int nacc : {24, 6}) {
        const int wgs = 256;
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        for (int rep = 0; rep < 2; ++rep) {
            hipEventRecord(e0);
            if (nacc == 24) hipLaunchKernelGGL(kf32_16<24>, dim3(wgs), dim3(256), 0, 0, (const float*)in, out, ticks, iters);
            else hipLaunchKernelGGL(kf32_16<6>, dim3(wgs), dim3(256), 0, 0, (const float*)in, out, ticks, iters * 4);
            hipEventRecord(e1); hipEventSynchronize(e1);
        }
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> t(wgs);
        hipMemcpy(t.data(), ticks, wgs * 8, hipMemcpyDeviceToHost);
        std::sort(t.begin(), t.end());
        const double nm = 24.0 * iters, flop = (double)wgs * 4 * nm * 2048.0;
        printf("f32 16x16x4 (%2d accumulators) %4d WGs: median %.1f ticks/MFMA per wave, %.3f ms, %.1f TFLOP/s, clock~%.2f GHz\n", nacc, wgs,
               t[wgs / 2] / nm, ms, flop / ms * 1e-9, t[wgs / 2] / (ms * 1e6));
    }
    return 0;
}
