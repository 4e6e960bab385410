// ldsbench -- development harness: bf16 MFMA loop (v_mfma_f32_32x32x16_bf16, 64 px x 96 co wave tile = 6 MFMAs
// per k16 step) with R fresh ds_read_b128 fragment reads per step out of an LDS image of random data: what the
// chip sustains (it is power-capped on random data) as a function of LDS bytes per MFMA.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/ldsbench tools/ldsbench.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

// TILE 0: 64x96 per wave (2 A + 3 B frags, 6 MFMAs); TILE 1: 128x96 per wave (4 A + 3 B, 12 MFMAs)
template <int READS, int TILE>
__global__ __launch_bounds__(256) void k(const bf16x8* src, float* out, int iters, long long* clk)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 16; i += 256) ((bf16x8*)lds)[i] = src[(i + blockIdx.x * 131) & 65535];
    __syncthreads();
    constexpr int NA = TILE ? 4 : 2;
    bf16x8 a[NA], b[3];
    for (int i = 0; i < NA; ++i) a[i] = src[(tid * 7 + i) & 65535];
    for (int i = 0; i < 3; ++i) b[i] = src[(tid * 7 + 4 + i) & 65535];
    f32x16 acc[NA][3] = {};
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const char* base = lds + lane * 16;
    for (int it = 0; it < iters; ++it) {
        const char* p = base + ((it * 5120) & 32767);
#pragma unroll
        for (int r = 0; r < READS; ++r) {
            const bf16x8 v = *(const bf16x8*)(p + r * 1024);
            if (r < NA) a[r] = v; else if (r < NA + 3) b[r - NA] = v;
        }
#pragma unroll
        for (int m = 0; m < NA; ++m)
#pragma unroll
            for (int n = 0; n < 3; ++n) acc[m][n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(b[n], a[m], acc[m][n], 0, 0, 0);
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0.f;
    for (int m = 0; m < NA; ++m) for (int n = 0; n < 3; ++n) for (int q = 0; q < 16; ++q) sum += acc[m][n][q];
    out[tid + blockIdx.x * 256] = sum;
    if (tid == 0) { clk[blockIdx.x * 2] = t1 - t0; clk[blockIdx.x * 2 + 1] = r1 - r0; }
}

template <int READS, int TILE>
void run(const bf16x8* d, float* o, long long* c, int iters, int wgs)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipFuncSetAttribute((const void*)k<READS, TILE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    float best = 1e9; long long hc[2];
    for (int rep = 0; rep < 3; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((k<READS, TILE>), dim3(wgs), dim3(256), 65536, 0, d, o, iters, c);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (ms < best) { best = ms; CK(hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost)); }
    }
    const int mf = TILE ? 12 : 6;
    const double flop = (double)wgs * 4 * iters * mf * 32.0 * 32 * 16 * 2;
    printf("tile %s  reads/step %d (%.2f KB LDS per MFMA)  %4d WGs: %.3f ms  %.0f TFLOP/s  cycles/step %.1f (MFMA %d)  clock %.2f GHz\n",
           TILE ? "128x96" : " 64x96", READS, READS * 1.0 / mf, wgs, best, flop / best * 1e-9, (double)hc[0] / iters, mf * 32,
           (double)hc[0] / hc[1] * 0.1);
}

int main(int argc, char** argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 20000;
    std::vector<unsigned short> h(65536 * 8);
    srand(1);
    for (auto& v : h) { float f = (rand() / (float)RAND_MAX) * 2.f - 1.f; unsigned u; __builtin_memcpy(&u, &f, 4); v = u >> 16; }
    bf16x8* d; float* o; long long* c;
    CK(hipMalloc(&d, h.size() * 2)); CK(hipMalloc(&o, 1024 * 256 * 4)); CK(hipMalloc(&c, 1024 * 16));
    CK(hipMemcpy(d, h.data(), h.size() * 2, hipMemcpyHostToDevice));
    for (int wgs : {256, 512}) {
        run<0, 0>(d, o, c, iters, wgs);
        run<2, 0>(d, o, c, iters, wgs);
        run<5, 0>(d, o, c, iters, wgs);
        if (wgs == 256) { run<0, 1>(d, o, c, iters / 2, wgs); run<7, 1>(d, o, c, iters / 2, wgs); }
    }
    return 0;
}
