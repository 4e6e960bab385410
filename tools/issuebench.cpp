// issuebench -- development microbenchmark: what ds_read_b128 / global_load_lds cost next to MFMAs
// on one wave per SIMD, clustered in front of a block of MFMAs or interleaved one per MFMA gap.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/issuebench tools/issuebench.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
// in-place accumulators by construction (the builtin lets the allocator rotate loop-carried
// accumulators through copies, which is what one then measures)
__device__ __forceinline__ f32x4 mfma_inplace(float a, float b, f32x4 c)
{
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
    return c;
}
#define MFMA(a, b, c) mfma_inplace((a), (b), (c))

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// MODE 0: MFMAs only; 1: NL ds_reads clustered before 24 MFMAs; 2: the same reads one per MFMA gap;
// 3: NG global_load_lds clustered; 4: one per second MFMA gap; 5: reads + glds clustered; 6: both interleaved
template <int MODE, int NL, int NG>
__global__ __launch_bounds__(256, 1) void k(const float* in, float* out, long long* ticks, int iters)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, w = tid >> 6;
    for (int i = tid; i < 32768 / 4; i += 256) ((float*)lds)[i] = in[i & 1023];
    __syncthreads();
    float a[6], b[4];
    for (int i = 0; i < 6; ++i) a[i] = in[tid + 256 * i];
    for (int i = 0; i < 4; ++i) b[i] = in[tid + 17 * i];
    f32x4 c[24], u[12];
    for (int i = 0; i < 24; ++i) c[i] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < 12; ++i) u[i] = f32x4{a[i % 6], a[(i + 1) % 6], a[(i + 2) % 6], a[(i + 3) % 6]};
    const char* lp = lds + (tid & 63) * 16;
    const char* gp = (const char*)in + tid * 16;
    char* ld = lds + 32768 + w * 1024;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 1 || MODE == 5) {
#pragma unroll
            for (int j = 0; j < NL; ++j) u[j] = *(const f32x4*)(lp + j * 1024);
        }
        if (MODE == 3 || MODE == 5) {
#pragma unroll
            for (int j = 0; j < NG; ++j) glds16(gp + j * 4096, ld + j * 4096);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < 24; ++i) {
            // consume the fragments of the previous iteration so the reads are real
            c[i] = MFMA(u[i % 12][i & 3], b[i % 4], c[i]);
            if ((MODE == 2 || MODE == 6) && i * NL / 24 != (i + 1) * NL / 24) u[i * NL / 24] = *(const f32x4*)(lp + (i * NL / 24) * 1024);
            if ((MODE == 4 || MODE == 6) && i * NG / 24 != (i + 1) * NG / 24) glds16(gp + (i * NG / 24) * 4096, ld + (i * NG / 24) * 4096);
            if (MODE == 2 || MODE == 4 || MODE == 6) __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    __syncthreads();
    float s = lds[32768 + tid];
    for (int i = 0; i < 24; ++i) for (int r = 0; r < 4; ++r) s += c[i][r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) ticks[blockIdx.x] = t1 - t0;
}

template <int MODE, int NL, int NG>
void run(const char* name, const float* in, float* out, long long* ticks)
{
    const int wgs = 256, iters = 1000;
    hipFuncSetAttribute((const void*)k<MODE, NL, NG>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<MODE, NL, NG>), dim3(wgs), dim3(256), 65536, 0, in, out, ticks, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    std::vector<long long> t(wgs);
    hipMemcpy(t.data(), ticks, wgs * 8, hipMemcpyDeviceToHost);
    std::sort(t.begin(), t.end());
    printf("%-44s %7.1f cycles / 24 MFMAs (768 ideal), %.3f ms\n", name, (double)t[wgs / 2] / iters, ms);
}

int main()
{
    float *in, *out; long long* ticks;
    hipMalloc(&in, 1 << 20); hipMalloc(&out, 256 * 256 * 4); hipMalloc(&ticks, 256 * 8);
    std::vector<float> h(1 << 18);
    for (auto& v : h) v = (rand() & 1023) / 1024.f;
    hipMemcpy(in, h.data(), 1 << 20, hipMemcpyHostToDevice);
    run<0, 0, 0>("MFMA only", in, out, ticks);
    run<1, 6, 0>("6 ds_read_b128 clustered", in, out, ticks);
    run<2, 6, 0>("6 ds_read_b128 interleaved", in, out, ticks);
    run<1, 12, 0>("12 ds_read_b128 clustered", in, out, ticks);
    run<2, 12, 0>("12 ds_read_b128 interleaved", in, out, ticks);
    run<3, 0, 4>("4 global_load_lds clustered", in, out, ticks);
    run<4, 0, 4>("4 global_load_lds interleaved", in, out, ticks);
    run<5, 6, 4>("6 ds_read + 4 glds clustered", in, out, ticks);
    run<6, 6, 4>("6 ds_read + 4 glds interleaved", in, out, ticks);
    return 0;
}
