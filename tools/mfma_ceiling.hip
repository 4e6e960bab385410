// tools/mfma_ceiling.hip -- what the matrix pipe sustains on this MI355X under its power management, per instruction and per
// operand behaviour.  512 workgroups x 4 waves (two waves per SIMD), every wave issues nothing but MFMAs on six independent
// accumulators; the operands are (0) all-zero registers, (1) one fixed set of random registers, (2) eight sets of random
// registers used in rotation (what a real kernel's operand stream looks like to the datapath).  Prints TFLOP/s and the
// shader clock (clock64 of workgroup 0 / wall time).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/mfma_ceiling.hip -o tools/bin/mfma_ceiling
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int INSTR, int MODE>
__global__ __launch_bounds__(256, 2) void mfma_loop(const float* src, float* out, long long* clk, int iters)
{
    const int lane = threadIdx.x & 63;
    f32x4 ra[8], rb[8];                                   // 8 operand sets of 16 bytes per lane
#pragma unroll
    for (int s = 0; s < 8; ++s) {
        const int set = MODE == 2 ? s : 0;
        ra[s] = MODE == 0 ? f32x4{0, 0, 0, 0} : *(const f32x4*)(src + ((set * 2 + 0) * 64 + lane) * 4);
        rb[s] = MODE == 0 ? f32x4{0, 0, 0, 0} : *(const f32x4*)(src + ((set * 2 + 1) * 64 + lane) * 4);
        asm volatile("" : "+v"(ra[s]), "+v"(rb[s]));      // distinct registers even when the contents are equal
    }
    f32x16 acc[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 8; ++s)
#pragma unroll
            for (int i = 0; i < 6; ++i) {
                if (INSTR == 0)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, ra[s]), __builtin_bit_cast(h8, rb[(s + i) & 7]), acc[i], 0, 0, 0);
                else if (INSTR == 1)
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(b8, ra[s]), __builtin_bit_cast(b8, rb[(s + i) & 7]), acc[i], 0, 0, 0);
                else
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s][i & 3], rb[(s + i) & 7][i & 3], acc[i], 0, 0, 0);
            }
    }
    const long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += acc[i][r];
    if (s == 12345.678f) out[threadIdx.x] = s;            // keeps the accumulators alive
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
}

template <int INSTR, int MODE>
static int run(const char* name, const float* src, float* out, long long* clk, int iters, double flop_per_mfma)
{
    hipEvent_t e0, e1;
    CK_(hipEventCreate(&e0));
    CK_(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((mfma_loop<INSTR, MODE>), dim3(512), dim3(256), 0, 0, src, out, clk, iters);
    CK_(hipDeviceSynchronize());
    const int reps = 10;
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((mfma_loop<INSTR, MODE>), dim3(512), dim3(256), 0, 0, src, out, clk, iters);
    CK_(hipEventRecord(e1, 0));
    CK_(hipEventSynchronize(e1));
    float ms = 0;
    CK_(hipEventElapsedTime(&ms, e0, e1));
    long long c = 0;
    CK_(hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost));
    const double us = ms / reps * 1e3;
    const double flops = 512.0 * 4 * iters * 48 * flop_per_mfma;
    printf("%-52s %8.1f us  %7.1f TFLOP/s  clock %.2f GHz  (%.1f cycles per MFMA and SIMD)\n", name, us, flops / us * 1e-6, c / us * 1e-3,
           (double)c / (iters * 48.0 * 2));
    fflush(stdout);
    return 0;
}

int main()
{
    std::vector<float> h(16 * 64 * 4);
    unsigned s = 99;
    // random halves in [-1, 1): as f16 / bf16 pairs the bit patterns are random mantissas with moderate exponents
    std::vector<_Float16> hh(h.size() * 2);
    for (auto& v : hh) { s = s * 1664525u + 1013904223u; v = (_Float16)(((s >> 8) & 0xffff) / 32768.f - 1.f); }
    float *src, *out;
    long long* clk;
    CK_(hipMalloc(&src, h.size() * 4));
    CK_(hipMalloc(&out, 4096));
    CK_(hipMalloc(&clk, 64));
    CK_(hipMemcpy(src, hh.data(), h.size() * 4, hipMemcpyHostToDevice));
    const int iters = 400;
    for (int round = 0; round < 2; ++round) {
        printf("-- round %d\n", round);
        if (run<0, 0>("f16 32x32x16, zero operands", src, out, clk, iters, 32768)) return 1;
        if (run<0, 1>("f16 32x32x16, one random operand set", src, out, clk, iters, 32768)) return 1;
        if (run<0, 2>("f16 32x32x16, eight random operand sets in rotation", src, out, clk, iters, 32768)) return 1;
        if (run<1, 0>("bf16 32x32x16, zero operands", src, out, clk, iters, 32768)) return 1;
        if (run<1, 1>("bf16 32x32x16, one random operand set", src, out, clk, iters, 32768)) return 1;
        if (run<1, 2>("bf16 32x32x16, eight random operand sets in rotation", src, out, clk, iters, 32768)) return 1;
        if (run<2, 0>("f32 32x32x2, zero operands", src, out, clk, iters, 4096)) return 1;
        if (run<2, 2>("f32 32x32x2, eight random operand sets in rotation", src, out, clk, iters, 4096)) return 1;
    }
    return 0;
}
