// tools/bench_bf16w.hip -- timing harness for the wide-tile bf16 3x3 conv kernel alone (trunk shape, 256 images) and its
// ablations (template parameter ABL of conv3x3_bf16w_kernel: pieces removed, results wrong, time only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 tools/bench_bf16w.hip -o tools/bench_bf16w
#include "conv3x3_bf16_wide.hip"

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using namespace srgen;

template <int ABL, bool RES, bool ACT>
static int run(const char* name, Conv3x3Args a, int reps)
{
    using namespace c3w;
    const int nitems = a.N * (64 / TH) * (64 / TW);
    const int grid = 256;
    auto k = conv3x3_bf16w_kernel<64, false, false, RES, ACT, ABL>;
    CK_(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES));
    hipEvent_t e0, e1;
    CK_(hipEventCreate(&e0));
    CK_(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS_BYTES, 0, a, nitems);
    CK_(hipDeviceSynchronize());
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL(k, dim3(grid), dim3(256), LDS_BYTES, 0, a, nitems);
    CK_(hipEventRecord(e1, 0));
    CK_(hipEventSynchronize(e1));
    CK_(hipGetLastError());
    float ms = 0;
    CK_(hipEventElapsedTime(&ms, e0, e1));
    printf("%-52s %8.1f us\n", name, ms / reps * 1e3);
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 256, reps = 20;
    const size_t act = (size_t)N * 64 * 64 * 96;
    std::vector<unsigned short> hx(act);
    unsigned s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3c00u + ((s >> 9) & 0x3ff)) ^ (unsigned short)((s >> 3) & 0x8000u); }
    std::vector<unsigned short> hw((size_t)9 * 96 * 96);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3a00u + ((s >> 9) & 0x1ff)) ^ (unsigned short)((s >> 3) & 0x8000u); }
    std::vector<float> hb(96, 0.01f), hs(96, 0.25f);
    void *x, *y, *r, *w;
    float *b, *sl;
    CK_(hipMalloc(&x, act * 2)); CK_(hipMalloc(&y, act * 2)); CK_(hipMalloc(&r, act * 2));
    CK_(hipMalloc(&w, hw.size() * 2)); CK_(hipMalloc(&b, 96 * 4)); CK_(hipMalloc(&sl, 96 * 4));
    CK_(hipMemcpy(x, hx.data(), act * 2, hipMemcpyHostToDevice));
    CK_(hipMemcpy(r, hx.data(), act * 2, hipMemcpyHostToDevice));
    CK_(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK_(hipMemcpy(b, hb.data(), 96 * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(sl, hs.data(), 96 * 4, hipMemcpyHostToDevice));
    Conv3x3Args a{};
    a.in = x; a.out = y; a.wpack = w; a.bias = b; a.slope = sl; a.res = nullptr; a.pool_partial = nullptr;
    a.N = N; a.H = 64; a.W = 64; a.cout_blocks = 1; a.shuffle = 0;
    Conv3x3Args ar = a;
    ar.res = r; ar.slope = nullptr;
    for (int round = 0; round < 2; ++round) {
        printf("-- round %d\n", round);
        if (run<0, false, true>("full (PReLU)", a, reps)) return 1;
        if (run<0, true, false>("full (residual)", ar, reps)) return 1;
        if (run<1, false, true>("no epilogue", a, reps)) return 1;
        if (run<2, false, true>("no MFMA", a, reps)) return 1;
        if (run<4, false, true>("no halo staging", a, reps)) return 1;
        if (run<8, false, true>("no tap barriers", a, reps)) return 1;
        if (run<16, false, true>("no fragment reads", a, reps)) return 1;
        if (run<32, false, true>("vmcnt(0) at every tap", a, reps)) return 1;
        if (run<1 + 4, false, true>("no epilogue, no staging", a, reps)) return 1;
        if (run<1 + 4 + 8, false, true>("no epilogue, no staging, no barriers", a, reps)) return 1;
        if (run<1 + 4 + 8 + 16, false, true>("MFMA + slab copies only", a, reps)) return 1;
    }
    return 0;
}
