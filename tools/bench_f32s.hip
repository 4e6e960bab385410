// tools/bench_f32s.hip -- timing harness for the split-f16 3x3 conv kernel alone (trunk shape, 256 images) and its
// ablations (template parameter ABL of conv3x3_f32s_kernel: pieces of the kernel removed, results wrong, time only).
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -I superresolution_amd/csrc tools/bench_f32s.hip -o tools/bench_f32s
#include "../superresolution_amd/csrc/srgen_conv3x3_f32s.hip"

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK_(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

using namespace srgen;

template <int ABL, bool RES, int TERMS = 3>
static int run(const char* name, Conv3x3Args a, int reps)
{
    using namespace c3s;
    const int nitems = a.N * (64 / TH) * (64 / TW);
    const int grid = nitems < MAX_WGS ? nitems & ~7 : MAX_WGS;
    hipEvent_t e0, e1;
    CK_(hipEventCreate(&e0));
    CK_(hipEventCreate(&e1));
    for (int i = 0; i < 10; ++i)
        hipLaunchKernelGGL((conv3x3_f32s_kernel<64, false, false, RES, ABL, TERMS>), dim3(grid), dim3(256), LDS_BYTES, 0, a, nitems);
    CK_(hipDeviceSynchronize());
    CK_(hipEventRecord(e0, 0));
    for (int i = 0; i < reps; ++i)
        hipLaunchKernelGGL((conv3x3_f32s_kernel<64, false, false, RES, ABL, TERMS>), dim3(grid), dim3(256), LDS_BYTES, 0, a, nitems);
    CK_(hipEventRecord(e1, 0));
    CK_(hipEventSynchronize(e1));
    CK_(hipGetLastError());
    float ms = 0;
    CK_(hipEventElapsedTime(&ms, e0, e1));
    long long clk = 0;
    if (a.zeros) CK_(hipMemcpy(&clk, a.zeros, 8, hipMemcpyDeviceToHost));
    printf("%-44s %8.1f us", name, ms / reps * 1e3);
    if ((ABL & 256) && a.zeros) printf("   workgroup 0: %lld cycles -> %.2f GHz", clk, clk / (ms / reps * 1e3) * 1e-3);
    printf("\n");
    fflush(stdout);
    return 0;
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 256, reps = 20;
    const size_t act = (size_t)N * 64 * 64 * 96;
    std::vector<float> hx(act);
    unsigned s = 12345;
    for (auto& v : hx) { s = s * 1664525u + 1013904223u; v = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
    std::vector<_Float16> hw((size_t)18 * c3s::WG_UNITS * 8 + 64, (_Float16)0.f);
    for (auto& v : hw) { s = s * 1664525u + 1013904223u; v = (_Float16)((((s >> 8) & 0xffff) / 65536.f - 0.5f) * 0.05f); }
    {                                                 // the image's amax tail: 32 floats
        float tail[32] = {1.f};
        memcpy(&hw[(size_t)18 * c3s::WG_UNITS * 8], tail, sizeof tail);
    }
    std::vector<float> hb(96, 0.01f), hs(96, 0.25f);
    float *x, *y, *r, *b, *sl;
    void* w;
    CK_(hipMalloc(&x, act * 4)); CK_(hipMalloc(&y, act * 4)); CK_(hipMalloc(&r, act * 4));
    CK_(hipMalloc(&w, hw.size() * 2)); CK_(hipMalloc(&b, 96 * 4)); CK_(hipMalloc(&sl, 96 * 4));
    CK_(hipMemcpy(x, hx.data(), act * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(r, hx.data(), act * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(w, hw.data(), hw.size() * 2, hipMemcpyHostToDevice));
    CK_(hipMemcpy(b, hb.data(), 96 * 4, hipMemcpyHostToDevice));
    CK_(hipMemcpy(sl, hs.data(), 96 * 4, hipMemcpyHostToDevice));
    Conv3x3Args a{};
    a.in = x; a.out = y; a.wpack = w; a.bias = b; a.slope = sl; a.res = nullptr; a.pool_partial = nullptr;
    a.N = N; a.H = 64; a.W = 64; a.cout_blocks = 1; a.shuffle = 0;
    void* clk;
    CK_(hipMalloc(&clk, 64));
    a.zeros = clk;
    Conv3x3Args ar = a;
    ar.res = r;
    if (argc > 2) {                                   // counter passes (rocprofv3 --pmc): the library's kernels, few launches
        if (run<0, false>("full", a, 2)) return 1;
        if (run<0, true>("full + residual", ar, 2)) return 1;
        return 0;
    }
    for (int round = 0; round < 3; ++round) {
        printf("-- round %d\n", round);
        if (run<256, false>("full", a, reps)) return 1;
        if (run<256, true>("full + residual", ar, reps)) return 1;
        if (run<256, false, 1>("one term (f16 products)", a, reps)) return 1;
        if (run<256, true, 1>("one term + residual", ar, reps)) return 1;
        if (run<256 + 4096, false, 1>("one term, no raised priority", a, reps)) return 1;
        if (run<256 + 4, false, 1>("one term, no MFMA", a, reps)) return 1;
        if (run<256 + 16, false, 1>("one term, no global loads", a, reps)) return 1;
        if (run<256 + 8192, false, 1>("one term, no output stores", a, reps)) return 1;
        if (run<256 + 8192, false, 3>("three terms, no output stores", a, reps)) return 1;
        if (run<256 + 8192 + 1, false, 1>("one term, no output stores, no pixel staging", a, reps)) return 1;
        if (run<256 + 2048, false, 1>("one term, pixel loads from L2 (image 0)", a, reps)) return 1;
        if (run<256 + 1, false, 1>("one term, no pixel staging", a, reps)) return 1;
        if (run<256 + 8, false, 1>("one term, no fragment LDS reads", a, reps)) return 1;
        if (run<256 + 64 + 2, false, 1>("one term, no barriers", a, reps)) return 1;
        if (run<256 + 1024, false, 1>("one term, vmcnt(0) in a chunk's first group", a, reps)) return 1;
        if (run<256 + 4096, false>("no raised priority in a chunk's last group", a, reps)) return 1;
        if (run<256 + 1, false>("no pixel staging (load, split, LDS write)", a, reps)) return 1;
        if (run<256 + 4, false>("no MFMA", a, reps)) return 1;
        if (run<256 + 8, false>("no fragment LDS reads", a, reps)) return 1;
        if (run<256 + 16, false>("no global loads", a, reps)) return 1;
        if (run<256 + 64 + 2, false>("no barriers", a, reps)) return 1;
        if (run<256 + 8 + 16 + 1, false>("MFMA + barriers only", a, reps)) return 1;
        if (run<256 + 8 + 16 + 1 + 64 + 2, false>("MFMA only", a, reps)) return 1;
    }
    return 0;
}
