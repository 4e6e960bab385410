// convbench -- development harness (not part of the product): times the bf16 3x3 conv kernel
// variants on random data in ONE process with interleaved rounds, and checks them against
// each other and against a CPU spot check.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/convbench tools/convbench.cpp
//   tools/bin/convbench [N=256] [rounds=5] [launches=20] [mode: 0 trunk, 1 trunk+res, 2 shuffle, 3 hr, 4 pool]
#define SRGEN_C3W_STAMPS 1
#include "../superresolution_amd/csrc/srgen_conv3x3_f32.hip"
#include "../superresolution_amd/csrc/srgen_conv3x3_f32_wino.hip"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <array>
#include <random>
#include <vector>

using namespace srgen;

#define CK(x)                                                                              \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            fprintf(stderr, "%s: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__);    \
            exit(1);                                                                       \
        }                                                                                  \
    } while (0)

static uint16_t f2bf(float f)
{
    uint32_t u;
    __builtin_memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t b)
{
    uint32_t u = (uint32_t)b << 16;
    float f;
    __builtin_memcpy(&f, &u, 4);
    return f;
}

int main(int argc, char** argv)
{
    const int N = argc > 1 ? atoi(argv[1]) : 256;
    const int rounds = argc > 2 ? atoi(argv[2]) : 5;
    const int launches = argc > 3 ? atoi(argv[3]) : 20;
    const int mode = argc > 4 ? atoi(argv[4]) : 1;
    const int variant = argc > 5 ? atoi(argv[5]) : 1;      // "v2" = 0: direct again, 1: Winograd (LDS), 2: Winograd (registers)
    const bool use_wino = variant == 1;
    const int H = mode == 3 ? 128 : 64;
    const int cb = mode == 2 ? 4 : 1;
    const size_t in_elems = (size_t)N * H * H * 96;
    const size_t out_elems = mode == 2 ? in_elems * 4 : in_elems;
    const size_t w_elems = (size_t)cb * 9 * 96 * 96;

    std::mt19937 rng(123);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<float> h_in(in_elems), h_res(in_elems), h_w(w_elems);
    const bool zero_data = getenv("C3_ZERO") != nullptr;      // all-zero operands: clock stays high (power test)
    for (auto& v : h_in) v = zero_data ? 0.f : U(rng);
    for (auto& v : h_res) v = U(rng);
    for (auto& v : h_w) v = zero_data ? 0.f : 0.06f * U(rng);
    std::vector<float> h_bias(96 * cb), h_slope(96 * cb);
    for (auto& v : h_bias) v = 0.1f * U(rng);
    for (auto& v : h_slope) v = 0.25f + 0.2f * U(rng);

    float *d_in, *d_res, *d_w, *d_out[2];
    float *d_bias, *d_slope, *d_pool[2];
    CK(hipMalloc(&d_in, in_elems * 4));
    CK(hipMalloc(&d_res, in_elems * 4));
    CK(hipMalloc(&d_w, w_elems * 4));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&d_out[i], out_elems * 4));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&d_pool[i], (size_t)N * 16 * 96 * 4));
    CK(hipMalloc(&d_bias, 96 * cb * 4));
    CK(hipMalloc(&d_slope, 96 * cb * 4));
    CK(hipMemcpy(d_in, h_in.data(), in_elems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_res, h_res.data(), in_elems * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), w_elems * 4, hipMemcpyHostToDevice));
    // Winograd weights U = G g Gt, float4 units [cb][nu 4][chunk 6][xi 4][ci/4 4][co 96]
    std::vector<float> h_u((size_t)cb * 16 * 96 * 96);
    {
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        for (int b = 0; b < cb; ++b)
            for (int co = 0; co < 96; ++co)
                for (int ci = 0; ci < 96; ++ci) {
                    double g[3][3];
                    for (int tap = 0; tap < 9; ++tap)
                        g[tap / 3][tap % 3] = h_w[((((size_t)b * 9 + tap) * 24 + ci / 4) * 96 + co) * 4 + (ci & 3)];
                    for (int xi = 0; xi < 4; ++xi)
                        for (int nu = 0; nu < 4; ++nu) {
                            double u = 0;
                            for (int a = 0; a < 3; ++a)
                                for (int c = 0; c < 3; ++c) u += G[xi][a] * g[a][c] * G[nu][c];
                            h_u[((((((size_t)b * 4 + nu) * 6 + ci / 16) * 4 + xi) * 4 + (ci / 4) % 4) * 96 + co) * 4 + (ci & 3)] = (float)u;
                        }
                }
    }
    // register-Winograd pack: float4 units [nu 4][chunk 6][xi 4][s 6][lane 64]; lane (m = lane & 15, kq = lane >> 4)
    // holds U[xi][nu][co = 24*(m >> 2) + 4*s + (m & 3)][ci = 16*chunk + 4*kq + j]
    std::vector<float> h_u2((size_t)16 * 96 * 96);
    if (cb == 1) {
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        for (int nu = 0; nu < 4; ++nu)
            for (int chunk = 0; chunk < 6; ++chunk)
                for (int xi = 0; xi < 4; ++xi)
                    for (int sx = 0; sx < 6; ++sx)
                        for (int lane = 0; lane < 64; ++lane)
                            for (int j = 0; j < 4; ++j) {
                                const int m = lane & 15, kq = lane >> 4;
                                const int co = 24 * (m >> 2) + 4 * sx + (m & 3), ci = 16 * chunk + 4 * kq + j;
                                double u = 0;
                                for (int a = 0; a < 3; ++a)
                                    for (int c = 0; c < 3; ++c)
                                        u += G[xi][a] * G[nu][c] *
                                             h_w[((((size_t)(a * 3 + c)) * 24 + ci / 4) * 96 + co) * 4 + (ci & 3)];
                                h_u2[((((((size_t)nu * 6 + chunk) * 4 + xi) * 6 + sx) * 64 + lane) * 4) + j] = (float)u;
                            }
    }
    float* d_u2;
    CK(hipMalloc(&d_u2, h_u2.size() * 4));
    CK(hipMemcpy(d_u2, h_u2.data(), h_u2.size() * 4, hipMemcpyHostToDevice));
    float *d_u, *d_zeros;
    CK(hipMalloc(&d_u, h_u.size() * 4));
    CK(hipMemcpy(d_u, h_u.data(), h_u.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_zeros, kWinoZeroBytes));
    CK(hipMemset(d_zeros, 0, kWinoZeroBytes));
    CK(hipMemcpy(d_bias, h_bias.data(), 96 * cb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_slope, h_slope.data(), 96 * cb * 4, hipMemcpyHostToDevice));

    auto args = [&](int which) {
        Conv3x3Args a;
        a.in = d_in; a.out = d_out[which]; a.wpack = d_w; a.bias = d_bias;
        a.slope = (mode == 1 || mode == 4) ? nullptr : d_slope;
        a.res = (mode == 1 || mode == 4) ? d_res : nullptr;
        a.pool_partial = mode == 4 ? d_pool[which] : nullptr;
        a.N = N; a.H = H; a.W = H; a.cout_blocks = cb; a.shuffle = mode == 2;
        a.zeros = d_zeros;
        if (which == 1 && use_wino) a.wpack = d_u;
        if (which == 1 && variant == 2) a.wpack = d_u2;
        return a;
    };
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto run = [&](int which) {
        if (which == 1 && variant == 2) return launch_conv3x3_f32_wreg(args(1), s);
        return which == 0 ? launch_conv3x3_f32(args(0), s) : (use_wino ? launch_conv3x3_f32_wino(args(1), s) : launch_conv3x3_f32(args(1), s));
    };
    for (int i = 0; i < 2; ++i) {
        CK(hipMemsetAsync(d_out[i], 0xff, out_elems * 4, s));
        CK(run(i));
    }
    CK(hipStreamSynchronize(s));

    // ---- correctness: v2 vs v1 (whole tensor), v1 vs CPU (spot check) ---------------------
    std::vector<float> o0(out_elems), o1(out_elems);
    CK(hipMemcpy(o0.data(), d_out[0], out_elems * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), d_out[1], out_elems * 4, hipMemcpyDeviceToHost));
    size_t ndiff = 0;
    double maxd = 0;
    for (size_t i = 0; i < out_elems; ++i) {
        if (o0[i] != o1[i]) ++ndiff;
        maxd = std::max(maxd, (double)fabsf(o0[i] - o1[i]));
    }
    printf("v2 vs v1: %zu / %zu elements differ, max abs diff %.3g\n", ndiff, out_elems, maxd);
    if (mode == 4) {
        std::vector<float> p0((size_t)N * 16 * 96), p1(p0.size());
        CK(hipMemcpy(p0.data(), d_pool[0], p0.size() * 4, hipMemcpyDeviceToHost));
        CK(hipMemcpy(p1.data(), d_pool[1], p1.size() * 4, hipMemcpyDeviceToHost));
        double md = 0;
        for (size_t i = 0; i < p0.size(); ++i) md = std::max(md, (double)fabsf(p0[i] - p1[i]));
        printf("pool partials max abs diff %.3g\n", md);
    }
    {
        std::uniform_int_distribution<int> Rn(0, N - 1), Rh(0, H - 1), Rc(0, 96 * cb - 1);
        double worst = 0;
        for (int t = 0; t < 4000; ++t) {
            const int n = Rn(rng), y = Rh(rng), x = Rh(rng), oc = Rc(rng);
            const int b = oc / 96, co = oc % 96;
            double acc = 0;
            for (int tap = 0; tap < 9; ++tap) {
                const int gy = y + tap / 3 - 1, gx = x + tap % 3 - 1;
                if (gy < 0 || gy >= H || gx < 0 || gx >= H) continue;
                for (int ci = 0; ci < 96; ++ci)
                    acc += (double)h_in[(((size_t)n * H + gy) * H + gx) * 96 + ci] *
                           h_w[((((size_t)b * 9 + tap) * 24 + ci / 4) * 96 + co) * 4 + (ci & 3)];
            }
            double v = acc + h_bias[oc];
            if (!(mode == 1 || mode == 4)) v = v >= 0 ? v : v * h_slope[oc];
            size_t o;
            if (mode == 2) o = (((size_t)n * 2 * H + 2 * y + (b >> 1)) * 2 * H + 2 * x + (b & 1)) * 96 + co;
            else o = (((size_t)n * H + y) * H + x) * 96 + co;
            if (mode == 1 || mode == 4) v += h_res[o];
            for (int k = 0; k < 2; ++k) {
                const double got = k ? o1[o] : o0[o];
                worst = std::max(worst, fabs(got - v) / (fabs(v) + 1.0));
            }
        }
        printf("CPU spot check (4000 samples, both variants): worst rel err %.3g %s\n", worst,
               worst < 2e-5 ? "OK" : "FAIL");
    }

    // ---- timing: interleaved rounds --------------------------------------------------------
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
#ifndef C3_REPEAT
#define C3_REPEAT 1
#endif
    const double flop = 2.0 * N * H * H * 96.0 * 864.0 * cb;
    
    std::vector<double> best(2, 1e30), med[2];
    for (int r = 0; r < rounds; ++r)
        for (int which = 0; which < 2; ++which) {
            CK(hipEventRecord(e0, s));
            for (int i = 0; i < launches; ++i) CK(run(which));
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms;
            CK(hipEventElapsedTime(&ms, e0, e1));
            med[which].push_back(ms / launches);
        }
    for (int which = 0; which < 2; ++which) {
        std::sort(med[which].begin(), med[which].end());
        const double m = med[which][med[which].size() / 2], b = med[which][0];
        printf("v%d: median %.4f ms (%.1f TFLOP/s)  min %.4f ms (%.1f TFLOP/s)\n", which + 1, m,
               flop / m * 1e-9, b, flop / b * 1e-9);
    }
    if (use_wino) {   // phase stamps of the Winograd kernel (one launch)
        const size_t nit = (size_t)N * (H / 4) * (H / 64) * cb;
        long long* d_st;
        CK(hipMalloc(&d_st, nit * 16 * 8));
        CK(hipMemset(d_st, 0, nit * 16 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c3w::g_wstamps), &d_st, sizeof(d_st)));
        CK(run(1));
        CK(hipStreamSynchronize(s));
        std::vector<long long> st(nit * 16);
        CK(hipMemcpy(st.data(), d_st, nit * 16 * 8, hipMemcpyDeviceToHost));
        const char* names[9] = {"nu0 loop", "nu0 fold", "nu1 loop", "nu1 fold", "nu2 loop", "nu2 fold", "nu3 loop", "nu3 fold", "epilogue"};
        for (int k = 0; k < 9; ++k) {
            std::vector<long long> d;
            for (size_t i = 0; i < nit; ++i) d.push_back(st[i * 16 + k + 1] - st[i * 16 + k]);
            std::sort(d.begin(), d.end());
            printf("  %-10s median %7lld  p10 %7lld  p90 %7lld ticks\n", names[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
        }
        {
            const char* nm[5] = {"top->stage+loadu issued", "xi0 mfma issue", "xi1+xi2 (to barrier)", "barrier", "xi3 + combine"};
            for (int k = 0; k < 5; ++k) {
                std::vector<long long> d;
                for (size_t i = 0; i < nit; ++i) d.push_back(st[i * 16 + 11 + k] - st[i * 16 + 10 + k]);
                std::sort(d.begin(), d.end());
                printf("  chunk(nu1,c2) %-26s median %6lld  p10 %6lld  p90 %6lld\n", nm[k], d[d.size() / 2], d[d.size() / 10], d[d.size() * 9 / 10]);
            }
        }
        void* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c3w::g_wstamps), &nul, sizeof(nul)));
    }
    return 0;
}
