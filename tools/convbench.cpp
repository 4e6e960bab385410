// convbench -- development harness (not part of the product): times the bf16 3x3 conv kernel
// variants on random data in ONE process with interleaved rounds, and checks them against
// each other and against a CPU spot check.
//   hipcc -O3 -std=c++17 --offload-arch=gfx950 -o tools/bin/convbench tools/convbench.cpp
//   tools/bin/convbench [N=256] [rounds=5] [launches=20] [mode: 0 trunk, 1 trunk+res, 2 shuffle, 3 hr, 4 pool]
#define SRGEN_C3_STAMPS 1
#include "conv3x3_bf16_v1.hip"
#include "../superresolution_amd/csrc/srgen_conv3x3_bf16.hip"

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
    const int H = mode == 3 ? 128 : 64;
    const int cb = mode == 2 ? 4 : 1;
    const size_t in_elems = (size_t)N * H * H * 96;
    const size_t out_elems = mode == 2 ? in_elems * 4 : in_elems;
    const size_t w_elems = (size_t)cb * 9 * 96 * 96;

    std::mt19937 rng(123);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    std::vector<uint16_t> h_in(in_elems), h_res(in_elems), h_w(w_elems);
    const bool zero_data = getenv("C3_ZERO") != nullptr;      // all-zero operands: clock stays high (power test)
    for (auto& v : h_in) v = zero_data ? 0 : f2bf(U(rng));
    for (auto& v : h_res) v = f2bf(U(rng));
    for (auto& v : h_w) v = zero_data ? 0 : f2bf(0.06f * U(rng));
    std::vector<float> h_bias(96 * cb), h_slope(96 * cb);
    for (auto& v : h_bias) v = 0.1f * U(rng);
    for (auto& v : h_slope) v = 0.25f + 0.2f * U(rng);

    uint16_t *d_in, *d_res, *d_w, *d_out[2];
    float *d_bias, *d_slope, *d_pool[2];
    CK(hipMalloc(&d_in, in_elems * 2));
    CK(hipMalloc(&d_res, in_elems * 2));
    CK(hipMalloc(&d_w, w_elems * 2));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&d_out[i], out_elems * 2));
    for (int i = 0; i < 2; ++i) CK(hipMalloc(&d_pool[i], (size_t)N * 16 * 96 * 4));
    CK(hipMalloc(&d_bias, 96 * cb * 4));
    CK(hipMalloc(&d_slope, 96 * cb * 4));
    CK(hipMemcpy(d_in, h_in.data(), in_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_res, h_res.data(), in_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_w, h_w.data(), w_elems * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_bias, h_bias.data(), 96 * cb * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(d_slope, h_slope.data(), 96 * cb * 4, hipMemcpyHostToDevice));

    auto args = [&](int which) {
        Conv3x3Args a;
        a.in = d_in; a.out = d_out[which]; a.wpack = d_w; a.bias = d_bias;
        a.slope = (mode == 1 || mode == 4) ? nullptr : d_slope;
        a.res = (mode == 1 || mode == 4) ? d_res : nullptr;
        a.pool_partial = mode == 4 ? d_pool[which] : nullptr;
        a.N = N; a.H = H; a.W = H; a.cout_blocks = cb; a.shuffle = mode == 2;
        return a;
    };
    hipStream_t s;
    CK(hipStreamCreate(&s));
    auto run = [&](int which) {
        return which == 0 ? v1::launch_conv3x3_bf16(args(0), s) : launch_conv3x3_bf16(args(1), s);
    };
    for (int i = 0; i < 2; ++i) {
        CK(hipMemsetAsync(d_out[i], 0xff, out_elems * 2, s));
        CK(run(i));
    }
    CK(hipStreamSynchronize(s));

    // ---- correctness: v2 vs v1 (whole tensor), v1 vs CPU (spot check) ---------------------
    std::vector<uint16_t> o0(out_elems), o1(out_elems);
    CK(hipMemcpy(o0.data(), d_out[0], out_elems * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o1.data(), d_out[1], out_elems * 2, hipMemcpyDeviceToHost));
    size_t ndiff = 0;
    double maxd = 0;
    for (size_t i = 0; i < out_elems; ++i) {
        if (o0[i] != o1[i]) ++ndiff;
        maxd = std::max(maxd, (double)fabsf(bf2f(o0[i]) - bf2f(o1[i])));
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
                    acc += (double)bf2f(h_in[(((size_t)n * H + gy) * H + gx) * 96 + ci]) *
                           bf2f(h_w[((((size_t)b * 9 + tap) * 12 + ci / 8) * 96 + co) * 8 + (ci & 7)]);
            }
            double v = acc + h_bias[oc];
            if (!(mode == 1 || mode == 4)) v = v >= 0 ? v : v * h_slope[oc];
            size_t o;
            if (mode == 2) o = (((size_t)n * 2 * H + 2 * y + (b >> 1)) * 2 * H + 2 * x + (b & 1)) * 96 + co;
            else o = (((size_t)n * H + y) * H + x) * 96 + co;
            if (mode == 1 || mode == 4) v += bf2f(h_res[o]);
            for (int k = 0; k < 2; ++k) {
                const double got = bf2f(k ? o1[o] : o0[o]);
                worst = std::max(worst, fabs(got - v) / (fabs(v) + 1.0));
            }
        }
        printf("CPU spot check (4000 samples, both variants): worst rel err %.3g %s\n", worst,
               worst < 8e-3 ? "OK" : "FAIL");
    }

    // ---- timing: interleaved rounds --------------------------------------------------------
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
#ifndef C3_REPEAT
#define C3_REPEAT 1
#endif
    const double flop = 2.0 * N * H * H * 96.0 * 864.0 * cb;
    printf("(v2 main loop repeated %d x: scale its TFLOP/s by that)\n", C3_REPEAT);
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
    // ---- phase stamps of v2 (one launch) -----------------------------------------------------
    {
        const size_t nwg = (size_t)N * (H / 4) * (H / 64) * cb;
        long long* d_st;
        CK(hipMalloc(&d_st, nwg * 8 * 8));
        CK(hipMemset(d_st, 0, nwg * 8 * 8));
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c3::g_stamps), &d_st, sizeof(d_st)));
        CK(run(1));
        CK(hipStreamSynchronize(s));
        std::vector<long long> st(nwg * 8);
        CK(hipMemcpy(st.data(), d_st, nwg * 8 * 8, hipMemcpyDeviceToHost));
        const char* names[5] = {"acc init", "chunk0 (9 taps)", "A chunk1 store", "chunk1 (9 taps)", "epilogue"};
        for (int k = 0; k < 5; ++k) {
            std::vector<long long> d;
            for (size_t i = 0; i < nwg; ++i) d.push_back(st[i * 8 + k + 1] - st[i * 8 + k]);
            std::sort(d.begin(), d.end());
            printf("  %-18s median %6lld  p10 %6lld  p90 %6lld  (s_memtime ticks)\n", names[k], d[d.size() / 2],
                   d[d.size() / 10], d[d.size() * 9 / 10]);
        }
        std::vector<long long> tot;
        long long tmin = st[0], tmax = 0;
        for (size_t i = 0; i < nwg; ++i) {
            tot.push_back(st[i * 8 + 5] - st[i * 8]);
            tmin = std::min(tmin, st[i * 8]);
            tmax = std::max(tmax, st[i * 8 + 5]);
        }
        std::sort(tot.begin(), tot.end());
        printf("  workgroup lifetime median %lld; kernel span %lld ticks; %zu workgroups\n", tot[tot.size() / 2],
               tmax - tmin, nwg);
        {   // per-CU overlap of the main-loop intervals [stamp1, stamp4]
            std::map<long long, std::vector<std::pair<long long, long long>>> cu;
            for (size_t i = 0; i < nwg; ++i) {
                const long long hw = st[i * 8 + 6], xcc = st[i * 8 + 7] & 0xf;
                const long long key = (xcc << 16) | (hw & 0xff00);     // se/sh/cu bits 8..15
                cu[key].push_back({st[i * 8 + 1], st[i * 8 + 4]});
            }
            double sum = 0, uni = 0, both = 0, span = 0;
            for (auto& kv : cu) {
                auto& v = kv.second;
                std::vector<std::pair<long long, int>> ev;
                for (auto& iv : v) { ev.push_back({iv.first, 1}); ev.push_back({iv.second, -1}); sum += iv.second - iv.first; }
                std::sort(ev.begin(), ev.end());
                int depth = 0;
                long long last = ev[0].first;
                for (auto& e : ev) {
                    if (depth >= 1) uni += e.first - last;
                    if (depth >= 2) both += e.first - last;
                    depth += e.second;
                    last = e.first;
                }
                span += ev.back().first - ev[0].first;
            }
            {   // timeline of one CU
                long long key0 = -1, t00 = 0;
                std::vector<std::array<long long, 7>> rows;
                for (size_t i = 0; i < nwg; ++i) {
                    const long long hw = st[i * 8 + 6], xcc = st[i * 8 + 7] & 0xf;
                    const long long key = (xcc << 16) | (hw & 0xff00);
                    if (key0 < 0) { key0 = key; t00 = st[i * 8]; }
                    if (key == key0) rows.push_back({st[i * 8], st[i * 8 + 1], st[i * 8 + 2], st[i * 8 + 3], st[i * 8 + 4], st[i * 8 + 5], (long long)i});
                }
                std::sort(rows.begin(), rows.end());
                for (auto& r : rows)
                    printf("    wg %5lld simd/wave %02llx: start %7lld  main %7lld  c0end %7lld  c1start %7lld  mainend %7lld  end %7lld\n", r[6],
                           st[r[6] * 8 + 6] & 0x3f, r[0] - t00, r[1] - t00, r[2] - t00, r[3] - t00, r[4] - t00, r[5] - t00);
            }
            double dsum[3] = {0, 0, 0};
            int dcnt[3] = {0, 0, 0};
            for (auto& kv : cu) {
                auto& v = kv.second;
                for (size_t i = 0; i < v.size(); ++i) {
                    long long ov = 0;
                    for (size_t k = 0; k < v.size(); ++k)
                        if (k != i) ov += std::max(0LL, std::min(v[i].second, v[k].second) - std::max(v[i].first, v[k].first));
                    const double f = (double)ov / (v[i].second - v[i].first);
                    const int b = f < 0.2 ? 0 : f < 0.8 ? 1 : 2;
                    dsum[b] += v[i].second - v[i].first;
                    dcnt[b]++;
                }
            }
            for (int b = 0; b < 3; ++b)
                printf("  main-loop overlap with the co-resident workgroup %s: %d workgroups, mean main-loop %.0f ticks\n",
                       b == 0 ? "<20%" : b == 1 ? "20-80%" : ">80%", dcnt[b], dcnt[b] ? dsum[b] / dcnt[b] : 0.0);
            printf("  %zu CUs seen; per-CU main-loop time: sum %.0f, union %.0f (>=2 concurrent %.0f), span %.0f ticks (means)\n",
                   cu.size(), sum / cu.size(), uni / cu.size(), both / cu.size(), span / cu.size());
        }
        void* nul = nullptr;
        CK(hipMemcpyToSymbol(HIP_SYMBOL(c3::g_stamps), &nul, sizeof(nul)));
    }
    return 0;
}
