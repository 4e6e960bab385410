// tools/conv3x3_bf16_wide.hip -- EXPERIMENT (round 3), not built into libsrgen.so: the bf16 3x3 conv 96 -> 96 (x cout_blocks)
// with a 96 co x 128 px wave tile, one workgroup per CU.  Parity-correct (run inside the library it passed every bf16 test
// of tests/test_gpu_generator.py, bit-identical to srgen_conv3x3_bf16.hip), but not faster: at batch 256 its main loop
// alone runs the trunk conv in 121 us (1.44 PFLOP/s, against 0.18-0.21 ms for the whole library kernel), and the epilogue
// (45 us), the halo staging (30 us) and the residual (55 us), which a lone wave per SIMD cannot hide behind anything, put
// it at 184 us (PReLU) / 240 us (residual) against 182 / 209 us.  Harness: tools/bench_bf16w.hip; numbers:
// profiles/r03_experiments/bf16_wide_tile_ablations.txt; discussion: DESIGN.md section 4.
//
// Reference sites: as srgen_conv3x3_bf16.hip (Generator.py:34,37,62,7,69 with BatchNorm folded; PReLU / residual /
// PixelShuffle in the epilogue).
//
// Why a second kernel.  The 64-pixel wave tile of srgen_conv3x3_bf16.hip reads five 16-byte LDS fragments per six MFMAs:
// at the bf16 MFMA rate that is 107 B/clk/CU of LDS reads, more than the LDS delivers beside everything else, and the
// measured energy per MFMA was twice the bare loop's (DESIGN.md section 4, round-2 measurements).  Here a wave owns TWO
// rows of 64 pixels x 96 output channels (4 x 3 tiles of 32 x 32 = 192 accumulator registers): a tap's three weight
// fragments serve twelve MFMAs instead of six -- seven fragment reads per twelve MFMAs, 30 % less LDS traffic per FLOP.
// 192 accumulators + double-buffered fragments + staging registers need the whole register file: ONE workgroup of four
// waves per CU (launch bounds 256 x 1, 512 registers per lane), so nothing hides a stall but the wave's own software
// pipeline -- fragments of step s + 1 are read in the gaps between the MFMAs of step s, weight slabs arrive by LDS-DMA two
// phases ahead, the next chunk's / item's halo tile is loaded into registers under the MFMAs and only written to LDS
// between two barriers.
//
// Work item = 8 rows x 64 pixels x 96 co (x cout block); K walk, weight slabs, halo-tile layout (112-byte pixel pitch),
// register epilogue and XCD-contiguous persistent item walk are those of srgen_conv3x3_bf16.hip; K is visited in the same
// order, so the two kernels produce bit-identical outputs and per-tile pool sums (tests: an image of a large batch equals
// the same image run alone, which takes the other kernel).
#include "../superresolution_amd/csrc/srgen_internal.h"

#include <type_traits>

namespace srgen {
namespace c3w {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
__device__ __forceinline__ void wait_vm0() { __builtin_amdgcn_s_waitcnt(0x0F70); }      // vmcnt(0) only

__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}
__device__ __forceinline__ void swap32(unsigned& x, unsigned& y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

constexpr int RPW = 2;                            // rows per wave
constexpr int TH = 4 * RPW, TW = 64, HH = TH + 2, HWP = 66;
constexpr int CK = 48;                            // ci per chunk
constexpr int PS = 112;                           // halo-tile pixel pitch, bytes (96 + 16 pad: odd multiple of 16)
constexpr int A_BYTES = HH * HWP * PS;            // 73,920
constexpr int SLAB = 6 * 96 * 16;                 // 9,216: one (tap, chunk) weight slab
constexpr int NRING = 6, AHEAD = 4;               // weight-slab ring; a slab is requested AHEAD taps before its tap
constexpr int BS_OFF = A_BYTES + NRING * SLAB;    // bias[4][96], slope[4][96] (fp32)
constexpr int LDS_BYTES = BS_OFF + 2 * 4 * 96 * 4;   // 132,288: one workgroup per CU
constexpr int NPX = HH * HWP;                     // 660 halo pixels
constexpr int ROUNDS = (NPX + 41) / 42;           // staging rounds of 42 pixels x 6 units (252 threads): 16
constexpr int NM = 2 * RPW;                       // 32-pixel tiles per wave: 4

struct Frags { bf16x8 a[NM], b[3]; };

struct Item { int n, ty, tx, cb; };

// ABL: timing ablations for tools/bench_bf16w.hip only (wrong results); the library instantiates ABL = 0
template <int HW, bool SHUFFLE, bool POOL, bool RES, bool ACT, int ABL = 0>
__global__ __launch_bounds__(256, 1) void conv3x3_bf16w_kernel(Conv3x3Args a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int tiles_x = HW / TW, tiles_y = HW / TH;
    constexpr int CB = SHUFFLE ? 4 : 1;

    const int per_xcd = nitems >> 3;                     // nitems is a multiple of 8
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int item_end = (xcd + 1) * per_xcd;
    int item = xcd * per_xcd + ((int)blockIdx.x >> 3);   // < item_end: grid <= nitems
    auto decode = [&](int it) {
        Item r;
        r.cb = it % CB;
        const int tile = it / CB;
        r.tx = tile % tiles_x;
        r.ty = (tile / tiles_x) % tiles_y;
        r.n = tile / (tiles_x * tiles_y);
        return r;
    };

    char* const ldsA = lds;
    char* const ldsB = lds + A_BYTES;
    float* const ldsBias = (float*)(lds + BS_OFF);
    float* const ldsSlope = ldsBias + 4 * 96;

    const int tid16 = tid * 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // ---- halo-tile staging, global -> registers -> LDS: thread t moves the 16-byte unit q = t % 6 of pixel
    // P = t / 6 + 42 it of the 10 x 66 tile in round it (threads 252..255 idle); everything that depends on (t, it) only
    // is a bit mask over it, formed once (srgen_conv3x3_bf16.hip) ----
    u32x4 stage[ROUNDS];
    const int p0 = tid / 6, q6 = tid - p0 * 6;
    const int a_g0 = p0 * kC + q6 * 8;
    char* const a_l0 = lds + p0 * PS + q6 * 16;
    unsigned m_wrap = 0, m_row0 = 0, m_rowL = 0, m_unused = 0, m_col0 = 0, m_colL = 0;
#pragma unroll
    for (int it = 0; it < ROUNDS; ++it) {
        const int rb = (42 * it) / HWP, rem = (42 * it) % HWP;
        const bool wrap = p0 + rem >= HWP;
        const int r = rb + (wrap ? 1 : 0), cc = p0 + rem - (wrap ? HWP : 0);
        const bool used = tid < 252 && p0 + 42 * it < NPX;
        m_wrap |= (wrap ? 1u : 0u) << it;
        m_row0 |= (r == 0 ? 1u : 0u) << it;
        m_rowL |= (r == HH - 1 ? 1u : 0u) << it;
        m_unused |= (used ? 0u : 1u) << it;
        m_col0 |= (cc == 0 ? 1u : 0u) << it;
        m_colL |= (cc == HWP - 1 ? 1u : 0u) << it;
    }
    // Every round ALWAYS issues its load (a pixel outside the image reads the image's first pixel instead and is zeroed when
    // the round is written to LDS): the counted vmcnt waits below rely on the number of loads a wave has in flight.
    unsigned bad_stage = 0;                           // rounds of the tile in `stage` that lie outside the image
    auto load_a = [&](const Item& t, int c) {
        const __bf16* in = (const __bf16*)a.in + (size_t)t.n * HW * HW * kC + c * CK;
        const int x0 = t.tx * TW, y0 = t.ty * TH;
        unsigned bad = m_unused;
        if (x0 == 0) bad |= m_col0;
        if (x0 + TW == HW) bad |= m_colL;
        if (y0 == 0) bad |= m_row0;
        if (y0 + TH == HW) bad |= m_rowL;
        bad_stage = bad;
        const int origin = ((y0 - 1) * HW + x0 - 1) * kC + a_g0;       // element offset of the tile origin (may lie outside)
#pragma unroll
        for (int it = 0; it < ROUNDS; ++it) {
            const int rb = (42 * it) / HWP, rem = (42 * it) % HWP;
            const int off = origin + (rb * HW + rem) * kC + (int)((m_wrap >> it) & 1u) * ((HW - HWP) * kC);
            stage[it] = *(const u32x4*)(in + (((bad >> it) & 1u) ? q6 * 8 : off));
        }
    };
    auto store_a = [&]() {
        if (tid < 252) {
#pragma unroll
            for (int it = 0; it < ROUNDS; ++it)
                if (it < ROUNDS - 1 || p0 + 42 * (ROUNDS - 1) < NPX) {
                    u32x4 v = stage[it];
                    if ((bad_stage >> it) & 1u) v = u32x4{0u, 0u, 0u, 0u};
                    *(u32x4*)(a_l0 + it * 42 * PS) = v;
                }
        }
    };
    // async slab copy, 576 x 16 B, slab j = (tap j % 9, chunk j / 9) of cout block cb into ring slot j % NRING (18 slabs
    // per item, NRING divides 18: the slot sequence continues across items); nb = 3 (wave 0) or 2 copy instructions
    auto load_b = [&](int cb, int j) {
        const int tap = j % 9, c = j / 9;
        const char* src = (const char*)a.wpack + ((size_t)((cb * 9 + tap) * 12 + 6 * c) * 96) * 16;
        const unsigned dst = lds_base + A_BYTES + (j % NRING) * SLAB + w * 1024;
        glds16_raw(src, (unsigned)tid16, dst);
        glds16_raw(src + 4096, (unsigned)tid16, dst + 4096);
        if (w == 0) glds16_raw(src + 8192, (unsigned)tid16, dst + 8192);
    };
    // residual of this lane's outputs: tile m = 2 rr + mx is pixel (row RPW w + rr, column 32 mx + p); per (m, nt) two
    // 16-byte pieces P = 0, 1 at channels 32 nt + 16 P + 8 h
    u32x4 rres[NM * 6];
    auto load_res = [&](const Item& t, int rr) {      // the 12 pieces of the wave's row rr (tiles m = 2 rr, 2 rr + 1)
        const int lv = opaque(lane);
        const __bf16* res = (const __bf16*)a.res +
                            (((size_t)t.n * HW + t.ty * TH + RPW * w + rr) * HW + t.tx * TW + (lv & 31)) * kC + 8 * (lv >> 5);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int mx = i / 6, nt = (i % 6) / 2, P = i % 2;
            rres[rr * 12 + i] = *(const u32x4*)(res + (size_t)(32 * mx) * kC + 32 * nt + 16 * P);
        }
    };

    // ---- counted waits.  gfx950 retires a wave's vector loads, LDS-DMA copies and stores through ONE in-order counter, so
    // "slab J has landed" = "at most as many operations are outstanding as this wave issued after J's copy".  Per tap L
    // (0..17 over both chunks) a wave issues, in order: the copy of slab L + AHEAD (nb instructions), then LOADS(L) loads
    // (chunk 1's halo tile at L = 2, the residual of the wave's first row at L = 15, the NEXT item's halo chunk 0 at
    // L = 16: the staging registers are free from the chunk boundary on); between tap 17 and the next item's tap 0 the
    // epilogue issues the residual loads of the second row (consumed there, so complete) and then EPI = 24 stores.  At the
    // end of tap L slab L + 2 must have landed (its first fragments are read under tap L + 1's last step); it was
    // requested at tap L - 2.  Nothing older than what is needed is ever waited for, so halo loads get three taps
    // (~3,500 cycles) and the epilogue's stores two taps of cover. ----
    constexpr int EPI = NM * 6;
    auto loads_at = [](int L) { return (L == 2 || L == 16 ? ROUNDS : 0) + (RES && L == 15 ? 12 : 0); };
    auto wait_vm = [&](int n) {                       // n: a compile-time constant after unrolling, one of the cases
        switch (n) {
#define C3W_CASE(v) case v: __builtin_amdgcn_s_waitcnt(((v) & 15) | (((v) >> 4) << 14) | 0x0F70); break;
            C3W_CASE(4) C3W_CASE(6) C3W_CASE(16) C3W_CASE(18) C3W_CASE(20) C3W_CASE(22) C3W_CASE(28) C3W_CASE(30)
            C3W_CASE(32) C3W_CASE(34) C3W_CASE(44) C3W_CASE(46)
#undef C3W_CASE
            default: wait_vm0(); break;
        }
    };
    auto lds_barrier = [&]() {                        // LDS traffic only: __syncthreads() would also wait vmcnt(0)
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };

    Item cur = decode(item);
    load_a(cur, 0);
    load_b(cur.cb, 0);
    load_b(cur.cb, 1);
    load_b(cur.cb, 2);
    load_b(cur.cb, 3);
    for (int i = tid; i < CB * 96; i += 256) {
        ldsBias[i] = a.bias[i];
        ldsSlope[i] = ACT ? a.slope[i] : 1.f;
    }
    store_a();
    wait_vm0();                                   // slabs 0 .. 3 (asm copies: not known to the compiler)
    __syncthreads();

    Frags f0, f1;
    const char* const Aw = ldsA + ((RPW * w) * HWP + p) * PS + h * 16;
    const char* const Bw = ldsB + (h * 96 + p) * 16;

    // fragment addresses of k16-step g of a tap whose halo offset / slab are Ab / Bb
    auto frag_a = [&](const char* Ab, int g, int m) { return *(const bf16x8*)(Ab + ((m >> 1) * HWP + 32 * (m & 1)) * PS + g * 32); };
    auto frag_b = [&](const char* Bb, int g, int nt) { return *(const bf16x8*)(Bb + (2 * g * 96 + 32 * nt) * 16); };

#pragma unroll 1
    for (;;) {
        const int nxt = item + wg_per_xcd;
        const bool has_next = nxt < item_end;
        const Item nx = decode(has_next ? nxt : item);

        // accumulators start at the bias (broadcast LDS reads: the address depends on h only)
        f32x16 acc[NM][3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                const f32x4 b4 = *(const f32x4*)(ldsBias + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                for (int m = 0; m < NM; ++m)
#pragma unroll
                    for (int e = 0; e < 4; ++e) acc[m][nt][4 * G + e] = b4[e];
            }

        // One k16-step = 12 MFMAs (nt-major: a weight fragment serves four pixel tiles) with the seven fragment reads of
        // the NEXT step one each in the gaps behind its first seven MFMAs, in the order the next step uses them (b0, a0..a3
        // for its first four MFMAs, b1 for the fifth, b2 for the ninth): alone in a gap a ds_read_b128 is free.
        auto mma_step_il = [&](const Frags& f, Frags& n, const char* Ab, const char* Bb, int g, bool next) {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int m = 0; m < NM; ++m) {
                    if (!(ABL & 2)) acc[m][nt] = MFMA_BF16(f.b[nt], f.a[m], acc[m][nt]);
                    const int k = nt * NM + m;
                    if (next && !(ABL & 16)) {
                        if (k == 0) n.b[0] = frag_b(Bb, g, 0);
                        else if (k <= NM) n.a[k - 1] = frag_a(Ab, g, k - 1);
                        else if (k == NM + 1) n.b[1] = frag_b(Bb, g, 1);
                        else if (k == NM + 2) n.b[2] = frag_b(Bb, g, 2);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        };

        // one ci-chunk = 9 phases (taps) x 3 k16-steps, fully unrolled; C is a compile-time chunk id so that the staging
        // registers and the residual registers have disjoint lifetimes
        auto run_chunk = [&](auto C) {
            constexpr int c = decltype(C)::value;
            {
                const char* Bb = Bw + ((c * 9) % NRING) * SLAB;
#pragma unroll
                for (int m = 0; m < NM; ++m) f0.a[m] = frag_a(Aw, 0, m);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) f0.b[nt] = frag_b(Bb, 0, nt);
            }
#pragma unroll
            for (int s = 0; s < 27; ++s) {
                const int tap = s / 3, g = s % 3;
                const int j = c * 9 + tap;
                if (g == 0) {
                    if (j + AHEAD < 18) load_b(cur.cb, j + AHEAD);
                    else load_b(nx.cb, j + AHEAD - 18);            // the next item's first slabs (nx = cur behind the last item)
                    if (j == 2 && !(ABL & 4)) load_a(cur, 1);
                    if (RES && j == 15) load_res(cur, 0);
                    if (j == 16 && !(ABL & 4)) load_a(nx, 0);       // (nx = cur behind the last item: loaded, not used)
                }
                {
                    const int tap1 = (s + 1) / 3, g1 = (s + 1) % 3;
                    const int dy = tap1 / 3, dx = tap1 % 3;
                    const char* Ab = Aw + (dy * HWP + dx) * PS;
                    const char* Bb = Bw + ((c * 9 + tap1) % NRING) * SLAB;
                    __builtin_amdgcn_sched_barrier(0);
                    if (s & 1) mma_step_il(f1, f0, Ab, Bb, g1, s + 1 < 27);
                    else mma_step_il(f0, f1, Ab, Bb, g1, s + 1 < 27);
                }
                if (g == 2) {                      // slab j + 2 landed (counted, see above); slab j is free behind the barrier
                    asm volatile("" ::: "memory");
                    const int younger = loads_at((j + 16) % 18) + loads_at((j + 17) % 18) + loads_at(j) + (j < 2 ? EPI : 0);
                    if (ABL & 32) wait_vm0();
                    else if (w == 0) wait_vm(younger + 6);
                    else wait_vm(younger + 4);
                    if (!(ABL & 8)) __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        };
        run_chunk(std::integral_constant<int, 0>{});
        if (!(ABL & 4)) store_a();                // every wave is past its last read of chunk 0
        lds_barrier();
        run_chunk(std::integral_constant<int, 1>{});

        // ---- epilogue (registers only): PReLU -> (+ residual) -> bf16 -> two permlane32 swaps per 16-byte piece -> NHWC ----
        const int y = cur.ty * TH + RPW * w, x0 = cur.tx * TW;
        __bf16* out = (__bf16*)a.out;
        constexpr int PXS = SHUFFLE ? 2 * kC : kC;            // element stride between this wave's pixels
        constexpr size_t ROWS_STRIDE = SHUFFLE ? (size_t)2 * (2 * HW) * kC : (size_t)HW * kC;      // ... and between its rows
        if (SHUFFLE)
            out += (((size_t)cur.n * 2 * HW + 2 * y + (cur.cb >> 1)) * 2 * HW + 2 * (x0 + p) + (cur.cb & 1)) * kC + 8 * h;
        else
            out += (((size_t)cur.n * HW + y) * HW + x0 + p) * kC + 8 * h;
        if (RES) load_res(cur, 1);                 // the second row's residual: in flight while the first row is finished
        // One pass per 32 x 32 tile, 16 accumulators at a time (a pass over ALL tiles per stage -- or a run-time `if` around
        // one -- makes the compiler move all 192 accumulators into vector registers at once, and spill).
        float psum[RPW][3][16];
#pragma unroll
        for (int m = 0; m < NM; ++m) {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                f32x16 c = acc[m][nt];
                if ((ABL & 1) && c[0] != 12345.678f) continue;
                if (ACT) {
#pragma unroll
                    for (int G = 0; G < 4; ++G) {
                        const f32x4 s4 = *(const f32x4*)(ldsSlope + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) c[4 * G + e] = c[4 * G + e] >= 0.f ? c[4 * G + e] : c[4 * G + e] * s4[e];
                    }
                }
                if (RES) {
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        const u32x4 r = rres[(m * 3 + nt) * 2 + P];
                        unsigned r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                        swap32(r0, r2);           // loaded piece = [X0 X1 Y0 Y1] in the stored (swapped) arrangement: undo it
                        swap32(r1, r3);
                        c[8 * P + 0] += bf16_lo(r0); c[8 * P + 1] += bf16_hi(r0);
                        c[8 * P + 2] += bf16_lo(r1); c[8 * P + 3] += bf16_hi(r1);
                        c[8 * P + 4] += bf16_lo(r2); c[8 * P + 5] += bf16_hi(r2);
                        c[8 * P + 6] += bf16_lo(r3); c[8 * P + 7] += bf16_hi(r3);
                    }
                }
                if (POOL) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) psum[m >> 1][nt][q] = (m & 1) == 0 ? c[q] : psum[m >> 1][nt][q] + c[q];
                }
#pragma unroll
                for (int P = 0; P < 2; ++P) {
                    unsigned x0d = pack_bf16(c[8 * P + 0], c[8 * P + 1]), x1d = pack_bf16(c[8 * P + 2], c[8 * P + 3]);
                    unsigned y0d = pack_bf16(c[8 * P + 4], c[8 * P + 5]), y1d = pack_bf16(c[8 * P + 6], c[8 * P + 7]);
                    swap32(x0d, y0d);
                    swap32(x1d, y1d);
                    const u32x4 ov = {x0d, x1d, y0d, y1d};
                    *(u32x4*)(out + (size_t)(m >> 1) * ROWS_STRIDE + (size_t)(32 * (m & 1)) * PXS + 32 * nt + 16 * P) = ov;
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (POOL) {
            // Per-tile channel sums (ChannelAttention's pool) in the order of srgen_conv3x3_bf16.hip: every ROW is summed by
            // the halving butterfly over its 64 pixels, then four consecutive rows are added as (r0 + r1) + (r2 + r3) -- the
            // 16 partial sums per image that kernel's 4-row tiles leave, bit for bit.  (Slab ring slot 2 is free here.)
            float* P = (float*)(ldsB + 4 * SLAB);      // slot of slab 16: read out, and not requested again before tap 0
#pragma unroll
            for (int rr = 0; rr < RPW; ++rr)
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    float (&t)[16] = psum[rr][nt];
#pragma unroll
                    for (int k = 8, b = 1; k >= 1; k >>= 1, b <<= 1)
#pragma unroll
                        for (int i = 0; i < k; ++i) {
                            const bool up = (lane & b) != 0;
                            const float send = up ? t[i] : t[i + k];
                            const float keep = up ? t[i + k] : t[i];
                            t[i] = keep + __shfl_xor(send, b);
                        }
                    t[0] += __shfl_xor(t[0], 16);
                    if ((lane & 16) == 0) {
                        const int q = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
                        P[(RPW * w + rr) * kC + 32 * nt + (q & 3) + 8 * (q >> 2) + 4 * h] = t[0];
                    }
                }
            lds_barrier();
            if (tid < 2 * kC) {
                const int half = tid / kC, ch = tid - half * kC;
                const float* Q = P + 4 * half * kC + ch;
                const float s = (Q[0] + Q[kC]) + (Q[2 * kC] + Q[3 * kC]);
                // 16 partials per 64 x 64 image: index = 4-row tile (POOL runs at HW = 64, one tile per row band)
                a.pool_partial[((size_t)cur.n * (2 * tiles_y * tiles_x) + (2 * cur.ty + half) * tiles_x + cur.tx) * kC + ch] = s;
            }
        }
        if (!has_next) break;
        if (ABL & 4) { lds_barrier(); item = nxt; cur = nx; continue; }
        // the next item's halo chunk 0 -> LDS.  No vmcnt(0) here: the compiler waits for the staging loads it stores (the
        // slab copies of the next item's first taps are older than those), and the 24 stores just issued stay in flight
        store_a();
        lds_barrier();
        item = nxt;
        cur = nx;
    }
    wait_vm0();                                   // (slab copies requested behind the last item)
}

template <int HW, bool SHUFFLE, bool POOL, bool RES, bool ACT>
hipError_t launch_act(const Conv3x3Args& a, hipStream_t s, int cus)
{
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv3x3_bf16w_kernel<HW, SHUFFLE, POOL, RES, ACT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int nitems = a.N * (HW / TH) * (HW / TW) * (SHUFFLE ? 4 : 1);     // a multiple of 8
    const int grid = nitems < cus ? nitems & ~7 : cus & ~7;
    hipLaunchKernelGGL((conv3x3_bf16w_kernel<HW, SHUFFLE, POOL, RES, ACT>), dim3((unsigned)grid), dim3(256), LDS_BYTES,
                       s, a, nitems);
    return hipGetLastError();
}

template <int HW, bool SHUFFLE, bool POOL, bool RES>
hipError_t launch_one(const Conv3x3Args& a, hipStream_t s, int cus)
{
    return a.slope ? launch_act<HW, SHUFFLE, POOL, RES, true>(a, s, cus) : launch_act<HW, SHUFFLE, POOL, RES, false>(a, s, cus);
}

}  // namespace c3w

// true when the wide-tile kernel serves this launch: enough 8-row items for one workgroup on every CU
bool conv3x3_bf16_wide_applies(const Conv3x3Args& a, int cus)
{
    const int nitems = a.N * (a.H / c3w::TH) * (a.H / c3w::TW) * (a.shuffle ? 4 : 1);
    return cus >= 8 && nitems >= cus;
}

hipError_t launch_conv3x3_bf16_wide(const Conv3x3Args& a, hipStream_t s, int cus)
{
    using namespace c3w;
    if (a.H != a.W || a.N <= 0) return hipErrorInvalidValue;
    const bool res = a.res != nullptr, pool = a.pool_partial != nullptr;
    if (a.H == kLR && !a.shuffle && a.cout_blocks == 1) {
        if (!pool && !res) return launch_one<kLR, false, false, false>(a, s, cus);
        if (!pool && res) return launch_one<kLR, false, false, true>(a, s, cus);
        if (pool && res) return launch_one<kLR, false, true, true>(a, s, cus);
    }
    if (a.H == kLR && a.shuffle && a.cout_blocks == 4 && !pool && !res) return launch_one<kLR, true, false, false>(a, s, cus);
    if (a.H == kHR && !a.shuffle && a.cout_blocks == 1 && !pool && !res) return launch_one<kHR, false, false, false>(a, s, cus);
    return hipErrorInvalidValue;
}

}  // namespace srgen
