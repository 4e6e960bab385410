// srgen_conv3x3_bf16.hip -- gfx950 bf16 3x3 conv 96 -> 96 (x cout_blocks), implicit GEMM on
// v_mfma_f32_32x32x16_bf16, software-pipelined.
//
// Reference sites (file:line into /root/reference): Generator.py:34,37 (ResBlock convs; eval
// BatchNorm :35,38 and the 0.1 of :42 are folded into w, b on the host), :62 (mid), :7
// (upsample conv, PixelShuffle :8 folded into the store address), :69 (output[0]); PReLU
// :36,:9,:70 and the residual adds :42,:78 are the epilogue.
//
// Structure (one workgroup = 4 waves = 4 rows x 64 pixels x 96 output channels; persistent, two
// workgroups per CU, so one's prologue / epilogue is covered by the other's MFMAs -- they drift apart by themselves):
//   * GEMM view: M = 96 co (the MFMA A operand: weights), N = pixels (B operand), K = 9 taps x 96 ci,
//     walked as 2 ci-chunks x 9 taps x 3 k16-steps = 54 steps of 6 MFMAs per wave; the accumulator
//     then holds 4 consecutive channels of one pixel in 4 consecutive registers;
//   * activations: zero-padded 6 x 66 pixel halo tile of one 48-ci chunk in LDS, pixel stride 112 B
//     (7 x 16 B, odd => conflict-free ds_read_b128), staged global -> registers -> LDS with every
//     (thread, round)-dependent quantity precomputed as bit masks; chunk 1 is fetched into registers
//     under the last tap of chunk 0, the next item's chunk 0 under the epilogue;
//   * weights: one 9 KB slab per (tap, chunk) copied global -> LDS with global_load_lds (wave-uniform
//     base + one per-lane offset) into a 3-deep ring, issued two phases ahead, so the fragments of the
//     next phase's first step can be read before the barrier that ends the current phase;
//   * fragments are double-buffered in registers and the five ds_reads of step s+1 sit one each in the
//     gaps between the six MFMAs of step s (alone in a gap a ds_read_b128 is free; clustered they stall
//     the wave ~30 cycles each, and a vector ALU instruction in the stream costs about as much:
//     tools/issuebench.cpp);
//   * epilogue, registers only: bias (accumulator start) -> PReLU -> (+ residual, prefetched before the
//     last tap) -> bf16 pack + two v_permlane32_swap per 16-byte piece -> NHWC stores; optional
//     per-tile channel sums for ChannelAttention's pool (fixed order => deterministic).
#include "srgen_internal.h"

#include <type_traits>

namespace srgen {
namespace c3 {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#ifndef C3_IL
#define C3_IL 1        // fragment reads interleaved one per MFMA gap (0: clustered in front of each step)
#endif
#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    // async 16-byte global -> LDS copy: LDS address = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

// The weight-slab copies go through inline asm: with an LDS-DMA instruction it can see inside a loop the
// compiler's wait-count pass waits lgkmcnt(0) before every use of an LDS read there, i.e. each k16-step
// would start by waiting for the fragment read issued one MFMA earlier.  Hidden from it, the copies are
// waited for explicitly (vmcnt(0) in front of every barrier that publishes a slab), and the fragment
// reads get counted waits.  sbase: wave-uniform; lds_addr: LDS byte address of the wave's 1 KB piece.
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

constexpr int TH = 4, TW = 64, HH = 6, HWP = 66;
constexpr int CK = 48;                            // ci per chunk
constexpr int PS = 112;                           // A pixel stride, bytes (96 + 16 pad)
constexpr int A_BYTES = HH * HWP * PS;            // 44,352
constexpr int SLAB = 6 * 96 * 16;                 // 9,216: one (tap, chunk) weight slab
constexpr int NRING = 3;
constexpr int BS_OFF = A_BYTES + NRING * SLAB;    // bias[4][96], slope[4][96] (fp32)
constexpr int RS_OFF = BS_OFF + 2 * 4 * 96 * 4;  // residual scales of the current / next item's image: [2][96] (fp32)
constexpr int LDS_BYTES = RS_OFF + 2 * 96 * 4;    // 75,840 (two workgroups per CU)
constexpr int MAX_WGS = 512;                      // persistent grid: 2 workgroups x 256 CUs

struct Frags { bf16x8 a0, a1, b0, b1, b2; };

#ifdef SRGEN_C3_STAMPS      // development only (tools/convbench): per-work-item phase timestamps
__device__ long long* g_stamps = nullptr;
#define C3_STAMP(k)                                                                       \
    do {                                                                                  \
        if (g_stamps && tid == 0) g_stamps[(size_t)item * 8 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define C3_STAMP(k) do { } while (0)
#endif

__device__ __forceinline__ void load_frags(Frags& f, const char* Ab, const char* Bb, int g)
{
    f.a0 = *(const bf16x8*)(Ab + g * 32);
    f.a1 = *(const bf16x8*)(Ab + 32 * PS + g * 32);
    f.b0 = *(const bf16x8*)(Bb + (2 * g * 96) * 16);
    f.b1 = *(const bf16x8*)(Bb + (2 * g * 96 + 32) * 16);
    f.b2 = *(const bf16x8*)(Bb + (2 * g * 96 + 64) * 16);
}

// The weights are the MFMA's A operand (rows = output channels) and the pixels its B operand
// (columns = pixels), so accumulator register q of lane l is
//     C[co = 32*nt + (q&3) + 8*(q>>2) + 4*(l>>5)][pixel = 32*mt + (l&31)]:
// 4 consecutive channels of ONE pixel sit in 4 consecutive registers -- bf16 packing, the
// residual add and the NHWC store need no transpose through LDS.
__device__ __forceinline__ void mma_step(f32x16 (&acc)[2][3], const Frags& f)
{
    acc[0][0] = MFMA_BF16(f.b0, f.a0, acc[0][0]);
    acc[1][0] = MFMA_BF16(f.b0, f.a1, acc[1][0]);
    acc[0][1] = MFMA_BF16(f.b1, f.a0, acc[0][1]);
    acc[1][1] = MFMA_BF16(f.b1, f.a1, acc[1][1]);
    acc[0][2] = MFMA_BF16(f.b2, f.a0, acc[0][2]);
    acc[1][2] = MFMA_BF16(f.b2, f.a1, acc[1][2]);
}

// One k16-step with the five fragment reads of the NEXT step in the gaps between its six MFMAs: a
// ds_read_b128 issued alone between two MFMAs costs nothing, five in a row in front of the block stall
// the wave -- and with it this SIMD's share of the matrix pipe -- for ~30 cycles each
// (tools/issuebench.cpp).  Read order = use order, so every fragment has five MFMAs to arrive.
__device__ __forceinline__ void mma_step_il(f32x16 (&acc)[2][3], const Frags& f, Frags& n, const char* Ab, const char* Bb,
                                            int g, bool next)
{
    acc[0][0] = MFMA_BF16(f.b0, f.a0, acc[0][0]);
    if (next) n.b0 = *(const bf16x8*)(Bb + (2 * g * 96) * 16);
    __builtin_amdgcn_sched_barrier(0);
    acc[1][0] = MFMA_BF16(f.b0, f.a1, acc[1][0]);
    if (next) n.a0 = *(const bf16x8*)(Ab + g * 32);
    __builtin_amdgcn_sched_barrier(0);
    acc[0][1] = MFMA_BF16(f.b1, f.a0, acc[0][1]);
    if (next) n.a1 = *(const bf16x8*)(Ab + 32 * PS + g * 32);
    __builtin_amdgcn_sched_barrier(0);
    acc[1][1] = MFMA_BF16(f.b1, f.a1, acc[1][1]);
    if (next) n.b1 = *(const bf16x8*)(Bb + (2 * g * 96 + 32) * 16);
    __builtin_amdgcn_sched_barrier(0);
    acc[0][2] = MFMA_BF16(f.b2, f.a0, acc[0][2]);
    if (next) n.b2 = *(const bf16x8*)(Bb + (2 * g * 96 + 64) * 16);
    __builtin_amdgcn_sched_barrier(0);
    acc[1][2] = MFMA_BF16(f.b2, f.a1, acc[1][2]);
}

// Returns v, but opaque to the optimiser: index arithmetic derived from it is recomputed where it
// is used instead of being hoisted out of the persistent loop and kept in (spilled) registers.
__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// v_permlane32_swap: lanes 32-63 of x trade places with lanes 0-31 of y.
__device__ __forceinline__ void swap32(unsigned& x, unsigned& y)
{
    const auto r = __builtin_amdgcn_permlane32_swap(x, y, false, false);
    x = r[0];
    y = r[1];
}

// Work item = (tile of 4 rows x 64 pixels, cout block).  Persistent workgroups: ids b, b+8, ...
// share an XCD (private L2) under round-robin placement, so XCD x owns the contiguous item
// range [x*per, (x+1)*per) and its workgroups walk it with stride (workgroups per XCD); the
// halo rows two neighbouring tiles share then come from that XCD's L2.  Speed only.
struct Item { int n, ty, tx, cb; };

// SC: a channel attention folded into this conv (eval forward: the block behind an attention, Generator.py:76-78):
// 1 = the INPUT is multiplied by a.in_scale[n][ci] when its halo tile is stored to LDS, 2 = the RESIDUAL by
// a.res_scale[n][co] in the epilogue -- each product rounded to bf16, exactly what the attention's in-place pass stored
template <int HW, bool SHUFFLE, bool POOL, bool RES, int SC = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(Conv3x3Args a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int tiles_x = HW / TW, tiles_y = HW / TH, tiles_img = tiles_x * tiles_y;
    constexpr int CB = SHUFFLE ? 4 : 1;
    const bool act = a.slope != nullptr;

    const int per_xcd = nitems >> 3;                     // nitems is a multiple of 16
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
        r.n = tile / tiles_img;
        return r;
    };

    char* const ldsA = lds;
    char* const ldsB = lds + A_BYTES;
    float* const ldsBias = (float*)(lds + BS_OFF);
    float* const ldsSlope = ldsBias + 4 * 96;

    const int tid16 = tid * 16;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    // ---- A staging: halo tile of one ci chunk, global -> registers -> LDS ----------------------------
    // Thread t moves the 16-byte unit q = t % 6 of pixel P = t/6 + 42*it of the 6 x 66 tile in round
    // it = 0..9 (threads 252..255 idle).  Everything that depends on (t, it) only is formed ONCE, as bit
    // masks over it: a vector ALU instruction inside the MFMA stream costs ~28 cycles, and the obvious
    // form (divide the piece index by 6 and by 66, test both coordinates, per round and per item) was
    // 7 % of the kernel.  Pixel P sits at row rb + wrap, column p0 + rem - 66*wrap with (rb, rem) =
    // divmod(42*it, 66) compile-time and wrap = (p0 + rem >= 66) a per-lane bit.
    u32x4 stage[10];
    const int p0 = tid / 6, q6 = tid - p0 * 6;
    const int a_g0 = p0 * kC + q6 * 8;                    // element offset of the unit inside its row segment
    char* const a_l0 = lds + p0 * PS + q6 * 16;           // LDS address of round 0 (round it: + it*42*PS)
    unsigned a_wtb = 0, a_col = 0;       // bits it: wrap | 10+it: row 0 | 20+it: row 5;  it: unused | 10+it: col 0 | 20+it: col 65
#pragma unroll
    for (int it = 0; it < 10; ++it) {
        const int rb = (42 * it) / HWP, rem = (42 * it) % HWP;
        const bool wrap = p0 + rem >= HWP;
        const int r = rb + (wrap ? 1 : 0), cc = p0 + rem - (wrap ? HWP : 0);
        const bool used = tid < 252 && p0 + 42 * it < HH * HWP;
        a_wtb |= (wrap ? 1u : 0u) << it | (r == 0 ? 1u : 0u) << (10 + it) | (r == HH - 1 ? 1u : 0u) << (20 + it);
        a_col |= (used ? 0u : 1u) << it | (cc == 0 ? 1u : 0u) << (10 + it) | (cc == HWP - 1 ? 1u : 0u) << (20 + it);
    }
    f32x4 sstage[2];                      // SC == 1: the scales of this thread's 8 channels of the chunk in `stage`
    auto load_a = [&](const Item& t, int c) {
        if (SC == 1) {
            const float* sp = a.in_scale + (size_t)t.n * kC + c * CK + q6 * 8;
            sstage[0] = *(const f32x4*)sp;
            sstage[1] = *(const f32x4*)(sp + 4);
        }
        const __bf16* in = (const __bf16*)a.in + (size_t)t.n * HW * HW * kC;
        const int x0 = t.tx * TW, y0 = t.ty * TH;
        // rounds whose pixel is outside the image (or unused): zero
        unsigned bad = a_col;
        if (x0 == 0) bad |= a_col >> 10;
        if (x0 + TW == HW) bad |= a_col >> 20;
        if (y0 == 0) bad |= a_wtb >> 10;
        if (y0 + TH == HW) bad |= a_wtb >> 20;
        const __bf16* base = in + ((ptrdiff_t)(y0 - 1) * HW + x0 - 1) * kC + c * CK;     // tile origin (may lie outside)
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int rb = (42 * it) / HWP, rem = (42 * it) % HWP;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (!((bad >> it) & 1u)) {
                const int off = a_g0 + (int)((a_wtb >> it) & 1u) * ((HW - HWP) * kC);
                v = *(const u32x4*)(base + (ptrdiff_t)(rb * HW + rem) * kC + off);
            }
            stage[it] = v;
        }
    };
    auto store_a = [&]() {
        if (tid < 252) {
#pragma unroll
            for (int it = 0; it < 10; ++it)
                if (it < 9 || p0 + 42 * 9 < HH * HWP) {
                    u32x4 v = stage[it];
                    if (SC == 1) {                // (zero pieces stay zero)
#pragma unroll
                        for (int j = 0; j < 4; ++j)
                            v[j] = pack_bf16(bf16_lo(v[j]) * sstage[j >> 1][2 * (j & 1)], bf16_hi(v[j]) * sstage[j >> 1][2 * (j & 1) + 1]);
                    }
                    *(u32x4*)(a_l0 + it * 42 * PS) = v;
                }
        }
    };
    // async slab copy, 576 x 16 B, slab j = (tap j%9, chunk j/9) of cout block cb into ring
    // slot j%3 (18 slabs per item, so the slot sequence continues across items)
    auto load_b = [&](int cb, int j) {
        const int tap = j % 9, c = j / 9;
        const char* src = (const char*)a.wpack + ((size_t)((cb * 9 + tap) * 12 + 6 * c) * 96) * 16;
        const unsigned dst = lds_base + A_BYTES + (j % NRING) * SLAB + w * 1024;
        // wave-uniform base + one per-lane 32-bit offset: global_load_lds v_off, s[base] with no vector
        // address arithmetic in the MFMA stream (a VALU instruction between MFMAs costs ~28 cycles);
        // the 64-unit tail of the slab is wave 0's alone (a scalar branch) instead of 16 lanes of each wave
        glds16_raw(src, (unsigned)tid16, dst);
        glds16_raw(src + 4096, (unsigned)tid16, dst + 4096);
        if (w == 0) glds16_raw(src + 8192, (unsigned)tid16, dst + 8192);
    };
    // Residual of this lane's outputs: pixel (y0 + w, x0 + 32*mt + p); per accumulator tile
    // (mt, nt) two 16-byte pieces P = 0, 1 at channels 32*nt + 16*P + 8*h (see the epilogue).
    u32x4 rres[12];
    auto load_res = [&](const Item& t) {
        const int lv = opaque(lane);
        const __bf16* res = (const __bf16*)a.res +
                            (((size_t)t.n * HW + t.ty * TH + w) * HW + t.tx * TW + (lv & 31)) * kC + 8 * (lv >> 5);
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            const int mt = i / 6, nt = (i % 6) / 2, P = i % 2;
            rres[i] = *(const u32x4*)(res + (size_t)(32 * mt) * kC + 32 * nt + 16 * P);
        }
    };

    Item cur = decode(item);
    load_a(cur, 0);
    load_b(cur.cb, 0);
    load_b(cur.cb, 1);
    for (int i = tid; i < CB * 96; i += 256) {
        ldsBias[i] = a.bias[i];
        ldsSlope[i] = act ? a.slope[i] : 1.f;
    }
    float* const ldsRS = (float*)(lds + RS_OFF);
    int rsb = 0;                                  // buffer of the current item's residual scales
    if (SC == 2 && tid < kC) ldsRS[tid] = a.res_scale[(size_t)cur.n * kC + tid];
    // (Rounds 1-2 de-phased the two workgroups that share a CU here -- the second half of the grid slept ~6 us -- so that one's
    // epilogue ran beside the other's MFMAs.  Measured again at the end of round 3: without the sleep the batch-256 forward
    // runs 7.85 instead of 7.97 ms and the 64-image 16-bit training step 10.65 instead of 10.85 ms: the workgroups drift apart
    // by themselves within the first item, the sleep only delayed the launch's end.)
    store_a();
    wait_vm0();                                   // slabs 0, 1 (asm copies: not known to the compiler)
    __syncthreads();

    Frags f0, f1;
    const char* const Aw = ldsA + (w * HWP + p) * PS + h * 16;
    const char* const Bw = ldsB + (h * 96 + p) * 16;

#pragma unroll 1
    for (;;) {
        const int nxt = item + wg_per_xcd;
        const bool has_next = nxt < item_end;
        const Item nx = decode(has_next ? nxt : item);
#ifdef SRGEN_C3_STAMPS
        if (g_stamps && tid == 0) {
            g_stamps[(size_t)item * 8 + 6] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_ID
            g_stamps[(size_t)item * 8 + 7] = __builtin_amdgcn_s_getreg((31 << 11) | 20);     // XCC_ID
        }
#endif
        C3_STAMP(0);

        // accumulators start at the bias (broadcast LDS reads: the address depends on h only)
        f32x16 acc[2][3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                const f32x4 b4 = *(const f32x4*)(ldsBias + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][nt][4 * G + e] = acc[1][nt][4 * G + e] = b4[e];
            }
        C3_STAMP(1);

        // one ci-chunk = 9 phases (taps) x 3 k16-steps, fully unrolled; C is a compile-time chunk
        // id so that the staging registers and the residual registers have disjoint lifetimes.
        auto run_chunk = [&](auto C) {
            constexpr int c = decltype(C)::value;
            load_frags(f0, Aw, Bw + ((c * 9) % NRING) * SLAB, 0);
#pragma unroll
            for (int s = 0; s < 27; ++s) {
                const int tap = s / 3, g = s % 3;
                const int j = c * 9 + tap;
                if (g == 0) {
                    if (j + 2 < 18) load_b(cur.cb, j + 2);
                    else if (has_next) load_b(nx.cb, j + 2 - 18);    // next item's slabs 0, 1
                    if (tap == 8 && c == 0) load_a(cur, 1);
                    if (RES && tap == 7 && c == 1) load_res(cur);
                }
#if C3_IL
                {
                    const int tap1 = (s + 1) / 3, g1 = (s + 1) % 3;
                    const int dy = tap1 / 3, dx = tap1 % 3;
                    const char* Ab = Aw + (dy * HWP + dx) * PS;
                    const char* Bb = Bw + ((c * 9 + tap1) % NRING) * SLAB;
                    __builtin_amdgcn_sched_barrier(0);
                    if (s & 1) mma_step_il(acc, f1, f0, Ab, Bb, g1, s + 1 < 27);
                    else mma_step_il(acc, f0, f1, Ab, Bb, g1, s + 1 < 27);
                }
#else
                if (s + 1 < 27) {
                    const int tap1 = (s + 1) / 3, g1 = (s + 1) % 3;
                    const int dy = tap1 / 3, dx = tap1 % 3;
                    const char* Ab = Aw + (dy * HWP + dx) * PS;
                    const char* Bb = Bw + ((c * 9 + tap1) % NRING) * SLAB;
                    if (s & 1) load_frags(f0, Ab, Bb, g1); else load_frags(f1, Ab, Bb, g1);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s & 1) mma_step(acc, f1); else mma_step(acc, f0);
#endif
                if (g == 2) {                      // slabs j+1, j+2 landed; slab j is free
                    asm volatile("" ::: "memory");
                    wait_vm0();
                    __builtin_amdgcn_s_barrier();
                    asm volatile("" ::: "memory");
                }
            }
        };
        run_chunk(std::integral_constant<int, 0>{});
        C3_STAMP(2);
        store_a();                                // every wave is past its last read of chunk 0
        __syncthreads();
        C3_STAMP(3);
        run_chunk(std::integral_constant<int, 1>{});
        C3_STAMP(4);

        // the A tile is free (barrier of the last phase): fetch the next item's chunk 0 into the
        // staging registers now, its latency is covered by the epilogue arithmetic below
        if (has_next) load_a(nx, 0);

        // ---- epilogue (registers only) -----------------------------------------------------
        // PReLU -> (+ residual) -> bf16 -> two permlane32 swaps per 16-byte piece -> NHWC stores.
        const int y = cur.ty * TH + w, x0 = cur.tx * TW;
        __bf16* out = (__bf16*)a.out;
        constexpr int PXS = SHUFFLE ? 2 * kC : kC;      // element stride between this wave's pixels
        if (SHUFFLE)
            out += (((size_t)cur.n * 2 * HW + 2 * y + (cur.cb >> 1)) * 2 * HW + 2 * (x0 + p) + (cur.cb & 1)) * kC + 8 * h;
        else
            out += (((size_t)cur.n * HW + y) * HW + x0 + p) * kC + 8 * h;
        float psum[3][16];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            if (act) {
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    const f32x4 s4 = *(const f32x4*)(ldsSlope + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const float v = acc[mt][nt][4 * G + e];
                            acc[mt][nt][4 * G + e] = v >= 0.f ? v : v * s4[e];
                        }
                }
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x16& c = acc[mt][nt];
                if (RES) {
#pragma unroll
                    for (int P = 0; P < 2; ++P) {
                        const u32x4 r = rres[(mt * 3 + nt) * 2 + P];
                        unsigned r0 = r[0], r1 = r[1], r2 = r[2], r3 = r[3];
                        // loaded piece = [X0 X1 Y0 Y1] in the stored (swapped) arrangement: undo it
                        swap32(r0, r2);
                        swap32(r1, r3);
                        if (SC == 2) {            // registers 8P + k: channel 32 nt + 16 P + 8 (k >> 2) + 4 h + (k & 3)
                            const float* rs = ldsRS + rsb * kC + 32 * nt + 16 * P + 4 * h;
                            const f32x4 sa = *(const f32x4*)rs, sb = *(const f32x4*)(rs + 8);
                            r0 = pack_bf16(bf16_lo(r0) * sa[0], bf16_hi(r0) * sa[1]);
                            r1 = pack_bf16(bf16_lo(r1) * sa[2], bf16_hi(r1) * sa[3]);
                            r2 = pack_bf16(bf16_lo(r2) * sb[0], bf16_hi(r2) * sb[1]);
                            r3 = pack_bf16(bf16_lo(r3) * sb[2], bf16_hi(r3) * sb[3]);
                        }
                        c[8 * P + 0] += bf16_lo(r0); c[8 * P + 1] += bf16_hi(r0);
                        c[8 * P + 2] += bf16_lo(r1); c[8 * P + 3] += bf16_hi(r1);
                        c[8 * P + 4] += bf16_lo(r2); c[8 * P + 5] += bf16_hi(r2);
                        c[8 * P + 6] += bf16_lo(r3); c[8 * P + 7] += bf16_hi(r3);
                    }
                }
                if (POOL) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) psum[nt][q] = mt == 0 ? c[q] : psum[nt][q] + c[q];
                }
#pragma unroll
                for (int P = 0; P < 2; ++P) {
                    // groups G = 2P (registers 8P..8P+3) and 2P+1 (8P+4..8P+7), 4 channels each
                    unsigned x0d = pack_bf16(c[8 * P + 0], c[8 * P + 1]), x1d = pack_bf16(c[8 * P + 2], c[8 * P + 3]);
                    unsigned y0d = pack_bf16(c[8 * P + 4], c[8 * P + 5]), y1d = pack_bf16(c[8 * P + 6], c[8 * P + 7]);
                    // lanes < 32 end with [own 2P | partner's 2P] = channels 16P .. 16P+7,
                    // lanes >= 32 with [partner's 2P+1 | own 2P+1] = channels 16P+8 .. 16P+15
                    swap32(x0d, y0d);
                    swap32(x1d, y1d);
                    const u32x4 ov = {x0d, x1d, y0d, y1d};
                    *(u32x4*)(out + (size_t)(32 * mt) * PXS + 32 * nt + 16 * P) = ov;
                }
            }
        }
        if (POOL) {
            // Per-tile channel sums (ChannelAttention's pool), fixed order.  Halving butterfly over
            // the 32 pixels of each lane half: after the 4 exchange steps lane l holds the 16-lane
            // sum of register idx(l) = 8*l0 + 4*l1 + 2*l2 + l3 (l_i = bit i of l); one more exchange
            // adds the other 16 lanes.  Then the 4 waves are summed through LDS (slab ring slot 2:
            // free between the last phase and the next item's phase 0).
            float* P = (float*)(ldsB + 2 * SLAB);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                float (&t)[16] = psum[nt];
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
                    P[w * kC + 32 * nt + (q & 3) + 8 * (q >> 2) + 4 * h] = t[0];
                }
            }
            __syncthreads();
            if (tid < kC) {
                const float s = (P[tid] + P[kC + tid]) + (P[2 * kC + tid] + P[3 * kC + tid]);
                a.pool_partial[((size_t)cur.n * tiles_img + cur.ty * tiles_x + cur.tx) * kC + tid] = s;
            }
        }
        C3_STAMP(5);
        if (!has_next) break;
        store_a();
        if (SC == 2) {                            // (the other buffer: slower waves still read this item's)
            rsb ^= 1;
            if (tid < kC) ldsRS[rsb * kC + tid] = a.res_scale[(size_t)nx.n * kC + tid];
        }
        wait_vm0();
        __syncthreads();                          // next A chunk 0 and its slabs 0, 1 are in LDS
        item = nxt;
        cur = nx;
    }
}

template <int HW, bool SHUFFLE, bool POOL, bool RES, int SC = 0>
hipError_t launch_one(const Conv3x3Args& a, hipStream_t s)
{
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv3x3_bf16_kernel<HW, SHUFFLE, POOL, RES, SC>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int nitems = a.N * (HW / TH) * (HW / TW) * (SHUFFLE ? 4 : 1);     // multiple of 16
    const int grid = nitems < MAX_WGS ? nitems & ~7 : MAX_WGS;
    hipLaunchKernelGGL((conv3x3_bf16_kernel<HW, SHUFFLE, POOL, RES, SC>), dim3((unsigned)grid), dim3(256), LDS_BYTES,
                       s, a, nitems);
    return hipGetLastError();
}

}  // namespace c3

hipError_t launch_conv3x3_bf16(const Conv3x3Args& a, hipStream_t s)
{
    using namespace c3;
    if (a.H != a.W || a.N <= 0) return hipErrorInvalidValue;
    const bool res = a.res != nullptr, pool = a.pool_partial != nullptr;
    if (a.in_scale && a.res_scale) return hipErrorInvalidValue;
    if (a.in_scale || a.res_scale) {                  // a folded channel attention: the three convs that can sit behind one
        if (a.H != kLR || a.shuffle || a.cout_blocks != 1 || pool) return hipErrorInvalidValue;
        if (a.in_scale && !res) return launch_one<kLR, false, false, false, 1>(a, s);
        if (a.in_scale && res) return launch_one<kLR, false, false, true, 1>(a, s);
        if (a.res_scale && res) return launch_one<kLR, false, false, true, 2>(a, s);
        return hipErrorInvalidValue;
    }
    if (a.H == kLR && !a.shuffle && a.cout_blocks == 1) {
        if (!pool && !res) return launch_one<kLR, false, false, false>(a, s);
        if (!pool && res) return launch_one<kLR, false, false, true>(a, s);
        if (pool && res) return launch_one<kLR, false, true, true>(a, s);
    }
    if (a.H == kLR && a.shuffle && a.cout_blocks == 4 && !pool && !res) return launch_one<kLR, true, false, false>(a, s);
    if (a.H == kHR && !a.shuffle && a.cout_blocks == 1 && !pool && !res) return launch_one<kHR, false, false, false>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace srgen
