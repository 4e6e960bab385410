// srgen_conv3x3_f32s.hip -- gfx950 3x3 conv 96 -> 96 (x cout_blocks) with fp32 tensors, fp32 accumulation and the
// products formed on the f16 matrix instruction from two-term splits of both operands ("split" path).
//
// Reference sites: as srgen_conv3x3_f32.hip (Generator.py:34,37,62,7,69 with the folded BatchNorm; PReLU / residual /
// PixelShuffle in the epilogue).
//
// Why.  The exact-fp32 MFMA (v_mfma_f32_32x32x2_f32) runs at 1/16 of the f16 one (v_mfma_f32_32x32x16_f16).  An fp32
// value x splits exactly into x = hi + lo + r with hi = f16(x), lo = f16(x - hi) and |r| <= 2^-22 |x| (f16 keeps 11
// significant bits; its subnormals are kept by the matrix pipe -- probed with tools/f16denorm.cpp), products of two f16
// numbers are exact in fp32, and the MFMA accumulates in fp32.  So
//     x * w  =  hi_x hi_w + hi_x lo_w + lo_x hi_w  + O(2^-22 |x w|)
// costs three f16 MFMAs = 3/16 of the exact-fp32 MFMA time (the Winograd kernel: 16/36 = 7/16), with the error of an
// fp32 product (a numpy emulation of one 864-term layer: 8.2e-7 of max against 6.1e-7 for plain fp32 accumulation), and
// none of Winograd's transform arithmetic or its 4 - 9x error growth.  Weights are split once on the host (fp64 fold ->
// fp32 -> hi / lo); activations are split when the halo tile is staged into LDS, so tensors in HBM stay fp32.
//
// Skeleton = the direct fp32 kernel: persistent workgroups of 4 waves, work item = 4 rows x 64 pixels x 96 co, wave = one
// row = 2 x 3 tiles of 32 x 32 (96 accumulators), weights = MFMA A operand so a lane ends with 4 consecutive channels of
// one pixel per register group (16-byte NHWC stores).  K = 9 taps x 96 ci runs as 6 chunks of 16 ci (one MFMA K) x 3
// kernel rows x 3 taps.  LDS: the halo tile of one chunk (6 x 66 pixels x {16 hi, 16 lo} f16, pixel pitch 80 B = 16 x odd:
// conflict-free ds_read_b128) and a two-slot ring of weight groups (one kernel row of one chunk: 3 taps x 3 co tiles x
// {hi, lo} x 64 lanes x 16 B = 18 KB, lane-ordered so a fragment is one ds_read_b128); the next group's weights and the
// next chunk's pixels are fetched into registers under the MFMAs and written to LDS behind them.  One barrier per group
// (54 MFMAs = 1,728 cycles), two at a chunk end; 71 KB of LDS, two workgroups per CU.
#include "srgen_internal.h"

#include <type_traits>

namespace srgen {
namespace c3s {

static_assert(kAmaxSlots == 4 * 256, "one float4 of partial maxima per thread");

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int TH = 4, TW = 64, HH = 6, HWP = 66;
constexpr int CK = 16, NCH = 6;                   // ci per chunk, chunks
constexpr int PSB = 80;                           // bytes per staged pixel: 16 hi | 16 lo | pad
constexpr int A_BYTES = HH * HWP * PSB;           // 31,680
constexpr int WG_UNITS = 3 * 3 * 2 * 64;          // 16-byte units of one weight group (kernel row of a chunk): 1,152
constexpr int WG_BYTES = WG_UNITS * 16;           // 18,432
constexpr int W_OFF = A_BYTES;
constexpr int BS_OFF = W_OFF + 2 * WG_BYTES;      // bias[4][96], slope[4][96]
constexpr int POOL_OFF = BS_OFF + 2 * 4 * 96 * 4;
constexpr int RS_OFF = POOL_OFF + 4 * 96 * 4;     // residual scales of the current / next item's image: [2][96] (a folded attention)
constexpr int LDS_BYTES = RS_OFF + 2 * 96 * 4;    // 73,920 (two workgroups per CU)
constexpr int A_ITEMS = HH * HWP * 4;             // float4 pieces per chunk (1,584)
constexpr int A_REGS = 7;                         // ceil(1584 / 256)
constexpr int MAX_WGS = 512;
constexpr int kSmallItems = 96;                   // fewer work items than this (x 3 co tiles: about one workgroup per CU) run the
                                                  // small-batch kernel: up to 5 images at 64 x 64, 1 at 128 x 128

struct Frags { h8 xh[2], xl[2], wh[3], wl[3]; };

__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

// acc += hi_w hi_x + lo_w hi_x + hi_w lo_x, small terms first (TERMS = 1: hi_w hi_x alone -- f16 products)
template <int TERMS>
__device__ __forceinline__ void mma_tap(f32x16 (&acc)[2][3], const Frags& f)
{
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (TERMS == 3) {
                acc[mt][nt] = MFMA_H(f.wl[nt], f.xh[mt], acc[mt][nt]);
                acc[mt][nt] = MFMA_H(f.wh[nt], f.xl[mt], acc[mt][nt]);
            }
            acc[mt][nt] = MFMA_H(f.wh[nt], f.xh[mt], acc[mt][nt]);
        }
}

struct Item { int n, ty, tx, cb; };

// ABL: timing ablations for tools/bench_f32s.hip only (wrong results); the library instantiates ABL = 0
// SC: a channel attention folded into this conv (eval forward, the block behind an attention -- Generator.py:76-78 -- as in
// srgen_conv3x3_bf16.hip): 1 = every input piece is multiplied by a.in_scale[n][ci] before it is split, 2 = the residual by
// a.res_scale[n][co] before it is added; one fp32 rounding each, the value the attention's in-place pass used to store
template <int HW, bool SHUFFLE, bool POOL, bool RES, int ABL = 0, int TERMS = 3, int SC = 0>
__global__ __launch_bounds__(256, 2) void conv3x3_f32s_kernel(Conv3x3Args a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int tiles_x = HW / TW, tiles_y = HW / TH, tiles_img = tiles_x * tiles_y;
    constexpr int CB = SHUFFLE ? 4 : 1;
    const bool act = a.slope != nullptr;

    // persistent item walk: XCD x owns items [x*per, (x+1)*per) (see srgen_conv3x3_bf16.hip)
    const int per_xcd = nitems >> 3;
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int item_end = (xcd + 1) * per_xcd;
    int item = xcd * per_xcd + ((int)blockIdx.x >> 3);
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
    char* const ldsW = lds + W_OFF;
    float* const ldsBias = (float*)(lds + BS_OFF);
    float* const ldsSlope = ldsBias + 4 * 96;
    float* const ldsPool = (float*)(lds + POOL_OFF);

    // ---- power-of-two scaling of both operands into f16's normal range (srgen_internal.h: split_exp) --------------
    // weights: scaled by 2^kw when the image was packed (the exponent of the amax in the image's tail); input: scaled by
    // 2^kx at the split, kx from a bound of max |in|: one for the whole tensor (training gradients: kAmaxSlots partial
    // maxima, reduced here once), one per image (eval forward: what the producing conv's waves left for that image by
    // atomic max, read whenever the workgroup moves to an item), or none (in_amax = nullptr: activations of
    // order 1).  The accumulators then hold 2^(kw + kx) times the conv: the bias goes in scaled up, the result comes
    // out scaled down.
    const int kw = split_tail_exp((const char*)a.wpack + (size_t)CB * kSplitImageBytes, lane);
    int kx_all = 0;
    if (a.in_amax && a.in_amax_per_image == 0) {      // maximum of the kAmaxSlots partial maxima
        const f32x4 q = *(const f32x4*)(a.in_amax + tid * 4);
        float m = fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) ldsPool[w] = m;
        __syncthreads();
        kx_all = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(ldsPool[0], ldsPool[1]), fmaxf(ldsPool[2], ldsPool[3]))));
        __syncthreads();
    }
    if (a.zero_amax && blockIdx.x == 0)
        for (int i = tid; i < a.N; i += 256) a.zero_amax[i] = 0.f;
    auto item_kx = [&](int n) {                       // wave-uniform: one scalar-cached load
        if (!a.in_amax || a.in_amax_per_image == 0) return kx_all;
        return __builtin_amdgcn_readfirstlane(split_exp(a.in_amax[n]));
    };
    int kx_stage = 0;                                 // of the item whose pixels are being staged (scalar register)
    float vmax = 0.f;                                 // max |stored value| of this lane in the current item (-> out_amax)

    // ---- halo tile staging: fp32 pieces -> registers -> split (under the MFMAs) -> LDS ----------------------------
    // Per item every thread keeps the image-relative offsets of its seven 16-byte pieces (clamped into the image) and a
    // validity bit each; a chunk's load is then seven loads at offset + 16 c.
    f32x4 stage[A_REGS];
    h4 chi[A_REGS], clo[A_REGS];
    int aoff[A_REGS];
    unsigned aok = 0;
    auto plan_a = [&](const Item& t) {
        const int x0 = t.tx * TW, y0 = t.ty * TH;
        aok = 0;
#pragma unroll
        for (int it = 0; it < A_REGS; ++it) {
            const int i = min(tid + it * 256, A_ITEMS - 1);
            const int pix = i >> 2, q = i & 3;
            const int r = pix / HWP, cc = pix - r * HWP;
            const int gy = y0 - 1 + r, gx = x0 - 1 + cc;
            const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const int cy = min(max(gy, 0), HW - 1), cx = min(max(gx, 0), HW - 1);
            aoff[it] = (cy * HW + cx) * kC + q * 4;
            aok |= (ok ? 1u : 0u) << it;
        }
    };
    f32x4 ssc = {1.f, 1.f, 1.f, 1.f};                 // SC == 1: the scales of this thread's four channels of the chunk in `stage`
    auto load_a = [&](const Item& t, int c) {
        // (in FRONT of the pixel loads: the counted waits below leave the seven youngest loads in flight)
        if (SC == 1) ssc = *(const f32x4*)(a.in_scale + (size_t)t.n * kC + c * CK + (tid & 3) * 4);
        const float* in = (const float*)a.in + (size_t)((ABL & 2048) ? 0 : t.n) * HW * HW * kC + c * CK;
#pragma unroll
        for (int it = 0; it < A_REGS; ++it) stage[it] = *(const f32x4*)(in + aoff[it]);
    };
    // split of one 16-byte piece: hi = f16(v * 2^kx), lo = f16(v * 2^kx - hi), two instructions per element
    // (v_fma_mixlo/hi_f16: fp32 x fp32 + f16 -> f16, each writing one half of the packed destination: no separate scaling,
    // conversion back to fp32, subtraction or packing); a piece outside the image is multiplied by 0 instead of selected
    auto convert_a = [&](int it) {
        const float m = ((aok >> it) & 1u) ? pow2f(kx_stage) : 0.f;
        unsigned h01, h23, l01 = 0, l23 = 0;      // (mixlo leaves the destination's high half for mixhi to fill: "=&v", then "+v")
        f32x4 v = stage[it];
        if (SC == 1) v *= ssc;                        // (a product of its own, rounded to fp32: what the attention pass stored)
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(h01) : "v"(v.x), "v"(m));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v.y), "v"(m));
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(h23) : "v"(v.z), "v"(m));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v.w), "v"(m));
        if (TERMS == 3) {
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(l01) : "v"(v.x), "v"(m), "v"(h01));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v.y), "v"(m), "v"(h01));
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(l23) : "v"(v.z), "v"(m), "v"(h23));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v.w), "v"(m), "v"(h23));
        }
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        chi[it] = __builtin_bit_cast(h4, (u32x2){h01, h23});
        if (TERMS == 3) clo[it] = __builtin_bit_cast(h4, (u32x2){l01, l23});
    };
    auto store_a = [&]() {
#pragma unroll
        for (int it = 0; it < A_REGS; ++it) {
            const int i = tid + it * 256;
            if (it < A_REGS - 1 || i < A_ITEMS) {
                char* rec = ldsA + (i >> 2) * PSB + (i & 3) * 8;
                *(h4*)rec = chi[it];
                if (TERMS == 3) *(h4*)(rec + 32) = clo[it];
            }
        }
    };
    // ---- weight groups: packed 16-byte units [cb][chunk 6][ky 3][kx 3][nt 3][hl 2][lane 64], copied global -> LDS
    // without registers: 18 pieces of 1 KB per group, wave w takes pieces w, w + 4, ... (see glds16_raw in
    // srgen_conv3x3_bf16.hip for why this is inline asm; the copies are waited for with vmcnt(0) before the barrier
    // that publishes the slot) ----
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    auto issue_w = [&](int cb, int g, int slot) {
        const char* src = (const char*)a.wpack + ((size_t)cb * 18 + g) * WG_BYTES;
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            // TERMS = 1: only the nine pieces of high parts (even piece indices)
            const int piece = TERMS == 3 ? w + 4 * k : 2 * (w + 4 * k);
            if (piece < 18) {
                const unsigned la = lds_base + W_OFF + slot * WG_BYTES + piece * 1024;
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                             :: "v"((unsigned)(piece * 1024 + lane * 16)), "s"(src), "s"(la) : "memory", "m0");
            }
        }
    };
    auto wait_all = [&]() { __builtin_amdgcn_s_waitcnt(0x0070); };            // vmcnt(0) lgkmcnt(0)
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // residual of this lane's outputs: per (32-pixel M tile, 32-channel N tile) four 16-byte pieces at channels
    // 32 nt + 8 G + 4 h.  Tile 0 is fetched under the last 18 MFMAs of the item (into registers the second fragment set
    // has just left), the other five behind the item's last barrier, when fragments and staging registers are free
    f32x4 rres[6][4];
    auto load_res = [&](const Item& t, int k) {
        const int mt = k / 3, nt = k % 3;
        const float* res = (const float*)a.res +
                           (((size_t)t.n * HW + t.ty * TH + w) * HW + t.tx * TW + 32 * mt + p) * kC + 4 * h + 32 * nt;
#pragma unroll
        for (int G = 0; G < 4; ++G) rres[k][G] = *(const f32x4*)(res + 8 * G);
    };

    long long t_start = 0;
    if (ABL & 256) t_start = clock64();
    Item cur = decode(item);
    int kx_cur = item_kx(cur.n), kx_nx = kx_cur;
    kx_stage = kx_cur;
    plan_a(cur);
    load_a(cur, 0);
    issue_w(cur.cb, 0, 0);
    issue_w(cur.cb, 1, 1);                            // (an item's group 1 is requested before the item starts: see the epilogue)
    for (int i = tid; i < CB * 96; i += 256) {
        ldsBias[i] = a.bias[i];
        ldsSlope[i] = act ? a.slope[i] : 1.f;
    }
    float* const ldsRS = (float*)(lds + RS_OFF);
    int rsb = 0;                                      // buffer of the current item's residual scales
    if (SC == 2 && tid < kC) ldsRS[tid] = a.res_scale[(size_t)cur.n * kC + tid];
#pragma unroll
    for (int it = 0; it < A_REGS; ++it) convert_a(it);
    store_a();
    wait_all();
    barrier();

    // this lane's fragment addresses: pixel (row w + ky, column p + kx [+ 32]) of the halo tile, half h of the chunk
    const char* const Ax = ldsA + (w * HWP + p) * PSB + 16 * h;
    const char* const Wl = ldsW + lane * 16;

#pragma unroll 1
    for (;;) {
        const int nxt = item + wg_per_xcd;
        const bool has_next = nxt < item_end;
        const Item nx = decode(has_next ? nxt : item);
        // the next item's bound: loaded here, first looked at five chunks later (no wait on its latency)
        const bool per_image = a.in_amax && a.in_amax_per_image;
        const float amax_nx = per_image ? a.in_amax[nx.n] : 0.f;

        // accumulators start at the (scaled) bias (broadcast LDS reads: the address depends on h only)
        const float acc_mul = pow2f(kw + kx_cur), out_mul = pow2f(-(kw + kx_cur));
        vmax = 0.f;
        f32x16 acc[2][3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                const f32x4 b4 = *(const f32x4*)(ldsBias + cur.cb * 96 + 32 * nt + 8 * G + 4 * h) * acc_mul;
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][nt][4 * G + e] = acc[1][nt][4 * G + e] = b4[e];
            }

        auto load_frags = [&](Frags& f, int slot, int ky, int kx) {
            const char* ab = Ax + (ky * HWP + kx) * PSB;
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f.xh[mt] = *(const h8*)(ab + mt * 32 * PSB);
                if (TERMS == 3) f.xl[mt] = *(const h8*)(ab + mt * 32 * PSB + 32);
            }
            const char* wb = Wl + slot * WG_BYTES + kx * (3 * 2 * 64 * 16);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                f.wh[nt] = *(const h8*)(wb + (nt * 2) * 1024);
                if (TERMS == 3) f.wl[nt] = *(const h8*)(wb + (nt * 2 + 1) * 1024);
            }
        };

        Frags f0, f1;
        constexpr int NMF = 6 * TERMS, NDS = TERMS == 3 ? 10 : 5;     // MFMAs / fragment reads of one tap
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
                const int g = c * 3 + ky, slot = g & 1;
                // the next group's weights (this item's, or group 0 of the next item) go to the other slot (last read
                // in the previous group, every wave is past the barrier that ended it); in a chunk's first group the
                // next chunk's pixels are fetched, in its last they are split into (hi, lo) between the MFMAs
                if (!(ABL & 16)) {
                    // (group 1's weights were requested in front of the previous item's stores, or in the prologue)
                    if (g == 0) { }
                    else if (g + 1 < 18) issue_w(cur.cb, g + 1, slot ^ 1);
                    else issue_w(nx.cb, 0, slot ^ 1);
                    if (ky == 0) {
                        if (c + 1 < NCH) load_a(cur, c + 1);
                        else {                        // the next item's first chunk: its image's scale from here on
                            kx_nx = per_image ? __builtin_amdgcn_readfirstlane(split_exp(amax_nx)) : kx_all;
                            kx_stage = kx_nx;
                            plan_a(nx);
                            load_a(nx, 0);
                        }
                    }
                }
                const bool cvt = ky == 2 && !(ABL & 1);
                // a chunk's last group carries the split arithmetic and the LDS writes of the next chunk's pixels: it
                // runs at raised priority, so of the two workgroups sharing a CU the one in plain MFMA groups yields
                // issue slots to it (measured 3 % on the PReLU instantiation, 2 % with the residual; tools/bench_f32s)
                if (ky == 2 && !(ABL & 4096)) __builtin_amdgcn_s_setprio(2);
                if (!(ABL & 8) || g == 0) load_frags(f0, slot, ky, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 8) || g == 0) load_frags(f1, slot, ky, 1);
                if (cvt) {
                    convert_a(0);
                    convert_a(1);
                }
                if (!(ABL & 4)) mma_tap<TERMS>(acc, f0);
                if (cvt) {
                    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? 1 : 3, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(ABL & 8)) load_frags(f0, slot, ky, 2);
                if (cvt) {
                    convert_a(2);
                    convert_a(3);
                    convert_a(4);
                }
                if (!(ABL & 4)) mma_tap<TERMS>(acc, f1);
                if (cvt) {
                    __builtin_amdgcn_sched_group_barrier(0x100, NDS, 0);
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? 2 : 5, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (cvt) {
                    convert_a(5);
                    convert_a(6);
                }
                if (RES && g == 17) load_res(cur, 0);
                if (!(ABL & 4)) mma_tap<TERMS>(acc, f0);
                if (cvt) {
#pragma unroll
                    for (int i = 0; i < NMF; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, TERMS == 3 ? 1 : 3, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if (ky == 2) {
                    if (!(ABL & 2)) barrier();        // every wave is past its last read of this chunk's pixels
                    if (!(ABL & 1)) store_a();
                }
                // this wave's weight copies (and pixel stores) have landed.  In a chunk's first group the seven pixel
                // loads issued after the copies may stay in flight (loads return in order): they are first needed in
                // the chunk's last group, and an HBM miss takes about as long as one group
                // An item's first group needs group 1's weights next, and those were requested BEFORE the previous item's
                // epilogue: the 24 output stores of that epilogue and this group's seven pixel loads, all younger, may
                // stay in flight (gfx950 retires vector loads and stores through ONE in-order counter: waiting for anything
                // younger than the stores would wait out every store acknowledgement of the burst all workgroups emit at
                // once).  Group 1 ends with vmcnt(0): the stores get two groups of MFMAs as cover instead of one.
                if (g == 0 && !(ABL & 1024)) __builtin_amdgcn_s_waitcnt((ABL & 8192) ? 0x0077 : 0x407F);   // vmcnt(31) lgkmcnt(0)
                else if (ky == 0 && !(ABL & 1024)) __builtin_amdgcn_s_waitcnt(0x0077);      // vmcnt(7) lgkmcnt(0)
                else wait_all();
                if (!(ABL & 64)) barrier();
                if (ky == 2 && !(ABL & 4096)) __builtin_amdgcn_s_setprio(0);
            }
        }

        // ---- epilogue (registers only): PReLU -> (+ residual) -> 16-byte NHWC stores ---------------
        const int y = cur.ty * TH + w, x0 = cur.tx * TW;
        float* out = (float*)a.out;
        constexpr int PXS = SHUFFLE ? 2 * kC : kC;
        if (SHUFFLE)
            out += (((size_t)cur.n * 2 * HW + 2 * y + (cur.cb >> 1)) * 2 * HW + 2 * (x0 + p) + (cur.cb & 1)) * kC + 4 * h;
        else
            out += (((size_t)cur.n * HW + y) * HW + x0 + p) * kC + 4 * h;
        float psum[3][16];
        // the next item's group 1 into slot 1 (group 17, its last reader, ended with a barrier), requested in front of the stores
        if (has_next && !(ABL & 16)) issue_w(nx.cb, 1, 1);
        if (RES) {
#pragma unroll
            for (int k = 1; k < 6; ++k) load_res(cur, k);
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                const int k = mt * 3 + nt;
                f32x16& c = acc[mt][nt];
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    f32x4 v = {c[4 * G], c[4 * G + 1], c[4 * G + 2], c[4 * G + 3]};
                    v *= out_mul;
                    if (act) {
                        const f32x4 s4 = *(const f32x4*)(ldsSlope + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * s4[e];
                    }
                    if (RES && SC == 2) {             // residual x scale, rounded, THEN added (no fused multiply-add)
                        const f32x4 s4 = *(const f32x4*)(ldsRS + rsb * kC + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            float t = rres[k][G][e] * s4[e];
                            asm volatile("" : "+v"(t));
                            v[e] += t;
                        }
                    } else if (RES) v += rres[k][G];
                    vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
                    if (POOL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) psum[nt][4 * G + e] = mt == 0 ? v[e] : psum[nt][4 * G + e] + v[e];
                    }
                    if (!(ABL & 8192) || v.x == 12345.678f) *(f32x4*)(out + (size_t)(32 * mt) * PXS + 32 * nt + 8 * G) = v;
                }
            }
        }
        if (POOL) {
            // per-tile channel sums (ChannelAttention's pool), fixed order (see srgen_conv3x3_f32.hip); the halo tile
            // and both weight slots already belong to the next item, so the four waves meet in a region of their own
            float* P = ldsPool;
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
            __syncthreads();                      // P is read before the next tile overwrites it
        }
        if (a.out_amax) {                             // one atomic per wave and item: non-negative floats order like
#pragma unroll                                        // their bit patterns
            for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            if (lane == 0 && vmax == vmax) atomicMax((unsigned*)a.out_amax + cur.n, __float_as_uint(vmax));
        }
        if (!has_next) break;
        if (SC == 2) {                                // (the other buffer: slower waves still read this item's; the next
            rsb ^= 1;                                 // item's epilogue is eighteen barriers away)
            if (tid < kC) ldsRS[rsb * kC + tid] = a.res_scale[(size_t)nx.n * kC + tid];
        }
        item = nxt;
        cur = nx;
        kx_cur = kx_nx;
    }
    if (ABL & 256) {
        if (tid == 0 && blockIdx.x == 0) ((long long*)a.zeros)[0] = clock64() - t_start;
    }
}

// ---- small batches (train.py:31-54 runs 4 images; the reference's CPU case runs 16) ----------------------------------------
// With fewer work items than compute units the persistent kernel above leaves most of the chip idle and every item waits
// out its 18 weight-group fetches one after the other (a group's 54 MFMAs per wave cover a fetch only when a second
// workgroup shares the CU).  Here a workgroup is ONE (work item, 32-channel output tile): three times the workgroups, a
// third of the MFMAs each, and ALL 18 weight groups of its tile (108 KB) requested at once into LDS before the first
// product -- one trip to L2 instead of eighteen -- and the pixel loads of all six chunks behind them (168 registers: one
// wave per SIMD has them to spare), so the only memory latency a workgroup waits out is the first.  A tap's fragments are
// read from LDS under the products of the tap before.  The products enter every accumulator in the order of the kernel above (chunk, kernel row,
// tap; low x high, high x low, high x high), so an image's result does not depend on which kernel its batch size selects.
constexpr int WS_BYTES = 3 * 2 * 64 * 16;             // one weight group of ONE co tile: 3 taps x {hi, lo} x 64 lanes x 16 B
constexpr int S_W_OFF = A_BYTES;
constexpr int S_POOL_OFF = S_W_OFF + 18 * WS_BYTES;
constexpr int S_LDS_BYTES = S_POOL_OFF + 4 * 96 * 4;  // 143,808: one workgroup per CU

template <int HW, bool SHUFFLE, bool POOL, bool RES>
__global__ __launch_bounds__(256, 1) void conv3x3_f32s_small_kernel(Conv3x3Args a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int tiles_x = HW / TW, tiles_y = HW / TH, tiles_img = tiles_x * tiles_y;
    constexpr int CB = SHUFFLE ? 4 : 1;
    const bool act = a.slope != nullptr;
    const int cot = (int)blockIdx.x % 3, it = (int)blockIdx.x / 3;
    Item cur;
    cur.cb = it % CB;
    {
        const int tile = it / CB;
        cur.tx = tile % tiles_x;
        cur.ty = (tile / tiles_x) % tiles_y;
        cur.n = tile / tiles_img;
    }
    char* const ldsA = lds;
    float* const ldsPool = (float*)(lds + S_POOL_OFF);

    // scaling exponents: as the kernel above
    const int kw = split_tail_exp((const char*)a.wpack + (size_t)CB * kSplitImageBytes, lane);
    int kx = 0;
    if (a.in_amax && a.in_amax_per_image == 0) {
        const f32x4 q = *(const f32x4*)(a.in_amax + tid * 4);
        float m = fmaxf(fmaxf(q.x, q.y), fmaxf(q.z, q.w));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) ldsPool[w] = m;
        __syncthreads();
        kx = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(ldsPool[0], ldsPool[1]), fmaxf(ldsPool[2], ldsPool[3]))));
        __syncthreads();
    } else if (a.in_amax) {
        kx = __builtin_amdgcn_readfirstlane(split_exp(a.in_amax[cur.n]));
    }
    if (a.zero_amax && blockIdx.x == 0)
        for (int i = tid; i < a.N; i += 256) a.zero_amax[i] = 0.f;

    // ---- every weight group of this co tile: 18 groups x 6 pieces of 1 KB, wave w takes pieces w, w + 4, ... of the 108 ----
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    {
        const char* src = (const char*)a.wpack + (size_t)cur.cb * 18 * WG_BYTES;
#pragma unroll
        for (int k = 0; k < 27; ++k) {
            const int piece = w + 4 * k;                         // = g * 6 + tap * 2 + hl
            const int g = piece / 6, r = piece - g * 6, tap = r >> 1, hl = r & 1;
            const unsigned goff = (unsigned)(g * WG_BYTES + ((tap * 3 + cot) * 2 + hl) * 1024 + lane * 16);
            const unsigned la = lds_base + S_W_OFF + piece * 1024;
            asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                         :: "v"(goff), "s"(src), "s"(la) : "memory", "m0");
        }
    }

    // ---- pixel staging (as above): fp32 pieces -> registers -> (hi, lo) -> LDS ----
    f32x4 stage[NCH][A_REGS];
    int aoff[A_REGS];
    unsigned aok = 0;
    {
        const int x0 = cur.tx * TW, y0 = cur.ty * TH;
#pragma unroll
        for (int i7 = 0; i7 < A_REGS; ++i7) {
            const int i = min(tid + i7 * 256, A_ITEMS - 1);
            const int pix = i >> 2, q = i & 3;
            const int r = pix / HWP, cc = pix - r * HWP;
            const int gy = y0 - 1 + r, gx = x0 - 1 + cc;
            const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const int cy = min(max(gy, 0), HW - 1), cx = min(max(gx, 0), HW - 1);
            aoff[i7] = (cy * HW + cx) * kC + q * 4;
            aok |= (ok ? 1u : 0u) << i7;
        }
    }
    const float* in = (const float*)a.in + (size_t)cur.n * HW * HW * kC;
    auto load_a = [&](int c) {
#pragma unroll
        for (int i7 = 0; i7 < A_REGS; ++i7) stage[c][i7] = *(const f32x4*)(in + c * CK + aoff[i7]);
    };
    auto split_store_a = [&](int c) {
#pragma unroll
        for (int i7 = 0; i7 < A_REGS; ++i7) {
            const float m = ((aok >> i7) & 1u) ? pow2f(kx) : 0.f;
            unsigned h01, h23, l01, l23;
            const f32x4 v = stage[c][i7];
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(h01) : "v"(v.x), "v"(m));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h01) : "v"(v.y), "v"(m));
            asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=&v"(h23) : "v"(v.z), "v"(m));
            asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h23) : "v"(v.w), "v"(m));
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(l01) : "v"(v.x), "v"(m), "v"(h01));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l01) : "v"(v.y), "v"(m), "v"(h01));
            asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=&v"(l23) : "v"(v.z), "v"(m), "v"(h23));
            asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l23) : "v"(v.w), "v"(m), "v"(h23));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const int i = tid + i7 * 256;
            if (i7 < A_REGS - 1 || i < A_ITEMS) {
                char* rec = ldsA + (i >> 2) * PSB + (i & 3) * 8;
                *(u32x2*)rec = u32x2{h01, h23};
                *(u32x2*)(rec + 32) = u32x2{l01, l23};
            }
        }
    };

#pragma unroll
    for (int c = 0; c < NCH; ++c) load_a(c);
    // accumulators start at the (scaled) bias
    const float acc_mul = pow2f(kw + kx), out_mul = pow2f(-(kw + kx));
    f32x16 acc[2];
#pragma unroll
    for (int G = 0; G < 4; ++G) {
        const f32x4 b4 = *(const f32x4*)(a.bias + cur.cb * 96 + 32 * cot + 8 * G + 4 * h) * acc_mul;
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[0][4 * G + e] = acc[1][4 * G + e] = b4[e];
    }
    split_store_a(0);
    // LDS-only barrier (__syncthreads() would also wait for the pixel loads that are meant to stay in flight)
    auto lds_barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);                      // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    const char* const Ax = ldsA + (w * HWP + p) * PSB + 16 * h;
    const char* const Wl = lds + S_W_OFF + lane * 16;
    struct SFrags { h8 wh, wl, xh[2], xl[2]; };
    auto load_sfrags = [&](SFrags& f, int c, int tap) {
        const int ky = tap / 3, kxx = tap - 3 * ky;
        const char* ab = Ax + (ky * HWP + kxx) * PSB;
        const char* wb = Wl + ((c * 3 + ky) * 6 + kxx * 2) * 1024;
        f.wh = *(const h8*)wb;
        f.wl = *(const h8*)(wb + 1024);
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            f.xh[mt] = *(const h8*)(ab + mt * 32 * PSB);
            f.xl[mt] = *(const h8*)(ab + mt * 32 * PSB + 32);
        }
    };
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        // chunk c's pixels are in LDS; for c = 0 every weight copy of this wave has to have landed too: the copies are the
        // oldest vector-memory operations of the wave and return in order, so vmcnt(35) -- everything but the pixel loads of
        // chunks 1 .. 5 -- covers them
        if (c == 0) __builtin_amdgcn_s_waitcnt(0x8F73);
        lds_barrier();
        SFrags f[2];
        load_sfrags(f[0], c, 0);
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            if (tap + 1 < 9) load_sfrags(f[(tap + 1) & 1], c, tap + 1);
            const SFrags& g = f[tap & 1];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt] = MFMA_H(g.wl, g.xh[mt], acc[mt]);
                acc[mt] = MFMA_H(g.wh, g.xl[mt], acc[mt]);
                acc[mt] = MFMA_H(g.wh, g.xh[mt], acc[mt]);
            }
        }
        if (c + 1 < NCH) {
            lds_barrier();                                       // every wave is past its reads of chunk c
            split_store_a(c + 1);
        }
    }

    // ---- epilogue (as above, for this workgroup's co tile) ----
    const int y = cur.ty * TH + w, x0 = cur.tx * TW;
    float* out = (float*)a.out;
    constexpr int PXS = SHUFFLE ? 2 * kC : kC;
    if (SHUFFLE)
        out += (((size_t)cur.n * 2 * HW + 2 * y + (cur.cb >> 1)) * 2 * HW + 2 * (x0 + p) + (cur.cb & 1)) * kC + 4 * h;
    else
        out += (((size_t)cur.n * HW + y) * HW + x0 + p) * kC + 4 * h;
    const float* res = nullptr;
    if (RES) res = (const float*)a.res + (((size_t)cur.n * HW + y) * HW + x0 + p) * kC + 4 * h + 32 * cot;
    float vmax = 0.f;
    float psum[16];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            f32x4 v = {acc[mt][4 * G], acc[mt][4 * G + 1], acc[mt][4 * G + 2], acc[mt][4 * G + 3]};
            v *= out_mul;
            if (act) {
                const f32x4 s4 = *(const f32x4*)(a.slope + cur.cb * 96 + 32 * cot + 8 * G + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * s4[e];
            }
            if (RES) v += *(const f32x4*)(res + (size_t)(32 * mt) * kC + 8 * G);
            vmax = fmaxf(fmaxf(vmax, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
            if (POOL) {
#pragma unroll
                for (int e = 0; e < 4; ++e) psum[4 * G + e] = mt == 0 ? v[e] : psum[4 * G + e] + v[e];
            }
            *(f32x4*)(out + (size_t)(32 * mt) * PXS + 32 * cot + 8 * G) = v;
        }
    }
    if (POOL) {                                                   // this tile's 32 channel sums, in the order of the kernel above
        float* P = ldsPool;
        float (&t)[16] = psum;
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
            P[w * 32 + (q & 3) + 8 * (q >> 2) + 4 * h] = t[0];
        }
        __syncthreads();
        if (tid < 32) {
            const float s = (P[tid] + P[32 + tid]) + (P[64 + tid] + P[96 + tid]);
            a.pool_partial[((size_t)cur.n * tiles_img + cur.ty * tiles_x + cur.tx) * kC + 32 * cot + tid] = s;
        }
    }
    if (a.out_amax) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax == vmax) atomicMax((unsigned*)a.out_amax + cur.n, __float_as_uint(vmax));
    }
}

template <int HW, bool SHUFFLE, bool POOL, bool RES>
hipError_t launch_small(const Conv3x3Args& a, int nitems, hipStream_t s)
{
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv3x3_f32s_small_kernel<HW, SHUFFLE, POOL, RES>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, S_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL((conv3x3_f32s_small_kernel<HW, SHUFFLE, POOL, RES>), dim3((unsigned)(nitems * 3)), dim3(256), S_LDS_BYTES,
                       s, a, nitems);
    return hipGetLastError();
}

template <int HW, bool SHUFFLE, bool POOL, bool RES>
hipError_t launch_one(const Conv3x3Args& a, hipStream_t s)
{
    const int nitems = a.N * (HW / TH) * (HW / TW) * (SHUFFLE ? 4 : 1);     // multiple of 16
    const int grid = nitems < MAX_WGS ? nitems & ~7 : MAX_WGS;
    if (a.in_scale || a.res_scale) {                  // a folded channel attention (persistent kernel, three terms only)
        if ((a.in_scale && a.res_scale) || a.terms != 3 || nitems < kSmallItems || HW != kLR || SHUFFLE || POOL) return hipErrorInvalidValue;
        if constexpr (HW == kLR && !SHUFFLE && !POOL) {
            if (a.in_scale)
                hipLaunchKernelGGL((conv3x3_f32s_kernel<HW, false, false, RES, 0, 3, 1>), dim3((unsigned)grid), dim3(256), LDS_BYTES, s, a, nitems);
            else if (RES)
                hipLaunchKernelGGL((conv3x3_f32s_kernel<HW, false, false, RES, 0, 3, 2>), dim3((unsigned)grid), dim3(256), LDS_BYTES, s, a, nitems);
            else
                return hipErrorInvalidValue;
            return hipGetLastError();
        }
        return hipErrorInvalidValue;
    }
    if (a.terms == 3 && nitems < kSmallItems) return launch_small<HW, SHUFFLE, POOL, RES>(a, nitems, s);
    if (a.terms == 1) {
        if (POOL) return hipErrorInvalidValue;        // the f16-product variant serves the training operators only
        hipLaunchKernelGGL((conv3x3_f32s_kernel<HW, SHUFFLE, false, RES, 0, 1>), dim3((unsigned)grid), dim3(256), LDS_BYTES,
                           s, a, nitems);
        return hipGetLastError();
    }
    hipLaunchKernelGGL((conv3x3_f32s_kernel<HW, SHUFFLE, POOL, RES>), dim3((unsigned)grid), dim3(256), LDS_BYTES,
                       s, a, nitems);
    return hipGetLastError();
}

}  // namespace c3s

hipError_t launch_conv3x3_f32_split(const Conv3x3Args& a, hipStream_t s)
{
    using namespace c3s;
    if (a.H != a.W || a.N <= 0 || (a.terms != 1 && a.terms != 3)) return hipErrorInvalidValue;
    const bool res = a.res != nullptr, pool = a.pool_partial != nullptr;
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
