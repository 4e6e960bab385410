// srgen_wgrad_f32s.hip -- gfx950 weight gradient of the 3x3 conv 96 -> 96 on the split arithmetic (fp32 tensors, fp32
// accumulation, products from two-term f16 splits on v_mfma_f32_32x32x16_f16; see srgen_conv3x3_f32s.hip).
//
// Reference site: what autograd does for F.conv2d(x, w, b, padding=1) in loss.backward() (train.py:95, Generator.py:34,37,62,69):
//     dW[co][ci][ky][kx] = sum over pixels dY[n][y][x][co] * X[n][y+ky-1][x+kx-1][ci]        (zero outside the image)
// Per tap a GEMM with M = co, N = ci and K = pixels: 9 x 3 = 27 f16 MFMA units per (co, ci, pixel) against the 64 of the
// Winograd-domain kernel on the fp32 MFMA (srgen_train_kernels.hip).
//
// K = pixels is the slow index of both NHWC operands, and the MFMA wants 8 consecutive K per lane: the operands are
// staged in LDS as [pixel][channel] f16 images (high and low parts apart) and read TRANSPOSED with ds_read_b64_tr_b16
// (a 16-lane group fetches 4 pixels x 16 channels and every lane receives one channel's 4 pixels; layout probed with
// tools/trprobe.hip).  Row pitches 192 B (dY: 96 channels) and 64 B (X: 32 channels) are 16 and 48 dwords mod 64: the four
// rows of a block fall on disjoint banks.
//
// Workgroup = (row slice, ci tile of 32, kernel row ky): 9 accumulator tiles of 32 x 32 per wave (3 kx x 3 co tiles,
// 144 registers); the four waves split K -- wave w takes pixels 16 w .. 16 w + 15 of every 64-pixel row segment -- and
// their partial sums meet in LDS at the end (fixed order).  Per segment: dY row (64 px x 96 co) and the one X row this
// ky needs (66 px x 32 ci) are fetched into registers a segment ahead (waves 0-2: dY, wave 3: X), split between the
// MFMAs (high part by truncation, low part = the rest, rounded), written to the other LDS buffer; one barrier per
// segment.  dY is a gradient (order 1e-6): scaled by a power of two from a bound of its maximum (dy_amax, as the
// data-gradient conv's in_amax) before the split, the sums scaled back.  Partial dW -> slab [slice][tap][co][ci], summed
// over slices in fixed order and transposed to OIHW by conv3x3_wgrad_reduce_kernel; the bias gradient (sum of dY) is
// accumulated from the staged values by the (ci tile 0, ky 1) workgroups.
#include "srgen_internal.h"

#include <cstdlib>

namespace srgen {
namespace wgs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(8)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int DY_PITCH = 192, X_PITCH = 64, X_ROWS = 66;
constexpr int DYH = 0, DYL = 64 * DY_PITCH, XH = 2 * 64 * DY_PITCH, XL = XH + X_ROWS * X_PITCH;
constexpr int BUF_BYTES = XL + X_ROWS * X_PITCH;              // 33,024
constexpr int LDS_BYTES = 2 * BUF_BYTES;                      // 66,048: two workgroups per CU
constexpr int RED_BYTES = 9 * 32 * 32 * 4;                    // 36,864: the four waves' partial tiles meet here
static_assert(RED_BYTES + 192 * 4 * 4 <= LDS_BYTES, "reduction buffers fit the staging buffers");
static_assert(kAmaxSlots == 4 * 256, "one float4 of partial maxima per thread");

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ h4 tr_read(const char* p)
{
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p));
}
__device__ __forceinline__ h8 cat(h4 a, h4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

// TERMS = 3: split products (hi hi + hi lo + lo hi); TERMS = 1: products of the operands rounded to f16 (the mixed-precision
// training step: the counterpart of the reference's autocast convs, train.py:91), a third of the MFMAs and no low parts
template <int TERMS>
__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_split_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                     int total_rows, int H, int W, int rows_per_slice,
                                                                     const float* __restrict__ dy_amax,
                                                                     float* __restrict__ slab, float* __restrict__ bias_slab,
                                                                     int abl)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the nine workgroups of a slice share an XCD (its L2 serves the slice's dY nine times, its X three times)
    const int xcd = blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int type = j % 9, slice = (j / 9) * 8 + xcd;
    const int ct = type % 3, ky = type / 3;
    const int nseg = W >> 6;

    // scale of dY: 2^kd with amax * 2^kd in [2^13, 2^14)  (srgen_internal.h: split_exp)
    int kd = 0;
    if (dy_amax) {
        const f32x4 q4 = *(const f32x4*)(dy_amax + tid * 4);
        float m = fmaxf(fmaxf(q4.x, q4.y), fmaxf(q4.z, q4.w));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* red = (float*)lds;
        if (lane == 0) red[w] = m;
        __syncthreads();
        kd = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        __syncthreads();
    }
    const float in_mul = pow2f(kd), out_mul = pow2f(-kd);

    f32x16 acc[3][3];                                 // [kx][co tile]
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};                // waves 0-2: sum of dY over this thread's pixels, its channel quad

    const int r0 = slice * rows_per_slice;
    const int r1 = min(r0 + rows_per_slice, total_rows);
    const int nunits = max(r1 - r0, 0) * nseg;
    // unit u = (row r0 + u / nseg, segment u % nseg); a unit whose X row (y + ky - 1) lies outside the image adds nothing
    auto valid = [&](int u) {
        const int yy = (r0 + u / nseg) % H + ky - 1;
        return (unsigned)yy < (unsigned)H;
    };
    auto next_valid = [&](int u) {
        while (u < nunits && !valid(u)) ++u;
        return u;
    };

    // ---- staging: waves 0-2 the dY row segment (thread: channel quad t % 24, pixels t / 24 + 8 it), wave 3 the X row
    // (lane: channel quad l % 8 of the ci tile, halo pixels l / 8 + 8 it of 66) ----
    f32x4 stage[9];
    const bool is_dy = w < 3;
    const bool do_bias = ct == 0 && ky == 1;          // the workgroup type that sees every row of the slice
    const int sq = is_dy ? tid % 24 : lane & 7, sp0 = is_dy ? tid / 24 : lane >> 3;
    auto load_unit = [&](int u) {
        const int row = (abl & 1) ? r0 + 1 : r0 + u / nseg, seg = u % nseg;      // abl: timing experiments only
        if (is_dy) {
            const float* src = dy + ((size_t)row * W + seg * 64) * 96;      // wave-uniform; the thread's part is 32 bits
            const int toff = sp0 * 96 + 4 * sq;
#pragma unroll
            for (int it = 0; it < 8; ++it) stage[it] = *(const f32x4*)(src + it * 8 * 96 + toff);
        } else {
            const float* src = x + (size_t)(row + ky - 1) * W * 96 + 32 * ct + 4 * sq;
#pragma unroll
            for (int it = 0; it < 9; ++it) {
                const int xx = min(max(seg * 64 + sp0 + 8 * it - 1, 0), W - 1);
                stage[it] = *(const f32x4*)(src + (size_t)xx * 96);
            }
        }
    };
    // split one staged piece and write it to buffer b (high part: 11 significant bits by truncation; low part: the
    // rest, rounded) -- converted pieces go to LDS at once, so only the fp32 pieces occupy registers
    auto split_store = [&](f32x4 v, char* rec, int hi_off, int lo_off) {
        h4 hi4, lo4;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (TERMS == 3) {
                const float hi = __uint_as_float(__float_as_uint(v[e]) & 0xFFFFE000u);
                hi4[e] = (_Float16)hi;
                lo4[e] = (_Float16)(v[e] - hi);
            } else {
                hi4[e] = (_Float16)v[e];              // rounded to nearest
            }
        }
        *(h4*)(rec + hi_off) = hi4;
        if (TERMS == 3) *(h4*)(rec + lo_off) = lo4;
    };
    auto convert = [&](int u, int it, int b) {        // it < 8 for dY, < 9 for X
        char* base = lds + b * BUF_BYTES;
        if (is_dy) {
            if (do_bias) bsum += stage[it];
            split_store(stage[it] * in_mul, base + (sp0 + 8 * it) * DY_PITCH + sq * 8, DYH, DYL);
        } else {
            const int r = sp0 + 8 * it;
            const int xx = (u % nseg) * 64 + r - 1;
            const bool ok = (it != 0 && it != 8) || (unsigned)xx < (unsigned)W;      // only halo pixels 0 and 65 can be outside
            if (r < X_ROWS) split_store(ok ? stage[it] : f32x4{0.f, 0.f, 0.f, 0.f}, base + r * X_PITCH + sq * 8, XH, XL);
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // this lane's transposed-read addresses: group g = lane >> 4 reads pixels 8 (g >> 1) + 4 t + {0..3}, channels
    // 16 (g & 1) + {0..15}; lane 4 q + p of the group supplies pixel q, channels 4 p .. 4 p + 3 (8 bytes)
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
    const int pix0 = 16 * w + 8 * (g >> 1) + q;
    const int offA = pix0 * DY_PITCH + (16 * (g & 1) + 4 * p4) * 2;
    const int offB = pix0 * X_PITCH + (16 * (g & 1) + 4 * p4) * 2;

    // pipeline: LDS holds unit u (being multiplied) and receives unit un (split from the registers under the MFMAs);
    // the registers then take the loads of the unit after that, issued at the end of the iteration: a load has about
    // one iteration of MFMAs to land before it is looked at
    int u = next_valid(0);
    int un = nunits;
    if (u < nunits) {
        load_unit(u);
#pragma unroll
        for (int it = 0; it < 9; ++it)
            if (it < 8 || !is_dy) convert(u, it, 0);
        un = next_valid(u + 1);
        if (un < nunits) load_unit(un);
    }
    barrier();

    int b = 0;
    while (u < nunits) {
        const bool more = un < nunits;
        const char* base = lds + b * BUF_BYTES;
        // B fragments (X, three taps), then per co tile its A fragments and nine MFMAs
        h8 bh[3], bl[3];
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const char* pb = base + offB + kx * X_PITCH;
            bh[kx] = cat(tr_read(pb + XH), tr_read(pb + XH + 4 * X_PITCH));
            if (TERMS == 3) bl[kx] = cat(tr_read(pb + XL), tr_read(pb + XL + 4 * X_PITCH));
        }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const char* pa = base + offA + c * 64;
            const h8 ah = cat(tr_read(pa + DYH), tr_read(pa + DYH + 4 * DY_PITCH));
            h8 al;
            if (TERMS == 3) al = cat(tr_read(pa + DYL), tr_read(pa + DYL + 4 * DY_PITCH));
            if (more && !(abl & 2)) {                 // a third of the next unit's split under these MFMAs
#pragma unroll
                for (int k = 0; k < 3; ++k)
                    if (3 * c + k < 8 || !is_dy) convert(un, 3 * c + k, b ^ 1);
            }
            if (!(abl & 4))
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                if (TERMS == 3) {
                    acc[kx][c] = MFMA_H(al, bh[kx], acc[kx][c]);
                    acc[kx][c] = MFMA_H(ah, bl[kx], acc[kx][c]);
                }
                acc[kx][c] = MFMA_H(ah, bh[kx], acc[kx][c]);
            }
        }
        const int unn = more ? next_valid(un + 1) : nunits;
        if (unn < nunits) load_unit(unn);
        barrier();                                    // (the other buffer was last read one unit ago: a barrier has passed)
        b ^= 1;
        u = un;
        un = unn;
    }

    // ---- the four K quarters meet in LDS (fixed order), then the partial tiles go to the slab ----
    float* red = (float*)lds;                         // [tile 9][co 32][ci 32]
    float* bred = red + 9 * 1024;                     // [192][4]
    const int hh = lane >> 5, pn = lane & 31;
    for (int r = 0; r < 4; ++r) {
        if (w == r) {
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
#pragma unroll
                for (int c = 0; c < 3; ++c)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        float* d = red + ((kx * 3 + c) * 32 + acc_row(i, hh)) * 32 + pn;
                        const float v = acc[kx][c][i] * out_mul;
                        *d = r == 0 ? v : *d + v;
                    }
        }
        if (r == 0 && is_dy) *(f32x4*)(bred + tid * 4) = bsum;
        __syncthreads();
    }
    float* o = slab + (size_t)slice * 9 * 9216;
    for (int i = tid; i < 9 * 32 * 8; i += 256) {     // float4 pieces: tile, co row, ci quad
        const int t9 = i >> 8, m = (i >> 3) & 31, n4 = i & 7;
        const int kx = t9 / 3, c = t9 % 3;
        *(f32x4*)(o + ((size_t)(ky * 3 + kx) * 96 + 32 * c + m) * 96 + 32 * ct + 4 * n4) = *(const f32x4*)(red + (t9 * 32 + m) * 32 + 4 * n4);
    }
    if (do_bias && tid < 96) {             // every row of the slice went through this workgroup
        const int quad = tid >> 2, e = tid & 3;
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) s += bred[(quad + 24 * k) * 4 + e];
        bias_slab[(size_t)slice * 96 + tid] = s;
    }
}

}  // namespace wgs

// ---- f16 products (TERMS = 1 of the training step's mixed-precision mode): a kernel of its own ---------------------------
// With one MFMA per product the kernel above is bound by its staging, not by the matrix pipe: its nine workgroup types
// per slice split the same dY row nine times and every X row three times (1.2 GB through the vector memory path per
// 64-image gradient, 150 us).  Here a workgroup = (slice, ci tile of 32) keeps ALL THREE kernel rows: wave ky = 0..2
// owns the 9 accumulator tiles (3 kx x 3 co tiles) of its kernel row and runs the whole K = 64 pixels of a unit (4 MFMA
// K steps), wave 3 only stages.  Per unit (one 64-pixel row segment of dY) the workgroup stages the dY row once (waves
// 0-2, pieces as above) and ONE new X row (wave 3) into a ring of four rows -- unit row R multiplies X rows R-1, R, R+1
// while row R+2 is written -- so dY is split three times per gradient and X once: a third of the staging for the same
// MFMAs, no K split over waves, no cross-wave reduction at the end.
// Two workgroups per CU, 80 slices x 3 = 240 workgroups (the slab the reduction reads stays small): 113 us + 23 us of
// reduction per 64-image gradient against 163 + 18 for TERMS = 1 of the kernel above.  The loop still runs near the
// memory latency (2 us per unit): a unit is 36 MFMAs per wave, shorter than a trip to HBM.  Tried: one workgroup per CU
// with three staging register sets (loads issued two units ahead): 128 us -- the compiler's vmcnt(0) at the loop head
// and, once the body was made branch-free, its hoisting of every load in sight (86 spilled registers) undid the depth.
// Slab layout, scaling and bias sums as above.
namespace wg1 {
using namespace wgs;

constexpr int DY_BYTES = 64 * DY_PITCH;                       // 12,288: one dY row segment, high parts
constexpr int X_BYTES = 72 * X_PITCH;                         // 4,608: one X row (66 halo pixels x 32 ci; 72 are written)
constexpr int XR = 2 * DY_BYTES;                              // ring of four X rows behind the two dY buffers
constexpr int XZ = XR + 4 * X_BYTES;                          // a row of zeros: what a kernel row outside the image multiplies
constexpr int LDS1_BYTES = XZ + X_BYTES;                      // 47,616
static_assert(192 * 4 * 4 <= LDS1_BYTES, "bias reduction fits");

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_f16_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                   int total_rows, int H, int W, int rows_per_slice,
                                                                   const float* __restrict__ dy_amax,
                                                                   float* __restrict__ slab, float* __restrict__ bias_slab)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // the three workgroups of a slice share an XCD (its L2 serves the slice's dY three times)
    const int xcd = blockIdx.x & 7, j = (int)blockIdx.x >> 3;
    const int ct = j % 3, slice = (j / 3) * 8 + xcd;
    const int nseg = W >> 6;

    int kd = 0;                                       // scale of dY: 2^kd with amax * 2^kd in [2^13, 2^14)
    if (dy_amax) {
        const f32x4 q4 = *(const f32x4*)(dy_amax + tid * 4);
        float m = fmaxf(fmaxf(q4.x, q4.y), fmaxf(q4.z, q4.w));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        float* red = (float*)lds;
        if (lane == 0) red[w] = m;
        __syncthreads();
        kd = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
        __syncthreads();
    }
    const float in_mul = pow2f(kd), out_mul = pow2f(-kd);

    f32x16 acc[3][3];                                 // [kx][co tile] of this wave's kernel row (waves 0-2)
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};

    const int r0 = min(slice * rows_per_slice, total_rows);
    const int r1 = min(r0 + rows_per_slice, total_rows);
    const bool is_dy = w < 3;
    const bool do_bias = ct == 0;
    const int ky = w;                                 // waves 0-2
    const int sq = is_dy ? tid % 24 : lane & 7, sp0 = is_dy ? tid / 24 : lane >> 3;

    // ---- staging: waves 0-2 the dY row segment of unit (row, seg), wave 3 X row row + 1 (the row that unit's
    // successor is the first to need), eight / nine 16-byte pieces per thread ----
    f32x4 st0[9];
    auto store_h4 = [&](f32x4 v, char* rec) {
        h4 r;
#pragma unroll
        for (int e = 0; e < 4; ++e) r[e] = (_Float16)v[e];    // rounded to nearest
        *(h4*)rec = r;
    };
    // (no branches below the wave level in here: the unit's MFMAs, fragment reads and splits are one basic block)
    const float bias_mul = do_bias ? 1.f : 0.f;
#define WG1_STAGE_OPS(LOAD, CONVERT, ARR)                                                                                 \
    auto LOAD = [&](int row, int seg) {                                                                                   \
        if (is_dy) {                                                                                                      \
            const float* src = dy + ((size_t)min(row, total_rows - 1) * W + seg * 64) * 96 + sp0 * 96 + 4 * sq;           \
            _Pragma("unroll") for (int it = 0; it < 8; ++it) ARR[it] = *(const f32x4*)(src + it * 8 * 96);                 \
        } else {                          /* any row: clamped into the tensor (a row nobody multiplies) */                \
            const int rr = min(max(row + 1, 0), total_rows - 1);                                                          \
            const float* src = x + (size_t)rr * W * 96 + 32 * ct + 4 * sq;                                                \
            _Pragma("unroll") for (int it = 0; it < 9; ++it) {                                                            \
                const int xx = min(max(seg * 64 + sp0 + 8 * it - 1, 0), W - 1);                                           \
                ARR[it] = *(const f32x4*)(src + (size_t)xx * 96);                                                         \
            }                                                                                                             \
        }                                                                                                                 \
    };                                                                                                                    \
    /* piece `it` (< 8) of dY unit (row, seg) -> dY buffer b */                                                           \
    auto CONVERT##_dy = [&](int it, int b, float bm) {      /* bm: 1 where this unit counts for the bias sums, else 0 */  \
        bsum += ARR[it] * bm;                                                                                             \
        store_h4(ARR[it] * in_mul, lds + b * DY_BYTES + (sp0 + 8 * it) * DY_PITCH + sq * 8);                              \
    };                                                                                                                    \
    /* piece `it` (< 9) of X row row + 1 -> its ring slot (halo pixels 66 .. 71 are written and never read) */            \
    auto CONVERT##_x = [&](int it, int row, int seg) {                                                                    \
        const int r = sp0 + 8 * it;                                                                                       \
        const int xx = seg * 64 + r - 1;                                                                                  \
        const bool ok = (it != 0 && it != 8) || (unsigned)xx < (unsigned)W;     /* only halo pixels 0, 65 can be outside */ \
        store_h4(ok ? ARR[it] : f32x4{0.f, 0.f, 0.f, 0.f}, lds + XR + ((row + 1) & 3) * X_BYTES + r * X_PITCH + sq * 8);  \
    }
    WG1_STAGE_OPS(load0, convert0, st0);
#undef WG1_STAGE_OPS
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // transposed-read addresses (see above); K step ks covers pixels 16 ks .. 16 ks + 15
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
    const int pixq = 8 * (g >> 1) + q;
    const int offA = pixq * DY_PITCH + (16 * (g & 1) + 4 * p4) * 2;
    const int offB = pixq * X_PITCH + (16 * (g & 1) + 4 * p4) * 2;

    // one unit of a wave ky: multiply dY(R) (buffer b) with X row R + ky - 1 while `cv` splits the next unit's dY into
    // buffer b ^ 1 (unconditionally: behind the slice's last unit it writes a row nobody reads)
    auto unit_dy = [&](int R, int b, auto&& cv) {
        const float bm = R + 1 < r1 ? bias_mul : 0.f;             // (the row behind the slice is not this slice's)
        const int yy = R % H + ky - 1;
        const bool mul = (unsigned)yy < (unsigned)H;              // else: the row of zeros
        const char* pa0 = lds + b * DY_BYTES + offA;
        const char* pb0 = lds + (mul ? XR + ((R + ky - 1) & 3) * X_BYTES : XZ) + offB;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            h8 bh[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const char* pb = pb0 + (16 * ks + kx) * X_PITCH;
                bh[kx] = cat(tr_read(pb), tr_read(pb + 4 * X_PITCH));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const char* pa = pa0 + 16 * ks * DY_PITCH + c * 64;
                const h8 ah = cat(tr_read(pa), tr_read(pa + 4 * DY_PITCH));
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) acc[kx][c] = MFMA_H(ah, bh[kx], acc[kx][c]);
            }
            cv(2 * ks, b ^ 1, bm);                    // a quarter of the next unit's dY under these MFMAs
            cv(2 * ks + 1, b ^ 1, bm);
            __builtin_amdgcn_sched_barrier(0);        // (fragments of one K step at a time: registers)
        }
    };

    for (int i = tid; i < X_BYTES / 4; i += 256) ((float*)(lds + XZ))[i] = 0.f;
    for (int seg = 0; seg < nseg && r0 < r1; ++seg) {
        // ---- a pass over the slice's rows.  Prologue: dY(r0) and X rows r0-1, r0, r0+1 into LDS, the loads of the next
        // unit on their way.  Waves 0-2 and wave 3 run loops of their own with the same number of barriers ----
        __syncthreads();                              // (the previous pass has finished reading)
        if (is_dy) {
            load0(r0, seg);
#pragma unroll
            for (int it = 0; it < 8; ++it) convert0_dy(it, 0, bias_mul);
            load0(r0 + 1, seg);
            barrier();
            int b = 0;
            for (int R = r0; R < r1; ++R) {
                unit_dy(R, b, convert0_dy);
                load0(R + 2, seg);
                barrier();
                b ^= 1;
            }
        } else {
            for (int rr = r0 - 2; rr <= r0; ++rr) {   // X rows r0-1, r0, r0+1
                load0(rr, seg);
#pragma unroll
                for (int it = 0; it < 9; ++it) convert0_x(it, rr, seg);
            }
            load0(r0 + 1, seg);
            barrier();
            for (int R = r0; R < r1; ++R) {           // unit R: X row R+2 into the slot row R-2 has left
#pragma unroll
                for (int it = 0; it < 9; ++it) convert0_x(it, R + 1, seg);
                load0(R + 2, seg);
                barrier();
            }
        }
    }

    // ---- partial tiles -> slab [slice][tap][co][ci]; bias partials ----
    const int hh = lane >> 5, pn = lane & 31;
    if (is_dy) {
        float* o = slab + (size_t)slice * 9 * 9216;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    o[((size_t)(ky * 3 + kx) * 96 + 32 * c + acc_row(i, hh)) * 96 + 32 * ct + pn] = acc[kx][c][i] * out_mul;
    }
    if (do_bias) {
        float* bred = (float*)lds;                    // [192][4]
        __syncthreads();
        if (is_dy) *(f32x4*)(bred + tid * 4) = bsum;
        __syncthreads();
        if (tid < 96) {
            const int quad = tid >> 2, e = tid & 3;
            float s = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) s += bred[(quad + 24 * k) * 4 + e];
            bias_slab[(size_t)slice * 96 + tid] = s;
        }
    }
}

}  // namespace wg1

// slices: a multiple of 8 (XCD mapping), at most 112; 56 x 9 = 504 workgroups = one round at two per CU (measured at 64
// images: 56 slices 240 us per gradient with its reduction, 80: 281, 112: 261)
constexpr int kWgradMaxSlices = kWgradSplitMaxSlices;
int conv3x3_wgrad_f16_slices(int total_rows);

int conv3x3_wgrad_split_slices(int total_rows)
{
    const char* e = getenv("SRGEN_WGRAD_SPLIT_SLICES");
    int s = e ? atoi(e) & ~7 : 56;
    if (s < 8 || s > 112) s = 112;
    while (s > 8 && total_rows < s) s -= 8;
    return s;
}

// the f16-product kernel: 3 workgroups per slice, one workgroup per CU: 80 slices = 240 workgroups
int conv3x3_wgrad_f16_slices(int total_rows)
{
    const char* e = getenv("SRGEN_WGRAD_F16_SLICES");
    int s = e ? atoi(e) & ~7 : 80;
    if (s < 8 || s > kWgradMaxSlices) s = kWgradMaxSlices;
    while (s > 8 && total_rows < s) s -= 8;
    return s;
}

size_t conv3x3_wgrad_split_scratch_bytes(int n, int h)
{
    (void)n; (void)h;
    return ((size_t)kWgradMaxSlices * 9 * 9216 + (size_t)kWgradMaxSlices * 96) * sizeof(float);        // the largest slice count
}

hipError_t launch_conv3x3_wgrad_split_partial(const float* x, const float* dy, int n, int h, const float* dy_amax,
                                              float* slab, float* bias_slab, int* slices_out, hipStream_t s, int terms)
{
    using namespace wgs;
    if (n <= 0 || h <= 0 || (h & 63) || (terms != 1 && terms != 3)) return hipErrorInvalidValue;
    const int total_rows = n * h;
    if (terms == 1) {                                 // f16 products: the (slice, ci tile) kernel
        const int slices1 = conv3x3_wgrad_f16_slices(total_rows);
        const int rps1 = (total_rows + slices1 - 1) / slices1;
        hipLaunchKernelGGL(wg1::conv3x3_wgrad_f16_kernel, dim3((unsigned)(slices1 * 3)), dim3(256), wg1::LDS1_BYTES, s, x, dy,
                           total_rows, h, h, rps1, dy_amax, slab, bias_slab);
        *slices_out = slices1;
        return hipGetLastError();
    }
    const int slices = conv3x3_wgrad_split_slices(total_rows);
    const int rps = (total_rows + slices - 1) / slices;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv3x3_wgrad_split_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int abl = getenv("SRGEN_WGRAD_ABL") ? atoi(getenv("SRGEN_WGRAD_ABL")) : 0;
    hipLaunchKernelGGL(conv3x3_wgrad_split_kernel<3>, dim3((unsigned)(slices * 9)), dim3(256), LDS_BYTES, s, x, dy, total_rows,
                       h, h, rps, dy_amax, slab, bias_slab, abl);
    *slices_out = slices;
    return hipGetLastError();
}

}  // namespace srgen
