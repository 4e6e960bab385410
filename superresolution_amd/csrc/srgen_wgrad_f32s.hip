// srgen_wgrad_f32s.hip -- gfx950 weight gradient of the 3x3 conv 96 -> 96 on the f16 matrix instruction: fp32 tensors, fp32
// accumulation, products either from two-term f16 splits of both operands (TERMS = 3, the fp32 training step; see
// srgen_conv3x3_f32s.hip) or of the operands rounded to f16 (TERMS = 1, the mixed-precision step: the counterpart of the
// reference's autocast convs, train.py:91), on v_mfma_f32_32x32x16_f16.
//
// Reference site: what autograd does for F.conv2d(x, w, b, padding=1) in loss.backward() (train.py:95, Generator.py:34,37,62,69):
//     dW[co][ci][ky][kx] = sum over pixels dY[n][y][x][co] * X[n][y+ky-1][x+kx-1][ci]        (zero outside the image)
// Per tap a GEMM with M = co, N = ci and K = pixels: 9 x 3 = 27 (TERMS = 3) or 9 f16 MFMA units per (co, ci, pixel) against
// the 64 of the Winograd-domain kernel on the fp32 MFMA (srgen_train_kernels.hip).
//
// K = pixels is the slow index of both NHWC operands, and the MFMA wants 8 consecutive K per lane: the operands are
// staged in LDS as [pixel][channel] f16 images (high and low parts apart) and read TRANSPOSED with ds_read_b64_tr_b16
// (a 16-lane group fetches 4 pixels x 16 channels and every lane receives one channel's 4 pixels; layout probed with
// tools/trprobe.hip).  Row pitches 192 B (dY: 96 channels) and 64 B (X: 32 channels) are 16 and 48 dwords mod 64: the four
// rows of a block fall on disjoint banks.
//
// Workgroup = (slice of rows, ci tile of 32) with ALL THREE kernel rows: wave ky = 0..2 owns the 9 accumulator tiles
// (3 kx x 3 co tiles of 32 x 32, 144 registers) of its kernel row and runs the whole K = 64 pixels of a unit (a 64-pixel
// row segment of dY: 4 MFMA K steps), wave 3 only stages.  Per unit the workgroup stages the dY row once (waves 0-2,
// eight 16-byte pieces per thread, fetched a unit ahead and split between the MFMAs into the other of two buffers)
// and ONE new X row (wave 3, nine pieces per lane) into a ring of four rows -- unit row R multiplies X rows R-1, R, R+1
// while row R+2 is written; a kernel row outside the image multiplies a row of zeros, so the unit body has no branch.
// One barrier per unit, no K split over waves, no cross-wave reduction.  dY is a gradient (order 1e-6): scaled by a
// power of two from a bound of its maximum (dy_amax, as the data-gradient conv's in_amax) before the split, the sums
// scaled back.  Partial dW -> slab [slice][tap][co][ci], summed over slices in fixed order and transposed to OIHW by
// conv3x3_wgrad_reduce_kernel; the bias gradient (sum of dY) is accumulated from the staged values by the ci tile 0
// workgroups.
//
// History (64-image gradient, with the reduction).  The first kernel of this file had workgroup = (slice, ci tile, ky),
// the four waves splitting K: nine workgroup types per slice split the same dY row nine times and every X row three
// times (1.2 GB through the vector memory path): 222 us (TERMS = 3) / 181 us (TERMS = 1, bound by staging alone).  This
// one splits dY three times and X once: 137 us / 69 us.  TERMS = 1: two workgroups per CU (48 KB of LDS); TERMS = 3: one
// (95 KB; a unit is 108 MFMAs per wave, about one trip to HBM).  Tried for TERMS = 1 and dropped: one workgroup per CU
// with three staging register sets (loads two units ahead): 128 us -- the compiler's vmcnt(0) at the loop head and, once
// the body was branch-free, its hoisting of every load in sight (86 spilled registers) undid the depth.
#include "srgen_internal.h"

#include <cstdlib>
#include <type_traits>

namespace srgen {
namespace wgs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef short s4v __attribute__((__vector_size__(8)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
#define MFMA_B(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf8, (a)), __builtin_bit_cast(bf8, (b)), (c), 0, 0, 0)

constexpr int DY_PITCH = 192, X_PITCH = 64, X_ROWS = 66;
static_assert(kAmaxSlots == 4 * 256, "one float4 of partial maxima per thread");

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ h4 tr_read(const char* p)
{
    return __builtin_bit_cast(h4, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p));
}
__device__ __forceinline__ h8 cat(h4 a, h4 b) { return __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7); }

}  // namespace wgs

namespace wg1 {
using namespace wgs;

// TERMS = 3 (split products: the fp32 step) keeps a low-part image behind every high-part image: twice the LDS, one
// workgroup per CU, three MFMAs per product -- a unit then is 108 MFMAs per wave, about one trip to HBM
constexpr int DY_HALF = 64 * DY_PITCH;                        // 12,288: one dY row segment, high (or low) parts
constexpr int X_HALF = 72 * X_PITCH;                          // 4,608: one X row (66 halo pixels x 32 ci; 72 are written)
template <int TERMS> struct Lds {
    static constexpr int DY_BYTES = (TERMS == 3 ? 2 : 1) * DY_HALF;
    static constexpr int X_BYTES = (TERMS == 3 ? 2 : 1) * X_HALF;
    static constexpr int XR = 2 * DY_BYTES;                   // ring of four X rows behind the two dY buffers
    static constexpr int XZ = XR + 4 * X_BYTES;               // a row of zeros: what a kernel row outside the image multiplies
    static constexpr int BYTES = XZ + X_BYTES;                // 47,616 (TERMS = 1) / 95,232
};
static_assert(192 * 4 * 4 <= Lds<1>::BYTES, "bias reduction fits");

// BF: x and dy are bf16 tensors (the 16-bit step): staged as they are (8-byte pieces, no conversion, no scaling), multiplied
// on v_mfma_f32_32x32x16_bf16; TERMS = 1 only
template <int TERMS, bool BF = false>
__global__ __launch_bounds__(256, TERMS == 3 ? 1 : 2) void conv3x3_wgrad_split_kernel(const void* __restrict__ x_, const void* __restrict__ dy_,
                                                                   int total_rows, int H, int W, int rows_per_slice,
                                                                   const float* __restrict__ dy_amax,
                                                                   float* __restrict__ slab, float* __restrict__ bias_slab)
{
    static_assert(!BF || TERMS == 1, "bf16 operands: one product term");
    using Piece = std::conditional_t<BF, u32x2, f32x4>;          // four consecutive channels
    constexpr int ES = BF ? 2 : 4;                                // bytes per element
    const char* const x = (const char*)x_;
    const char* const dy = (const char*)dy_;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    constexpr int DY_BYTES = Lds<TERMS>::DY_BYTES, X_BYTES = Lds<TERMS>::X_BYTES, XR = Lds<TERMS>::XR, XZ = Lds<TERMS>::XZ;
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
    Piece st0[9];
    // one piece -> its f16 record; TERMS = 3: high part by truncation (11 significant bits) at rec, the rest, rounded, in the
    // low image `lo_off` behind it; TERMS = 1: rounded to nearest
    auto store_h4 = [&](Piece pv, char* rec, int lo_off) {
        if constexpr (BF) {
            *(u32x2*)rec = pv;                        // four bf16 channels as they are
        } else {
            const f32x4 v = pv;
            h4 r, l;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (TERMS == 3) {
                    const float hi = __uint_as_float(__float_as_uint(v[e]) & 0xFFFFE000u);
                    r[e] = (_Float16)hi;
                    l[e] = (_Float16)(v[e] - hi);
                } else {
                    r[e] = (_Float16)v[e];
                }
            }
            *(h4*)rec = r;
            if (TERMS == 3) *(h4*)(rec + lo_off) = l;
        }
    };
    auto piece_f32 = [&](Piece pv) -> f32x4 {
        if constexpr (BF)
            return f32x4{__uint_as_float(pv.x << 16), __uint_as_float(pv.x & 0xffff0000u), __uint_as_float(pv.y << 16),
                         __uint_as_float(pv.y & 0xffff0000u)};
        else
            return pv;
    };
    auto piece_scaled = [&](Piece pv) -> Piece {      // dY times 2^kd (fp32 tensors only)
        if constexpr (BF) return pv;
        else return pv * in_mul;
    };
    // (no branches below the wave level in here: the unit's MFMAs, fragment reads and splits are one basic block)
    const float bias_mul = do_bias ? 1.f : 0.f;
#define WG1_STAGE_OPS(LOAD, CONVERT, ARR)                                                                                 \
    auto LOAD = [&](int row, int seg) {                                                                                   \
        if (is_dy) {                                                                                                      \
            const char* src = dy + (((size_t)min(row, total_rows - 1) * W + seg * 64) * 96 + sp0 * 96 + 4 * sq) * ES;     \
            _Pragma("unroll") for (int it = 0; it < 8; ++it) ARR[it] = *(const Piece*)(src + (size_t)it * 8 * 96 * ES);   \
        } else {                          /* any row: clamped into the tensor (a row nobody multiplies) */                \
            const int rr = min(max(row + 1, 0), total_rows - 1);                                                          \
            const char* src = x + ((size_t)rr * W * 96 + 32 * ct + 4 * sq) * ES;                                          \
            _Pragma("unroll") for (int it = 0; it < 9; ++it) {                                                            \
                const int xx = min(max(seg * 64 + sp0 + 8 * it - 1, 0), W - 1);                                           \
                ARR[it] = *(const Piece*)(src + (size_t)xx * 96 * ES);                                                    \
            }                                                                                                             \
        }                                                                                                                 \
    };                                                                                                                    \
    /* piece `it` (< 8) of dY unit (row, seg) -> dY buffer b */                                                           \
    auto CONVERT##_dy = [&](int it, int b, float bm) {      /* bm: 1 where this unit counts for the bias sums, else 0 */  \
        bsum += piece_f32(ARR[it]) * bm;                                                                                  \
        store_h4(piece_scaled(ARR[it]), lds + b * DY_BYTES + (sp0 + 8 * it) * DY_PITCH + sq * 8, DY_HALF);                \
    };                                                                                                                    \
    /* piece `it` (< 9) of X row row + 1 -> its ring slot (halo pixels 66 .. 71 are written and never read) */            \
    auto CONVERT##_x = [&](int it, int row, int seg) {                                                                    \
        const int r = sp0 + 8 * it;                                                                                       \
        const int xx = seg * 64 + r - 1;                                                                                  \
        const bool ok = (it != 0 && it != 8) || (unsigned)xx < (unsigned)W;     /* only halo pixels 0, 65 can be outside */ \
        store_h4(ok ? ARR[it] : Piece{}, lds + XR + ((row + 1) & 3) * X_BYTES + r * X_PITCH + sq * 8, X_HALF);            \
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
            h8 bh[3], bl[3];
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const char* pb = pb0 + (16 * ks + kx) * X_PITCH;
                bh[kx] = cat(tr_read(pb), tr_read(pb + 4 * X_PITCH));
                if (TERMS == 3) bl[kx] = cat(tr_read(pb + X_HALF), tr_read(pb + X_HALF + 4 * X_PITCH));
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const char* pa = pa0 + 16 * ks * DY_PITCH + c * 64;
                const h8 ah = cat(tr_read(pa), tr_read(pa + 4 * DY_PITCH));
                h8 al;
                if (TERMS == 3) al = cat(tr_read(pa + DY_HALF), tr_read(pa + DY_HALF + 4 * DY_PITCH));
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (TERMS == 3) {
                        acc[kx][c] = MFMA_H(al, bh[kx], acc[kx][c]);
                        acc[kx][c] = MFMA_H(ah, bl[kx], acc[kx][c]);
                    }
                    if constexpr (BF) acc[kx][c] = MFMA_B(ah, bh[kx], acc[kx][c]);
                    else acc[kx][c] = MFMA_H(ah, bh[kx], acc[kx][c]);
                }
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

// slices: a multiple of 8 (XCD mapping); 3 workgroups per slice.  TERMS = 3 (one workgroup per CU): 80 slices = 240
// workgroups (120: 179 us, 168: 154 us against 137); TERMS = 1 (two per CU): 168 slices = 504 workgroups (68 us against 73)
int conv3x3_wgrad_split_slices(int total_rows, int terms)
{
    int s = terms == 1 ? 168 : 80;
    static_assert(168 <= kWgradSplitMaxSlices, "slab size");
    while (s > 8 && total_rows < s) s -= 8;
    return s;
}

size_t conv3x3_wgrad_split_scratch_bytes(int n, int h)
{
    (void)n; (void)h;
    return ((size_t)kWgradSplitMaxSlices * 9 * 9216 + (size_t)kWgradSplitMaxSlices * 96) * sizeof(float);     // the largest slice count
}

hipError_t launch_conv3x3_wgrad_split_partial(const void* x, const void* dy, int n, int h, const float* dy_amax,
                                              float* slab, float* bias_slab, int* slices_out, hipStream_t s, int terms, bool bf16)
{
    using namespace wg1;
    if (n <= 0 || h <= 0 || (h & 63) || (terms != 1 && terms != 3) || (bf16 && terms != 1)) return hipErrorInvalidValue;
    const int total_rows = n * h;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv3x3_wgrad_split_kernel<3>, hipFuncAttributeMaxDynamicSharedMemorySize, Lds<3>::BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int slices = conv3x3_wgrad_split_slices(total_rows, terms);
    const int rps = (total_rows + slices - 1) / slices;
    if (bf16)
        hipLaunchKernelGGL((conv3x3_wgrad_split_kernel<1, true>), dim3((unsigned)(slices * 3)), dim3(256), Lds<1>::BYTES, s, x, dy,
                           total_rows, h, h, rps, (const float*)nullptr, slab, bias_slab);
    else if (terms == 1)
        hipLaunchKernelGGL(conv3x3_wgrad_split_kernel<1>, dim3((unsigned)(slices * 3)), dim3(256), Lds<1>::BYTES, s, x, dy,
                           total_rows, h, h, rps, dy_amax, slab, bias_slab);
    else
        hipLaunchKernelGGL(conv3x3_wgrad_split_kernel<3>, dim3((unsigned)(slices * 3)), dim3(256), Lds<3>::BYTES, s, x, dy,
                           total_rows, h, h, rps, dy_amax, slab, bias_slab);
    *slices_out = slices;
    return hipGetLastError();
}

}  // namespace srgen
