// srgen_head_f32s.hip -- gfx950 head: Conv2d(96, 3, 9, padding=4) + crop [:125,:125] (Generator.py:71,80) with fp32
// tensors, fp32 accumulation and the products on the f16 matrix instruction from two-term splits of both operands
// (the "split" path, see srgen_conv3x3_f32s.hip for the arithmetic and its error).
//
// Formulation as in the fp32 / bf16 heads (srgen_kernels_f32.hip): three output channels are no MFMA shape, so the kx taps
// move into N,
//     T[y][x'][kx*3+co] = sum_{ky,ci} in[y+ky-4][x'][ci] * w[co][ci][ky][kx]            (N = 27 -> 32, K = 9 * 96)
//     out[y][x][co]     = bias[co] + sum_kx T[y][x+kx-4][kx*3+co]                         (9-term shifted sum through LDS)
// x' runs over the 128 image columns only (T outside the image is identically zero).
//
// Workgroup = 8 waves = 4 output rows x 128 columns; wave (row r, half mh) owns T of row y0 + r, columns 64 mh .. 64 mh + 63
// as two 32 x 32 accumulator tiles.  K runs as 6 chunks of 16 ci x 9 ky.  Per chunk the 12 x 128 halo pixels are staged
// in LDS as {16 hi, 16 lo} f16 records (pitch 80 B as in the 3x3 kernel: 120 KB, one workgroup per CU, two waves per
// SIMD).  The next chunk's pixels are fetched into registers at the start of a chunk's MFMAs and split into (hi, lo)
// under its last ones, so only the LDS writes sit between the chunk's two barriers.  The weight fragments of a chunk
// (9 ky x {hi, lo}, 16 B per lane each) live in registers and are re-loaded for the next chunk one ky at a time as they
// fall free.
#include "srgen_internal.h"

namespace srgen {
namespace hfs {


typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int ROWS = 12, COLS = 128, PSB = 80, NTHR = 512;
constexpr int A_BYTES = ROWS * COLS * PSB;            // 122,880
constexpr int PIECES = ROWS * COLS * 4 / NTHR;        // 16-byte fp32 pieces per thread and chunk: 12
constexpr int TS = 29;                                // T row stride in floats (odd: conflict-free reads)
constexpr int T_BYTES = 4 * COLS * TS * 4;            // 59,392 (re-uses the pixel buffer)
static_assert(T_BYTES <= A_BYTES, "T fits the pixel buffer");

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(NTHR, 1) void head_f32s_kernel(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row = w & 3, mh = w >> 2;
    // XCD-aware mapping: each XCD (private L2) gets a contiguous run of row blocks (shared halo rows are L2 hits)
    const int nblk = gridDim.x;
    const int blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int ty = blk % (kHR / 4), n = blk / (kHR / 4);
    const int y0 = ty * 4;
    const float* in = (const float*)a.in + (size_t)n * kHR * kHR * kC;
    const h8* wp = (const h8*)a.wpack;                // units [chunk 6][ky 9][hi|lo][lane 64], then the amax tail
    // the weights were scaled by 2^kw when the image was packed (srgen_internal.h: split_exp)
    const int kw = split_tail_exp((const char*)a.wpack + (size_t)6 * 9 * 2 * 64 * 16, lane);
    // ... and the input is scaled by 2^kx from the bound its producer left (in_amax: one bound per image)
    const int kx = a.in_amax ? __builtin_amdgcn_readfirstlane(split_exp(a.in_amax[n])) : 0;
    const float in_mul = pow2f(kx), out_mul = pow2f(-(kw + kx));

    f32x16 acc[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // staging: 12 x 128 pixels x 16 ci (fp32) = 6,144 16-byte pieces per chunk, 12 per thread: piece `it` of a thread is
    // halo row `it`, column tid / 4, channels 4 (tid % 4) .. + 3 of the chunk -- one per-thread offset, the row offsets
    // and the row validity (rows above / below the image are the conv's zero padding) are wave-uniform
    f32x4 stage[PIECES];
    h4 chi[PIECES], clo[PIECES];
    const int voff = (tid >> 2) * kC + (tid & 3) * 4;
    auto row_ok = [&](int it) { return (unsigned)(y0 - 4 + it) < (unsigned)kHR; };
    auto load_stage = [&](int g) {
#pragma unroll
        for (int it = 0; it < PIECES; ++it) {
            const int cy = min(max(y0 - 4 + it, 0), kHR - 1);
            stage[it] = *(const f32x4*)(in + (size_t)cy * kHR * kC + g * 16 + voff);
        }
    };
    auto convert = [&](int it) {
        const bool ok = row_ok(it);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = ok ? stage[it][e] * in_mul : 0.f;
            chi[it][e] = (_Float16)v;
            clo[it][e] = (_Float16)(v - (float)chi[it][e]);
        }
    };
    char* const rec0 = lds + (tid >> 2) * PSB + (tid & 3) * 8;
    auto store_stage = [&]() {
#pragma unroll
        for (int it = 0; it < PIECES; ++it) {
            *(h4*)(rec0 + it * COLS * PSB) = chi[it];
            *(h4*)(rec0 + it * COLS * PSB + 32) = clo[it];
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): this wave's LDS traffic is done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    h8 bh[9], bl[9];
#pragma unroll
    for (int ky = 0; ky < 9; ++ky) {
        bh[ky] = wp[(ky * 2 + 0) * 64 + lane];
        bl[ky] = wp[(ky * 2 + 1) * 64 + lane];
    }
    load_stage(0);
#pragma unroll
    for (int it = 0; it < PIECES; ++it) convert(it);
    store_stage();
    barrier();

    // this lane's pixel fragments: row (row + ky), column 64 mh + 32 mt + p, K half h
    const char* const Ax = lds + (row * COLS + 64 * mh + p) * PSB + 16 * h;

#pragma unroll 1
    for (int g = 0; g < 6; ++g) {
        const bool more = g + 1 < 6;
        if (more) load_stage(g + 1);
        h8 xh[2][2], xl[2][2];                        // [buffer][mt]
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            xh[0][mt] = *(const h8*)(Ax + mt * 32 * PSB);
            xl[0][mt] = *(const h8*)(Ax + mt * 32 * PSB + 32);
        }
#pragma unroll
        for (int ky = 0; ky < 9; ++ky) {
            const int b = ky & 1;
            if (ky + 1 < 9) {
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    xh[b ^ 1][mt] = *(const h8*)(Ax + ((ky + 1) * COLS + mt * 32) * PSB);
                    xl[b ^ 1][mt] = *(const h8*)(Ax + ((ky + 1) * COLS + mt * 32) * PSB + 32);
                }
            }
            // the last three ky of a chunk also split the next chunk's pixels (four pieces each)
            if (more && ky >= 6) {
#pragma unroll
                for (int k = 0; k < 4; ++k) convert((ky - 6) * 4 + k);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                acc[mt] = MFMA_H(xl[b][mt], bh[ky], acc[mt]);
                acc[mt] = MFMA_H(xh[b][mt], bl[ky], acc[mt]);
                acc[mt] = MFMA_H(xh[b][mt], bh[ky], acc[mt]);
            }
            if (more) {                               // this ky's weights of the next chunk
                bh[ky] = wp[(((g + 1) * 9 + ky) * 2 + 0) * 64 + lane];
                bl[ky] = wp[(((g + 1) * 9 + ky) * 2 + 1) * 64 + lane];
            }
        }
        barrier();                                    // every wave is past its reads of chunk g
        if (more) {
            store_stage();
            barrier();
        }
    }

    // T -> LDS: Tl[row][x' 0..127][n 0..28]
    float* Tl = (float*)lds + row * COLS * TS;
    if (p < 27) {
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Tl[(64 * mh + 32 * mt + acc_row(r, h)) * TS + p] = acc[mt][r];
    }
    barrier();
    const int y = y0 + row;
    if (y < kOut) {
        float* yo = (float*)a.y + (size_t)n * 3 * kOut * kOut;
        // the two waves of a row split its 3 x 125 outputs
        for (int i = lane + 64 * mh; i < 3 * kOut; i += 128) {
            const int co = i / kOut, x = i - co * kOut;
            float s = 0.f;
#pragma unroll
            for (int kx = 0; kx < 9; ++kx) {
                const int xs = x + kx - 4;            // x' of this tap; outside the image T is zero
                if ((unsigned)xs < (unsigned)COLS) s += Tl[xs * TS + kx * 3 + co];
            }
            yo[((size_t)co * kOut + y) * kOut + x] = a.bias[co] + s * out_mul;
        }
    }
}

}  // namespace hfs

hipError_t launch_head_f32_split(const HeadArgs& a, hipStream_t s)
{
    using namespace hfs;
    static bool attr_set[64] = {};      // per device: > 64 KB dynamic LDS needs the opt-in once
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (a.N <= 0) return hipErrorInvalidValue;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)head_f32s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, A_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(head_f32s_kernel, dim3((unsigned)(a.N * (kHR / 4))), dim3(NTHR), A_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace srgen
