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
// Workgroup = 8 waves = 4 output rows x 128 columns.  K runs as 6 chunks of 16 ci x 9 ky.  Per chunk the 12 x 128 halo
// pixels are staged in LDS as {16 hi, 16 lo} f16 records (pitch 80 B as in the 3x3 kernel: 120 KB, one workgroup per CU,
// two waves per SIMD).  Halo row i feeds output row r through kernel row ky = i - r: a wave is INPUT-row stationary --
// wave (column tile ct of 32, half ih) reads the fragments of halo rows 6 ih .. 6 ih + 5 of its columns ONCE per chunk and
// multiplies each into every output row it reaches (1 .. 4 of them), so a chunk costs a wave 12 fragment reads for its
// 54 MFMAs (the output-row-stationary first version read 36: with N = 32 a pixel fragment is used by three MFMAs only and
// the LDS pipe was as busy as the matrix pipe -- 1.35 ms per batch-256 launch).  The two halves cover 18 (row, ky) pairs
// each (ih = 0: ky 0 .. 5, ih = 1: ky 3 .. 8 -- six kernel rows of weight fragments in registers instead of nine; half 1 is
// half 0 mirrored, so both run one instruction stream) and hold four accumulator tiles (one per output row); their sums
// meet in the T buffer.  The next chunk's pixels are
// fetched into registers at the start of a chunk's MFMAs and split into (hi, lo) under its last ones, so only the LDS
// writes sit between the chunk's two barriers.  A kernel row's weight fragments are re-loaded for the next chunk as
// soon as its last product of the chunk is issued.
#include "srgen_internal.h"
#include <cstdlib>
#include <cstring>

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

    const int ct = w & 3, ih = w >> 2;                // main loop: column tile, half of the halo rows
    f32x16 acc[4];                                    // one 32 x 32 tile of T per output row
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[r][e] = 0.f;

    // staging: 12 x 128 pixels x 16 ci (fp32) = 6,144 16-byte pieces per chunk, 12 per thread: piece `it` of a thread is
    // halo row `it`, column tid / 4, channels 4 (tid % 4) .. + 3 of the chunk -- one per-thread offset, the row offsets
    // and the row validity (rows above / below the image are the conv's zero padding) are wave-uniform
    f32x4 stage[PIECES];
    h4 chi[PIECES], clo[PIECES];
    const unsigned voffb = (unsigned)(((tid >> 2) * kC + (tid & 3) * 4) * 4);     // bytes; the row pointer is wave-uniform (scalar)
    auto row_ok = [&](int it) { return (unsigned)(y0 - 4 + it) < (unsigned)kHR; };
    auto load_stage = [&](int g) {
        unsigned vo = voffb + (unsigned)g * 64u;      // ONE per-thread offset; the twelve row pointers stay scalar
        asm volatile("" : "+v"(vo));                  // (opaque: else the twelve sums are hoisted as 64-bit vector addresses)
#pragma unroll
        for (int it = 0; it < PIECES; ++it) {
            const int cy = min(max(y0 - 4 + it, 0), kHR - 1);
            const char* rowp = (const char*)(in + (size_t)cy * kHR * kC);
            stage[it] = *(const f32x4*)(rowp + vo);
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
    const unsigned rec0 = (unsigned)((tid >> 2) * PSB + (tid & 3) * 8);
    auto store_stage = [&]() {
        unsigned r0 = rec0;                           // (opaque: one address register and twelve adds per chunk instead of
        asm volatile("" : "+v"(r0));                  // twelve hoisted address registers -- the row pitch exceeds a DS offset)
#pragma unroll
        for (int it = 0; it < PIECES; ++it) {
            *(h4*)(lds + r0 + it * COLS * PSB) = chi[it];
            *(h4*)(lds + r0 + it * COLS * PSB + 32) = clo[it];
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);           // lgkmcnt(0): this wave's LDS traffic is done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    load_stage(0);
#pragma unroll
    for (int it = 0; it < PIECES; ++it) convert(it);
    store_stage();

    // Half 1 is half 0 mirrored -- halo rows 11 - ii instead of ii, output rows 3 - r, kernel rows 8 - k -- so both run
    // ONE instruction stream with the same register indices; the half enters through addresses only:
    //   step ii = 0 .. 5 reads halo row (ih ? 11 - ii : ii) and multiplies it into accumulators r = 0 .. min(ii, 3)
    //   with kernel-row slot k = ii - r (0 .. 5), slot k holding kernel row (ih ? 8 - k : k)
    const int kys = ih ? -1 : 1, ky0 = ih ? 8 : 0;    // kernel row of slot k: ky0 + kys * k
    h8 bh[6], bl[6];
    // slot k of chunk g: a wave-uniform (scalar) pointer + this lane's 16 bytes
    auto load_w = [&](int k, int g) {
        unsigned lo = (unsigned)lane * 16u;
        asm volatile("" : "+v"(lo));
        const char* base = (const char*)wp + (size_t)((g * 9 + ky0 + kys * k) * 2) * 1024;
        bh[k] = *(const h8*)(base + lo);
        bl[k] = *(const h8*)(base + 1024 + lo);
    };
#pragma unroll
    for (int k = 0; k < 6; ++k) load_w(k, 0);
    barrier();                                        // chunk 0 is in LDS
    // this lane's pixel fragments: column 32 ct + p, K half h; step ii at Ax + ii * xstep
    const char* const Ax = lds + ((ih ? 11 : 0) * COLS + 32 * ct + p) * PSB + 16 * h;
    const int xstep = ih ? -COLS * PSB : COLS * PSB;
#pragma unroll 1
    for (int g = 0; g < 6; ++g) {
        const bool more = g + 1 < 6;
        if (more) load_stage(g + 1);
        h8 xh[2], xl[2];
        xh[0] = *(const h8*)Ax;
        xl[0] = *(const h8*)(Ax + 32);
#pragma unroll
        for (int ii = 0; ii < 6; ++ii) {
            const int b = ii & 1;
            if (ii + 1 < 6) {
                xh[b ^ 1] = *(const h8*)(Ax + (ii + 1) * xstep);
                xl[b ^ 1] = *(const h8*)(Ax + (ii + 1) * xstep + 32);
            }
            // the last three steps of a chunk also split the next chunk's pixels (four pieces each)
            if (more && ii >= 3) {
#pragma unroll
                for (int k = 0; k < 4; ++k) convert((ii - 3) * 4 + k);
            }
            __builtin_amdgcn_sched_barrier(0);
            // term by term over the output rows this halo row reaches: independent accumulators back to back
#pragma unroll
            for (int term = 0; term < 3; ++term)
#pragma unroll
                for (int r = 0; r <= (ii < 3 ? ii : 3); ++r) {
                    const int k = ii - r;
                    acc[r] = term == 0 ? MFMA_H(xl[b], bh[k], acc[r])
                           : term == 1 ? MFMA_H(xh[b], bl[k], acc[r]) : MFMA_H(xh[b], bh[k], acc[r]);
                }
            if (more) {                               // slots whose last product of this chunk was just issued: min(k + 3, 5) == ii
#pragma unroll
                for (int k = 0; k < 6; ++k)
                    if ((k + 3 < 5 ? k + 3 : 5) == ii) load_w(k, g + 1);
            }
        }
        barrier();                                    // every wave is past its reads of chunk g
        if (more) {
            store_stage();
            barrier();
        }
    }

    // T -> LDS: Tl[row][x' 0..127][n 0..28] = (half 0) + (half 1), in that order
    float* const T0 = (float*)lds;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        if (ih == half && p < 27) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    float* t = T0 + (((ih ? 3 - r : r) * COLS) + 32 * ct + acc_row(e, h)) * TS + p;
                    *t = half == 0 ? acc[r][e] : *t + acc[r][e];
                }
        }
        barrier();
    }
    const float* Tl = T0 + row * COLS * TS;
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
    {
        const char* e = getenv("SRGEN_HEAD_F32S");            // development switch, read per call: "tile" = this file's kernel
        if (!(e && !strcmp(e, "tile"))) return launch_head_f32_split_stream(a, s);
    }
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
