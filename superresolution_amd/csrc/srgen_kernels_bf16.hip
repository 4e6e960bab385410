// srgen_kernels_bf16.hip -- gfx950 bf16 kernels of the SRGAN generator forward.
//
// Activations are NHWC bf16 in HBM, weights bf16, accumulation fp32 on
// v_mfma_f32_32x32x16_bf16 (dense peak ~2.5 PFLOP/s).  At that rate operand delivery, not
// arithmetic, is the limit, so compared with the fp32 kernels:
// the 3x3 convs live in srgen_conv3x3_bf16.hip; this file holds the stem, the head and the
// channel attention.  Their epilogues transpose the accumulators through LDS so every global
// access is a 16-byte row-contiguous store.
//
// Operand maps, v_mfma_f32_32x32x16_bf16: lane l (r = l&31, h = l>>5) holds A[row r][k 8h+j]
// and B[k 8h+j][col r], j = 0..7; accumulator register q of lane l is
// C[row (q&3) + 8*(q>>2) + 4h][col r].
#include "srgen_internal.h"
#include <cstdlib>
#include <cstring>

namespace srgen {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    // async 16-byte global -> LDS copy: LDS address = wave-uniform base + lane*16
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

__device__ __forceinline__ float bf16_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf16_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ unsigned pack_bf16(float lo, float hi)
{
    typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
    bf16x2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}

// -----------------------------------------------------------------------------------------
// Shared epilogue: accumulators of one wave (2 M tiles x 3 N tiles = 64 pixels x 96 channels,
// already biased / activated) -> LDS transpose -> (+ residual) -> bf16 rows to global.
// E is this wave's private fp32 scratch of 32 x 100 floats.  px_off(px) gives the element
// offset of output pixel `px` (0..63 along the wave's row) in out / res.
// -----------------------------------------------------------------------------------------
constexpr int E_RS = 100;                 // floats per pixel row in the transpose scratch
constexpr int E_WAVE_BYTES = 32 * E_RS * 4;

template <bool POOL, typename OutOff>
__device__ __forceinline__ void epilogue_rows(f32x16 (&acc)[2][3], float* E, int lane,
                                              __bf16* out, const __bf16* res, OutOff out_off,
                                              float (&colsum)[3][8])
{
    const int h = lane >> 5, p = lane & 31;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        __syncthreads();                       // scratch free (previous readers / main loop done)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) E[acc_row(r, h) * E_RS + nt * 32 + p] = acc[mt][nt][r];
        __syncthreads();
#pragma unroll
        for (int it = 0; it < 6; ++it) {
            const int px = (lane >> 2) + 16 * (it / 3);
            const int g = (lane & 3) + 4 * (it % 3);
            const f32x4 v0 = *(const f32x4*)(E + px * E_RS + g * 8);
            const f32x4 v1 = *(const f32x4*)(E + px * E_RS + g * 8 + 4);
            float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
            const size_t o = out_off(32 * mt + px) + g * 8;
            if (res) {
                const u32x4 rv = *(const u32x4*)(res + o);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[2 * j] += bf16_lo(rv[j]);
                    v[2 * j + 1] += bf16_hi(rv[j]);
                }
            }
            if (POOL) {
#pragma unroll
                for (int j = 0; j < 8; ++j) colsum[it % 3][j] += v[j];
            }
            u32x4 ov;
#pragma unroll
            for (int j = 0; j < 4; ++j) ov[j] = pack_bf16(v[2 * j], v[2 * j + 1]);
            *(u32x4*)(out + o) = ov;
        }
    }
}

// =========================================================================================
// Stem, bf16: Conv2d(3, 96, 9, padding=4) + PReLU() -- Generator.py:50-51.
// The input tile lives in LDS as HWC4 (ci padded 3 -> 4, 8 bytes per pixel), so 8 consecutive
// k = 2 neighbouring pixels = two aligned ds_read_b64.  K per ky is kx 0..11 (9 real) x 4
// = 48 = three k16 steps; the packed weights are zero at kx >= 9 and ci == 3.
// =========================================================================================
constexpr int SB_ROWS = 12, SB_RS = 76;           // pixels per LDS row (64 + 4 + 8 reach)
constexpr int SB_TILE_BYTES = SB_ROWS * SB_RS * 8;       // 7,296
constexpr int SB_LDS_BYTES = 4 * E_WAVE_BYTES;           // epilogue scratch dominates (51,200)

__global__ __launch_bounds__(256, 2) void stem_bf16_kernel(StemArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ty = blockIdx.x % (kLR / 4), n = blockIdx.x / (kLR / 4);
    const int y0 = ty * 4;
    const __bf16* x = (const __bf16*)a.x + (size_t)n * 3 * kLR * kLR;

    for (int i = tid; i < SB_TILE_BYTES / 4; i += 256) ((unsigned*)lds)[i] = 0u;
    __syncthreads();
    __bf16* tile = (__bf16*)lds;
    for (int i = tid; i < 3 * SB_ROWS * kLR; i += 256) {
        const int gx = i & 63, r = (i >> 6) % SB_ROWS, c = i / (64 * SB_ROWS);
        const int gy = y0 - 4 + r;
        if ((unsigned)gy < (unsigned)kLR) tile[(r * SB_RS + gx + 4) * 4 + c] = x[((size_t)c * kLR + gy) * kLR + gx];
    }
    __syncthreads();

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const bf16x8* wp = (const bf16x8*)a.wpack;
#pragma unroll
    for (int ky = 0; ky < 9; ++ky) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const char* Ab = lds + ((w + ky) * SB_RS + p + 4 * s + 2 * h) * 8;
            bf16x8 av[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const bf16x4 lo = *(const bf16x4*)(Ab + 32 * 8 * mt);
                const bf16x4 hi = *(const bf16x4*)(Ab + 32 * 8 * mt + 8);
                av[mt] = __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            bf16x8 bv[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) bv[nt] = wp[((ky * 3 + s) * 2 + h) * 96 + nt * 32 + p];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                acc[0][nt] = MFMA_BF16(av[0], bv[nt], acc[0][nt]);
                acc[1][nt] = MFMA_BF16(av[1], bv[nt], acc[1][nt]);
            }
        }
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int co = nt * 32 + p;
        const float bias = a.bias[co], slope = a.slope[co];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r] + bias;
                acc[mt][nt][r] = v >= 0.f ? v : v * slope;
            }
    }
    float colsum[3][8];
    float* E = (float*)lds + w * (E_WAVE_BYTES / 4);
    __bf16* out = (__bf16*)a.out + (((size_t)n * kLR + y0 + w) * kLR) * kC;
    epilogue_rows<false>(acc, E, lane, out, (const __bf16*)nullptr, [](int px) { return (size_t)px * kC; },
                         colsum);
}

hipError_t launch_stem_bf16(const StemArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(stem_bf16_kernel, dim3((unsigned)(a.N * (kLR / 4))), dim3(256), SB_LDS_BYTES, s, a);
    return hipGetLastError();
}

// =========================================================================================
// ChannelAttention, bf16 activations (Generator.py:15-27): same as the fp32 kernel, the
// in-place scale moves 16 bytes (8 channels) per lane.
// =========================================================================================
constexpr int CAB_SLICES = 8;

__global__ __launch_bounds__(256) void chan_attn_bf16_kernel(ChanAttnArgs a)
{
    __shared__ float s_pool[kC], s_hid[12], s_scale[kC];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int nsl = a.scale_only ? 1 : CAB_SLICES;
    const int n = blockIdx.x / nsl, slice = blockIdx.x % nsl;
    if (tid < kC) {
        float s = 0.f;
        for (int t = 0; t < a.tiles; ++t) s += a.pool_partial[((size_t)n * a.tiles + t) * kC + tid];
        s_pool[tid] = s / (float)a.hw;
    }
    __syncthreads();
    for (int r = wv; r < 12; r += 4) {
        float s = a.w1[r * kC + lane] * s_pool[lane];
        if (lane < 32) s += a.w1[r * kC + 64 + lane] * s_pool[64 + lane];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) s += __shfl_xor(s, off);
        if (lane == 0) s_hid[r] = fmaxf(s + a.b1[r], 0.f);
    }
    __syncthreads();
    if (tid < kC) {
        float s = a.b2[tid];
#pragma unroll
        for (int r = 0; r < 12; ++r) s += a.w2[tid * 12 + r] * s_hid[r];
        s_scale[tid] = 1.f / (1.f + expf(-s));
    }
    __syncthreads();
    if (slice == 0 && tid < kC) a.scale_out[(size_t)n * kC + tid] = s_scale[tid];
    if (a.scale_only) return;
    const int per_slice = a.hw * kC / 8 / CAB_SLICES;          // 16-byte pieces per slice
    u32x4* x = (u32x4*)a.x + (size_t)n * a.hw * kC / 8 + (size_t)slice * per_slice;
    for (int i = tid; i < per_slice; i += 256) {
        const int c = (i % (kC / 8)) * 8;
        u32x4 v = x[i];
#pragma unroll
        for (int j = 0; j < 4; ++j)
            v[j] = pack_bf16(bf16_lo(v[j]) * s_scale[c + 2 * j], bf16_hi(v[j]) * s_scale[c + 2 * j + 1]);
        x[i] = v;
    }
}

hipError_t launch_chan_attn_bf16(const ChanAttnArgs& a, hipStream_t s)
{
    if ((a.hw * kC / 8) % CAB_SLICES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chan_attn_bf16_kernel, dim3((unsigned)(a.N * (a.scale_only ? 1 : CAB_SLICES))), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =========================================================================================
// Head, bf16: Conv2d(96, 3, 9, padding=4) + crop -- Generator.py:71,80.  Same (kx -> N)
// reformulation as the fp32 head: T[y][x'][kx*3+co] = sum_{ky,ci} in[y+ky-4][x'][ci] w[...],
// out[y][x][co] = bias + sum_kx T[y][x+kx-4][kx*3+co].
// x' runs over the 128 image columns only (T outside the image is zero: see the fp32 head).
// LDS: 12 rows x 128 cols x 16 ci per chunk as two planes [k half][pixel][16 B] (49,152 B),
// re-used for T (4 x 128 x 29 floats = 59,392 B) in the epilogue.
// =========================================================================================
constexpr int HB_ROWS = 12, HB_COLS = 128, HB_MT = HB_COLS / 32, HB_PLANE = HB_ROWS * HB_COLS * 16;
constexpr int HB_TS = 29;
constexpr int HB_LDS_BYTES = 4 * HB_COLS * HB_TS * 4;     // 59,392 >= 2*HB_PLANE = 49,152
constexpr int HB_PIECES = HB_ROWS * HB_COLS * 2 / 256;    // 16-byte pieces per thread and chunk: 12

// OUT_F32: y is an fp32 tensor (the 16-bit training step hands the image to the fp32 loss kernels)
template <bool OUT_F32>
__global__ __launch_bounds__(256, 2) void head_bf16_kernel(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware mapping: ids b, b+8, ... share an XCD (private L2); give each XCD a contiguous run
    // of row blocks so the 8 halo rows two neighbouring blocks share are L2 hits
    const int nblk = gridDim.x;
    const int blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int ty = blk % (kHR / 4), n = blk / (kHR / 4);
    const int y0 = ty * 4;
    const __bf16* in = (const __bf16*)a.in + (size_t)n * kHR * kHR * kC;
    const bf16x8* wp = (const bf16x8*)a.wpack;

    f32x16 acc[HB_MT];
#pragma unroll
    for (int mt = 0; mt < HB_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // 12 x 128 pixel tile (row halo only) of one 16-ci chunk = 3,072 16-byte pieces, 12 per thread.  The
    // loads are branch-free (clamped row + a validity bit applied at the LDS store) and are
    // issued one chunk ahead, so their latency sits under the previous chunk's MFMAs.
    u32x4 stage[HB_PIECES];
    unsigned ok_bits = 0;
#pragma unroll
    for (int it = 0; it < HB_PIECES; ++it) {
        const int i = tid + it * 256, pix = i >> 1;
        const int r = pix / HB_COLS, c = pix - r * HB_COLS;
        const int gy = y0 - 4 + r;
        (void)c;
        ok_bits |= ((unsigned)gy < (unsigned)kHR ? 1u : 0u) << it;
    }
    auto load_stage = [&](int g) {
#pragma unroll
        for (int it = 0; it < HB_PIECES; ++it) {
            const int i = tid + it * 256, half = i & 1, pix = i >> 1;
            const int r = pix / HB_COLS, c = pix - r * HB_COLS;
            const int cy = min(max(y0 - 4 + r, 0), kHR - 1), cx = c;
            stage[it] = *(const u32x4*)(in + ((size_t)cy * kHR + cx) * kC + g * 16 + half * 8);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int it = 0; it < HB_PIECES; ++it) {
            const int i = tid + it * 256, half = i & 1, pix = i >> 1;
            u32x4 v = stage[it];
            if (!((ok_bits >> it) & 1u)) v = u32x4{0u, 0u, 0u, 0u};
            *(u32x4*)(lds + half * HB_PLANE + pix * 16) = v;
        }
    };

    load_stage(0);
#pragma unroll 1
    for (int g = 0; g < 6; ++g) {
        __syncthreads();                          // every wave is past its reads of chunk g-1
        bf16x8 bv[9];
#pragma unroll
        for (int ky = 0; ky < 9; ++ky) bv[ky] = wp[((ky * 6 + g) * 2 + h) * 32 + p];
        store_stage();
        __syncthreads();
        if (g + 1 < 6) load_stage(g + 1);
        const char* ab = lds + h * HB_PLANE + (w * HB_COLS + p) * 16;
        bf16x8 a0[HB_MT], a1[HB_MT];
#pragma unroll
        for (int mt = 0; mt < HB_MT; ++mt) a0[mt] = *(const bf16x8*)(ab + 32 * 16 * mt);
#pragma unroll
        for (int ky = 0; ky < 9; ++ky) {
            if (ky + 1 < 9) {
#pragma unroll
                for (int mt = 0; mt < HB_MT; ++mt) {
                    const bf16x8 v = *(const bf16x8*)(ab + ((ky + 1) * HB_COLS + 32 * mt) * 16);
                    if (ky & 1) a0[mt] = v; else a1[mt] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < HB_MT; ++mt) acc[mt] = MFMA_BF16((ky & 1) ? a1[mt] : a0[mt], bv[ky], acc[mt]);
        }
    }
    __syncthreads();
    float* Tl = (float*)lds + w * HB_COLS * HB_TS;
    if (p < 27) {
#pragma unroll
        for (int mt = 0; mt < HB_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Tl[(32 * mt + acc_row(r, h)) * HB_TS + p] = acc[mt][r];
    }
    __syncthreads();
    const int y = y0 + w;
    if (y < kOut) {
        __bf16* yo = (__bf16*)a.y + (size_t)n * 3 * kOut * kOut;
        float* yf = (float*)a.y + (size_t)n * 3 * kOut * kOut;
        for (int i = lane; i < 3 * kOut; i += 64) {
            const int co = i / kOut, x = i - co * kOut;
            float s = a.bias[co];
#pragma unroll
            for (int kx = 0; kx < 9; ++kx) {
                const int xs = x + kx - 4;                   // x' of this tap; outside the image T is zero
                if ((unsigned)xs < (unsigned)HB_COLS) s += Tl[xs * HB_TS + kx * 3 + co];
            }
            if constexpr (OUT_F32) yf[((size_t)co * kOut + y) * kOut + x] = s;
            else yo[((size_t)co * kOut + y) * kOut + x] = (__bf16)s;
        }
    }
}

hipError_t launch_head_bf16(const HeadArgs& a, hipStream_t s)
{
    {
        const char* e = getenv("SRGEN_HEAD_BF16");            // development switch, read per call: "tile" = this file's kernel
        if (!(e && !strcmp(e, "tile"))) return launch_head_bf16_stream(a, s);
    }
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)head_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HB_LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)head_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                HB_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (a.y_f32) hipLaunchKernelGGL(head_bf16_kernel<true>, dim3((unsigned)(a.N * (kHR / 4))), dim3(256), HB_LDS_BYTES, s, a);
    else hipLaunchKernelGGL(head_bf16_kernel<false>, dim3((unsigned)(a.N * (kHR / 4))), dim3(256), HB_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace srgen
