// srgen_kernels_f32.hip -- gfx950 fp32 kernels of the SRGAN generator forward.
//
// (The 3x3 convs live in srgen_conv3x3_f32.hip; this file holds the stem, the head and the
// channel attention.)
// All convolutions are implicit GEMMs on the exact-fp32 matrix instruction
// v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD; the fp32 roofline on MI355X is the 157 TF
// matrix peak).  M = output pixels (32 consecutive x of one row per MFMA tile),
// N = output channels, K = (tap, ci).  Activations are NHWC fp32 in HBM; each workgroup
// stages a zero-padded halo tile of the input in LDS once per ci chunk and re-reads it for
// all taps.  Weights are pre-packed on the host so every B fragment is one 16-byte (or
// 8-byte) coalesced global load that hits L1/L2; at fp32 MFMA rates (one 4 KFLOP
// instruction per 64 cycles) operand bandwidth is <5 % of L1, so B needs no LDS.
//
// MFMA operand maps (cdna_hip_programming.md section 3): for 32x32x2 f32, lane l holds
// A[row l&31][k l>>5] and B[k l>>5][col l&31]; accumulator register r of lane l is
// C[row (r&3) + 8*(r>>2) + 4*(l>>5)][col l&31].  K may be visited in any order as long as A
// and B agree, which is what lets one lane fetch 4 (or 2) consecutive ci with one wide load
// and spend them on consecutive MFMAs.
#include "srgen_internal.h"

namespace srgen {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// =========================================================================================
// Stem: Conv2d(3, 96, 9, padding=4) + PReLU() -- Generator.py:50-51.
// K = (ky, kx*3+ci): 9 rows of 27 contiguous floats in an HWC3 LDS image of the input
// tile (row padded to 28 => 252, the 28th weight is zero).  Reads NCHW, writes NHWC.
// =========================================================================================
constexpr int TS_ROWS = 12, TS_RS = 220;     // 4+8 halo rows; 72 cols * 3 ch = 216 (+4 pad)

__global__ __launch_bounds__(256, 2) void stem_f32_kernel(StemArgs a)
{
    __shared__ __attribute__((aligned(16))) float smem[TS_ROWS * TS_RS];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, p = lane & 31;
    const int ty = blockIdx.x % (kLR / 4), n = blockIdx.x / (kLR / 4);
    const int y0 = ty * 4;
    const float* x = (const float*)a.x + (size_t)n * 3 * kLR * kLR;

    for (int i = tid; i < TS_ROWS * TS_RS; i += 256) smem[i] = 0.f;
    __syncthreads();
    for (int i = tid; i < 3 * TS_ROWS * kLR; i += 256) {
        const int gx = i & 63, r = (i >> 6) % TS_ROWS, c = i / (64 * TS_ROWS);
        const int gy = y0 - 4 + r;
        if ((unsigned)gy < (unsigned)kLR)
            smem[r * TS_RS + (gx + 4) * 3 + c] = x[((size_t)c * kLR + gy) * kLR + gx];
    }
    __syncthreads();

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const float* wp = (const float*)a.wpack;
#pragma unroll
    for (int ky = 0; ky < 9; ++ky) {
        const float* arow = smem + (w + ky) * TS_RS + p * 3 + h;
        const float* brow = wp + (size_t)(ky * 28 + h) * 96 + p;
#pragma unroll
        for (int js = 0; js < 14; ++js) {
            const float a0 = arow[2 * js], a1 = arow[32 * 3 + 2 * js];
            float bv[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) bv[nt] = brow[(2 * js) * 96 + nt * 32];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                acc[0][nt] = MFMA_F32(a0, bv[nt], acc[0][nt]);
                acc[1][nt] = MFMA_F32(a1, bv[nt], acc[1][nt]);
            }
        }
    }

    const int y = y0 + w;
    float* out = (float*)a.out;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int co = nt * 32 + p;
        const float bias = a.bias[co], slope = a.slope[co];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int xx = 32 * mt + acc_row(r, h);
                float v = acc[mt][nt][r] + bias;
                v = v >= 0.f ? v : v * slope;
                out[(((size_t)n * kLR + y) * kLR + xx) * kC + co] = v;
            }
    }
}

hipError_t launch_stem_f32(const StemArgs& a, hipStream_t s)
{
    hipLaunchKernelGGL(stem_f32_kernel, dim3((unsigned)(a.N * (kLR / 4))), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =========================================================================================
// ChannelAttention -- Generator.py:15-27.  The global average pool (:18) arrives as per-tile
// channel sums written by the preceding conv's epilogue; every workgroup finishes the pool
// (fixed-order sum over tiles), runs FC(96->12)+ReLU+FC(12->96)+sigmoid (:19-24) with wave
// reductions, and scales its slice of the image in place (:27).  HBM-bound: one read and
// one write of the activation.
// =========================================================================================
constexpr int CA_SLICES = 8;    // workgroups per image

template <typename T>
__device__ __forceinline__ void ca_fc(const ChanAttnArgs& a, int n, float* s_pool, float* s_hid,
                                      float* s_scale)
{
    const int tid = threadIdx.x;
    if (tid < kC) {
        float s = 0.f;
        for (int t = 0; t < a.tiles; ++t) s += a.pool_partial[((size_t)n * a.tiles + t) * kC + tid];
        s_pool[tid] = s / (float)a.hw;
    }
    __syncthreads();
    // hidden[r] = relu(b1[r] + sum_c w1[r][c] * pool[c]); one wave per group of rows
    const int lane = tid & 63, wv = tid >> 6;
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
}

__global__ __launch_bounds__(256) void chan_attn_f32_kernel(ChanAttnArgs a)
{
    __shared__ float s_pool[kC], s_hid[12], s_scale[kC];
    const int nsl = a.scale_only ? 1 : CA_SLICES;
    const int n = blockIdx.x / nsl, slice = blockIdx.x % nsl;
    ca_fc<float>(a, n, s_pool, s_hid, s_scale);
    if (slice == 0 && threadIdx.x < kC) a.scale_out[(size_t)n * kC + threadIdx.x] = s_scale[threadIdx.x];
    if (a.amax && slice == 0 && threadIdx.x < 64) {            // |x * s| <= amax * max_c s: the refreshed per-image bound
        const int lane = threadIdx.x;
        float m = fmaxf(s_scale[lane], lane < 32 ? s_scale[64 + lane] : 0.f);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off));
        if (lane == 0) a.amax[n] *= m;
    }
    if (a.scale_only) return;                                 // (the convs behind apply the scales: Conv3x3Args::in_scale)
    const int per_slice = a.hw * kC / 4 / CA_SLICES;          // float4 per slice
    f32x4* x = (f32x4*)a.x + (size_t)n * a.hw * kC / 4 + (size_t)slice * per_slice;
    for (int i = threadIdx.x; i < per_slice; i += 256) {
        const int c = (i % (kC / 4)) * 4;
        f32x4 v = x[i];
        v.x *= s_scale[c]; v.y *= s_scale[c + 1]; v.z *= s_scale[c + 2]; v.w *= s_scale[c + 3];
        x[i] = v;
    }
}

hipError_t launch_chan_attn_f32(const ChanAttnArgs& a, hipStream_t s)
{
    if ((a.hw * kC / 4) % CA_SLICES) return hipErrorInvalidValue;
    hipLaunchKernelGGL(chan_attn_f32_kernel, dim3((unsigned)(a.N * (a.scale_only ? 1 : CA_SLICES))), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =========================================================================================
// Head: Conv2d(96, 3, 9, padding=4) + crop [:125,:125] -- Generator.py:71,80.
// N = 3 output channels is a hopeless MFMA shape, so the kx taps are moved into N:
//   T[y][x'][kx*3+co] = sum_{ky,ci} in[y+ky-4][x'][ci] * w[co][ci][ky][kx]     (N = 27 -> 32)
//   out[y][x][co]     = bias[co] + sum_kx T[y][x+kx-4][kx*3+co]
// i.e. a GEMM with K = 9*96 followed by a 9-term shifted sum done through LDS.  Same FLOPs
// as the direct form, 84 % of the MFMA columns useful.  Wave w owns output row y0+w and
// x' in [0, 128) as four 32-wide M tiles: T at x' outside the image is identically zero (the
// input there is the conv's zero padding), so those columns are neither staged nor multiplied --
// the shifted sum simply skips them.  Only rows/cols < 125 are stored.
// LDS: 12 rows x 128 cols x 8 ci per chunk as 4 planes [ci pair][pixel][2]  (49,152 B);
// re-used for T (4 x 128 x 29 floats = 59,392 B) in the epilogue.
// =========================================================================================
constexpr int TH_ROWS = 12, TH_COLS = 128, TH_MT = TH_COLS / 32, TH_PLANE = TH_ROWS * TH_COLS * 2;
constexpr int TH_TS = 29;                       // T row stride (odd: conflict-free reads)
constexpr int TH_LDS_BYTES = 4 * TH_COLS * TH_TS * 4;   // 59,392 >= 4*TH_PLANE*4 = 49,152
constexpr int TH_PIECES = TH_ROWS * TH_COLS * 2 / 256;  // 16-byte pieces per thread and chunk: 12

__global__ __launch_bounds__(256, 2) void head_f32_kernel(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // XCD-aware mapping: each XCD (private L2) gets a contiguous run of row blocks, so the 8 halo
    // rows two neighbouring blocks share are L2 hits
    const int nblk = gridDim.x;
    const int blk = (blockIdx.x & 7) * (nblk >> 3) + (blockIdx.x >> 3);
    const int ty = blk % (kHR / 4), n = blk / (kHR / 4);
    const int y0 = ty * 4;
    const float* in = (const float*)a.in + (size_t)n * kHR * kHR * kC;
    const f32x2* wp = (const f32x2*)a.wpack;

    f32x16 acc[TH_MT];
#pragma unroll
    for (int mt = 0; mt < TH_MT; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    // 12 x 128 pixels x 8 ci per chunk = 3,072 16-byte pieces, 12 per thread; branch-free loads
    // (clamped row, validity bit applied at the LDS store) issued one chunk ahead.
    f32x4 stage[TH_PIECES];
    unsigned ok_bits = 0;
#pragma unroll
    for (int it = 0; it < TH_PIECES; ++it) {
        const int i = tid + it * 256, pix = i >> 1;
        const int r = pix / TH_COLS, c = pix - r * TH_COLS;
        const int gy = y0 - 4 + r;
        (void)c;
        ok_bits |= ((unsigned)gy < (unsigned)kHR ? 1u : 0u) << it;
    }
    auto load_stage = [&](int c8) {
#pragma unroll
        for (int it = 0; it < TH_PIECES; ++it) {
            const int i = tid + it * 256, half = i & 1, pix = i >> 1;
            const int r = pix / TH_COLS, c = pix - r * TH_COLS;
            const int cy = min(max(y0 - 4 + r, 0), kHR - 1), cx = c;
            stage[it] = *(const f32x4*)(in + ((size_t)cy * kHR + cx) * kC + c8 * 8 + half * 4);
        }
    };
    auto store_stage = [&]() {
#pragma unroll
        for (int it = 0; it < TH_PIECES; ++it) {
            const int i = tid + it * 256, half = i & 1, pix = i >> 1;
            f32x4 v = stage[it];
            if (!((ok_bits >> it) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
            const f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
            *(f32x2*)(smem + (2 * half) * TH_PLANE + pix * 2) = lo;
            *(f32x2*)(smem + (2 * half + 1) * TH_PLANE + pix * 2) = hi;
        }
    };

    load_stage(0);
#pragma unroll 1
    for (int c8 = 0; c8 < 12; ++c8) {
        __syncthreads();                          // every wave is past its reads of chunk c8-1
        f32x2 bv[18];
#pragma unroll
        for (int s = 0; s < 18; ++s) bv[s] = wp[((((size_t)(s >> 1) * 12 + c8) * 2 + (s & 1)) * 2 + h) * 32 + p];
        store_stage();
        __syncthreads();
        if (c8 + 1 < 12) load_stage(c8 + 1);
        f32x2 a0[TH_MT], a1[TH_MT];
        {
            const float* ab = smem + h * TH_PLANE + (w * TH_COLS + p) * 2;
#pragma unroll
            for (int mt = 0; mt < TH_MT; ++mt) a0[mt] = *(const f32x2*)(ab + 64 * mt);
        }
#pragma unroll
        for (int s = 0; s < 18; ++s) {                // s = ky*2 + u
            if (s + 1 < 18) {
                const int ky = (s + 1) >> 1, u = (s + 1) & 1;
                const float* ab = smem + (2 * u + h) * TH_PLANE + ((w + ky) * TH_COLS + p) * 2;
#pragma unroll
                for (int mt = 0; mt < TH_MT; ++mt) {
                    const f32x2 v = *(const f32x2*)(ab + 64 * mt);
                    if (s & 1) a0[mt] = v; else a1[mt] = v;
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int e = 0; e < 2; ++e)
#pragma unroll
                for (int mt = 0; mt < TH_MT; ++mt)
                    acc[mt] = MFMA_F32((s & 1) ? a1[mt][e] : a0[mt][e], bv[s][e], acc[mt]);
        }
    }
    __syncthreads();
    // T -> LDS: Tl[w][x' 0..127][n 0..28]
    float* Tl = smem + w * TH_COLS * TH_TS;
    if (p < 27) {
#pragma unroll
        for (int mt = 0; mt < TH_MT; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) Tl[(32 * mt + acc_row(r, h)) * TH_TS + p] = acc[mt][r];
    }
    __syncthreads();
    const int y = y0 + w;
    if (y < kOut) {
        float* yo = (float*)a.y + (size_t)n * 3 * kOut * kOut;
        for (int i = lane; i < 3 * kOut; i += 64) {
            const int co = i / kOut, x = i - co * kOut;
            float s = a.bias[co];
#pragma unroll
            for (int kx = 0; kx < 9; ++kx) {
                const int xs = x + kx - 4;                   // x' of this tap; outside the image T is zero
                if ((unsigned)xs < (unsigned)TH_COLS) s += Tl[xs * TH_TS + kx * 3 + co];
            }
            yo[((size_t)co * kOut + y) * kOut + x] = s;
        }
    }
}

hipError_t launch_head_f32(const HeadArgs& a, hipStream_t s)
{
    static bool attr_set[64] = {};      // per device: >64 KB dynamic LDS needs the opt-in once
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)head_f32_kernel,
                                hipFuncAttributeMaxDynamicSharedMemorySize, TH_LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    hipLaunchKernelGGL(head_f32_kernel, dim3((unsigned)(a.N * (kHR / 4))), dim3(256), TH_LDS_BYTES, s, a);
    return hipGetLastError();
}

}  // namespace srgen
