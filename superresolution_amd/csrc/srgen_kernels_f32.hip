// srgen_kernels_f32.hip -- gfx950 fp32 kernels of the SRGAN generator forward.
//
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
// 3x3 conv 96 -> 96 (x cout_blocks), zero pad 1.  Reference: nn.Conv2d at Generator.py:34,37
// (ResBlock, with BatchNorm :35,:38 and the 0.1 of :42 folded into w/b), :62 (mid), :7
// (upsample, with PixelShuffle :8 folded into the store) and :69 (output[0]); PReLU
// (:36, :9, :70) and the residual adds (:42, :78) are the epilogue.
//
// Workgroup = 4 waves = 4 rows x 64 cols of output; wave w owns row w as two 32-pixel M
// tiles x three 32-channel N tiles (96 accumulator VGPRs).  LDS: 6 x 66 halo pixels x 32 ci
// per chunk, pixel stride 36 floats (4*odd => the 16-lane groups of ds_read_b128 hit 16
// distinct 16-byte slots: conflict-free).
// =========================================================================================
constexpr int T3_H = 4, T3_W = 64, T3_HH = 6, T3_HW = 66, T3_CK = 32, T3_PS = 36;
constexpr int T3_LDS_BYTES = T3_HH * T3_HW * T3_PS * 4;   // 57,024

// HW: image side (64 trunk / 128 output[0]); SHUFFLE: PixelShuffle(2) store (upsample conv).
// Distinct instantiations give the trunk, upsample and output[0] launches their own names in
// rocprofv3 kernel traces.
template <int HW, bool SHUFFLE>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_kernel(Conv3x3Args a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, p = lane & 31;
    constexpr int tiles_x = HW / T3_W, tiles_y = HW / T3_H;
    int b = blockIdx.x;
    const int tx = b % tiles_x; b /= tiles_x;
    const int ty = b % tiles_y;
    const int n = b / tiles_y;
    const int cb = blockIdx.y;
    const int x0 = tx * T3_W, y0 = ty * T3_H;

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    const float* in = (const float*)a.in + (size_t)n * HW * HW * kC;
    const f32x4* wp = (const f32x4*)a.wpack + (size_t)cb * 9 * 24 * 96;

    for (int ck = 0; ck < 3; ++ck) {
        __syncthreads();
        // ---- stage the halo tile of this ci chunk: 396 pixels x 8 float4 ------------------
#pragma unroll
        for (int it = 0; it < 13; ++it) {
            const int i = tid + it * 256;
            if (i < T3_HH * T3_HW * 8) {
                const int pix = i >> 3, q = i & 7;
                const int r = pix / T3_HW, c = pix - r * T3_HW;
                const int gy = y0 - 1 + r, gx = x0 - 1 + c;
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if ((unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW)
                    v = *(const f32x4*)(in + ((size_t)gy * HW + gx) * kC + ck * T3_CK + q * 4);
                *(f32x4*)(smem + pix * T3_PS + q * 4) = v;
            }
        }
        __syncthreads();
        // ---- 9 taps x 4 groups of 8 ci: 2 A reads + 3 B loads feed 24 MFMAs --------------
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int dy = tap / 3, dx = tap % 3;
            const float* abase = smem + ((w + dy) * T3_HW + (p + dx)) * T3_PS + 4 * h;
            const f32x4* bbase = wp + ((size_t)tap * 24 + ck * 8 + h) * 96 + p;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                f32x4 av[2], bv[3];
                av[0] = *(const f32x4*)(abase + 8 * j);
                av[1] = *(const f32x4*)(abase + 32 * T3_PS + 8 * j);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) bv[nt] = bbase[(2 * j) * 96 + nt * 32];
#pragma unroll
                for (int t = 0; t < 4; ++t)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                        for (int nt = 0; nt < 3; ++nt)
                            acc[mt][nt] = MFMA_F32(av[mt][t], bv[nt][t], acc[mt][nt]);
            }
        }
    }

    // ---- epilogue: bias, PReLU, residual, store (plain or pixel-shuffled), pool sums -----
    const int y = y0 + w;
    float bias[3], slope[3], pool[3];
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int co = cb * kC + nt * 32 + p;
        bias[nt] = a.bias[co];
        slope[nt] = a.slope ? a.slope[co] : 1.f;
        pool[nt] = 0.f;
    }
    const float* res = a.res ? (const float*)a.res + (size_t)n * HW * HW * kC : nullptr;
    float* out = (float*)a.out;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int x = x0 + 32 * mt + acc_row(r, h);
            size_t o;
            if (SHUFFLE)
                o = (((size_t)n * 2 * HW + 2 * y + (cb >> 1)) * 2 * HW + 2 * x + (cb & 1)) * kC;
            else
                o = (((size_t)n * HW + y) * HW + x) * kC;
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                float v = acc[mt][nt][r] + bias[nt];
                v = v >= 0.f ? v : v * slope[nt];
                if (res) v += res[((size_t)y * HW + x) * kC + nt * 32 + p];
                out[o + nt * 32 + p] = v;
                pool[nt] += v;
            }
        }
    if (a.pool_partial) {
        // channel sums of this tile: lanes l and l^32 hold the same channels (cross-lane add),
        // then the 4 waves are combined through LDS in fixed order (deterministic).
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 3; ++nt) {
            const float s = pool[nt] + __shfl_xor(pool[nt], 32);
            if (h == 0) smem[w * kC + nt * 32 + p] = s;
        }
        __syncthreads();
        if (tid < kC) {
            const float s = (smem[tid] + smem[kC + tid]) + (smem[2 * kC + tid] + smem[3 * kC + tid]);
            a.pool_partial[((size_t)n * (tiles_x * tiles_y) + ty * tiles_x + tx) * kC + tid] = s;
        }
    }
}

hipError_t launch_conv3x3_f32(const Conv3x3Args& a, hipStream_t s)
{
    if (a.H != a.W) return hipErrorInvalidValue;
    dim3 grid((unsigned)(a.N * (a.H / T3_H) * (a.W / T3_W)), (unsigned)a.cout_blocks);
    if (a.H == kLR && !a.shuffle)
        hipLaunchKernelGGL((conv3x3_f32_kernel<kLR, false>), grid, dim3(256), T3_LDS_BYTES, s, a);
    else if (a.H == kLR && a.shuffle)
        hipLaunchKernelGGL((conv3x3_f32_kernel<kLR, true>), grid, dim3(256), T3_LDS_BYTES, s, a);
    else if (a.H == kHR && !a.shuffle)
        hipLaunchKernelGGL((conv3x3_f32_kernel<kHR, false>), grid, dim3(256), T3_LDS_BYTES, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

// =========================================================================================
// Stem: Conv2d(3, 96, 9, padding=4) + PReLU() -- Generator.py:50-51.
// K = (ky, kx*3+ci): 9 rows of 27 contiguous floats in an HWC3 LDS image of the input
// tile (row padded to 28 => 252, the 28th weight is zero).  Reads NCHW, writes NHWC.
// =========================================================================================
constexpr int TS_ROWS = 12, TS_RS = 220;     // 4+8 halo rows; 72 cols * 3 ch = 216 (+4 pad)
constexpr int TS_LDS_BYTES = TS_ROWS * TS_RS * 4;

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
    const int n = blockIdx.x / CA_SLICES, slice = blockIdx.x % CA_SLICES;
    ca_fc<float>(a, n, s_pool, s_hid, s_scale);
    if (slice == 0 && threadIdx.x < kC) a.scale_out[(size_t)n * kC + threadIdx.x] = s_scale[threadIdx.x];
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
    hipLaunchKernelGGL(chan_attn_f32_kernel, dim3((unsigned)(a.N * CA_SLICES)), dim3(256), 0, s, a);
    return hipGetLastError();
}

// =========================================================================================
// Head: Conv2d(96, 3, 9, padding=4) + crop [:125,:125] -- Generator.py:71,80.
// N = 3 output channels is a hopeless MFMA shape, so the kx taps are moved into N:
//   T[y][x'][kx*3+co] = sum_{ky,ci} in[y+ky-4][x'][ci] * w[co][ci][ky][kx]     (N = 27 -> 32)
//   out[y][x][co]     = bias[co] + sum_kx T[y][x+kx-4][kx*3+co]
// i.e. a GEMM with K = 9*96 followed by a 9-term shifted sum done through LDS.  Same FLOPs
// as the direct form, 84 % of the MFMA columns useful.  Wave w owns output row y0+w and
// x' in [-4, 156) as five 32-wide M tiles.  Only rows/cols < 125 are stored.
// LDS: 12 rows x 160 cols x 8 ci per chunk as 4 planes [ci pair][pixel][2]  (61,440 B);
// re-used for T (4 x 160 x 29 floats = 74,240 B) in the epilogue.
// =========================================================================================
constexpr int TH_ROWS = 12, TH_COLS = 160, TH_PLANE = TH_ROWS * TH_COLS * 2;
constexpr int TH_TS = 29;                       // T row stride (odd: conflict-free reads)
constexpr int TH_LDS_BYTES = 4 * TH_COLS * TH_TS * 4;   // 74,240 >= 4*TH_PLANE*4 = 61,440

__global__ __launch_bounds__(256, 2) void head_f32_kernel(HeadArgs a)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, p = lane & 31;
    const int ty = blockIdx.x % (kHR / 4), n = blockIdx.x / (kHR / 4);
    const int y0 = ty * 4;
    const float* in = (const float*)a.in + (size_t)n * kHR * kHR * kC;
    const f32x2* wp = (const f32x2*)a.wpack;

    f32x16 acc[5];
#pragma unroll
    for (int mt = 0; mt < 5; ++mt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

    for (int c8 = 0; c8 < 12; ++c8) {
        __syncthreads();
        // stage 12 x 160 pixels x 8 ci: each item = one pixel's half (4 ci = 2 planes)
        for (int i = tid; i < TH_ROWS * TH_COLS * 2; i += 256) {
            const int half = i & 1, pix = i >> 1;
            const int r = pix / TH_COLS, c = pix - r * TH_COLS;
            const int gy = y0 - 4 + r, gx = c - 4;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)kHR && (unsigned)gx < (unsigned)kHR)
                v = *(const f32x4*)(in + ((size_t)gy * kHR + gx) * kC + c8 * 8 + half * 4);
            f32x2 lo = {v.x, v.y}, hi = {v.z, v.w};
            *(f32x2*)(smem + (2 * half) * TH_PLANE + pix * 2) = lo;
            *(f32x2*)(smem + (2 * half + 1) * TH_PLANE + pix * 2) = hi;
        }
        __syncthreads();
#pragma unroll
        for (int ky = 0; ky < 9; ++ky) {
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const f32x2 bv = wp[((((size_t)ky * 12 + c8) * 2 + u) * 2 + h) * 32 + p];
                const float* ab = smem + (2 * u + h) * TH_PLANE + ((w + ky) * TH_COLS + p) * 2;
                f32x2 av[5];
#pragma unroll
                for (int mt = 0; mt < 5; ++mt) av[mt] = *(const f32x2*)(ab + 64 * mt);
#pragma unroll
                for (int e = 0; e < 2; ++e)
#pragma unroll
                    for (int mt = 0; mt < 5; ++mt) acc[mt] = MFMA_F32(av[mt][e], bv[e], acc[mt]);
            }
        }
    }
    __syncthreads();
    // T -> LDS: Tl[w][x' index 0..159][n 0..28]
    float* Tl = smem + w * TH_COLS * TH_TS;
    if (p < 27) {
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
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
            for (int kx = 0; kx < 9; ++kx) s += Tl[(x + kx) * TH_TS + kx * 3 + co];
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
