// Development reference: the round-1 v1 fp32 3x3 conv kernel, kept only so that
// tools/convbench_f32 can A/B new kernels against it.  Not built into libsrgen.so.
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
#include "../superresolution_amd/csrc/srgen_internal.h"

namespace srgen {
namespace v1f {

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

}  // namespace v1f
}  // namespace srgen
