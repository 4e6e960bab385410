// Development reference: the first (round-1 v1) bf16 3x3 conv kernel, kept only so that
// tools/convbench can A/B new kernels against it.  Not built into libsrgen.so.
// srgen_kernels_bf16.hip -- gfx950 bf16 kernels of the SRGAN generator forward.
//
// Activations are NHWC bf16 in HBM, weights bf16, accumulation fp32 on
// v_mfma_f32_32x32x16_bf16 (dense peak ~2.5 PFLOP/s).  At that rate operand delivery, not
// arithmetic, is the limit, so compared with the fp32 kernels:
//   * weights (B) go through LDS too: one 9 KB slab per (tap, 48-ci chunk) is copied
//     global->LDS asynchronously (global_load_lds, 16 B/lane) into a 2-deep ring while the
//     previous slab is being multiplied;
//   * the activation halo tile (A) is staged per 48-ci chunk with a 112-byte pixel stride
//     (7 x 16 B, odd => conflict-free ds_read_b128 for 32 consecutive pixels);
//   * two 4-wave workgroups per CU (62.8 KB LDS each) so one workgroup's barrier / fill
//     stalls are covered by the other's MFMAs;
//   * the epilogue transposes the accumulators through LDS so every global access is a
//     16-byte row-contiguous load/store (residual in, bf16 out);
//   * workgroup ids are remapped so each XCD (private L2) owns a contiguous run of tiles.
//
// Operand maps, v_mfma_f32_32x32x16_bf16: lane l (r = l&31, h = l>>5) holds A[row r][k 8h+j]
// and B[k 8h+j][col r], j = 0..7; accumulator register q of lane l is
// C[row (q&3) + 8*(q>>2) + 4h][col r].
#include "../superresolution_amd/csrc/srgen_internal.h"

namespace srgen {
namespace v1 {

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
// 3x3 conv 96 -> 96 (x cout_blocks), bf16.  Same reference sites as the fp32 kernel:
// Generator.py:34,37 (+BN :35,38, 0.1 of :42 folded), :62, :7 (+PixelShuffle :8), :69;
// PReLU :36,:9,:70; residual adds :42,:78.
// =========================================================================================
constexpr int B3_H = 4, B3_W = 64, B3_HH = 6, B3_HW = 66;
constexpr int B3_CK = 48;                         // ci per chunk
constexpr int B3_PS = 112;                        // A pixel stride, bytes (96 + 16 pad)
constexpr int B3_A_BYTES = B3_HH * B3_HW * B3_PS; // 44,352
constexpr int B3_SLAB = 6 * 96 * 16;              // 9,216: one (tap, chunk) weight slab
constexpr int B3_LDS_BYTES = B3_A_BYTES + 2 * B3_SLAB;   // 62,784
static_assert(4 * E_WAVE_BYTES <= B3_LDS_BYTES, "epilogue scratch must fit");
constexpr int B3_A_ITEMS = B3_HH * B3_HW * 6;     // 16-byte pieces per chunk (2,376)

template <int HW, bool SHUFFLE, bool POOL>
__global__ __launch_bounds__(256, 2) void conv3x3_bf16_kernel(Conv3x3Args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int tiles_x = HW / B3_W, tiles_y = HW / B3_H, tiles_img = tiles_x * tiles_y;
    // XCD-aware mapping: ids b, b+8, b+16, ... share an XCD, give them consecutive tiles
    const int ntiles = gridDim.x;
    const int tile = (blockIdx.x & 7) * (ntiles >> 3) + (blockIdx.x >> 3);
    const int tx = tile % tiles_x, ty = (tile / tiles_x) % tiles_y, n = tile / tiles_img;
    const int cb = blockIdx.y;
    const int x0 = tx * B3_W, y0 = ty * B3_H;

    const __bf16* in = (const __bf16*)a.in + (size_t)n * HW * HW * kC;
    const char* wp = (const char*)a.wpack + (size_t)cb * 9 * 12 * 96 * 16;
    char* const ldsA = lds;
    char* const ldsB = lds + B3_A_BYTES;

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    u32x4 stage[10];
    auto load_a = [&](int c) {
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int i = tid + it * 256;
            u32x4 v = {0u, 0u, 0u, 0u};
            if (i < B3_A_ITEMS) {
                const int pix = i / 6, q = i - pix * 6;
                const int r = pix / B3_HW, cc = pix - r * B3_HW;
                const int gy = y0 - 1 + r, gx = x0 - 1 + cc;
                if ((unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW)
                    v = *(const u32x4*)(in + ((size_t)gy * HW + gx) * kC + c * B3_CK + q * 8);
            }
            stage[it] = v;
        }
    };
    auto store_a = [&]() {
#pragma unroll
        for (int it = 0; it < 10; ++it) {
            const int i = tid + it * 256;
            if (i < B3_A_ITEMS) {
                const int pix = i / 6, q = i - pix * 6;
                *(u32x4*)(ldsA + pix * B3_PS + q * 16) = stage[it];
            }
        }
    };
    auto load_b = [&](int tap, int c, int buf) {      // async slab copy, 576 x 16 B
        const char* src = wp + ((size_t)(tap * 12 + 6 * c) * 96) * 16;
        char* dst = ldsB + buf * B3_SLAB;
        glds16(src + (size_t)tid * 16, dst + w * 1024);
        glds16(src + (size_t)(256 + tid) * 16, dst + (256 + w * 64) * 16);
        if (w == 0) glds16(src + (size_t)(512 + tid) * 16, dst + 512 * 16);
    };

    load_a(0);
    load_b(0, 0, 0);
    store_a();
    __syncthreads();

    for (int c = 0; c < 2; ++c) {
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int buf = (c + tap) & 1;
            if (tap < 8) load_b(tap + 1, c, buf ^ 1);
            else if (c == 0) { load_b(0, 1, buf ^ 1); load_a(1); }
            const int dy = tap / 3, dx = tap % 3;
            const char* Ab = ldsA + ((w + dy) * B3_HW + (p + dx)) * B3_PS + h * 16;
            const char* Bb = ldsB + buf * B3_SLAB + (h * 96 + p) * 16;
#pragma unroll
            for (int g = 0; g < 3; ++g) {
                const bf16x8 a0 = *(const bf16x8*)(Ab + g * 32);
                const bf16x8 a1 = *(const bf16x8*)(Ab + 32 * B3_PS + g * 32);
                bf16x8 bv[3];
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) bv[nt] = *(const bf16x8*)(Bb + (2 * g * 96 + nt * 32) * 16);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) {
                    acc[0][nt] = MFMA_BF16(a0, bv[nt], acc[0][nt]);
                    acc[1][nt] = MFMA_BF16(a1, bv[nt], acc[1][nt]);
                }
            }
            __syncthreads();                  // slab tap+1 landed (vmcnt) and slab `buf` is free
        }
        if (c == 0) {
            store_a();                        // every wave is past its last read of chunk 0
            __syncthreads();
        }
    }

    // ---- epilogue ------------------------------------------------------------------------
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
        const int co = cb * kC + nt * 32 + p;
        const float bias = a.bias[co];
        const float slope = a.slope ? a.slope[co] : 1.f;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[mt][nt][r] + bias;
                acc[mt][nt][r] = v >= 0.f ? v : v * slope;
            }
    }
    const int y = y0 + w;
    float colsum[3][8];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) colsum[i][j] = 0.f;
    float* E = (float*)lds + w * (E_WAVE_BYTES / 4);
    __bf16* out = (__bf16*)a.out;
    if (SHUFFLE) {
        out += (((size_t)n * 2 * HW + 2 * y + (cb >> 1)) * 2 * HW + 2 * x0 + (cb & 1)) * kC;
        epilogue_rows<false>(acc, E, lane, out, (const __bf16*)nullptr,
                             [](int px) { return (size_t)(2 * px) * kC; }, colsum);
    } else {
        const size_t row = (((size_t)n * HW + y) * HW + x0) * kC;
        const __bf16* res = a.res ? (const __bf16*)a.res + row : nullptr;
        epilogue_rows<POOL>(acc, E, lane, out + row, res, [](int px) { return (size_t)px * kC; }, colsum);
    }
    if (POOL) {
        // lanes with equal (lane & 3) hold the same channel groups: butterfly over lane bits 2..5,
        // then the 4 waves are summed in fixed order through LDS.
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = colsum[i][j];
                s += __shfl_xor(s, 4); s += __shfl_xor(s, 8); s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
                colsum[i][j] = s;
            }
        __syncthreads();
        float* P = (float*)lds;
        if (lane < 4) {
#pragma unroll
            for (int i = 0; i < 3; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) P[w * kC + (lane + 4 * i) * 8 + j] = colsum[i][j];
        }
        __syncthreads();
        if (tid < kC) {
            const float s = (P[tid] + P[kC + tid]) + (P[2 * kC + tid] + P[3 * kC + tid]);
            a.pool_partial[((size_t)n * tiles_img + ty * tiles_x + tx) * kC + tid] = s;
        }
    }
}

hipError_t launch_conv3x3_bf16(const Conv3x3Args& a, hipStream_t s)
{
    if (a.H != a.W) return hipErrorInvalidValue;
    const unsigned tiles = (unsigned)(a.N * (a.H / B3_H) * (a.W / B3_W));    // multiple of 16
    dim3 grid(tiles, (unsigned)a.cout_blocks);
    if (a.H == kLR && !a.shuffle && !a.pool_partial)
        hipLaunchKernelGGL((conv3x3_bf16_kernel<kLR, false, false>), grid, dim3(256), B3_LDS_BYTES, s, a);
    else if (a.H == kLR && !a.shuffle)
        hipLaunchKernelGGL((conv3x3_bf16_kernel<kLR, false, true>), grid, dim3(256), B3_LDS_BYTES, s, a);
    else if (a.H == kLR && a.shuffle && !a.pool_partial && !a.res)
        hipLaunchKernelGGL((conv3x3_bf16_kernel<kLR, true, false>), grid, dim3(256), B3_LDS_BYTES, s, a);
    else if (a.H == kHR && !a.shuffle && !a.pool_partial)
        hipLaunchKernelGGL((conv3x3_bf16_kernel<kHR, false, false>), grid, dim3(256), B3_LDS_BYTES, s, a);
    else
        return hipErrorInvalidValue;
    return hipGetLastError();
}

}  // namespace v1
}  // namespace srgen
