// srgen_stem_f32s.hip -- gfx950 stem of the generator on the split arithmetic: Conv2d(3, 96, 9, padding=4) + PReLU()
// (Generator.py:50-51) with fp32 tensors, fp32 accumulation and every product formed from two-term f16 splits of both
// operands (three v_mfma_f32_32x32x16_f16 per product: srgen_conv3x3_f32s.hip explains the arithmetic and its error).
//
// K layout of the bf16 stem: per kernel row ky the taps kx = 0..11 (9 real) x ci = 0..3 (3 real) = 48 = three k16 steps, so
// the 8 k-values of a lane are two neighbouring pixels of an HWC4 image of the input tile in LDS (two planes: high and low
// parts).  The input tile is scaled by a power of two taken from ITS OWN maximum (exact, per workgroup: a workgroup only
// ever multiplies the 12 x 64 input pixels it staged), the weights by the exponent in the image's tail.  Weights are the
// MFMA A operand (rows = output channels), pixels the B operand: a lane ends with 4 consecutive channels of one pixel per
// register group and stores 16-byte NHWC pieces straight from registers.  The epilogue also leaves the per-image maximum
// of |output| (atomic max on the bit pattern) -- the bound the first 3x3 conv of the split path scales its input by, which
// used to cost a pass of its own over the stem's output.
//
// The same kernel serves the training step's 9x9 convs against a 3-channel image (srgen_conv9x9_c3_nhwc: the stem forward
// from the live weights, and the head's data gradient -- d out (125 x 125) against the flipped head weights at 128 x 128):
// any input size (zero outside), output a multiple of 4 rows x 64 columns, optional bias / PReLU, and a weight image packed
// on the device from the [243][96] tap matrix (stem_split_pack_kernel; the flip is a permutation of the taps there).
#include "srgen_internal.h"
#include <cstdlib>
#include <cstring>

namespace srgen {
namespace sfs {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int ROWS = 12, RS = 76;                  // 4 + 8 halo rows; pixels per LDS row (4 + 64 + 8 reach)
constexpr int PLANE = ROWS * RS * 8;               // bytes of one (high | low) HWC4 f16 plane: 7,296
constexpr int NVAL = 3 * ROWS * RS;                // input values of a tile: 2,736
constexpr int PER_THREAD = (NVAL + 255) / 256;     // ... per thread: 11

// OUT_BF16: the NHWC output is stored as bf16 (the 16-bit training step), 8-byte pieces.  TERMS = 1 (with OUT_BF16 only): the
// operands rounded to f16 after the range scaling, ONE MFMA per product -- the 16-bit step's arithmetic class (the reference
// runs these convs under bf16 autocast, train.py:91: operands rounded to 8 significant bits; f16 keeps 11), a third of the
// matrix work: the head's data gradient at 128 x 128 was the step's largest 9x9 kernel (240 us of MFMAs for 40 us of stores)
template <bool OUT_BF16, int TERMS = 3>
__global__ __launch_bounds__(256, 2) void stem_f32s_kernel(Stem9Args a)
{
    __shared__ __attribute__((aligned(16))) char lds[2 * PLANE];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = a.ow >> 6, tiles_y = a.oh >> 2;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int y0 = ty * 4, x0 = tx * 64;
    const float* x = a.x + (size_t)n * 3 * a.ih * a.iw;

    // ---- input tile (12 rows x 76 columns x 3 channels, zero outside the image): load, bound, split, HWC4 planes ------
    float v[PER_THREAD];
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = tid + 256 * k;
        const int cc = i % RS, r = (i / RS) % ROWS, c = i / (RS * ROWS);
        const int gy = y0 - 4 + r, gx = x0 - 4 + cc;
        const bool ok = i < NVAL && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw;
        v[k] = ok ? x[((size_t)c * a.ih + gy) * a.iw + gx] : 0.f;
        m = fmaxf(m, fabsf(v[k]));
    }
    for (int i = tid; i < 2 * PLANE / 4; i += 256) ((unsigned*)lds)[i] = 0u;          // (the unused fourth channel must be 0)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[w] = m;
    __syncthreads();
    const int kx_in = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    const int kw = split_tail_exp((const char*)a.wpack + kStemSplitHalfs * 2, lane);
    const float in_mul = pow2f(kx_in), out_mul = pow2f(-(kw + kx_in));
    _Float16* const hi_plane = (_Float16*)lds;
    _Float16* const lo_plane = (_Float16*)(lds + PLANE);
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = tid + 256 * k;
        if (i < NVAL) {
            const int cc = i % RS, r = (i / RS) % ROWS, c = i / (RS * ROWS);
            const float s = v[k] * in_mul;
            const _Float16 hi = (_Float16)s;
            hi_plane[(r * RS + cc) * 4 + c] = hi;
            if (TERMS == 3) lo_plane[(r * RS + cc) * 4 + c] = (_Float16)(s - (float)hi);
        }
    }
    __syncthreads();

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

    // packed weights: 16-byte units [ky 9][s 3][nt 3][hi | lo][lane 64]: lane l holds w[co = 32 nt + (l & 31)][ci = e & 3]
    // [ky][kx = 4 s + 2 (l >> 5) + (e >> 2)], zero at kx >= 9 and ci == 3
    const h8* wp = (const h8*)a.wpack + lane;
#pragma unroll
    for (int ky = 0; ky < 9; ++ky) {
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const char* xb = lds + ((w + ky) * RS + p + 4 * s + 2 * h) * 8;
            h8 xh[2], xl[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const h4 a0 = *(const h4*)(xb + 32 * 8 * mt), a1 = *(const h4*)(xb + 32 * 8 * mt + 8);
                xh[mt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                if (TERMS == 3) {
                    const h4 b0 = *(const h4*)(xb + PLANE + 32 * 8 * mt), b1 = *(const h4*)(xb + PLANE + 32 * 8 * mt + 8);
                    xl[mt] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
            }
            h8 wh[3], wl[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                wh[nt] = wp[(((ky * 3 + s) * 3 + nt) * 2 + 0) * 64];
                if (TERMS == 3) wl[nt] = wp[(((ky * 3 + s) * 3 + nt) * 2 + 1) * 64];
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {          // small terms first
                    if (TERMS == 3) {
                        acc[mt][nt] = MFMA_H(wl[nt], xh[mt], acc[mt][nt]);
                        acc[mt][nt] = MFMA_H(wh[nt], xl[mt], acc[mt][nt]);
                    }
                    acc[mt][nt] = MFMA_H(wh[nt], xh[mt], acc[mt][nt]);
                }
        }
    }

    // ---- epilogue (registers only): scale back, bias, PReLU, 16-byte NHWC stores, per-image maximum -----------------------
    const size_t out_off = (((size_t)n * a.oh + y0 + w) * a.ow + x0 + p) * kC + 4 * h;
    float* out = a.out + out_off;
    bf16_t* outb = (bf16_t*)a.out + out_off;
    float vmax = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, s4 = {1.f, 1.f, 1.f, 1.f};
            if (a.bias) b4 = *(const f32x4*)(a.bias + 32 * nt + 8 * G + 4 * h);
            if (a.slope) s4 = *(const f32x4*)(a.slope + 32 * nt + 8 * G + 4 * h);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[mt][nt][4 * G + e] * out_mul + b4[e];
                    o[e] = t >= 0.f ? t : t * s4[e];
                    vmax = fmaxf(vmax, fabsf(o[e]));
                }
                if constexpr (OUT_BF16) {
                    const float o4[4] = {o[0], o[1], o[2], o[3]};
                    st4(outb + (size_t)(32 * mt) * kC + 32 * nt + 8 * G, o4);
                } else {
                    *(f32x4*)(out + (size_t)(32 * mt) * kC + 32 * nt + 8 * G) = o;
                }
            }
        }
    if (a.out_amax) {                                 // non-negative floats order like their bit patterns
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax == vmax) atomicMax((unsigned*)a.out_amax + n, __float_as_uint(vmax));
    }
}

// ---- the three-term kernel with its weights on an LDS ring ---------------------------------------------------------------
// stem_f32s_kernel reads a k step's six weight fragments from L2 right where it multiplies them, and the compiler cannot
// hoist 162 loads far enough: 31 us per workgroup for 3.7 us of MFMAs (0.29 ms per batch-256 launch).  Here the weights of
// one kernel row (3 s x 3 nt x {hi, lo} = 18 KB) are a slab copied global -> LDS without registers (global_load_lds), two
// kernel rows ahead, into a three-slot ring; one barrier per kernel row; fragments come from LDS.  Same products, same order
// per accumulator: bit-identical to stem_f32s_kernel.  68.6 KB of LDS: two workgroups per CU.
constexpr int SLAB = 18 * 1024;
constexpr int RING_OFF = 2 * PLANE;                       // ring behind the two input planes
constexpr int LDSR_BYTES = RING_OFF + 3 * SLAB;           // 69,888

__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}
template <int N>
__device__ __forceinline__ void wait_vm_only()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

template <bool OUT_BF16>
__global__ __launch_bounds__(256, 2) void stem_f32s_ring_kernel(Stem9Args a)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[4];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = a.ow >> 6, tiles_y = a.oh >> 2;
    int b = blockIdx.x;
    const int tx = b % tiles_x;
    b /= tiles_x;
    const int ty = b % tiles_y, n = b / tiles_y;
    const int y0 = ty * 4, x0 = tx * 64;
    const float* x = a.x + (size_t)n * 3 * a.ih * a.iw;

    // slab ky: pieces 0 .. 17 of 1 KB; wave w copies pieces w, w + 4, ... (waves 0, 1: five, waves 2, 3: four)
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    const int npc = w < 2 ? 5 : 4;
    auto issue_slab = [&](int ky) {
        const char* src = (const char*)a.wpack + (size_t)ky * SLAB;
        const unsigned dst = lds_base + RING_OFF + (ky % 3) * SLAB;
#pragma unroll
        for (int k = 0; k < 5; ++k)
            if (k < npc) glds16_raw(src + (w + 4 * k) * 1024, lane16, dst + (w + 4 * k) * 1024);
    };
    issue_slab(0);
    issue_slab(1);

    // ---- input tile: load, bound, split, HWC4 planes (as stem_f32s_kernel) ----
    float v[PER_THREAD];
    float m = 0.f;
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = tid + 256 * k;
        const int cc = i % RS, r = (i / RS) % ROWS, c = i / (RS * ROWS);
        const int gy = y0 - 4 + r, gx = x0 - 4 + cc;
        const bool ok = i < NVAL && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw;
        v[k] = ok ? x[((size_t)c * a.ih + gy) * a.iw + gx] : 0.f;
        m = fmaxf(m, fabsf(v[k]));
    }
    for (int i = tid; i < 2 * PLANE / 4; i += 256) ((unsigned*)lds)[i] = 0u;          // (the unused fourth channel must be 0)
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if (lane == 0) red[w] = m;
    __syncthreads();
    const int kx_in = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]))));
    const int kw = split_tail_exp((const char*)a.wpack + kStemSplitHalfs * 2, lane);
    const float in_mul = pow2f(kx_in), out_mul = pow2f(-(kw + kx_in));
    _Float16* const hi_plane = (_Float16*)lds;
    _Float16* const lo_plane = (_Float16*)(lds + PLANE);
#pragma unroll
    for (int k = 0; k < PER_THREAD; ++k) {
        const int i = tid + 256 * k;
        if (i < NVAL) {
            const int cc = i % RS, r = (i / RS) % ROWS, c = i / (RS * ROWS);
            const float sv = v[k] * in_mul;
            const _Float16 hi = (_Float16)sv;
            hi_plane[(r * RS + cc) * 4 + c] = hi;
            lo_plane[(r * RS + cc) * 4 + c] = (_Float16)(sv - (float)hi);
        }
    }

    f32x16 acc[2][3];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;

#pragma unroll 1
    for (int ky = 0; ky < 9; ++ky) {
        // this wave's pieces of slab ky have landed (only the pieces of slab ky + 1 may still be on their way: copies retire
        // in order, and the input loads above are older than every copy that matters here)
        if (ky + 1 < 9) { if (w < 2) wait_vm_only<5>(); else wait_vm_only<4>(); }
        else wait_vm_only<0>();
        __builtin_amdgcn_s_waitcnt(0xC07F);               // lgkmcnt(0): the planes (ky = 0), this wave's reads of slab ky - 1
        __builtin_amdgcn_s_barrier();                     // slab ky complete; every wave is past slab ky - 1
        asm volatile("" ::: "memory");
        if (ky + 2 < 9) issue_slab(ky + 2);               // into the slot of slab ky - 1
        const char* const wl = lds + RING_OFF + (ky % 3) * SLAB + lane * 16;
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const char* xb = lds + ((w + ky) * RS + p + 4 * s + 2 * h) * 8;
            h8 xh[2], xl[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const h4 a0 = *(const h4*)(xb + 32 * 8 * mt), a1 = *(const h4*)(xb + 32 * 8 * mt + 8);
                const h4 b0 = *(const h4*)(xb + PLANE + 32 * 8 * mt), b1 = *(const h4*)(xb + PLANE + 32 * 8 * mt + 8);
                xh[mt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                xl[mt] = __builtin_shufflevector(b0, b1, 0, 1, 2, 3, 4, 5, 6, 7);
            }
            h8 wh[3], wlo[3];
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                wh[nt] = *(const h8*)(wl + ((s * 3 + nt) * 2 + 0) * 1024);
                wlo[nt] = *(const h8*)(wl + ((s * 3 + nt) * 2 + 1) * 1024);
            }
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {          // small terms first (the order of stem_f32s_kernel)
                    acc[mt][nt] = MFMA_H(wlo[nt], xh[mt], acc[mt][nt]);
                    acc[mt][nt] = MFMA_H(wh[nt], xl[mt], acc[mt][nt]);
                    acc[mt][nt] = MFMA_H(wh[nt], xh[mt], acc[mt][nt]);
                }
        }
    }

    // ---- epilogue (registers only): scale back, bias, PReLU, 16-byte NHWC stores, per-image maximum ----
    const size_t out_off = (((size_t)n * a.oh + y0 + w) * a.ow + x0 + p) * kC + 4 * h;
    float* out = a.out + out_off;
    bf16_t* outb = (bf16_t*)a.out + out_off;
    float vmax = 0.f;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt)
#pragma unroll
        for (int G = 0; G < 4; ++G) {
            f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, s4 = {1.f, 1.f, 1.f, 1.f};
            if (a.bias) b4 = *(const f32x4*)(a.bias + 32 * nt + 8 * G + 4 * h);
            if (a.slope) s4 = *(const f32x4*)(a.slope + 32 * nt + 8 * G + 4 * h);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float t = acc[mt][nt][4 * G + e] * out_mul + b4[e];
                    o[e] = t >= 0.f ? t : t * s4[e];
                    vmax = fmaxf(vmax, fabsf(o[e]));
                }
                if constexpr (OUT_BF16) {
                    const float o4[4] = {o[0], o[1], o[2], o[3]};
                    st4(outb + (size_t)(32 * mt) * kC + 32 * nt + 8 * G, o4);
                } else {
                    *(f32x4*)(out + (size_t)(32 * mt) * kC + 32 * nt + 8 * G) = o;
                }
            }
        }
    if (a.out_amax) {                                 // non-negative floats order like their bit patterns
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        if (lane == 0 && vmax == vmax) atomicMax((unsigned*)a.out_amax + n, __float_as_uint(vmax));
    }
}

// ---- the 16-bit step's form: ONE term, weights resident in LDS, persistent workgroups --------------------------------------
// The kernel above reads its weight fragments from L2 inside the k loop (81 or 162 16-byte loads per lane and tile, each
// k step behind the latency of its own loads): 31 us per workgroup for 1.2 us of MFMAs -- the head's data gradient of the
// 16-bit step (64 images, 128 x 128) took 250 us whether it ran one or three terms.  With one term the 81 high-part
// fragments are 81 KB: a persistent workgroup (one per CU) copies them to LDS once and walks tiles, the next tile's input
// values in registers under the current tile's MFMAs, two input planes alternating; per tile: 81 MFMAs per wave against
// 189 LDS reads (the LDS array, then the stores, bound it).  Arithmetic: operands rounded to f16 after the power-of-two range
// scaling, ONE v_mfma_f32_32x32x16_f16 per product -- the reference runs these convs under bf16 autocast (train.py:91:
// operands rounded to 8 significant bits; f16 keeps 11).
constexpr int WL_BYTES = 81 * 1024;
constexpr int LDS1_BYTES = WL_BYTES + 2 * PLANE;          // 97,536

template <bool OUT_BF16>
__global__ __launch_bounds__(256, 1) void stem_f16_lds_kernel(Stem9Args a, int total_tiles)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    __shared__ float red[2][4];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = a.ow >> 6, tiles_y = a.oh >> 2;

    {   // high-part units of the packed image ([ky 9][s 3][nt 3][hi | lo][lane 64]) -> LDS [ky][s][nt][lane]; planes zeroed once
        const h8* wg = (const h8*)a.wpack;               // (the unused fourth channel of a pixel stays 0 for good)
        for (int i = tid; i < 81 * 64; i += 256) *(h8*)(lds + i * 16) = wg[((i >> 6) * 2) * 64 + (i & 63)];
        for (int i = tid; i < 2 * PLANE / 4; i += 256) ((unsigned*)(lds + WL_BYTES))[i] = 0u;
    }
    const int kw = split_tail_exp((const char*)a.wpack + kStemSplitHalfs * 2, lane);

    float v[PER_THREAD];
    auto load_tile = [&](int t) {                        // -> v (nothing here may USE v: the loads stay in flight under the MFMAs)
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int y0 = ty * 4, x0 = tx * 64;
        const float* x = a.x + (size_t)n * 3 * a.ih * a.iw;
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const int i = tid + 256 * k;
            const int cc = i % RS, r = (i / RS) % ROWS, c = i / (RS * ROWS);
            const int gy = y0 - 4 + r, gx = x0 - 4 + cc;
            const bool ok = i < NVAL && (unsigned)gy < (unsigned)a.ih && (unsigned)gx < (unsigned)a.iw;
            v[k] = ok ? x[((size_t)c * a.ih + gy) * a.iw + gx] : 0.f;
        }
    };

    int t = blockIdx.x, b = 0;
    if (t < total_tiles) load_tile(t);
    for (; t < total_tiles; t += gridDim.x, b ^= 1) {
        const int tx = t % tiles_x, ty = (t / tiles_x) % tiles_y, n = t / (tiles_x * tiles_y);
        const int y0 = ty * 4, x0 = tx * 64;
        // ---- this tile's bound, its f16 plane ----
        float m = 0.f;
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) m = fmaxf(m, fabsf(v[k]));
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
        if (lane == 0) red[b][w] = m;
        __syncthreads();                                  // (also: every wave is past the MFMAs of the tile before)
        const int kx_in = __builtin_amdgcn_readfirstlane(split_exp(fmaxf(fmaxf(red[b][0], red[b][1]), fmaxf(red[b][2], red[b][3]))));
        const float in_mul = pow2f(kx_in), out_mul = pow2f(-(kw + kx_in));
        _Float16* const plane = (_Float16*)(lds + WL_BYTES + b * PLANE);
#pragma unroll
        for (int k = 0; k < PER_THREAD; ++k) {
            const int i = tid + 256 * k;
            if (i < NVAL) {
                const int cc = i % RS, r = (i / RS) % ROWS, c = i / (RS * ROWS);
                plane[(r * RS + cc) * 4 + c] = (_Float16)(v[k] * in_mul);
            }
        }
        __syncthreads();
        // ---- the next tile's values on their way under this tile's MFMAs ----
        const int tn = t + gridDim.x;
        if (tn < total_tiles) load_tile(tn);

        f32x16 acc[2][3];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][nt][r] = 0.f;
        const char* const wl = lds + lane * 16;
#pragma unroll
        for (int ky = 0; ky < 9; ++ky) {
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const char* xb = (const char*)plane + ((w + ky) * RS + p + 4 * s + 2 * h) * 8;
                h8 xh[2], wh[3];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const h4 a0 = *(const h4*)(xb + 32 * 8 * mt), a1 = *(const h4*)(xb + 32 * 8 * mt + 8);
                    xh[mt] = __builtin_shufflevector(a0, a1, 0, 1, 2, 3, 4, 5, 6, 7);
                }
#pragma unroll
                for (int nt = 0; nt < 3; ++nt) wh[nt] = *(const h8*)(wl + ((ky * 3 + s) * 3 + nt) * 1024);
#pragma unroll
                for (int nt = 0; nt < 3; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) acc[mt][nt] = MFMA_H(wh[nt], xh[mt], acc[mt][nt]);
            }
        }

        // ---- epilogue (registers only): scale back, bias, PReLU, NHWC stores, per-image maximum ----
        const size_t out_off = (((size_t)n * a.oh + y0 + w) * a.ow + x0 + p) * kC + 4 * h;
        float* out = a.out + out_off;
        bf16_t* outb = (bf16_t*)a.out + out_off;
        float vmax = 0.f;
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                f32x4 b4 = {0.f, 0.f, 0.f, 0.f}, s4 = {1.f, 1.f, 1.f, 1.f};
                if (a.bias) b4 = *(const f32x4*)(a.bias + 32 * nt + 8 * G + 4 * h);
                if (a.slope) s4 = *(const f32x4*)(a.slope + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    f32x4 o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float tt = acc[mt][nt][4 * G + e] * out_mul + b4[e];
                        o[e] = tt >= 0.f ? tt : tt * s4[e];
                        vmax = fmaxf(vmax, fabsf(o[e]));
                    }
                    if constexpr (OUT_BF16) {
                        const float o4[4] = {o[0], o[1], o[2], o[3]};
                        st4(outb + (size_t)(32 * mt) * kC + 32 * nt + 8 * G, o4);
                    } else {
                        *(f32x4*)(out + (size_t)(32 * mt) * kC + 32 * nt + 8 * G) = o;
                    }
                }
            }
        if (a.out_amax) {
#pragma unroll
            for (int o = 32; o >= 1; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            if (lane == 0 && vmax == vmax) atomicMax((unsigned*)a.out_amax + n, __float_as_uint(vmax));
        }
    }
}

// Weight image of the kernel above from the tap matrix wm[j = (ci, ky, kx)][96] (fp32, on the device): one workgroup finds
// max |w|, then writes the units [ky' 9][s 3][nt 3][hi | lo][lane 64] (element e of lane l: kx' = 4 s + 2 (l >> 5) + (e >> 2),
// ci = e & 3, co = 32 nt + (l & 31); zero where kx' >= 9 or ci == 3) and the tail.  flip: position (ky', kx') carries tap
// (8 - ky', 8 - kx') -- the correlation with the flipped kernel that conv9x9_c3 calls flip = 1.
constexpr int PACK_UNITS = 9 * 3 * 3 * 64;            // (ky', s, nt, lane); each has a high and a low unit
__global__ __launch_bounds__(256) void stem_split_pack_kernel(const float* __restrict__ wm, int flip, _Float16* __restrict__ out)
{
    __shared__ float red[4];
    const int tid = threadIdx.x;
    float m = 0.f;                                        // every workgroup finds max |w| for itself (93 KB, cache resident)
    for (int i = tid; i < 243 * 96 / 4; i += 256) {
        const f32x4 q = *(const f32x4*)(wm + 4 * i);
        m = fmaxf(fmaxf(m, fmaxf(fabsf(q.x), fabsf(q.y))), fmaxf(fabsf(q.z), fabsf(q.w)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
    const float sc = pow2f(split_exp(m));
    const int u = blockIdx.x * 256 + tid;
    if (u < PACK_UNITS) {
        const int lane = u & 63, nt = (u >> 6) % 3, s = (u / 192) % 3, kyp = u / 576;
        h8 hi, lo;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int kxp = 4 * s + 2 * (lane >> 5) + (e >> 2), ci = e & 3, co = 32 * nt + (lane & 31);
            float v = 0.f;
            if (kxp < 9 && ci < 3) {
                const int ky = flip ? 8 - kyp : kyp, kx = flip ? 8 - kxp : kxp;
                v = wm[(size_t)(ci * 81 + ky * 9 + kx) * 96 + co] * sc;
            }
            hi[e] = (_Float16)v;
            lo[e] = (_Float16)(v - (float)hi[e]);
        }
        const size_t unit = ((((size_t)kyp * 3 + s) * 3 + nt) * 2) * 64 + lane;
        *(h8*)(out + unit * 8) = hi;
        *(h8*)(out + (unit + 64) * 8) = lo;
    }
    if (blockIdx.x == 0 && tid < 32) ((float*)(out + kStemSplitHalfs))[tid] = tid == 0 ? m : 0.f;
}

}  // namespace sfs

hipError_t launch_conv9x9_c3_split(const Stem9Args& a, hipStream_t s)
{
    if (a.N <= 0 || a.ih <= 0 || a.iw <= 0 || a.oh <= 0 || a.ow <= 0 || (a.oh & 3) || (a.ow & 63)) return hipErrorInvalidValue;
    const long long blocks = (long long)a.N * (a.oh >> 2) * (a.ow >> 6);
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    if (a.out_bf16) {                                 // the 16-bit step: one term, weights in LDS, one persistent workgroup per CU
        static bool attr_set[64] = {};
        static int cu_count[64] = {};
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev < 0 || dev >= 64 || !attr_set[dev]) {
            int cus = 0;
            if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && dev >= 0 && dev < 64) cu_count[dev] = cus;
            e = hipFuncSetAttribute((const void*)sfs::stem_f16_lds_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, sfs::LDS1_BYTES);
            if (e != hipSuccess) return e;
            if (dev >= 0 && dev < 64) attr_set[dev] = true;
        }
        const int cus = (dev >= 0 && dev < 64 && cu_count[dev] > 0) ? cu_count[dev] : 256;
        const unsigned grid = (unsigned)(blocks < cus ? blocks : cus);
        hipLaunchKernelGGL(sfs::stem_f16_lds_kernel<true>, dim3(grid), dim3(256), sfs::LDS1_BYTES, s, a, (int)blocks);
    }
    else {
        const char* ev = getenv("SRGEN_STEM_F32S");       // development switch, read per call: "direct" = weights straight from L2
        if (ev && !strcmp(ev, "direct")) {
            hipLaunchKernelGGL(sfs::stem_f32s_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, a);
        } else {
            static bool ring_attr[64] = {};
            int dev = 0;
            hipError_t e = hipGetDevice(&dev);
            if (e != hipSuccess) return e;
            if (dev < 0 || dev >= 64 || !ring_attr[dev]) {
                e = hipFuncSetAttribute((const void*)sfs::stem_f32s_ring_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                        sfs::LDSR_BYTES);
                if (e != hipSuccess) return e;
                if (dev >= 0 && dev < 64) ring_attr[dev] = true;
            }
            hipLaunchKernelGGL(sfs::stem_f32s_ring_kernel<false>, dim3((unsigned)blocks), dim3(256), sfs::LDSR_BYTES, s, a);
        }
    }
    return hipGetLastError();
}

hipError_t launch_stem_f32_split(const StemArgs& a, hipStream_t s)
{
    Stem9Args g{(const float*)a.x, (float*)a.out, a.wpack, a.bias, a.slope, a.N, kLR, kLR, kLR, kLR, a.out_amax, 0};
    return launch_conv9x9_c3_split(g, s);
}

hipError_t launch_stem_split_pack(const float* wm, int flip, void* out, hipStream_t s)
{
    hipLaunchKernelGGL(sfs::stem_split_pack_kernel, dim3((sfs::PACK_UNITS + 255) / 256), dim3(256), 0, s, wm, flip, (_Float16*)out);
    return hipGetLastError();
}

}  // namespace srgen
