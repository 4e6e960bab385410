// srgen_conv3x3_f32.hip -- gfx950 fp32 3x3 conv 96 -> 96 (x cout_blocks), implicit GEMM on the
// exact-fp32 matrix instruction v_mfma_f32_32x32x2_f32 (64 FLOP/clk/SIMD: the fp32 roofline on
// MI355X is the 157 TF matrix peak, ~138 TF at the clock the chip holds under this load).
//
// Reference sites (file:line into /root/reference): Generator.py:34,37 (ResBlock convs; eval
// BatchNorm :35,38 and the 0.1 of :42 are folded into w, b on the host), :62 (mid), :7
// (upsample conv, PixelShuffle :8 folded into the store address), :69 (output[0]); PReLU
// :36,:9,:70 and the residual adds :42,:78 are the epilogue.
//
// Same skeleton as the bf16 kernel (srgen_conv3x3_bf16.hip): persistent workgroups of 4 waves
// (4 rows x 64 pixels x 96 output channels per work item, two workgroups per CU), weights as
// the MFMA's A operand so the accumulator holds 4 consecutive channels of one pixel in 4
// consecutive registers (16-byte NHWC stores straight from registers), fragments
// double-buffered in registers, next item's halo tile fetched under the epilogue.
// Differences that follow from the 16x lower matrix rate: one 16-byte operand load feeds 4
// MFMAs = 256 cycles, so the weights (B of the GEMM) are read straight from L1/L2 -- pre-packed
// so each fragment is one coalesced 16-byte load -- and only the halo tile goes through LDS
// (32-ci chunks, pixel stride 36 floats = 4*odd => conflict-free ds_read_b128); there is no
// per-tap barrier, only one per ci-chunk.
//
// Operand maps, v_mfma_f32_32x32x2_f32: lane l holds A[row l&31][k l>>5], B[k l>>5][col l&31];
// accumulator register q of lane l is C[row (q&3) + 8*(q>>2) + 4*(l>>5)][col l&31].  K may be
// visited in any order as long as A and B agree: lane half h fetches ci 8j+4h .. 8j+4h+3 with
// one 16-byte access and spends them on 4 consecutive MFMAs.
#include "srgen_internal.h"

#include <type_traits>

namespace srgen {
namespace c3f {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define MFMA_F32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

constexpr int TH = 4, TW = 64, HH = 6, HWP = 66;
constexpr int CK = 32;                            // ci per chunk
constexpr int PS = 36;                            // A pixel stride, floats
constexpr int A_BYTES = HH * HWP * PS * 4;        // 57,024
constexpr int BS_OFF = A_BYTES;                   // bias[4][96], slope[4][96]
constexpr int LDS_BYTES = BS_OFF + 2 * 4 * 96 * 4;   // 60,096 (two workgroups per CU)
constexpr int A_ITEMS = HH * HWP * 8;             // 16-byte pieces per chunk (3,168)
constexpr int A_REGS = 13;                        // ceil(3168 / 256)
constexpr int MAX_WGS = 512;                      // persistent grid: 2 workgroups x 256 CUs

struct Frags { f32x4 a0, a1, b0, b1, b2; };

__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void mma_step(f32x16 (&acc)[2][3], const Frags& f)
{
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        acc[0][0] = MFMA_F32(f.b0[t], f.a0[t], acc[0][0]);
        acc[1][0] = MFMA_F32(f.b0[t], f.a1[t], acc[1][0]);
        acc[0][1] = MFMA_F32(f.b1[t], f.a0[t], acc[0][1]);
        acc[1][1] = MFMA_F32(f.b1[t], f.a1[t], acc[1][1]);
        acc[0][2] = MFMA_F32(f.b2[t], f.a0[t], acc[0][2]);
        acc[1][2] = MFMA_F32(f.b2[t], f.a1[t], acc[1][2]);
    }
}

struct Item { int n, ty, tx, cb; };

template <int HW, bool SHUFFLE, bool POOL, bool RES>
__global__ __launch_bounds__(256, 2) void conv3x3_f32_kernel(Conv3x3Args a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int tiles_x = HW / TW, tiles_y = HW / TH, tiles_img = tiles_x * tiles_y;
    constexpr int CB = SHUFFLE ? 4 : 1;
    const bool act = a.slope != nullptr;

    // persistent item walk: XCD x owns items [x*per, (x+1)*per) (see srgen_conv3x3_bf16.hip)
    const int per_xcd = nitems >> 3;
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int item_end = (xcd + 1) * per_xcd;
    int item = xcd * per_xcd + ((int)blockIdx.x >> 3);
    auto decode = [&](int it) {
        Item r;
        r.cb = it % CB;
        const int tile = it / CB;
        r.tx = tile % tiles_x;
        r.ty = (tile / tiles_x) % tiles_y;
        r.n = tile / tiles_img;
        return r;
    };

    float* const ldsA = (float*)lds;
    float* const ldsBias = (float*)(lds + BS_OFF);
    float* const ldsSlope = ldsBias + 4 * 96;

    // Halo tile staging, branch-free on the load side (control flow inside the unrolled MFMA
    // stream wrecks register allocation): out-of-image pieces load a clamped, valid address and
    // are zeroed when they are written to LDS (stage_ok bit per piece).
    f32x4 stage[A_REGS];
    unsigned stage_ok = 0;
    auto load_a = [&](const Item& t, int c) {
        const float* in = (const float*)a.in + (size_t)t.n * HW * HW * kC + c * CK;
        const int x0 = t.tx * TW, y0 = t.ty * TH;
        const int tv = opaque(tid);
        stage_ok = 0;
#pragma unroll
        for (int it = 0; it < A_REGS; ++it) {
            const int i = min(tv + it * 256, A_ITEMS - 1);
            const int pix = i >> 3, q = i & 7;
            const int r = pix / HWP, cc = pix - r * HWP;
            const int gy = y0 - 1 + r, gx = x0 - 1 + cc;
            const bool ok = (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const int cy = min(max(gy, 0), HW - 1), cx = min(max(gx, 0), HW - 1);
            stage[it] = *(const f32x4*)(in + (cy * HW + cx) * kC + q * 4);
            stage_ok |= (ok ? 1u : 0u) << it;
        }
    };
    auto store_a = [&]() {
        const int tv = opaque(tid);
#pragma unroll
        for (int it = 0; it < A_REGS; ++it) {
            const int i = tv + it * 256;
            const bool ok = (stage_ok >> it) & 1u;
            f32x4 v = stage[it];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.f;
            if (i < A_ITEMS) *(f32x4*)(ldsA + (i >> 3) * PS + (i & 7) * 4) = v;
        }
    };
    // residual of this lane's outputs, one 32-pixel M tile at a time: pixel (y0 + w, x0 + 32*mt
    // + p), per (nt, G) one 16-byte piece at channels 32*nt + 8*G + 4*h
    f32x4 rres[12];
    auto load_res = [&](const Item& t, int mt) {
        const int lv = opaque(lane);
        const float* res = (const float*)a.res +
                           (((size_t)t.n * HW + t.ty * TH + w) * HW + t.tx * TW + 32 * mt + (lv & 31)) * kC + 4 * (lv >> 5);
#pragma unroll
        for (int i = 0; i < 12; ++i) rres[i] = *(const f32x4*)(res + 32 * (i >> 2) + 8 * (i & 3));
    };

    Item cur = decode(item);
    load_a(cur, 0);
    for (int i = tid; i < CB * 96; i += 256) {
        ldsBias[i] = a.bias[i];
        ldsSlope[i] = act ? a.slope[i] : 1.f;
    }
    store_a();
    __syncthreads();

    Frags f0, f1;
    const float* const Aw = ldsA + (w * HWP + p) * PS + 4 * h;

#pragma unroll 1
    for (;;) {
        const int nxt = item + wg_per_xcd;
        const bool has_next = nxt < item_end;
        const Item nx = decode(has_next ? nxt : item);
        // packed weights: float4 units [cb][tap 9][ci/4 24][co 96]
        const f32x4* const wq = (const f32x4*)a.wpack + (size_t)cur.cb * 9 * 24 * 96 + h * 96 + opaque(p);
        auto load_frags = [&](Frags& f, int c, int tap, int j) {
            const int dy = tap / 3, dx = tap % 3;
            const float* Ab = Aw + (dy * HWP + dx) * PS + 8 * j;
            f.a0 = *(const f32x4*)Ab;
            f.a1 = *(const f32x4*)(Ab + 32 * PS);
            const f32x4* Bb = wq + (size_t)(tap * 24 + c * 8 + 2 * j) * 96;
            f.b0 = Bb[0];
            f.b1 = Bb[32];
            f.b2 = Bb[64];
        };

        // accumulators start at the bias (broadcast LDS reads: the address depends on h only)
        f32x16 acc[2][3];
#pragma unroll
        for (int nt = 0; nt < 3; ++nt)
#pragma unroll
            for (int G = 0; G < 4; ++G) {
                const f32x4 b4 = *(const f32x4*)(ldsBias + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                for (int e = 0; e < 4; ++e) acc[0][nt][4 * G + e] = acc[1][nt][4 * G + e] = b4[e];
            }

        // one ci-chunk = 9 taps x 4 steps of 8 ci (24 MFMAs each), fully unrolled
        auto run_chunk = [&](auto C) {
            constexpr int c = decltype(C)::value;
            load_frags(f0, c, 0, 0);
#pragma unroll
            for (int s = 0; s < 36; ++s) {
                if (RES && s == 30 && c == 2) load_res(cur, 0);
                if (s + 1 < 36) {
                    if (s & 1) load_frags(f0, c, (s + 1) / 4, (s + 1) % 4);
                    else load_frags(f1, c, (s + 1) / 4, (s + 1) % 4);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (s & 1) mma_step(acc, f1); else mma_step(acc, f0);
            }
            __syncthreads();                      // every wave is past its last read of this chunk
        };
        // The next chunk's halo tile is fetched at the chunk boundary: its latency (~2k cycles per
        // 55k-cycle chunk) is covered by the co-resident workgroup; holding it in registers across
        // the last steps of the chunk instead costs more in register pressure than it saves.
        run_chunk(std::integral_constant<int, 0>{});
        load_a(cur, 1);
        store_a();
        __syncthreads();
        run_chunk(std::integral_constant<int, 1>{});
        load_a(cur, 2);
        store_a();
        __syncthreads();
        run_chunk(std::integral_constant<int, 2>{});

        // the A tile is free: fetch the next item's chunk 0 now, under the epilogue
        if (has_next) load_a(nx, 0);

        // ---- epilogue (registers only): PReLU -> (+ residual) -> 16-byte NHWC stores -----------
        const int y = cur.ty * TH + w, x0 = cur.tx * TW;
        float* out = (float*)a.out;
        constexpr int PXS = SHUFFLE ? 2 * kC : kC;
        if (SHUFFLE)
            out += (((size_t)cur.n * 2 * HW + 2 * y + (cur.cb >> 1)) * 2 * HW + 2 * (x0 + p) + (cur.cb & 1)) * kC + 4 * h;
        else
            out += (((size_t)cur.n * HW + y) * HW + x0 + p) * kC + 4 * h;
        float psum[3][16];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            if (RES && mt == 1) load_res(cur, 1);
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                f32x16& c = acc[mt][nt];
#pragma unroll
                for (int G = 0; G < 4; ++G) {
                    f32x4 v = {c[4 * G], c[4 * G + 1], c[4 * G + 2], c[4 * G + 3]};
                    if (act) {
                        const f32x4 s4 = *(const f32x4*)(ldsSlope + cur.cb * 96 + 32 * nt + 8 * G + 4 * h);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * s4[e];
                    }
                    if (RES) v += rres[nt * 4 + G];
                    if (POOL) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) psum[nt][4 * G + e] = mt == 0 ? v[e] : psum[nt][4 * G + e] + v[e];
                    }
                    *(f32x4*)(out + (size_t)(32 * mt) * PXS + 32 * nt + 8 * G) = v;
                }
            }
        }
        if (POOL) {
            // per-tile channel sums (ChannelAttention's pool), fixed order: halving butterfly over
            // the 32 pixels of each lane half (see srgen_conv3x3_bf16.hip), then the 4 waves through
            // LDS (the A tile is free here: the next item's chunk 0 is still in registers)
            float* P = ldsA;
#pragma unroll
            for (int nt = 0; nt < 3; ++nt) {
                float (&t)[16] = psum[nt];
#pragma unroll
                for (int k = 8, b = 1; k >= 1; k >>= 1, b <<= 1)
#pragma unroll
                    for (int i = 0; i < k; ++i) {
                        const bool up = (lane & b) != 0;
                        const float send = up ? t[i] : t[i + k];
                        const float keep = up ? t[i + k] : t[i];
                        t[i] = keep + __shfl_xor(send, b);
                    }
                t[0] += __shfl_xor(t[0], 16);
                if ((lane & 16) == 0) {
                    const int q = 8 * (lane & 1) + 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + ((lane >> 3) & 1);
                    P[w * kC + 32 * nt + (q & 3) + 8 * (q >> 2) + 4 * h] = t[0];
                }
            }
            __syncthreads();
            if (tid < kC) {
                const float s = (P[tid] + P[kC + tid]) + (P[2 * kC + tid] + P[3 * kC + tid]);
                a.pool_partial[((size_t)cur.n * tiles_img + cur.ty * tiles_x + cur.tx) * kC + tid] = s;
            }
            __syncthreads();                      // P is read before the next tile overwrites it
        }
        if (!has_next) break;
        store_a();
        __syncthreads();
        item = nxt;
        cur = nx;
    }
}

template <int HW, bool SHUFFLE, bool POOL, bool RES>
hipError_t launch_one(const Conv3x3Args& a, hipStream_t s)
{
    const int nitems = a.N * (HW / TH) * (HW / TW) * (SHUFFLE ? 4 : 1);     // multiple of 16
    const int grid = nitems < MAX_WGS ? nitems & ~7 : MAX_WGS;
    hipLaunchKernelGGL((conv3x3_f32_kernel<HW, SHUFFLE, POOL, RES>), dim3((unsigned)grid), dim3(256), LDS_BYTES,
                       s, a, nitems);
    return hipGetLastError();
}

}  // namespace c3f

hipError_t launch_conv3x3_f32(const Conv3x3Args& a, hipStream_t s)
{
    using namespace c3f;
    if (a.H != a.W || a.N <= 0) return hipErrorInvalidValue;
    const bool res = a.res != nullptr, pool = a.pool_partial != nullptr;
    if (a.H == kLR && !a.shuffle && a.cout_blocks == 1) {
        if (!pool && !res) return launch_one<kLR, false, false, false>(a, s);
        if (!pool && res) return launch_one<kLR, false, false, true>(a, s);
        if (pool && res) return launch_one<kLR, false, true, true>(a, s);
    }
    if (a.H == kLR && a.shuffle && a.cout_blocks == 4 && !pool && !res) return launch_one<kLR, true, false, false>(a, s);
    if (a.H == kHR && !a.shuffle && a.cout_blocks == 1 && !pool && !res) return launch_one<kHR, false, false, false>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace srgen
