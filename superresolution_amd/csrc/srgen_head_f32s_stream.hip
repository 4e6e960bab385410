// srgen_head_f32s_stream.hip -- gfx950 head of the fp32 forward, row-streaming form: Conv2d(96, 3, 9, padding=4) + crop
// [:125,:125] (/root/reference/Generator.py:71,80) with fp32 tensors, fp32 accumulation and the products on the f16
// matrix instruction from two-term splits of both operands (three MFMAs per product: srgen_conv3x3_f32s.hip).
//
// Arithmetic, weight image and scaling of head_f32s_kernel (srgen_head_f32s.hip: kx taps in N, T[y][x'][kx*3+co], 9-term
// shifted sum); the walk of head_stream_bf16_kernel (srgen_head_bf16_stream.hip): a workgroup of 4 waves goes DOWN a band
// of rows, wave w owning image columns 32 w .. 32 w + 31, every input row is read once, nine accumulator tiles hold the
// partial output rows r - 4 .. r + 4, a finished row's T tile crosses LDS for the kx sum.  The tile kernel stages 12 halo
// rows for 4 output rows (each input row crosses L2 -> LDS three times as fp32, 4.8 GB per batch-256 launch) behind two
// barriers per 16-channel chunk, one workgroup per CU in lockstep: 1.18 ms, MFMA a third busy.  What differs from the bf16
// walk:
//   * the pixels never touch LDS: lane (h, p) loads ITS pixel's 8 channels of a chunk (two 16-byte loads) a full row ahead
//     into registers and splits them into (hi, lo) f16 fragments there, between the MFMAs of the following chunk;
//   * all 108 weight fragments (6 chunks x 9 ky x {hi, lo}) sit in LDS (108 KB) and are read two (ky) steps ahead of their
//     three MFMAs -- 2 ds_read_b128 per 3 MFMAs and wave, a third of what the LDS array delivers;
//   * the three terms of a product go to one accumulator back to back (a chain of this MFMA needs no interleaving).
// One wave per SIMD (144 accumulator registers), so everything that is not an MFMA is spread between MFMAs by scheduling
// groups.  LDS: weights 108 KB + T 2 x 17.5 KB.
#include "srgen_internal.h"
#include <cstdlib>

namespace srgen {
namespace hss {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

#define MFMA_H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)

constexpr int ROW_BYTES = kHR * kC * 4;           // 49,152: one fp32 input row
constexpr int TS = 33;                            // T row stride in floats (odd; >= 32: all MFMA columns written, no branch)
constexpr int T_ROWS = kHR + 8;                   // x' = -4 .. 131: four rows of zeros on either side
constexpr int T_BYTES = T_ROWS * TS * 4;          // 17,952
constexpr int W_BYTES = 6 * 9 * 2 * 1024;         // 110,592: [chunk 6][ky 9][hi | lo][lane 64][16 B] (head_pack_split)
constexpr int T_OFF = W_BYTES;
constexpr int LDS_BYTES = T_OFF + 2 * T_BYTES;    // 146,496 of 163,840
constexpr int NSTEP = 54;                         // (chunk, ky) steps per row; weights are read two steps ahead

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

__global__ __launch_bounds__(256, 1) void head_stream_f32s_kernel(HeadArgs a, int nbands, int R, int total)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ids b, b + 8, ... share an XCD (private L2): each XCD takes a contiguous run of (image, band) pairs
    const int per_xcd = (total + 7) >> 3;
    const int blk = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (blk >= total) return;
    const int n = __builtin_amdgcn_readfirstlane(blk / nbands);
    const int band = blk - n * nbands;
    const int yb = band * R, ye = min(yb + R, kOut);          // output rows [yb, ye)
    const int r_lo = max(yb - 4, 0), r_end = ye + 4;           // input rows walked: [r_lo, r_end); rows >= 128 are zero

    for (int i = tid; i < 2 * 8 * TS; i += 256) {              // T's zero rows of both buffers
        const int b = i / (8 * TS), k = i - b * (8 * TS);
        ((float*)(lds + T_OFF + b * T_BYTES))[k < 4 * TS ? k : (kHR + 4) * TS + (k - 4 * TS)] = 0.f;
    }
    const h8* wp = (const h8*)a.wpack;
    for (int i = tid; i < W_BYTES / 16; i += 256) *(h8*)(lds + i * 16) = wp[i];
    // the weights were scaled by 2^kw when the image was packed, the input is scaled by 2^kx from its producer's bound
    const int kw = split_tail_exp((const char*)a.wpack + W_BYTES, lane);
    const int kx = a.in_amax ? __builtin_amdgcn_readfirstlane(split_exp(a.in_amax[n])) : 0;
    const float in_mul = pow2f(kx), out_mul = pow2f(-(kw + kx));
    int b0i = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[0]));
    int b1i = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[1]));
    int b2i = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[2]));
    asm volatile("" : "+s"(b0i), "+s"(b1i), "+s"(b2i));       // (pinned: not re-loaded inside the row loop)
    const float b0 = __builtin_bit_cast(float, b0i), b1 = __builtin_bit_cast(float, b1i), b2 = __builtin_bit_cast(float, b2i);
    const float b01 = h ? b1 : b0;
    float* const yout = (float*)a.y;

    // ---- pixels: raw[c] = this lane's 8 fp32 channels of chunk c for the NEXT use of that chunk; ah / al = the split in use ----
    const char* const img = (const char*)a.in + (size_t)n * kHR * ROW_BYTES;
    const unsigned voff = (unsigned)(((32 * w + p) * kC + 8 * h) * 4);
    f32x4 raw[6][2];
    h8 ah[6], al[6];
    auto load_raw = [&](int c, int r) {                        // (rows below the image: a valid address, zeroed in split)
        const char* rowp = img + (size_t)min(r, kHR - 1) * ROW_BYTES + c * 64;
        raw[c][0] = *(const f32x4*)(rowp + voff);
        raw[c][1] = *(const f32x4*)(rowp + voff + 16);
    };
    auto split = [&](int c, int r) {
        // rows below the image are zero: a bit mask, not a factor 0 (the clamped row may hold inf) and not a select (the
        // compiler turns one into branches or exec-mask regions that cut the MFMA scheduling region into pieces)
        unsigned keep = r < kHR ? ~0u : 0u;
        asm volatile("" : "+v"(keep));
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float x = raw[c][e >> 2][e & 3];             // (a scalar first: bit_cast straight on the vector element took
            const float v = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, x) & keep) * in_mul;    // element 0 every time)
            const _Float16 hi = (_Float16)v;
            ah[c][e] = hi;
            al[c][e] = (_Float16)(v - (float)hi);
        }
    };

    // ---- weights: step t = chunk * 9 + (8 - ky); fragments {hi, lo} of step t + 2 are read while step t multiplies ----
    const char* const wl = lds + lane * 16;
    h8 wh[3], wlo[3];
    auto load_w = [&](int t, int slot) {
        const int g = t / 9, ky = 8 - (t - 9 * g);
        wh[slot] = *(const h8*)(wl + ((g * 9 + ky) * 2) * 1024);
        wlo[slot] = *(const h8*)(wl + ((g * 9 + ky) * 2 + 1) * 1024);
    };

    f32x16 acc[9], zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 9; ++s) acc[s] = zero16;

    // ---- output row y = r - 4 (complete when input row r is done): see srgen_head_bf16_stream.hip ----
    auto t_write = [&](f32x16& c, int y) {
        float* T = (float*)(lds + T_OFF + (y & 1) * T_BYTES) + 4 * TS;
#pragma unroll
        for (int e = 0; e < 16; ++e) T[(32 * w + acc_row(e, h)) * TS + p] = c[e];
    };
    float tv[2][9];
    auto t_sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0) only (the pixel loads stay in flight)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto t_read = [&](int y) {
        const float* Tb = (const float*)(lds + T_OFF + (y & 1) * T_BYTES) + (32 * w + p) * TS;
#pragma unroll
        for (int kx9 = 0; kx9 < 9; ++kx9) {
            tv[0][kx9] = Tb[kx9 * (TS + 3) + h];
            tv[1][kx9] = Tb[kx9 * (TS + 3) + 2];
        }
    };
    float sv[2];
    auto t_sum = [&]() {
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float s = 0.f;
#pragma unroll
            for (int kx9 = 0; kx9 < 9; ++kx9) s += tv[it][kx9];
            sv[it] = (it == 0 ? b01 : b2) + s * out_mul;
        }
        asm volatile("" : "+v"(sv[0]), "+v"(sv[1]));           // (computed HERE, not sunk into the store's branch)
    };
    const int xo = 32 * w + p;
    const bool st0 = xo < kOut, st1 = xo < kOut && h == 0;
    const size_t o0 = ((size_t)n * 3 + h) * kOut * kOut + xo, o1 = ((size_t)n * 3 + 2) * kOut * kOut + xo;
    auto t_store = [&](int y) {
        if (y >= yb && y < ye) {                               // (uniform)
            if (st0) yout[o0 + (size_t)y * kOut] = sv[0];
            if (st1) yout[o1 + (size_t)y * kOut] = sv[1];
        }
    };

    // prologue: chunks 0 .. 4 of the first row split, its chunk 5 and the second row's chunks 0 .. 4 raw
#pragma unroll
    for (int c = 0; c < 6; ++c) load_raw(c, r_lo);
#pragma unroll
    for (int c = 0; c < 5; ++c) {
        split(c, r_lo);
        load_raw(c, r_lo + 1);
    }
    __syncthreads();                                           // T's zero rows, the weights
    load_w(0, 0);
    load_w(1, 1);

#pragma unroll 1
    for (int rb = r_lo; rb < r_end; rb += 9) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int r = rb + j;
            if (r < r_end) {                                   // (uniform)
#pragma unroll
                for (int g = 0; g < 6; ++g) {
                    if (g == 0) t_sync();                      // T of output row r - 5 is complete
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 0) t_read(r - 5);
                    // the chunk whose products were issued last gets its next pixels: chunk 5 is used later in THIS row
                    const int c = (g + 5) % 6, rc = g == 0 ? r : r + 1;
                    split(c, rc);
                    load_raw(c, rc + 1);
#pragma unroll
                    for (int ky = 8; ky >= 0; --ky) {          // (ky = 8 first: its slot is the one t_write reads)
                        const int t = g * 9 + (8 - ky), sl = t % 3;
                        load_w((t + 2) % NSTEP, (t + 2) % 3);
                        const int s = (j - ky + 13) % 9;       // output row r - ky + 4
                        // (chunk 0, ky = 0 is the first product of output row r + 4: C = 0 instead of clearing the slot)
                        acc[s] = MFMA_H(al[g], wh[sl], (g == 0 && ky == 0) ? zero16 : acc[s]);
                        acc[s] = MFMA_H(ah[g], wlo[sl], acc[s]);
                        acc[s] = MFMA_H(ah[g], wh[sl], acc[s]);
                    }
                    if (g == 1) t_sum();
                    if (g == 5) t_write(acc[(j + 5) % 9], r - 4);
                    // 27 MFMAs; between them: 18 weight reads, 2 pixel loads, ~45 split instructions and the chunk's
                    // share of the output work
                    if (g == 5) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
#pragma unroll
                        for (int i = 0; i < 23; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < 27; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 1) t_store(r - 5);
                }
            }
        }
    }
    t_sync();                                                  // the band's last output row
    t_read(r_end - 5);
    t_sum();
    t_store(r_end - 5);
}

static int env_bands()
{
    const char* e = getenv("SRGEN_HEAD_BANDS");               // development switch (tools/, tests): 1..8, else the default
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 8 ? v : 0;
}

}  // namespace hss

hipError_t launch_head_f32_split_stream(const HeadArgs& a, hipStream_t s)
{
    using namespace hss;
    if (a.N < 1) return hipErrorInvalidValue;
    static bool attr_set[64] = {};
    static int cu_count[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && dev >= 0 && dev < 64) cu_count[dev] = cus;
        e = hipFuncSetAttribute((const void*)head_stream_f32s_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    // bands per image: as launch_head_bf16_stream (rounds of one workgroup per CU x rows per band + start-up)
    int nbands = env_bands();
    if (!nbands) {
        const int cus = (dev >= 0 && dev < 64 && cu_count[dev] > 0) ? cu_count[dev] : 256;
        long long best = 0;
        for (int nb = 1; nb <= 8; ++nb) {
            const long long rounds = ((long long)a.N * nb + cus - 1) / cus;
            const long long cost = rounds * ((kOut + nb - 1) / nb + (nb > 1 ? 8 : 3) + 4);
            if (!nbands || cost < best) { best = cost; nbands = nb; }
        }
    }
    const int R = (kOut + nbands - 1) / nbands;
    nbands = (kOut + R - 1) / R;                               // (no empty band)
    const long long total = (long long)a.N * nbands;
    if (total > 0x3fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((total + 7) / 8 * 8);
    hipLaunchKernelGGL(head_stream_f32s_kernel, dim3(grid), dim3(256), LDS_BYTES, s, a, nbands, R, (int)total);
    return hipGetLastError();
}

}  // namespace srgen
