// srgen_head_bf16_stream.hip -- gfx950 head of the bf16 forward, row-streaming form: Conv2d(96, 3, 9, padding=4) + crop
// [:125,:125] (/root/reference/Generator.py:71,80) on bf16 NHWC input, fp32 accumulation.
//
// Same arithmetic as head_bf16_kernel (srgen_kernels_bf16.hip) -- the kx taps move into N,
//     T[y][x'][kx*3+co] = sum_{ky,ci} in[y+ky-4][x'][ci] * w[co][ci][ky][kx]            (N = 27 -> 32, K = 9 * 96)
//     out[y][x][co]     = bias[co] + sum_kx T[y][x+kx-4][kx*3+co]
// -- and the same weight image; what changes is who walks what.  The tile kernel stages 12 halo rows for 4 output rows (every
// input row crosses L2 -> LDS three times: 2.4 GB per batch-256 launch) and spends two barriers and 12 register-staged
// pieces per thread on each of its six 16-channel chunks; it ran at 0.46 PFLOP/s, neither HBM- nor matrix-bound.  Here a
// workgroup of 4 waves walks DOWN a band of rows, wave w owning image columns 32 w .. 32 w + 31 (4 x 32 = the whole
// row), and every input row is fetched once:
//   * a wave copies its own 6 KB strip of a row global -> LDS without registers (6 global_load_lds of 1 KB), D = 4
//     rows ahead, waited for with a counted vmcnt; nobody else reads that strip, so the main loop has no barrier for it;
//   * the weight fragments of kernel rows 3 .. 8 (36 of 9 ky x 6 chunks) live in registers (one wave per SIMD: 512
//     registers per lane), those of rows 0 .. 2 in LDS, read one MFMA group ahead (all 54 in registers left the compiler
//     no room: the T sum ran as a chain of single reads, each waited for);
//   * input row r contributes through kernel row ky to output row r - ky + 4: nine accumulator tiles form a ring of
//     partial output rows (slot = row mod 9, static after unrolling nine rows); when row r is done, output row r - 4 is
//     complete: its T tile goes through LDS (the 9-term kx sum reaches into the neighbours' columns: one barrier per
//     row, T double-buffered) and the slot starts over for output row r + 5 (first product with C = 0).
// A band of R output rows reads R + 8 input rows; bands x images = workgroups (one per CU: 155 KB of LDS).
#include "srgen_internal.h"
#include <cstdlib>
#include <cstring>

namespace srgen {
namespace hst {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int D = 4;                              // ring of input rows (a power of two)
constexpr int ROW_BYTES = kHR * kC * 2;           // 24,576
constexpr int STRIP = 32 * kC * 2;                // 6,144: a wave's 32 pixels of one row
constexpr int TS = 33;                            // T row stride in floats (odd: conflict-free column walks; >= 32: a lane
                                                  // writes all 32 MFMA columns, the five unused ones included, without a branch)
constexpr int T_ROWS = kHR + 8;                   // x' = -4 .. 131: four rows of zeros on either side (T outside the image is zero)
constexpr int T_BYTES = T_ROWS * TS * 4;          // 17,952
constexpr int ZERO_OFF = D * ROW_BYTES;           // one strip of zeros: the rows below the image
constexpr int T_OFF = ZERO_OFF + STRIP;
constexpr int W_OFF = T_OFF + 2 * T_BYTES;        // weight fragments of kernel rows 0 .. KY_LDS - 1: [ky][chunk 6][lane 64][16 B]
constexpr int KY_LDS = 3;
constexpr int LDS_BYTES = W_OFF + KY_LDS * 6 * 1024;   // 158,784 of 163,840
static_assert((D & (D - 1)) == 0 && D >= 3, "ring size");

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// s_waitcnt vmcnt(n) (lgkmcnt / expcnt untouched), then nothing of the compiler's may move above it
template <int N>
__device__ __forceinline__ void wait_vm()
{
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    asm volatile("" ::: "memory");
}

// async 16-byte-per-lane global -> LDS copy (inline asm: see srgen_conv3x3_bf16.hip); lds_addr = the wave's 1 KB piece
__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

template <bool OUT_F32>
__global__ __launch_bounds__(256, 1) void head_stream_bf16_kernel(HeadArgs a, int nbands, int R, int total)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ids b, b + 8, ... share an XCD (private L2): each XCD takes a contiguous run of (image, band) pairs, so the eight
    // input rows two neighbouring bands share are L2 hits
    const int per_xcd = (total + 7) >> 3;
    const int blk = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (blk >= total) return;
    const int n = __builtin_amdgcn_readfirstlane(blk / nbands);             // (the division runs on the vector ALU)
    const int band = blk - n * nbands;
    const int yb = band * R, ye = min(yb + R, kOut);          // output rows [yb, ye)
    const int r_lo = max(yb - 4, 0), r_end = ye + 4;           // input rows walked: [r_lo, r_end) ...
    const int r_data = min(r_end, kHR);                        // ... of which [r_lo, r_data) exist (below: the zero strip)

    for (int i = tid * 16; i < STRIP; i += 256 * 16) *(u32x4*)(lds + ZERO_OFF + i) = u32x4{0u, 0u, 0u, 0u};
    for (int i = tid; i < 2 * 8 * TS; i += 256) {              // T's zero rows x' = -4 .. -1 and 128 .. 131 of both buffers
        const int b = i / (8 * TS), k = i - b * (8 * TS);
        ((float*)(lds + T_OFF + b * T_BYTES))[k < 4 * TS ? k : (kHR + 4) * TS + (k - 4 * TS)] = 0.f;
    }

    const bf16x8* wp = (const bf16x8*)a.wpack;                 // [ky 9][chunk 6][lane 64] fragments (pack_head_bf16)
    bf16x8 wf[9 - KY_LDS][6];
#pragma unroll
    for (int ky = KY_LDS; ky < 9; ++ky)
#pragma unroll
        for (int g = 0; g < 6; ++g) wf[ky - KY_LDS][g] = wp[(ky * 6 + g) * 64 + lane];
    for (int i = tid; i < KY_LDS * 6 * 64; i += 256) *(bf16x8*)(lds + W_OFF + i * 16) = wp[i];
    const char* const wl = lds + W_OFF + lane * 16;
    // (three wave-uniform values pinned in scalar registers: left to itself the compiler re-loads them inside the row loop,
    // and every load it waits for also waits for the row copies in flight)
    int b0i = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[0]));
    int b1i = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[1]));
    int b2i = __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, a.bias[2]));
    asm volatile("" : "+s"(b0i), "+s"(b1i), "+s"(b2i));
    const float b0 = __builtin_bit_cast(float, b0i), b1 = __builtin_bit_cast(float, b1i), b2 = __builtin_bit_cast(float, b2i);
    const float b01 = h ? b1 : b0;
    void* const yout = a.y;                                    // (no lambda below touches `a`: its address stays untaken)

    const char* const img = (const char*)a.in + (size_t)n * kHR * ROW_BYTES + w * STRIP;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_row = [&](int r) {
        const unsigned long long sa = (unsigned long long)(img + (size_t)r * ROW_BYTES);
        const char* src = (const char*)(((unsigned long long)__builtin_amdgcn_readfirstlane((unsigned)(sa >> 32)) << 32) |
                                        (unsigned)__builtin_amdgcn_readfirstlane((unsigned)sa));
        const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + (unsigned)(((r - r_lo) & (D - 1)) * ROW_BYTES + w * STRIP));
#pragma unroll
        for (int k = 0; k < 6; ++k) glds16_raw(src + k * 1024, lane16, dst + k * 1024);
    };
    // row q of this wave's strip has landed when at most 6 x (rows issued after q) copies are outstanding
    // (loads and stores retire through ONE in-order counter: the two output stores a row issues sit between the copy groups
    // and are not counted here, so the wait is stricter than it looks -- by two copies of the next row -- never laxer)
    auto wait_rows_after = [&](int k) {
        if (k >= 3) wait_vm<18>();
        else if (k == 2) wait_vm<12>();
        else if (k == 1) wait_vm<6>();
        else wait_vm<0>();
    };
    const int aoff = p * (kC * 2) + h * 16;
    auto row_base = [&](int r) -> const char* {
        return r < r_data ? lds + ((r - r_lo) & (D - 1)) * ROW_BYTES + w * STRIP + aoff : lds + ZERO_OFF + aoff;
    };

    f32x16 acc[9], zero16;
#pragma unroll
    for (int e = 0; e < 16; ++e) zero16[e] = 0.f;
#pragma unroll
    for (int s = 0; s < 9; ++s)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[s][e] = 0.f;

    // Output row y = r - 4 is complete when input row r is done.  Its T tile goes to LDS at the end of row r (t_write);
    // the barrier, the reads of the 9-term kx sum (t_read: all 18 issued at once) and the sum + store (t_store) sit inside
    // row r + 1's MFMA stream -- with one wave per SIMD nobody else would fill the matrix pipe while a wave does them.
    // Straight-line code: rows outside the band run it too and only mask the store.
    // (The buffer of row y is written again for row y + 2, after the barrier of row y + 1, which every wave reaches with
    // its reads of row y consumed.)
    auto t_write = [&](f32x16& c, int y) {
        float* T = (float*)(lds + T_OFF + (y & 1) * T_BYTES) + 4 * TS;
#pragma unroll
        for (int e = 0; e < 16; ++e) T[(32 * w + acc_row(e, h)) * TS + p] = c[e];
        // (no clearing: the slot's next product -- the next input row's chunk 0, ky = 0 -- starts from zero by itself)
    };
    float tv[2][9];
    auto t_sync = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);                    // lgkmcnt(0) only: __syncthreads would also wait for the row copies
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    auto t_read = [&](int y) {
        // items (co, x): lanes 0..63 -> co = h, then lanes 0..31 -> co = 2; x = 32 w + p; x' = x + kx - 4 (rows -4 .. 131 exist)
        const float* Tb = (const float*)(lds + T_OFF + (y & 1) * T_BYTES) + (32 * w + p) * TS;
#pragma unroll
        for (int kx = 0; kx < 9; ++kx) {
            tv[0][kx] = Tb[kx * (TS + 3) + h];
            tv[1][kx] = Tb[kx * (TS + 3) + 2];
        }
    };
    float sv[2];
    auto t_sum = [&]() {                                       // (branch-free: it runs between MFMAs)
#pragma unroll
        for (int it = 0; it < 2; ++it) {
            float s = it == 0 ? b01 : b2;
#pragma unroll
            for (int kx = 0; kx < 9; ++kx) s += tv[it][kx];
            sv[it] = s;
        }
        asm volatile("" : "+v"(sv[0]), "+v"(sv[1]));           // (computed HERE, not sunk into the store's branch)
    };
    const int xo = 32 * w + p;
    const bool st0 = xo < kOut, st1 = xo < kOut && h == 0;
    const size_t o0 = ((size_t)n * 3 + h) * kOut * kOut + xo, o1 = ((size_t)n * 3 + 2) * kOut * kOut + xo;
    auto t_store = [&](int y) {
        if (y >= yb && y < ye) {                               // (uniform)
            if constexpr (OUT_F32) {
                if (st0) ((float*)yout)[o0 + (size_t)y * kOut] = sv[0];
                if (st1) ((float*)yout)[o1 + (size_t)y * kOut] = sv[1];
            } else {
                if (st0) ((__bf16*)yout)[o0 + (size_t)y * kOut] = (__bf16)sv[0];
                if (st1) ((__bf16*)yout)[o1 + (size_t)y * kOut] = (__bf16)sv[1];
            }
        }
    };

    __syncthreads();                                           // the zero strip, T's zero rows, the LDS weights
#pragma unroll
    for (int i = 0; i < D; ++i)
        if (r_lo + i < r_data) issue_row(r_lo + i);
    wait_rows_after(min(r_lo + D - 1, r_data - 1) - r_lo);
    bf16x8 af[6];                                              // the fragments of the row being multiplied; chunk g is
    {                                                          // replaced by the next row's as soon as its products are issued
        const char* base = row_base(r_lo);
#pragma unroll
        for (int g = 0; g < 6; ++g) af[g] = *(const bf16x8*)(base + g * 32);
    }
    bf16x8 wt[2][KY_LDS];                                      // kernel rows 0 .. KY_LDS - 1 of chunk g, read one chunk ahead
#pragma unroll
    for (int ky = 0; ky < KY_LDS; ++ky) wt[0][ky] = *(const bf16x8*)(wl + (ky * 6) * 1024);

#pragma unroll 1
    for (int rb = r_lo; rb < r_end; rb += 9) {
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int r = rb + j;
            if (r < r_end) {                                   // (uniform)
                const char* next = row_base(r + 1);
#pragma unroll
                for (int g = 0; g < 6; ++g) {
#pragma unroll
                    for (int ky = 0; ky < KY_LDS; ++ky)
                        wt[(g + 1) & 1][ky] = *(const bf16x8*)(wl + (ky * 6 + (g + 1) % 6) * 1024);
                    if (g == 0) {
                        t_sync();                              // T of output row r - 5 is complete
                        // row r + D goes into row r's own slot: its fragments are in registers (read during row r - 1,
                        // complete at the latest at t_sync's lgkmcnt(0))
                        if (r + D < r_data) issue_row(r + D);
                    }
                    // One scheduling region per chunk: the nine MFMAs and the chunk's share of the output work, interleaved
                    // by the group barriers below -- with one wave per SIMD a cluster of other instructions idles the matrix
                    // pipe, spread out (a few per 32-cycle MFMA) they are free
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 0) t_read(r - 5);
#pragma unroll
                    for (int ky = 8; ky >= 0; --ky) {          // (ky = 8 first: its slot is the one t_write reads)
                        const int s = (j - ky + 13) % 9;       // output row r - ky + 4
                        // (chunk 0, ky = 0 is the first product of output row r + 4: C = 0 instead of clearing the slot)
                        acc[s] = MFMA_BF16(af[g], ky >= KY_LDS ? wf[ky >= KY_LDS ? ky - KY_LDS : 0][g] : wt[g & 1][ky < KY_LDS ? ky : 0],
                                           (g == 0 && ky == 0) ? zero16 : acc[s]);
                    }
                    if (g == 1) t_sum();
                    if (g == 5) t_write(acc[(j + 5) % 9], r - 4);
                    if (g == 0) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                        }
                    } else if (g == 1) {
#pragma unroll
                        for (int i = 0; i < 9; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
                        }
                    } else if (g == 5) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
#pragma unroll
                        for (int i = 0; i < 7; ++i) {
                            __builtin_amdgcn_sched_group_barrier(0x200, 3, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    if (g == 1) t_store(r - 5);
                    if (g == 0 && r + 1 < r_data) wait_rows_after(min(r + D, r_data - 1) - (r + 1));   // row r + 1 has landed
                    af[g] = *(const bf16x8*)(next + g * 32);
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
    const char* e = getenv("SRGEN_HEAD_BANDS");               // development switch (tools/): 1..8, else the default
    const int v = e ? atoi(e) : 0;
    return v >= 1 && v <= 8 ? v : 0;
}

}  // namespace hst

hipError_t launch_head_bf16_stream(const HeadArgs& a, hipStream_t s)
{
    using namespace hst;
    if (a.N < 1) return hipErrorInvalidValue;
    static bool attr_set[64] = {};
    static int cu_count[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && dev >= 0 && dev < 64) cu_count[dev] = cus;
        e = hipFuncSetAttribute((const void*)head_stream_bf16_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)head_stream_bf16_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    // Bands per image (1 .. 8): one workgroup is resident per CU, so the launch takes ceil(workgroups / CUs) rounds of
    // (band rows + 8 halo rows + about 6 rows' worth of start-up) row times; the cheapest count wins (measured, 256 images:
    // one band 234 us, two 250, four 286; 64 images: four bands 65 us, one 158)
    int nbands = env_bands();
    if (!nbands) {
        const int cus = (dev >= 0 && dev < 64 && cu_count[dev] > 0) ? cu_count[dev] : 256;
        long long best = 0;
        for (int nb = 1; nb <= 8; ++nb) {
            const long long rounds = ((long long)a.N * nb + cus - 1) / cus;
            const long long cost = rounds * ((kOut + nb - 1) / nb + (nb > 1 ? 8 : 3) + 6);
            if (!nbands || cost < best) { best = cost; nbands = nb; }
        }
    }
    const int R = (kOut + nbands - 1) / nbands;
    nbands = (kOut + R - 1) / R;                               // (no empty band)
    const long long total = (long long)a.N * nbands;
    if (total > 0x3fffffffLL) return hipErrorInvalidValue;
    const unsigned grid = (unsigned)((total + 7) / 8 * 8);
    if (a.y_f32) hipLaunchKernelGGL(head_stream_bf16_kernel<true>, dim3(grid), dim3(256), LDS_BYTES, s, a, nbands, R, (int)total);
    else hipLaunchKernelGGL(head_stream_bf16_kernel<false>, dim3(grid), dim3(256), LDS_BYTES, s, a, nbands, R, (int)total);
    return hipGetLastError();
}

}  // namespace srgen
