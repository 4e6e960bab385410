// srgen_wgrad9_bf16.hip -- gfx950 weight gradient of the generator's 9x9 convs against a 3-channel image on the bf16 matrix
// instruction (the 16-bit training step: bf16 activation / gradient tensors, fp32 accumulation).
//
// Reference sites: what autograd does inside the autocast of /root/reference/train.py:91-95 for Generator.py:50 (stem,
// Conv2d(3, 96, 9, padding=4): act = d x1, img = the lr input, flip = 0) and Generator.py:71 (head, Conv2d(96, 3, 9, padding=4):
// act = the head's input, img = d out, flip = 1):
//     r[c][j] = sum over pixels act[n][y][x][c] * img[n][c3][y + s (ky - 4)][x + s (kx - 4)],  j = (c3, ky, kx), s = +1 | -1
// (zero outside the ih x iw image) -- a GEMM with M = 96 channels, N = 243 taps (-> 256) and K = pixels.  The fp32-MFMA kernel
// (srgen_train_kernels.hip: conv9x9_c3_wgrad_kernel) runs it at the fp32 matrix rate: 0.78 ms for the head at 64 images, 6 % of
// the 16-bit step.
//
// K = pixels has to be the per-lane-contiguous index of both MFMA operands:
//   * act (the A operand, 32 channels x 16 pixels): a row of aw pixels is copied global -> LDS as it is (bf16, no
//     registers: global_load_lds with per-lane source offsets) into three [pixel][32 channel] images of 64-byte records and
//     read TRANSPOSED with ds_read_b64_tr_b16 (srgen_wgrad_f32s.hip);
//   * the taps (the B operand, 16 pixels x 32 taps): lane (tap j, K half h) needs 8 consecutive pixels of image row
//     y + s (ky - 4), channel c3, starting at column x0 + 8 h + s (kx - 4) -- 8 consecutive bf16, but at any 2-byte
//     alignment.  Every staged image row therefore exists in EIGHT copies shifted by 0 .. 7 pixels (it is 3 x 144 values:
//     the copies cost 7 KB per row), and a lane reads its fragment with one aligned ds_read_b128 from the copy its kx
//     selects.  Rows live in a ring of ten slots (nine are in use by an output row, the tenth is written for the next).
//   * column 243 of the tap matrix is all ones: r[c][243] = sum of act = the bias gradient, out of the same MFMAs.
// Workgroup = a slice of rows; wave w = 96 channels x taps [64 w, 64 w + 64) = six accumulator tiles; a unit = one row of
// aw pixels = aw / 16 K-steps of six MFMAs per wave, one barrier per unit.  Partial sums -> slab [slice][96][256], added in
// fixed order by conv9x9_c3_wgrad_reduce_kernel.
#include "srgen_internal.h"

namespace srgen {
namespace w9 {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s4v __attribute__((__vector_size__(8)));
typedef short s8v __attribute__((__vector_size__(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int acc_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }
__device__ __forceinline__ s4v tr_read(const char* p)
{
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)p);
}
__device__ __forceinline__ bf8 cat(s4v a, s4v b)
{
    const s8v v = __builtin_shufflevector(a, b, 0, 1, 2, 3, 4, 5, 6, 7);
    return __builtin_bit_cast(bf8, v);
}
__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

constexpr int PAD = 8;                            // padded row: index i holds image column i - PAD
constexpr int COPY_EL = 144, COPY_B = COPY_EL * 2;            // one shifted copy: values i + shift, i < 144
constexpr int CH_B = 8 * COPY_B, ROW_B = 3 * CH_B;            // 2,304 per channel, 6,912 per ring slot
constexpr int NSLOT = 10;
constexpr int CONST_B = 32;                       // 16 bytes of zeros, 16 bytes of bf16 ones

template <int AW>
struct Lds {
    static constexpr int SUB = AW * 64;           // one [pixel][32 channel] image
    static constexpr int ACT = 3 * SUB;           // one act row: 24,576 (AW = 128)
    static constexpr int RING = 2 * ACT;
    static constexpr int CST = RING + NSLOT * ROW_B;
    static constexpr int BYTES = CST + CONST_B;   // 118,304 (AW = 128) / 93,728 (AW = 64)
};

template <int AW>
__global__ __launch_bounds__(256, 1) void conv9x9_c3_wgrad_bf16_kernel(const bf16_t* __restrict__ act, const float* __restrict__ img,
                                                                       int total_rows, int ah, int ih, int iw, int flip,
                                                                       int rows_per_slice, float* __restrict__ slab,
                                                                       float* __restrict__ bias_slab)
{
    using L = Lds<AW>;
    constexpr int NSTEP = AW / 16, NDMA = 3 * AW * 4 / 64 / 4;     // K-steps per row; act copies per wave and row (6 / 3)
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, hh = lane >> 5, pn = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int slice = blockIdx.x;
    const int r0 = min(slice * rows_per_slice, total_rows), r1 = min(r0 + rows_per_slice, total_rows);
    const int sg = flip ? -1 : 1;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;

    if (tid < 8) ((unsigned*)(lds + L::CST))[tid] = tid < 4 ? 0u : 0x3F803F80u;

    // ---- act row copies: 16-byte unit u of sub-image s = pixel u / 4, channels 32 s + 8 (u % 4) ..; wave w issues copies
    // k = w, w + 4, ... of the row's 3 AW / 16 (one copy = 64 units = 1 KB of LDS) ----
    unsigned avoff[NDMA];
#pragma unroll
    for (int i = 0; i < NDMA; ++i) {
        const int k = w + 4 * i;                          // copy index: sub-image k / (AW / 16), units 64 (k % (AW / 16)) + lane
        const int s = k / (AW / 16), u = 64 * (k % (AW / 16)) + lane;
        avoff[i] = (unsigned)(((u >> 2) * 96 + 32 * s + 8 * (u & 3)) * 2);
    }
    auto issue_act = [&](int row, int buf) {
        const char* src = (const char*)(act + (size_t)min(row, total_rows - 1) * AW * 96);
#pragma unroll
        for (int i = 0; i < NDMA; ++i) glds16_raw(src, avoff[i], lds_base + buf * L::ACT + (w + 4 * i) * 1024);
    };
    // ---- image rows: 8 shifted bf16 copies per (row, channel); unit q of copy sh = padded values 8 q + sh .. + 7 ----
    auto stage_img_row = [&](int n, int srow) {           // any srow: rows outside the image are rows of zeros
        const bool inside = (unsigned)srow < (unsigned)ih;
        const float* rp = img + ((size_t)n * 3 * ih + (inside ? srow : 0)) * iw;
        char* dst = lds + L::RING + ((srow + 2 * NSLOT) % NSLOT) * ROW_B;
        for (int i = tid; i < 3 * 8 * (COPY_EL / 8); i += 256) {
            const int c3 = i / (8 * (COPY_EL / 8)), rem = i - c3 * (8 * (COPY_EL / 8));
            const int sh = rem / (COPY_EL / 8), q = rem - sh * (COPY_EL / 8);
            const float* cp = rp + (size_t)c3 * ih * iw;
            unsigned pk[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c0 = 8 * q + sh + 2 * e - PAD, c1 = c0 + 1;
                const float v0 = inside && (unsigned)c0 < (unsigned)iw ? cp[c0] : 0.f;
                const float v1 = inside && (unsigned)c1 < (unsigned)iw ? cp[c1] : 0.f;
                pk[e] = pack2_bf16(v0, v1);
            }
            *(u32x4*)(dst + c3 * CH_B + sh * COPY_B + q * 16) = u32x4{pk[0], pk[1], pk[2], pk[3]};
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0x0070);               // vmcnt(0) lgkmcnt(0): this wave's copies and LDS writes are done
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- this lane's two tap columns ----
    int tdy[2], tcol[2];                                   // row offset s (ky - 4); byte offset of the fragment at x0 = 0, h = 0 inside a ring slot
    int tkind[2];                                          // 0 tap, 1 ones (column 243: the bias sums), 2 zeros
#pragma unroll
    for (int jt = 0; jt < 2; ++jt) {
        const int j = 64 * w + 32 * jt + pn;
        const int jc = j < 243 ? j : 0;
        const int c3 = jc / 81, rem = jc - 81 * c3, ky = rem / 9, kx = rem - 9 * ky;
        const int dx = sg * (kx - 4), st = dx + PAD, sh = st & 7;
        tdy[jt] = sg * (ky - 4);
        tcol[jt] = c3 * CH_B + sh * COPY_B + (st - sh) * 2 + hh * 16;
        tkind[jt] = j < 243 ? 0 : j == 243 ? 1 : 2;
    }
    // transposed-read offset of this lane inside a 16-pixel x 32-channel block (srdisc_wgrad_bf16.hip)
    const int g = lane >> 4, q4 = (lane >> 2) & 3, p4 = lane & 3;
    const int troff = (8 * (g >> 1) + q4) * 64 + (16 * (g & 1) + 4 * p4) * 2;

    f32x16 acc[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[a][b][e] = 0.f;

    int buf = 0;
    if (r0 < r1) issue_act(r0, 0);
#pragma unroll 1
    for (int row = r0; row < r1; ++row) {
        const int n = row / ah, y = row - n * ah;
        if (row == r0 || y == 0) {                         // (wave-uniform) a slice or an image starts: all nine rows
            barrier();                                     // (the ring may still be read by the unit before)
            for (int d = -4; d <= 4; ++d) stage_img_row(n, y + d);
        }
        barrier();                                         // act row `row`, the image rows of this unit are in LDS
        if (row + 1 < r1) {
            issue_act(row + 1, buf ^ 1);
            if (y + 1 < ah) stage_img_row(n, y + 1 + 4);   // into the slot no tap of this unit reads (row y - 5's)
        }
        int baddr[2];
#pragma unroll
        for (int jt = 0; jt < 2; ++jt) {
            const int slot = (y + tdy[jt] + 2 * NSLOT) % NSLOT;
            baddr[jt] = tkind[jt] == 0 ? L::RING + slot * ROW_B + tcol[jt] : L::CST + (tkind[jt] == 1 ? 16 : 0);
        }
        const int bstep0 = tkind[0] == 0 ? 32 : 0, bstep1 = tkind[1] == 0 ? 32 : 0;       // bytes per K-step (16 pixels)
        const char* abase = lds + buf * L::ACT + troff;
#pragma unroll
        for (int ks = 0; ks < NSTEP; ++ks) {
            const bf8 b0 = *(const bf8*)(lds + baddr[0] + ks * bstep0);
            const bf8 b1 = *(const bf8*)(lds + baddr[1] + ks * bstep1);
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const char* pa = abase + a * L::SUB + 16 * ks * 64;
                const bf8 av = cat(tr_read(pa), tr_read(pa + 4 * 64));
                acc[a][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b0, acc[a][0], 0, 0, 0);
                acc[a][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, b1, acc[a][1], 0, 0, 0);
            }
        }
        buf ^= 1;
    }

    // ---- partial sums -> slab [slice][96][256]; the ones column -> bias partials ----
    float* o = slab + (size_t)slice * 96 * 256;
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int c = 32 * a + acc_row(i, hh), j = 64 * w + 32 * b + pn;
                o[(size_t)c * 256 + j] = j < 243 ? acc[a][b][i] : 0.f;
                if (j == 243) bias_slab[(size_t)slice * 96 + c] = acc[a][b][i];
            }
}

}  // namespace w9

// act (n, ah, aw, 96) bf16 with aw = 64 | 128; scratch as conv9x9_c3_wgrad_scratch_bytes
hipError_t launch_conv9x9_c3_wgrad_bf16_mfma(const void* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                             float* dbias, void* scratch, int slices, hipStream_t s)
{
    using namespace w9;
    if ((aw != 64 && aw != 128) || n <= 0 || ah <= 0 || slices < 1) return hipErrorInvalidValue;
    const int total_rows = n * ah;
    const int rps = (total_rows + slices - 1) / slices;
    float* slab = (float*)scratch;
    float* bias_slab = slab + (size_t)slices * 96 * 256;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv9x9_c3_wgrad_bf16_kernel<128>, hipFuncAttributeMaxDynamicSharedMemorySize, Lds<128>::BYTES);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)conv9x9_c3_wgrad_bf16_kernel<64>, hipFuncAttributeMaxDynamicSharedMemorySize, Lds<64>::BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (aw == 128)
        hipLaunchKernelGGL(conv9x9_c3_wgrad_bf16_kernel<128>, dim3((unsigned)slices), dim3(256), Lds<128>::BYTES, s, (const bf16_t*)act, img,
                           total_rows, ah, ih, iw, flip, rps, slab, bias_slab);
    else
        hipLaunchKernelGGL(conv9x9_c3_wgrad_bf16_kernel<64>, dim3((unsigned)slices), dim3(256), Lds<64>::BYTES, s, (const bf16_t*)act, img,
                           total_rows, ah, ih, iw, flip, rps, slab, bias_slab);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_conv9x9_c3_wgrad_reduce(slab, bias_slab, slices, r, dbias, s);
}

}  // namespace srgen
