// srdisc_wgrad_bf16.hip -- gfx950 weight gradient of the discriminator's 64..512-channel 3x3 convs with bf16 products
// (the GAN step's 16-bit arithmetic: under the bf16 autocast of /root/reference/train.py:148-152 the backward of every
// conv multiplies bf16 operands), fp32 tensors in HBM, fp32 accumulation.
//
// Reference site: what autograd does for Discriminator.py:9 (F.conv2d(x, w, b, stride, padding=1)) in d_loss.backward()
// (train.py:152):   dW[co][ci][ky][kx] = sum over output pixels dpre[n][oy][ox][co] * in[n][s oy + ky - 1][s ox + kx - 1][ci]
// Per tap a GEMM with M = co, N = ci and K = pixels.  The fp32-product kernel (srdisc_train_kernels.hip) feeds
// v_mfma_f32_16x16x4_f32 straight from memory: 64 - 84 TFLOP/s, 5.2 ms for the eight layers at 128 images -- two thirds of
// the discriminator half-step once its convs ran on bf16 products.
//
// K = pixels is the slow index of both NHWC operands and v_mfma_f32_32x32x16_bf16 wants 8 consecutive K per lane: as in
// srgen_wgrad_f32s.hip the operands are staged in LDS as [pixel][32 channels] bf16 images (64-byte pixel records: the four
// pixel rows a 16-lane group reads fall on disjoint banks) and read TRANSPOSED with ds_read_b64_tr_b16.
//
// Workgroup = (slice of units, 64 co, 64 ci) with all nine taps; wave (a, b) = co half a x ci half b = nine accumulator
// tiles of 32 x 32 (144 registers: one workgroup per CU).  Unit = a spatial tile of output pixels -- 8 x 16 (stride 1) or
// 4 x 16 (stride 2) -- so the input halo costs 1.4x / 1.16x the pixels instead of the 3x of row-by-row staging, and a
// tile row is one K step: its 16 output pixels are 16 consecutive records of the dpre image, and for tap (ky, kx) 16
// consecutive records of the input image (stride 2: even columns are stored in front of the odd ones).  Per unit: the
// next unit's pieces are fetched into registers (fp32, 16 bytes each) before the MFMAs, rounded to bf16 and written to
// the other of two LDS buffers behind them; one barrier per unit.  Partial dW -> slab [slice][tap][Cout][Cin], summed in
// fixed order and transposed to OIHW by disc_wgrad_reduce_kernel; the ci block 0 workgroups also sum dpre (bias).
#include "srgen_internal.h"

namespace srgen {
namespace dwg {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short s4v __attribute__((__vector_size__(8)));
typedef short s8v __attribute__((__vector_size__(16)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

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

template <int S>
struct Geo {
    static constexpr int TH = S == 1 ? 8 : 4, TW = 16, UP = TH * TW;              // output pixels of a unit
    static constexpr int HR = (TH - 1) * S + 3, HC = (TW - 1) * S + 3, NPX = HR * HC;
    static constexpr int CE = (HC + 1) / 2;                                       // stride 2: even columns first
    static constexpr int ND = UP * 16 / 256;                                      // dpre pieces per thread (8 / 4)
    static constexpr int NX = (NPX * 16 + 255) / 256;                             // input pieces per thread (12 / 19)
    static constexpr int D_IMG = UP * 64;                                         // one 32-channel dpre image
    static constexpr int X_IMG = (NPX * 64 + 255) & ~255;                         // one 32-channel input image
    static constexpr int X_OFF = 2 * D_IMG;
    static constexpr int BUF = X_OFF + 2 * X_IMG;                                 // 39,424 / 46,592
};

template <int S>
__global__ __launch_bounds__(256, 1) void disc_wgrad_bf16_kernel(const float* __restrict__ in, const float* __restrict__ dpre, int N,
                                                                 int H, int W, int OH, int OW, int Cin, int Cout, int tiles_x,
                                                                 int tiles_y, int units, int units_per_slice,
                                                                 float* __restrict__ slab, float* __restrict__ bias_slab)
{
    using G = Geo<S>;
    constexpr int TH = G::TH, TW = G::TW, HC = G::HC, NPX = G::NPX, CE = G::CE, ND = G::ND, NX = G::NX;
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wa = w & 1, wb = w >> 1;
    const int ncib = Cin >> 6, ncob = Cout >> 6;
    int bid = blockIdx.x;
    const int cib = bid % ncib;
    bid /= ncib;
    const int cob = bid % ncob, slice = bid / ncob;
    const int co0 = 64 * cob, ci0 = 64 * cib;
    const int u0 = min(slice * units_per_slice, units), u1 = min(u0 + units_per_slice, units);
    const bool do_bias = cib == 0;

    // ---- staging plan: a piece = 4 channels (16 bytes of fp32) of one pixel; thread t always meets channel group t % 16 ----
    const int sub = tid & 15, pcol = tid >> 4;
    const int rec = (sub >> 3) * 1 /* image */, chb = (sub & 7) * 8;
    // dpre: piece k = row k, column pcol of the unit -> record 16 k + pcol of image sub / 8
    const int d_lds = rec * G::D_IMG + pcol * 64 + chb;
    // input: piece k = halo pixel pcol + 16 k
    unsigned xinfo[NX];                              // LDS address | halo row << 16 | halo column << 24
    int xoff[NX];                                    // (halo row * W + halo column) * Cin
#pragma unroll
    for (int k = 0; k < NX; ++k) {
        const int pi = pcol + 16 * k;
        const bool used = pi < NPX;
        const int pc = used ? pi : 0;
        const int hr = pc / HC, hc = pc - hr * HC;
        const int cm = S == 1 ? hc : (hc & 1) * CE + (hc >> 1);
        const int la = G::X_OFF + rec * G::X_IMG + (hr * HC + cm) * 64 + chb;
        xinfo[k] = used ? (unsigned)la | (unsigned)hr << 16 | (unsigned)hc << 24 : 0xFFFFFFFFu;      // (a thread's last piece may not exist)
        xoff[k] = (hr * W + hc) * Cin;
    }
    f32x4 sd[ND], sx[NX];
    unsigned dmask = 0, xmask = 0;
    auto load_unit = [&](int u) {
        const int tx = u % tiles_x, t2 = u / tiles_x;
        const int ty = t2 % tiles_y, n = t2 / tiles_y;
        const int oy0 = ty * TH, ox0 = tx * TW;
        dmask = 0;
        const bool cok = ox0 + pcol < OW;
        const float* dp = dpre + ((size_t)(n * OH + oy0) * OW + ox0 + pcol) * Cout + co0 + sub * 4;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const bool ok = cok && oy0 + k < OH;
            dmask |= (ok ? 1u : 0u) << k;
            sd[k] = *(const f32x4*)(ok ? dp + (size_t)k * OW * Cout : dpre);
        }
        xmask = 0;
        const int iy0 = oy0 * S - 1, ix0 = ox0 * S - 1;
        const float* xp = in + ((ptrdiff_t)(n * H + iy0) * W + ix0) * Cin + ci0 + sub * 4;
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const int hr = (int)((xinfo[k] >> 16) & 255u), hc = (int)(xinfo[k] >> 24);
            const bool ok = xinfo[k] != 0xFFFFFFFFu && (unsigned)(iy0 + hr) < (unsigned)H && (unsigned)(ix0 + hc) < (unsigned)W;
            xmask |= (ok ? 1u : 0u) << k;
            sx[k] = *(const f32x4*)(ok ? xp + xoff[k] : in);
        }
    };
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    auto store_unit = [&](int buf) {
        char* base = lds + buf * G::BUF;
#pragma unroll
        for (int k = 0; k < ND; ++k) {
            const bool ok = (dmask >> k) & 1u;
            const f32x4 v = ok ? sd[k] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (do_bias) bsum += v;
            *(u32x2*)(base + d_lds + k * (16 * 64)) = u32x2{pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w)};
        }
#pragma unroll
        for (int k = 0; k < NX; ++k) {
            const bool ok = (xmask >> k) & 1u;
            const f32x4 v = ok ? sx[k] : f32x4{0.f, 0.f, 0.f, 0.f};
            if (xinfo[k] != 0xFFFFFFFFu) *(u32x2*)(base + (xinfo[k] & 0xFFFFu)) = u32x2{pack2_bf16(v.x, v.y), pack2_bf16(v.z, v.w)};
        }
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[t][e] = 0.f;

    // transposed-read addresses (srgen_wgrad_f32s.hip, tools/trprobe.hip): a 16-lane group fetches 4 pixels x 16 channels,
    // every lane receives one channel's 4 pixels; two reads = the 8 consecutive K of a lane
    const int g = lane >> 4, q = (lane >> 2) & 3, p4 = lane & 3;
    const int troff = (8 * (g >> 1) + q) * 64 + (16 * (g & 1) + 4 * p4) * 2;
    const int offA = wa * G::D_IMG + troff, offB = G::X_OFF + wb * G::X_IMG + troff;

    auto compute = [&](int buf) {
        const char* base = lds + buf * G::BUF;
#pragma unroll
        for (int r = 0; r < TH; ++r) {
            const char* pa = base + offA + 16 * r * 64;
            const bf8 av = cat(tr_read(pa), tr_read(pa + 4 * 64));
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int k0 = S == 1 ? (r + ky) * HC + kx : (2 * r + ky) * HC + (kx & 1) * CE + (kx >> 1);
                    const char* pb = base + offB + k0 * 64;
                    const bf8 bv = cat(tr_read(pb), tr_read(pb + 4 * 64));
                    acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, bv, acc[ky * 3 + kx], 0, 0, 0);
                }
        }
    };

    if (u0 < u1) {
        load_unit(u0);
        store_unit(0);
        barrier();
        int cur = 0;
#pragma unroll 1
        for (int u = u0; u < u1; ++u) {
            const bool has_next = u + 1 < u1;
            if (has_next) load_unit(u + 1);
            compute(cur);
            if (has_next) store_unit(cur ^ 1);
            barrier();
            cur ^= 1;
        }
    }

    // ---- partial tiles -> slab [slice][tap][Cout][Cin]; bias partials ----
    const int hh = lane >> 5, pn = lane & 31;
    float* o = slab + (size_t)slice * 9 * Cout * Cin;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i)
            o[((size_t)t * Cout + co0 + 32 * wa + acc_row(i, hh)) * Cin + ci0 + 32 * wb + pn] = acc[t][i];
    if (do_bias) {
        float* bred = (float*)lds;                   // [256][4]: thread t = 16 j + sub holds channels 4 sub .. 4 sub + 3
        *(f32x4*)(bred + tid * 4) = bsum;
        __syncthreads();
        if (tid < 64) {
            const int sb = tid >> 2, e = tid & 3;
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 16; ++j) s += bred[(16 * j + sb) * 4 + e];
            bias_slab[(size_t)slice * Cout + co0 + tid] = s;
        }
    }
}

static int slices_for(int units, int cin, int cout)
{
    const int tiles = (cin / 64) * (cout / 64);
    int sl = (256 + tiles - 1) / tiles;
    if (sl > units) sl = units;
    return sl < 1 ? 1 : sl;
}

}  // namespace dwg

// an upper bound for any batch and image size (the slice count is capped by 256 / tiles)
size_t disc_wgrad_bf16_scratch_bytes(int cin, int cout)
{
    const int tiles = (cin / 64) * (cout / 64);
    const size_t sl = (size_t)((256 + tiles - 1) / tiles);
    return (sl * 9 * cin * cout + sl * cout) * sizeof(float);
}

hipError_t launch_disc_wgrad_bf16(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride, float* dw,
                                  float* dbias, void* scratch, hipStream_t s)
{
    using namespace dwg;
    if (cin % 64 || cout % 64 || (stride != 1 && stride != 2) || n <= 0 || h < 1 || w < 1) return hipErrorInvalidValue;
    const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
    if ((long long)n * h * w * cin > 0x7fffffffLL || (long long)n * oh * ow * cout > 0x7fffffffLL) return hipErrorInvalidValue;
    const int th = stride == 1 ? Geo<1>::TH : Geo<2>::TH;
    const int tiles_x = (ow + 15) / 16, tiles_y = (oh + th - 1) / th;
    const long long units_ll = (long long)n * tiles_x * tiles_y;
    if (units_ll > 0x7fffffffLL) return hipErrorInvalidValue;
    const int units = (int)units_ll;
    int slices = slices_for(units, cin, cout);
    const int ups = (units + slices - 1) / slices;
    slices = (units + ups - 1) / ups;
    float* slab = (float*)scratch;
    float* bias_slab = slab + (size_t)slices * 9 * cin * cout;
    const unsigned grid = (unsigned)(slices * (cin / 64) * (cout / 64));
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)disc_wgrad_bf16_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Geo<1>::BUF);
        if (e != hipSuccess) return e;
        e = hipFuncSetAttribute((const void*)disc_wgrad_bf16_kernel<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * Geo<2>::BUF);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    if (stride == 1)
        hipLaunchKernelGGL(disc_wgrad_bf16_kernel<1>, dim3(grid), dim3(256), 2 * Geo<1>::BUF, s, in, dpre, n, h, w, oh, ow, cin, cout,
                           tiles_x, tiles_y, units, ups, slab, bias_slab);
    else
        hipLaunchKernelGGL(disc_wgrad_bf16_kernel<2>, dim3(grid), dim3(256), 2 * Geo<2>::BUF, s, in, dpre, n, h, w, oh, ow, cin, cout,
                           tiles_x, tiles_y, units, ups, slab, bias_slab);
    e = hipGetLastError();
    if (e != hipSuccess) return e;
    return launch_disc_wgrad_reduce(slab, bias_slab, slices, cin, cout, dw, dbias, s);
}

}  // namespace srgen
