// srdisc_conv_bf16.hip -- gfx950 3x3 conv of the PatchGAN discriminator with bf16 products (the GAN step's 16-bit
// arithmetic: /root/reference/train.py:148-150,166-168 run the discriminator under bf16 autocast), LDS-tiled.
//
// Reference sites: Discriminator.py:9 (DiscBlock conv, stride 1 | 2, pad 1), :10 (LeakyReLU 0.2, the epilogue), :23-31,35
// (the eight 64..512-channel layers this kernel serves); their data gradients are the same kernel on flipped / transposed
// packs (stride 1) or on one output-parity class of the input grid at a time (stride 2; srdisc_train_kernels.hip).
//
// Tensors stay fp32 NHWC in HBM (what the fp32-product kernel of srdisc_kernels.hip reads and writes, so every other
// operator of the step -- masks, weight gradients, stem, tail -- is untouched); an activation is rounded to bf16 when its
// halo tile is staged into LDS, the weights come from a bf16 pack, accumulation is fp32 on v_mfma_f32_32x32x16_bf16.
//
// Why LDS.  The register-only kernel reads a lane's operand fragments straight from L1 / L2: right for the fp32 MFMA
// (one 16-byte load feeds 256 cycles of matrix work) but with bf16 products the same loads feed 1/8 of the cycles, every
// tap re-reads the pixels (9x the tensor from L2) and the kernel ran at the L2's rate (0.22 ms for the 64 -> 128 layer at
// 128 images, 340 TFLOP/s).  Here a workgroup stages the halo of its 128 output pixels ONCE per 32-channel chunk and
// reads the nine shifted windows from LDS.
//
// Structure.  Workgroup = 4 waves = 128 output pixels (8 x 16 of one image, or 8 x 8 of two images when the grid is
// 8 wide) x BN output channels (128, or 64 for the 64-channel layers); wave (wm, wn) = 64 pixels x BN/2 channels = 2 x
// BN/64 accumulator tiles of 32 x 32.  GEMM view as srgen_conv3x3_bf16.hip: weights = MFMA A operand (rows = output
// channels), pixels = B operand, so a lane ends with 4 consecutive channels of one pixel per register group (16-byte NHWC
// stores).  K = Cin/32 chunks x taps x 2 k16-steps:
//   * pixels: halo tile ((8-1) s + 3) x ((TW-1) s + 3) of one chunk, fp32 -> registers (one chunk ahead, issued in the
//     chunk's first tap) -> bf16 -> LDS at a pixel pitch of 80 B (5 x 16 B: conflict-free ds_read_b128); for stride 2 the
//     even columns are stored in front of the odd ones, so the 16 pixels of a fragment row stay 80 B apart;
//   * weights: one slab per (chunk, tap) -- 2 k16-steps x BN/32 tiles x 1 KB in lane order, a fragment = one
//     ds_read_b128 -- copied global -> LDS without registers (global_load_lds) into a three-slot ring, two taps ahead,
//     waited for with a COUNTED vmcnt (every wave issues the same number of copies; loads return in order) so the copy
//     issued in this tap and the pixel loads stay in flight across the barrier that ends it.
// One barrier per tap, two at a chunk end; 40 - 71 KB of LDS: two or three workgroups per CU cover each other's waits.
#include "srgen_internal.h"

namespace srgen {
namespace dcl {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

constexpr int PITCH = 80;                         // staged pixel: 32 ci x bf16 = 64 B + 16 B pad
constexpr int NRING = 3;

template <int S, bool NARROW>
struct Geo {
    static constexpr int TW = NARROW ? 8 : 16, TH = 8, TI = NARROW ? 2 : 1;       // 128 output pixels
    static constexpr int HR = (TH - 1) * S + 3, HC = (TW - 1) * S + 3;            // halo rows / columns per image
    static constexpr int NPX = TI * HR * HC;
    static constexpr int A_BYTES = (NPX * PITCH + 15) & ~15;
    static constexpr int NA = (NPX * 8 + 255) / 256;                              // 16-byte fp32 pieces per thread and chunk
    static constexpr int CE = (HC + 1) / 2;                                       // stride 2: even columns first
};

// s_waitcnt vmcnt(n) lgkmcnt(0)
constexpr int wait_imm(int n) { return (n & 15) | (7 << 4) | ((n >> 4) << 14); }
template <int N>
__device__ __forceinline__ void wait_vm_lgkm0()
{
    static_assert(N >= 0 && N < 64, "vmcnt is six bits");
    __builtin_amdgcn_s_waitcnt(wait_imm(N));
}

// async 16-byte-per-lane global -> LDS copy (see srgen_conv3x3_bf16.hip: inline asm so that the compiler's wait-count pass
// does not serialise every LDS read behind it); sbase wave-uniform, lds_addr = LDS byte address of the wave's 1 KB piece
__device__ __forceinline__ void glds16_raw(const void* sbase, unsigned voff, unsigned lds_addr)
{
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1"
                 :: "v"(voff), "s"(sbase), "s"(lds_addr) : "memory", "m0");
}

// One launch runs up to four convs over the same input with different tap lists, weight packs and output pixel sets:
// the four output-parity classes of a stride-2 data gradient (1, 2, 2 and 4 taps: separate launches left the 16 x 16 and
// 8 x 8 layers with a quarter of the chip each); a plain conv is one class.
struct ConvClass {
    int OH, OW;                        // the grid this class runs over
    int NT, py, px;                    // taps; output pixel (oy, ox) of the class is pixel (2 oy + py, 2 ox + px) (parity classes)
    unsigned long long taps;           // 4-bit tap ids (ky * 3 + kx)
    long long wbytes;                  // byte offset of the class's weight pack behind a.wpack
};
// Work item ids run (image group, tile row, tile column, class, channel block), the last fastest, over the tiling of the
// LARGEST class grid (class 0's): the classes of one tile read the same input halo, so they sit next to each other in one
// XCD's run and the halo comes from HBM once (class-major order had different XCDs read the gradient four times: 355 us
// for the 125 x 125 layer at 128 images, 1.75 GB moved for 1.15 GB); a class whose own grid ends before a tile leaves at once
struct ConvClasses { ConvClass c[4]; int n, tiles_x, tiles_y; };

template <int S, bool NARROW, int BN>
__global__ __launch_bounds__(256, 2) void disc_conv_lds_kernel(DiscConvArgs a, ConvClasses cls, int total_wgs)
{
    using G = Geo<S, NARROW>;
    constexpr int TW = G::TW, TH = G::TH, TI = G::TI, HR = G::HR, HC = G::HC, NPX = G::NPX, NA = G::NA, CE = G::CE;
    constexpr int A_BYTES = G::A_BYTES;
    constexpr int SLAB = BN * 64;                 // one (chunk, tap): 2 k16-steps x BN/32 tiles x 1 KB
    constexpr int NB = BN / 64;                   // 1 KB pieces of a slab per wave
    constexpr int NTL = BN / 64;                  // 32-channel tiles per wave
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, p = lane & 31;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wm = w & 1, wn = w >> 1;

    // workgroup ids b, b + 8, ... share an XCD (private L2): XCD x takes a contiguous run of (tile, channel block) pairs,
    // the channel blocks of one tile next to each other, so a halo tile is fetched into one L2
    const int per_xcd = (total_wgs + 7) >> 3;
    int lid = ((int)blockIdx.x & 7) * per_xcd + ((int)blockIdx.x >> 3);
    if (lid >= total_wgs) return;
    const int ncb = a.Cout / BN;
    const int cb = lid % ncb;
    int tile = lid / ncb;
    const int ci = tile % cls.n;
    tile /= cls.n;
    ConvClass k = cls.c[0];
#pragma unroll
    for (int j = 1; j < 4; ++j)
        if (ci == j) k = cls.c[j];                                  // (wave-uniform)
    const int OH = k.OH, OW = k.OW, tiles_x = cls.tiles_x, tiles_y = cls.tiles_y, NT = k.NT;
    const unsigned long long taps = k.taps;
    const int tx = tile % tiles_x;
    tile /= tiles_x;
    const int ty = tile % tiles_y, ig = tile / tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, img0 = ig * TI;
    if (oy0 >= OH || ox0 >= OW) return;                             // (whole workgroup: no barrier has been reached)
    const bool par = a.FH > 0;

    // ---- pixel staging plan: piece i = tid + 256 it is the 16-byte unit i % 8 (4 channels) of halo pixel i / 8 ----
    unsigned goff[NA];                            // byte offsets: one scalar base + a 32-bit vector offset per load
    int ldsa[NA];
    unsigned okbits = 0, usedbits = 0;
#pragma unroll
    for (int it = 0; it < NA; ++it) {
        const int i = tid + 256 * it, pix = i >> 3, sub = i & 7;
        const bool used = pix < NPX;
        const int pc = used ? pix : 0;
        const int ti = pc / (HR * HC), rem = pc - ti * (HR * HC);
        const int hr = rem / HC, hc = rem - hr * HC;
        const int img = img0 + ti, iy = oy0 * S - 1 + hr, ix = ox0 * S - 1 + hc;
        const bool ok = used && img < a.N && (unsigned)iy < (unsigned)a.H && (unsigned)ix < (unsigned)a.W;
        goff[it] = ok ? (unsigned)(((img * a.H + iy) * a.W + ix) * a.Cin + sub * 4) * 4u : 0u;
        const int cm = S == 1 ? hc : (hc & 1) * CE + (hc >> 1);
        ldsa[it] = ((ti * HR + hr) * HC + cm) * PITCH + sub * 8;
        okbits |= (ok ? 1u : 0u) << it;
        usedbits |= (used ? 1u : 0u) << it;
    }
    f32x4 stage[NA];
    auto load_a = [&](int c) {
        const char* in = (const char*)(a.in + c * 32);
#pragma unroll
        for (int it = 0; it < NA; ++it) stage[it] = *(const f32x4*)(in + goff[it]);
    };
    auto store_a = [&]() {
#pragma unroll
        for (int it = 0; it < NA; ++it) {
            const bool ok = (okbits >> it) & 1u;
            const f32x4 v = stage[it];
            const u32x2 o = {ok ? pack2_bf16(v.x, v.y) : 0u, ok ? pack2_bf16(v.z, v.w) : 0u};
            if ((usedbits >> it) & 1u) *(u32x2*)(lds + ldsa[it]) = o;
        }
    };

    // ---- weight slabs: [cb][chunk][tap] x SLAB bytes, wave w copies pieces w, w + 4 ----
    const int nchunks = a.Cin >> 5, nslabs = nchunks * NT;
    const char* const wsrc = (const char*)a.wpack + k.wbytes + (size_t)cb * nslabs * SLAB;
    const unsigned lds_base = (unsigned)(size_t)(__attribute__((address_space(3))) char*)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    auto issue_slab = [&](int j, int slot) {
        const char* src = wsrc + (size_t)j * SLAB;
        const unsigned dst = lds_base + A_BYTES + slot * SLAB;
#pragma unroll
        for (int k = 0; k < NB; ++k) glds16_raw(src + (w + 4 * k) * 1024, lane16, dst + (w + 4 * k) * 1024);
    };
    auto barrier = [&]() {
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };

    // ---- this lane's fragment addresses and output pixels ----
    int abase[2], ooff[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int q = 64 * wm + 32 * mt + p;
        const int c = q % TW, r = (q / TW) % TH, ti = q / (TW * TH);
        abase[mt] = ((ti * HR + r * S) * HC + c) * PITCH + h * 16;
        const int img = img0 + ti, oy = oy0 + r, ox = ox0 + c;
        const bool valid = img < a.N && oy < OH && ox < OW;
        const int pix = par ? (img * a.FH + 2 * oy + k.py) * a.FW + 2 * ox + k.px : (img * OH + oy) * OW + ox;
        ooff[mt] = valid ? pix * a.Cout + cb * BN + wn * (BN / 2) + 4 * h : -1;
    }

    f32x16 acc[2][NTL];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[mt][i][e] = 0.f;

    issue_slab(0, 0);
    if (nslabs > 1) issue_slab(1, 1);
    load_a(0);
    store_a();
    wait_vm_lgkm0<0>();
    barrier();

    int t = 0, c = 0, slot = 0;
#pragma unroll 1
    for (int j = 0; j < nslabs; ++j) {
        const bool more_slab = j + 2 < nslabs;
        const bool first = t == 0, last = t == NT - 1, more_chunk = c + 1 < nchunks;
        if (more_slab) issue_slab(j + 2, slot >= 1 ? slot - 1 : 2);          // (slot + 2) % 3: read last in tap j - 1
        if (first && more_chunk) {
            load_a(c + 1);
            asm volatile("" ::: "memory");        // the loads are issued HERE: the counted wait below relies on it
        }
        const int id = (int)((taps >> (4 * t)) & 15ull);
        const int dy = id / 3, dx = id - 3 * dy;
        const int toff = (S == 1 ? dy * HC + dx : dy * HC + (dx & 1) * CE + (dx >> 1)) * PITCH;
        const char* const A0 = lds + abase[0] + toff;
        const char* const A1 = lds + abase[1] + toff;
        const char* const Wb = lds + A_BYTES + slot * SLAB + wn * NTL * 1024 + lane * 16;
        bf16x8 x[2][2], wf[2][NTL];
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            x[s][0] = *(const bf16x8*)(A0 + 32 * s);
            x[s][1] = *(const bf16x8*)(A1 + 32 * s);
#pragma unroll
            for (int i = 0; i < NTL; ++i) wf[s][i] = *(const bf16x8*)(Wb + (s * (BN / 32) + i) * 1024);
        }
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int i = 0; i < NTL; ++i)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) acc[mt][i] = MFMA_BF16(wf[s][i], x[s][mt], acc[mt][i]);
        if (last && more_chunk) {
            __builtin_amdgcn_s_waitcnt(0xC07F);   // lgkmcnt(0): this wave's reads of the chunk are done
            barrier();                            // ... and every other wave's
            store_a();
        }
        // slab j + 1 has landed (it is older than everything that may stay in flight: the copy issued in this tap and,
        // in a chunk's first tap, the pixel loads behind it), this wave's LDS writes are done
        if (more_slab) {
            if (first && more_chunk) wait_vm_lgkm0<NB + NA>();
            else wait_vm_lgkm0<NB>();
        } else {
            if (first && more_chunk) wait_vm_lgkm0<NA>();
            else wait_vm_lgkm0<0>();
        }
        barrier();
        slot = slot == 2 ? 0 : slot + 1;
        if (last) { t = 0; ++c; } else ++t;
    }

    // ---- epilogue: (+ bias, LeakyReLU) or (x LeakyReLU' at the saved activation), 16-byte NHWC stores ----
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        if (ooff[mt] < 0) continue;
        float* o = a.out + ooff[mt];
#pragma unroll
        for (int i = 0; i < NTL; ++i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                f32x4 v = {acc[mt][i][4 * g], acc[mt][i][4 * g + 1], acc[mt][i][4 * g + 2], acc[mt][i][4 * g + 3]};
                if (a.mask) {
                    const f32x4 mv = *(const f32x4*)(a.mask + ooff[mt] + 32 * i + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = mv[e] > 0.f ? v[e] : v[e] * a.slope;
                } else {
                    const f32x4 bv = *(const f32x4*)(a.bias + cb * BN + wn * (BN / 2) + 4 * h + 32 * i + 8 * g);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float s = v[e] + bv[e];
                        v[e] = s > 0.f ? s : s * a.slope;
                    }
                }
                *(f32x4*)(o + 32 * i + 8 * g) = v;
            }
    }
}

template <int S, bool NARROW, int BN>
static hipError_t launch_one(const DiscConvArgs& a, ConvClasses cls, int nclasses, hipStream_t s)
{
    using G = Geo<S, NARROW>;
    constexpr int lds_bytes = G::A_BYTES + NRING * BN * 64;
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)disc_conv_lds_kernel<S, NARROW, BN>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                lds_bytes);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int groups = (a.N + G::TI - 1) / G::TI;
    int oh = 0, ow = 0;
    for (int j = 0; j < nclasses; ++j) {
        if (cls.c[j].OH > oh) oh = cls.c[j].OH;
        if (cls.c[j].OW > ow) ow = cls.c[j].OW;
    }
    cls.n = nclasses;
    cls.tiles_x = (ow + G::TW - 1) / G::TW;
    cls.tiles_y = (oh + G::TH - 1) / G::TH;
    const long long total = (long long)groups * cls.tiles_y * cls.tiles_x * nclasses * (a.Cout / BN);
    if (total > 0x3fffffffLL) return hipErrorInvalidValue;
    if (total <= 0) return hipSuccess;
    const unsigned grid = (unsigned)((total + 7) / 8 * 8);
    hipLaunchKernelGGL((disc_conv_lds_kernel<S, NARROW, BN>), dim3(grid), dim3(256), lds_bytes, s, a, cls, (int)total);
    return hipGetLastError();
}

static hipError_t dispatch(const DiscConvArgs& a, const ConvClasses& cls, int nclasses, int stride, bool narrow, hipStream_t s)
{
    const bool wide_n = a.Cout % 128 == 0;
#define DCL_GO(S_, NAR_, BN_) return launch_one<S_, NAR_, BN_>(a, cls, nclasses, s)
    if (stride == 1) {
        if (narrow) { if (wide_n) DCL_GO(1, true, 128); else DCL_GO(1, true, 64); }
        if (wide_n) DCL_GO(1, false, 128); else DCL_GO(1, false, 64);
    }
    if (stride == 2) {
        if (narrow) { if (wide_n) DCL_GO(2, true, 128); else DCL_GO(2, true, 64); }
        if (wide_n) DCL_GO(2, false, 128); else DCL_GO(2, false, 64);
    }
#undef DCL_GO
    return hipErrorInvalidValue;
}

static bool sizes_ok(const DiscConvArgs& a, long long out_px)
{
    if (a.Cin % 32 || a.Cout % 64 || a.N < 1 || a.H < 1 || a.W < 1) return false;
    if ((long long)a.N * a.H * a.W * a.Cin > 0x3fffffffLL) return false;       // 32-bit byte offsets
    return out_px * a.Cout <= 0x7fffffffLL;
}

}  // namespace dcl

// the a.bf16 convs of launch_disc_conv (srdisc_kernels.hip): wpack = launch_disc_pack(.., mode 0 | 1, bf16 = true)
hipError_t launch_disc_conv_bf16(const DiscConvArgs& a, int OH, int OW, hipStream_t s)
{
    using namespace dcl;
    if (OH < 1 || OW < 1 || !sizes_ok(a, (long long)a.N * OH * OW) || (a.stride != 1 && a.stride != 2)) return hipErrorInvalidValue;
    DiscConvArgs b = a;
    b.FH = b.FW = 0;                                 // (not a parity class)
    ConvClasses cls = {};
    cls.c[0].OH = OH;
    cls.c[0].OW = OW;
    cls.c[0].NT = 9;
    cls.c[0].taps = 0x876543210ull;
    return dispatch(b, cls, 1, a.stride, OW <= 8, s);
}

// the a0.bf16 data gradient of a stride-2 layer (launch_disc_dgrad2): the four output-parity classes of the fh x fw input
// grid in ONE launch; a0.H x a0.W = the layer's output grid (dpre), wpack4 = the four class packs back to back
// (launch_disc_pack mode 4, bf16: 1, 2, 2 and 4 taps)
hipError_t launch_disc_dgrad2_bf16(const DiscConvArgs& a0, const void* wpack4, int fh, int fw, hipStream_t s)
{
    using namespace dcl;
    if (fh < 1 || fw < 1 || !a0.mask || !sizes_ok(a0, (long long)a0.N * fh * fw)) return hipErrorInvalidValue;
    DiscConvArgs a = a0;
    a.stride = 1;
    a.wpack = (const float*)wpack4;
    a.FH = fh;
    a.FW = fw;
    ConvClasses cls = {};
    long long woff = 0;
    int n = 0, max_cw = 0;
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const int ntaps = (py + 1) * (px + 1);
        const int CH = (fh - py + 1) / 2, CW = (fw - px + 1) / 2;          // rows / columns of that parity in the image
        if (CH > 0 && CW > 0) {
            ConvClass& k = cls.c[n++];
            k.OH = CH;
            k.OW = CW;
            k.NT = ntaps;
            k.py = py;
            k.px = px;
            k.wbytes = woff * 2;                     // bf16
            int t = 0;
            for (int ky = 1; ky <= 1 + py; ++ky)
                for (int kx = 1; kx <= 1 + px; ++kx) k.taps |= (unsigned long long)(ky * 3 + kx) << (4 * t++);
            if (CW > max_cw) max_cw = CW;
        }
        woff += (long long)ntaps * a0.Cin * a0.Cout;
    }
    if (n == 0) return hipSuccess;
    return dispatch(a, cls, n, 1, max_cw <= 8, s);
}

}  // namespace srgen
