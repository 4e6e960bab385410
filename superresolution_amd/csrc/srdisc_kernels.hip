// srdisc_kernels.hip -- gfx950 kernels of the PatchGAN discriminator forward
// (/root/reference/Discriminator.py:17-50) and the relativistic-average loss (:53-72).
//
// Activations between layers are NHWC fp32.  Every conv is 3x3, pad 1 (Discriminator.py:9,23,35,37):
//   disc_stem_kernel   3 -> 64, stride 1: K = 27 padded to 28 on the same MFMA; reads the caller's NCHW
//                      input, writes NHWC through an LDS transpose (the layer is store-bound).
//   disc_conv_kernel   Cin -> Cout, stride 1 or 2: implicit GEMM on v_mfma_f32_16x16x4_f32.
//                      A = weights (rows = 16 output channels), B = activations (columns = 16 output
//                      pixels), so a lane ends up with 4 consecutive channels of one pixel per
//                      accumulator and, through the channel permutation of the weight pack, 16
//                      consecutive channels over its four accumulators: 64-byte contiguous NHWC
//                      stores.  A lane's B fragment for four k-steps is ONE 16-byte load (4 consecutive
//                      input channels of its pixel), its A fragment one 16-byte load of the pack; no
//                      LDS: the four waves of a workgroup share the weight fragments through L1.
//   disc_tail_kernel   Cin -> 1: one wave per output pixel, lanes over input channels.
//   ragan_loss_kernel  both losses of RelativisticAverageLoss in one single-workgroup launch.
#include "srgen_internal.h"

#include <type_traits>

#include <cstdlib>

namespace srgen {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float lrelu(float v, float slope) { return v > 0.f ? v : v * slope; }

// ---- stem -------------------------------------------------------------------------------------
// 3 -> 64 on the same MFMA, K = 27 padded to 28 = seven k-steps: lane (l16, kq) feeds k = 4*j + kq of
// pixel l16 in step j, k = (ky*3 + kx)*3 + ci, gathered straight from the caller's NCHW planes (the
// input is 3 floats per pixel against 64 written: the gather lives in L1).  A wave keeps its 28
// weight values in registers and walks STEM_GROUPS consecutive 16-pixel groups.  The layer is bound
// by its 256-byte-per-pixel output, so the accumulators (a lane holds 16 channels of one pixel: 16-byte
// pieces 256 bytes apart across lanes) are transposed through LDS and leave as 1 KB-contiguous stores.
constexpr int STEM_GROUPS = 8;                  // 16-pixel groups per wave
constexpr int STEM_ROW = 272;                   // LDS row pitch in bytes: 256 + 16, conflict-free both ways

__global__ __launch_bounds__(256) void disc_stem_kernel(DiscStemArgs a, int total)
{
    __shared__ __attribute__((aligned(16))) char lds[4 * 16 * STEM_ROW];
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int hw = a.H * a.W;

    // weights: [j 7][s 4][lane 64], bias: the lane's 16 channels kq*16 .. kq*16 + 15
    float w[7][4];
#pragma unroll
    for (int j = 0; j < 7; ++j)
#pragma unroll
        for (int s = 0; s < 4; ++s) w[j][s] = a.w[(j * 4 + s) * 64 + lane];
    f32x4 bias[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) bias[s] = *reinterpret_cast<const f32x4*>(a.bias + kq * 16 + s * 4);

    // this lane's seven k's: plane/row/column offset of tap k relative to the output pixel
    int koff[7], kdy[7], kdx[7];
#pragma unroll
    for (int j = 0; j < 7; ++j) {
        const int k = 4 * j + kq;
        const int tap = k / 3, ci = k - 3 * tap;
        const int ky = tap / 3, kx = tap - 3 * ky;
        kdy[j] = k < 27 ? ky - 1 : -100000;     // k == 27 is the zero pad: never inside the image
        kdx[j] = kx - 1;
        koff[j] = (ci * a.H + ky - 1) * a.W + kx - 1;
    }

    char* my = lds + wv * 16 * STEM_ROW;
    const int first = (blockIdx.x * 4 + wv) * (16 * STEM_GROUPS);
#pragma unroll 1
    for (int gi = 0; gi < STEM_GROUPS; ++gi) {
        const int p0 = first + gi * 16;
        if (p0 >= total) break;                                  // wave-uniform
        const int p = p0 + l16;
        const bool pv = p < total;
        const int pc = pv ? p : 0;
        const int img = pc / hw, r = pc - img * hw;
        const int y = r / a.W, x = r - y * a.W;
        const float* src = a.x + (size_t)img * 3 * hw + r;
        float xk[7];
#pragma unroll
        for (int j = 0; j < 7; ++j) {
            const int iy = y + kdy[j], ix = x + kdx[j];
            const bool v = pv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const float t = src[v ? koff[j] : 0];
            xk[j] = v ? t : 0.f;
        }
        f32x4 acc[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) acc[s] = bias[s];
#pragma unroll
        for (int j = 0; j < 7; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc[s] = __builtin_amdgcn_mfma_f32_16x16x4f32(w[j][s], xk[j], acc[s], 0, 0, 0);
        // accumulator (row m = 4*kq + r of tile s, column l16) = channel kq*16 + s*4 + r of pixel l16
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const f32x4 v = {lrelu(acc[s][0], a.slope), lrelu(acc[s][1], a.slope), lrelu(acc[s][2], a.slope),
                             lrelu(acc[s][3], a.slope)};
            *reinterpret_cast<f32x4*>(my + l16 * STEM_ROW + kq * 64 + s * 16) = v;
        }
        // same wave wrote and reads (LDS executes a wave's accesses in order): no barrier
        float* o = a.out + (size_t)p0 * 64;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int px = i * 4 + kq;
            const f32x4 v = *reinterpret_cast<const f32x4*>(my + px * STEM_ROW + l16 * 16);
            if (p0 + px < total) *reinterpret_cast<f32x4*>(o + px * 64 + l16 * 4) = v;
        }
    }
}

// ---- Cin -> Cout implicit GEMM ------------------------------------------------------------------
// workgroup = 4 waves = 64*G output pixels x 64 output channels; wave = 16*G pixels (G 16-pixel column
// groups) x 64 channels (four 16-row tiles) = 4*G accumulators.  k runs over (tap, 16-channel chunk);
// per chunk a lane loads G activation and 4 weight float4s for 16*G MFMAs, the next chunk's loads
// issued before the current chunk's MFMAs.
//
// VALU instructions between MFMAs cost issue slots the matrix pipe then idles for (the first version
// of this kernel computed the tap address and zero-masked the fragment per chunk and ran at 75 % of
// the MFMA rate), so the chunk loop carries none: chunks advance in steps of four with immediate
// offsets from one per-lane pointer, a lane's nine tap validities are one bit mask computed up front,
// and an out-of-image tap points the lane at a page of zeros (DiscConvArgs::zeros) instead of
// masking the loaded values.  Per step: one 64-bit pointer update per pixel group; per tap: a
// select between the image address and the zero page.
template <int G>
struct ConvTile {
    int p[G];                   // output pixel of each 16-pixel group (may lie past M)
    unsigned vmask[G];          // bit t: tap t of that pixel lies inside the image (0 past M)
    const float* base[G];       // address of tap (0,0), channel 4*kq (may be out of image)
    const char* wp;             // this lane's weight fragments of the tile's 64-channel block (16-byte units)
    int cb;
};

// PAR (stride-2 data gradient, one output parity class per launch; srdisc_train_kernels.hip): the conv runs over the
// class grid OH x OW of the upper layer's dpre with the 1, 2 or 4 taps of a.taps (4-bit tap ids, real taps first, id 15 =
// padding that reads zeros), and output pixel (oy, ox) of the class is pixel (2 oy + py, 2 ox + px) of the FH x FW image.
//
template <int STRIDE, int G, bool PAR = false>
__global__ __launch_bounds__(256, G == 4 ? 1 : 2) void disc_conv_kernel(DiscConvArgs a, int OH, int OW, int M,
                                                                        int lg_nst, int ntiles)
{
    using WFrag = f32x4;
    constexpr int WU = sizeof(WFrag);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int ncb = a.Cout >> 6;
    const int nst = 1 << lg_nst;               // steps (of 4 chunks = 64 input channels) per tap
    const int total = (PAR ? a.ntaps : 9) * nst;
    const float* zero = a.zeros + kq * 4;

    auto setup = [&](ConvTile<G>& T, int t) {
        T.cb = t % ncb;
        const int tile = t / ncb;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int p = tile * (64 * G) + wv * (16 * G) + g * 16 + l16;
            const bool pv = p < M;
            const int pc = pv ? p : 0;
            const int img = pc / (OH * OW);
            const int r = pc - img * (OH * OW);
            const int oy = r / OW, ox = r - oy * OW;
            const int iy0 = oy * STRIDE - 1, ix0 = ox * STRIDE - 1;
            unsigned m = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                const int iy = iy0 + k / 3, ix = ix0 + k % 3;
                if (pv && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) m |= 1u << k;
            }
            T.p[g] = !PAR ? p : pv ? (img * a.FH + 2 * oy + a.py) * a.FW + 2 * ox + a.px : -1;
            T.vmask[g] = m;
            T.base[g] = a.in + (((long long)img * a.H + iy0) * a.W + ix0) * a.Cin + kq * 4;
        }
        T.wp = reinterpret_cast<const char*>(a.wpack) + ((size_t)T.cb * total * 1024 + lane) * WU;
    };

    const float* cur[G];                        // the activation pointers (chunk 0) of the step being loaded
    auto set_step = [&](const ConvTile<G>& T, int st) {
        const int c4 = st & (nst - 1);
        const int tap = PAR ? (int)((a.taps >> (4 * (st >> lg_nst))) & 15u) : st >> lg_nst;      // 15: padding tap
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int toff = (ky * a.W + kx) * a.Cin + c4 * 64;       // wave-uniform
#pragma unroll
        for (int g = 0; g < G; ++g) cur[g] = ((T.vmask[g] >> tap) & 1u) ? T.base[g] + toff : zero + c4 * 64;
    };
    // activation fragments: ring of four chunks, issued three chunks ahead (they come from HBM / the
    // other XCDs' traffic); weight fragments (L2-resident: an XCD sees one or two 64-channel blocks):
    // two buffers, one chunk ahead.  Weights are issued BEFORE the activations of the same slot: vmcnt
    // retires in order, so waiting for w(i) must not also wait for the younger x(i+2).
    f32x4 xb[4][G];
    WFrag wb[2][4];
    auto load_x = [&](int slot, int c) {
#pragma unroll
        for (int g = 0; g < G; ++g) xb[slot][g] = *reinterpret_cast<const f32x4*>(cur[g] + c * 16);
    };
    auto load_w = [&](int slot, const char* wp, int st, int c) {
        if (PAR && (st >> lg_nst) >= a.nreal) st &= nst - 1;     // a padding tap multiplies zeros: any real weights do
        const WFrag* w = reinterpret_cast<const WFrag*>(wp) + ((size_t)st * 4 + c) * 256;
#pragma unroll
        for (int s = 0; s < 4; ++s) wb[slot][s] = w[s * 64];
    };

    f32x4 acc[4][G];
    auto clear = [&]() {
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g) acc[s][g] = f32x4{0.f, 0.f, 0.f, 0.f};
    };
    auto mma = [&](int ws, int xs) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int g = 0; g < G; ++g)
                    acc[s][g] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[ws][s][j], xb[xs][g][j], acc[s][g], 0, 0, 0);
    };

    // One step = four chunks of tile T at step st; its prefetches reach into step sn of tile N (N is T
    // except on a tile's last step, where it is the workgroup's next tile: the pipeline never drains
    // between tiles).  sched_barrier: keep the compiler from regrouping the loads (it otherwise
    // clusters several chunks' loads and waits on them a few MFMAs later).
    auto do_step = [&](const ConvTile<G>& T, int st, const ConvTile<G>& N, int sn) {
        load_w(1, T.wp, st, 1);
        load_x(3, 3);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        load_w(0, T.wp, st, 2);
        set_step(N, sn);
        load_x(0, 0);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        load_w(1, T.wp, st, 3);
        load_x(1, 1);
        __builtin_amdgcn_sched_barrier(0);
        mma(0, 2);
        __builtin_amdgcn_sched_barrier(0);
        load_w(0, N.wp, sn, 0);
        load_x(2, 2);
        __builtin_amdgcn_sched_barrier(0);
        mma(1, 3);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto epilogue = [&](const ConvTile<G>& T) {
#pragma unroll
        for (int g = 0; g < G; ++g) {
            if (PAR ? T.p[g] < 0 : T.p[g] >= M) continue;
            float* o = a.out + (size_t)T.p[g] * a.Cout + T.cb * 64 + kq * 16;
            if (a.mask) {
                // data-gradient use (srdisc_train_kernels.hip): no bias; times the LeakyReLU derivative at the
                // saved activation of the layer below, which has this output's shape
                const float* mk = a.mask + (size_t)T.p[g] * a.Cout + T.cb * 64 + kq * 16;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const f32x4 mv = *reinterpret_cast<const f32x4*>(mk + s * 4);
                    f32x4 r = acc[s][g];
#pragma unroll
                    for (int e = 0; e < 4; ++e) r[e] = mv[e] > 0.f ? r[e] : r[e] * a.slope;
                    *reinterpret_cast<f32x4*>(o + s * 4) = r;
                }
                continue;
            }
            const float* b = a.bias + T.cb * 64 + kq * 16;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const f32x4 bv = *reinterpret_cast<const f32x4*>(b + s * 4);
                f32x4 r = acc[s][g] + bv;
                r = f32x4{lrelu(r[0], a.slope), lrelu(r[1], a.slope), lrelu(r[2], a.slope), lrelu(r[3], a.slope)};
                *reinterpret_cast<f32x4*>(o + s * 4) = r;
            }
        }
    };

    // Persistent: a workgroup walks tiles blockIdx.x, + gridDim.x, ...  The tile loop is rotated by one
    // step -- its body is steps 1 .. total-1 of a tile, the epilogue, and step 0 of the next tile -- so
    // that the conservative wait the compiler places at a loop header (everything but the current
    // chunk's loads) never sits right behind the epilogue's stores or the next tile's first loads.
    // Inner trips of three steps for the same reason (one such wait per twelve chunks).
    ConvTile<G> T;
    setup(T, blockIdx.x);
    set_step(T, 0);
    load_x(0, 0);
    load_x(1, 1);
    load_w(0, T.wp, 0, 0);
    load_x(2, 2);
    clear();
    do_step(T, 0, T, 1);
    for (int t = blockIdx.x; t < ntiles; t += gridDim.x) {
        if (PAR) {                               // total is 2 .. 32, not a multiple of three: plain loop
            for (int st = 1; st < total - 1; ++st) do_step(T, st, T, st + 1);
        } else {
            do_step(T, 1, T, 2);
            do_step(T, 2, T, 3);
            for (int st = 3; st < total - 3; st += 3) {
                do_step(T, st, T, st + 1);
                do_step(T, st + 1, T, st + 2);
                do_step(T, st + 2, T, st + 3);
            }
            do_step(T, total - 3, T, total - 2);
            do_step(T, total - 2, T, total - 1);
        }
        const int tn = t + gridDim.x;
        ConvTile<G> N;
        setup(N, tn < ntiles ? tn : t);         // past the last tile: harmless re-loads of this one
        do_step(T, total - 1, N, 0);
        epilogue(T);
        T = N;
        clear();
        if (tn < ntiles) do_step(T, 0, T, 1);
    }
}

// ---- Cin -> 1 tail ------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void disc_tail_kernel(DiscTailArgs a, int total)
{
    const int lane = threadIdx.x & 63;
    const int p = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (p >= total) return;                                  // wave-uniform
    const int hw = a.H * a.W;
    const int img = p / hw, r = p - img * hw;
    const int y = r / a.W, x = r - y * a.W;
    float sum = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int iy = y + ky - 1, ix = x + kx - 1;
        if (iy < 0 || iy >= a.H || ix < 0 || ix >= a.W) continue;
        const float* src = a.in + (((size_t)img * a.H + iy) * a.W + ix) * a.Cin;
        const float* w = a.w + (size_t)tap * a.Cin;
        for (int c = lane * 4; c < a.Cin; c += 256) {
            const f32x4 xv = *reinterpret_cast<const f32x4*>(src + c);
            const f32x4 wv = *reinterpret_cast<const f32x4*>(w + c);
            sum = fmaf(xv[0], wv[0], sum);
            sum = fmaf(xv[1], wv[1], sum);
            sum = fmaf(xv[2], wv[2], sum);
            sum = fmaf(xv[3], wv[3], sum);
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) a.y[p] = sum + a.bias[0];
}

// ---- relativistic-average loss ---------------------------------------------------------------------
// nn.BCEWithLogitsLoss()(x, t) per element: max(x, 0) - x*t + log1p(exp(-|x|))
__device__ __forceinline__ float bce_logits(float x, float t)
{
    return fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
}

template <int N>
__device__ void block_sum(double (&v)[N], double (*red)[N])
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
    __syncthreads();                                         // red may still be read from the last use
    if (lane == 0)
        for (int i = 0; i < N; ++i) red[wv][i] = v[i];
    __syncthreads();
    for (int i = 0; i < N; ++i) {
        double t = 0.0;
        for (int w = 0; w < nw; ++w) t += red[w][i];
        v[i] = t;
    }
}

__global__ __launch_bounds__(1024) void ragan_loss_kernel(const float* __restrict__ real,
                                                          const float* __restrict__ fake, size_t nr, size_t nf,
                                                          float real_label, float fake_label, float* out2)
{
    __shared__ double red[16][4];
    double m[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t i = threadIdx.x; i < nr; i += blockDim.x) m[0] += (double)real[i];
    for (size_t i = threadIdx.x; i < nf; i += blockDim.x) m[1] += (double)fake[i];
    block_sum(m, red);
    const float mr = (float)(m[0] / (double)nr), mf = (float)(m[1] / (double)nf);   // real.mean(), fake.mean()
    double t[4] = {0.0, 0.0, 0.0, 0.0};
    for (size_t i = threadIdx.x; i < nr; i += blockDim.x) {
        const float x = real[i] - mf;                        // r_vs_f           Discriminator.py:61,68
        t[0] += (double)bce_logits(x, real_label);           // :63
        t[2] += (double)bce_logits(x, 0.f);                  // :70
    }
    for (size_t i = threadIdx.x; i < nf; i += blockDim.x) {
        const float x = fake[i] - mr;                        // f_vs_r           :62,69
        t[1] += (double)bce_logits(x, fake_label);           // :64
        t[3] += (double)bce_logits(x, 1.f);                  // :71
    }
    block_sum(t, red);
    if (threadIdx.x == 0) {
        out2[0] = (float)((t[0] / (double)nr + t[1] / (double)nf) * 0.5);   // :65
        out2[1] = (float)((t[2] / (double)nr + t[3] / (double)nf) * 0.5);   // :72
    }
}

}  // namespace

hipError_t launch_disc_stem(const DiscStemArgs& a, hipStream_t s)
{
    const long long total = (long long)a.N * a.H * a.W;
    if (total <= 0) return hipSuccess;
    if (total > 0x7fffffffLL) return hipErrorInvalidValue;
    const long long per_wg = 4 * 16 * STEM_GROUPS;
    hipLaunchKernelGGL(disc_stem_kernel, dim3((unsigned)((total + per_wg - 1) / per_wg)), dim3(256), 0, s, a,
                       (int)total);
    return hipGetLastError();
}

static hipError_t launch_conv_grid(const DiscConvArgs& a, int OH, int OW, long long M, bool par, hipStream_t s)
{
    if (a.bf16) return par ? hipErrorInvalidValue : launch_disc_conv_bf16(a, OH, OW, s);      // bf16 products: the LDS-tiled kernel
    const int nst = a.Cin >> 6;                 // steps of 64 input channels per tap
    int lg = 0;
    while ((1 << lg) < nst) ++lg;
    // 64 pixels per wave (16 accumulators) when that still gives every CU a workgroup, 32 otherwise
    // (small batches / the late layers)
    constexpr int wide_min = 256;
    const bool wide = ((M + 255) / 256) * (a.Cout >> 6) >= wide_min;
    const int wg_px = wide ? 256 : 128;
    const long long blocks = ((M + wg_px - 1) / wg_px) * (a.Cout >> 6);
    if (M > 0x7fffffffLL || blocks > 0x7fffffffLL || (long long)a.N * a.H * a.W > 0x7fffffffLL)
        return hipErrorInvalidValue;
    // Occupancy: ONE workgroup (one wave per SIMD) per CU for the 64-pixel variant -- its 268 registers
    // allow no more, and that is the fast point: at batch 256 the L2-fed layers ran 135-140 TFLOP/s that
    // way against 107-120 with two or three workgroups per CU taking turns on the matrix pipe and
    // evicting each other's weight fragments from L1; the prefetch ring hides the latency alone.
    const int lds_pad = 0;
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const int per_cu = wide ? 1 : 2;
    const long long resident = (long long)cus * per_cu;
    const dim3 grid((unsigned)(blocks < resident ? blocks : resident)), block(256);
    const int ntiles = (int)blocks;
#define DISC_LAUNCH(ST, GG, PP) hipLaunchKernelGGL((disc_conv_kernel<ST, GG, PP>), grid, block, lds_pad, s, a, OH, OW, (int)M, lg, ntiles)
    if (par && wide) DISC_LAUNCH(1, 4, true);
    else if (par) DISC_LAUNCH(1, 2, true);
    else if (a.stride == 1 && wide) DISC_LAUNCH(1, 4, false);
    else if (a.stride == 1) DISC_LAUNCH(1, 2, false);
    else if (wide) DISC_LAUNCH(2, 4, false);
    else DISC_LAUNCH(2, 2, false);
#undef DISC_LAUNCH
    return hipGetLastError();
}

static bool conv_args_ok(const DiscConvArgs& a)
{
    const int nst = a.Cin >> 6;
    return !(a.Cin % 64 || a.Cin > kDiscZeroFloats || a.Cout % 64 || (nst & (nst - 1)) || !a.zeros || a.H < 1 || a.W < 1);
}

hipError_t launch_disc_conv(const DiscConvArgs& a, hipStream_t s)
{
    if (a.N <= 0) return hipSuccess;
    if (!conv_args_ok(a) || (a.stride != 1 && a.stride != 2)) return hipErrorInvalidValue;
    const int OH = (a.H - 1) / a.stride + 1, OW = (a.W - 1) / a.stride + 1;
    return launch_conv_grid(a, OH, OW, (long long)a.N * OH * OW, false, s);
}

// Data gradient of a stride-2 layer: a0.in = dpre (N, H, W, Cin) of that layer (H x W its OUTPUT grid), a0.out / a0.mask
// (N, fh, fw, Cout) its input grid; wpack4 = the four class packs back to back (disc_pack mode 4 .. 7: 1, 2, 2, 4 taps).
hipError_t launch_disc_dgrad2(const DiscConvArgs& a0, const float* wpack4, int fh, int fw, hipStream_t s)
{
    if (a0.N <= 0) return hipSuccess;
    if (!conv_args_ok(a0) || !a0.mask || (fh - 1) / 2 + 1 != a0.H || (fw - 1) / 2 + 1 != a0.W) return hipErrorInvalidValue;
    if (a0.bf16) return launch_disc_dgrad2_bf16(a0, wpack4, fh, fw, s);        // one launch for the four classes (srdisc_conv_bf16.hip)
    const int nst = a0.Cin >> 6;
    size_t woff = 0;
    for (int c = 0; c < 4; ++c) {
        const int py = c >> 1, px = c & 1;
        const int ntaps = (py + 1) * (px + 1);
        DiscConvArgs a = a0;
        a.stride = 1;
        a.wpack = a0.bf16 ? (const float*)((const unsigned short*)wpack4 + woff) : wpack4 + woff;
        woff += (size_t)ntaps * a0.Cin * a0.Cout;
        const int CH = (fh - py + 1) / 2, CW = (fw - px + 1) / 2;           // rows / columns of that parity in the image
        if (CH <= 0 || CW <= 0) continue;
        unsigned long long taps = 0;
        int k = 0;
        for (int ky = 1; ky <= 1 + py; ++ky)
            for (int kx = 1; kx <= 1 + px; ++kx) taps |= (unsigned long long)(ky * 3 + kx) << (4 * k++);
        a.nreal = ntaps;
        while (k * nst < 2) taps |= 15ull << (4 * k++);                     // the pipeline wants two steps
        a.taps = taps;
        a.ntaps = k;
        a.py = py; a.px = px; a.FH = fh; a.FW = fw;
        hipError_t e = launch_conv_grid(a, CH, CW, (long long)a.N * CH * CW, true, s);
        if (e != hipSuccess) return e;
    }
    return hipSuccess;
}

hipError_t launch_disc_tail(const DiscTailArgs& a, hipStream_t s)
{
    const long long total = (long long)a.N * a.H * a.W;
    if (total <= 0) return hipSuccess;
    if (total > 0x7fffffffLL || a.Cin % 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(disc_tail_kernel, dim3((unsigned)((total + 3) / 4)), dim3(256), 0, s, a, (int)total);
    return hipGetLastError();
}

hipError_t launch_ragan_loss(const float* real, const float* fake, size_t n_real, size_t n_fake,
                             float real_label, float fake_label, float* out2, hipStream_t s)
{
    hipLaunchKernelGGL(ragan_loss_kernel, dim3(1), dim3(1024), 0, s, real, fake, n_real, n_fake, real_label,
                       fake_label, out2);
    return hipGetLastError();
}

}  // namespace srgen
