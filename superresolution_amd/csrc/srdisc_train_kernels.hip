// srdisc_train_kernels.hip -- gfx950 kernels of the GAN phase's backward pass (/root/reference/train.py:145-176):
// the discriminator (Discriminator.py:17-50) differentiated layer by layer, the gradient of the
// relativistic-average losses (Discriminator.py:53-72) and clip_grad_norm_ (train.py:175).
//
// Every conv of the discriminator is 3x3 / pad 1 followed by LeakyReLU(0.2) (all but the last).  With
// out_i = lrelu(pre_i) saved by the forward, dpre_i = dout_i * (out_i > 0 ? 1 : 0.2) -- the sign of out_i is the
// sign of pre_i -- so no pre-activation is kept.  Per layer i:
//   data gradient    dout_{i-1} = conv_transpose(dpre_i, W_i): the FORWARD kernel (srdisc_kernels.hip) on a
//                    device-made pack of the flipped, transposed weights (disc_pack_kernel mode 1), its epilogue
//                    multiplying by lrelu'(out_{i-1}); a stride-2 layer runs one launch per output parity (y mod 2,
//                    x mod 2) with the 1, 2 or 4 taps that reach that parity (disc_pack_par_kernel, PAR kernel variant);
//   weight gradient  disc_wgrad_kernel: GEMM M = co, N = ci, K = output pixels per tap on v_mfma_f32_16x16x4_f32.
// Activations are NHWC fp32, weights and their gradients OIHW fp32 (the torch parameters themselves).
#include "srgen_internal.h"

#include <cstdlib>

namespace srgen {
namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- weight packs made on the device from the live OIHW parameters ----------------------------------------------
// fp32 (disc_conv_kernel, srdisc_kernels.hip):
//   mode 0: forward pack of pack_disc_conv (srdisc_api.cpp): float4 units [cout/64][tap 9][cin/16][s 4][lane 64]
//   mode 1: the same layout for the data-gradient conv: cin' = cout, cout' = cin, w'[co'][ci'][tap] = w[ci'][co'][8 - tap]
__global__ __launch_bounds__(256) void disc_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout,
                                                        int mode, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;           // one packed float
    if (idx >= total) return;
    const int pcin = mode ? cout : cin;                                  // the packed conv's input / output channels
    const int nch = pcin / 16;
    const int j = (int)(idx & 3);
    size_t unit = idx >> 2;
    const int lane = (int)(unit & 63); unit >>= 6;
    const int s = (int)(unit & 3); unit >>= 2;
    const int ch = (int)(unit % nch); unit /= nch;
    const int tap = (int)(unit % 9);
    const int cb = (int)(unit / 9);
    const int m = lane & 15, kq = lane >> 4;
    const int co = 64 * cb + 16 * (m >> 2) + 4 * s + (m & 3);
    const int ci = 16 * ch + 4 * kq + j;
    out[idx] = mode ? w[((size_t)ci * cin + co) * 9 + (8 - tap)] : w[((size_t)co * cin + ci) * 9 + tap];
}

// Packs of a stride-2 layer's data gradient, one per output parity class (py, px) (disc_conv_kernel<.., PAR>): input
// pixel y = 2a + py receives dpre[a + dy] * w[ky] with ky = py + 1 - 2 dy, dy in {0} (py = 0) or {0, 1} (py = 1), and the
// same along x: 1, 2, 2, 4 taps instead of the nine a zero-stuffed stride-1 conv would run.  Tap k of the class =
// (ky', kx') = (1 + dy, 1 + dx), ky' outer; layout [cout'/64][k][cin'/16][s 4][lane 64] with cin' = cout, cout' = cin.
__global__ __launch_bounds__(256) void disc_pack_par_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int cout,
                                                            int py, int px, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int nch = cout / 16, ntaps = (py + 1) * (px + 1);
    const int j = (int)(idx & 3);
    size_t unit = idx >> 2;
    const int lane = (int)(unit & 63); unit >>= 6;
    const int s = (int)(unit & 3); unit >>= 2;
    const int ch = (int)(unit % nch); unit /= nch;
    const int k = (int)(unit % ntaps);
    const int cb = (int)(unit / ntaps);
    const int m = lane & 15, kq = lane >> 4;
    const int cop = 64 * cb + 16 * (m >> 2) + 4 * s + (m & 3);          // packed output channel = the layer's ci
    const int cip = 16 * ch + 4 * kq + j;                                // packed input channel = the layer's co
    const int dy = k / (px + 1), dx = k % (px + 1);
    const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
    out[idx] = w[((size_t)cip * cin + cop) * 9 + ky * 3 + kx];
}

// bf16 (disc_conv_lds_kernel, srdisc_conv_bf16.hip): the same three kinds of pack (kind 0 forward, 1 data gradient of a
// stride-1 layer, 2 parity class (py, px) of a stride-2 layer's data gradient -- taps, flips and transposes as above) in
// that kernel's layout: bf16 [cout'/BN][cin'/32][tap][k16 2][tile BN/32][lane 64][8], BN = 128 when cout' is a multiple of
// 128, else 64; lane (r, h) of tile nt holds w'[co' = BN cb + 32 nt + r][ci' = 32 chunk + 16 k16 + 8 h + 0..7]: the MFMA
// A-operand fragment of v_mfma_f32_32x32x16_bf16, one ds_read_b128 per lane.
__global__ __launch_bounds__(256) void disc_pack_lds_kernel(const float* __restrict__ w, bf16_t* __restrict__ out, int cin, int cout,
                                                            int kind, int py, int px, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;           // one packed bf16
    if (idx >= total) return;
    const int pcin = kind ? cout : cin, pcout = kind ? cin : cout;
    const int bn = pcout % 128 == 0 ? 128 : 64;
    const int ntaps = kind == 2 ? (py + 1) * (px + 1) : 9;
    const int e = (int)(idx & 7), lane = (int)((idx >> 3) & 63);
    size_t unit = idx >> 9;
    const int nt = (int)(unit % (bn / 32)); unit /= bn / 32;
    const int k16 = (int)(unit & 1); unit >>= 1;
    const int tap = (int)(unit % ntaps); unit /= ntaps;
    const int chunk = (int)(unit % (pcin / 32));
    const int cb = (int)(unit / (pcin / 32));
    const int cop = bn * cb + 32 * nt + (lane & 31);
    const int cip = 32 * chunk + 16 * k16 + 8 * (lane >> 5) + e;
    float v;
    if (kind == 0) v = w[((size_t)cop * cin + cip) * 9 + tap];
    else if (kind == 1) v = w[((size_t)cip * cin + cop) * 9 + (8 - tap)];
    else {
        const int dy = tap / (px + 1), dx = tap % (px + 1);
        const int ky = py + 1 - 2 * dy, kx = px + 1 - 2 * dx;
        v = w[((size_t)cip * cin + cop) * 9 + ky * 3 + kx];
    }
    out[idx].bits = (unsigned short)(pack2_bf16(v, 0.f) & 0xffffu);
}

// stem pack [j 7][s 4][lane 64] (pack_disc_stem) and tail pack [tap 9][cin] (pack_disc_tail)
__global__ __launch_bounds__(256) void disc_pack_small_kernel(const float* __restrict__ w, float* __restrict__ out, int cin, int tail)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (tail) {
        if (idx < 9 * cin) out[idx] = w[(size_t)(idx % cin) * 9 + idx / cin];
        return;
    }
    if (idx >= 7 * 4 * 64) return;
    const int lane = idx & 63, s = (idx >> 6) & 3, j = idx >> 8;
    const int m = lane & 15, kq = lane >> 4;
    const int co = 16 * (m >> 2) + 4 * s + (m & 3);
    const int k = 4 * j + kq;
    out[idx] = k < 27 ? w[((size_t)co * 3 + k % 3) * 9 + k / 3] : 0.f;
}

// ---- weight gradient of a Cin -> Cout layer (both multiples of 64), stride 1 or 2 -------------------------------------
// dW[co][ci][ky][kx] = sum over output pixels dpre[n][oy][ox][co] * in[n][s*oy + ky - 1][s*ox + kx - 1][ci].
// Workgroup = (slice of output rows, tap, 64 co, 64 ci), two waves of 32 co x 64 ci = 2 x 4 MFMA tiles.  A = dpre^T:
// lane (i, kq) loads dpre[pixel kq][co0 + 2i .. +1] (tile a: row i = channel co0 + 2i + a); B = in: lane (kq, j) loads
// in[shifted pixel kq][ci0 + 4j .. +3] (tile b: column j = channel ci0 + 4j + b) -- one wide load each, the channel
// permutation undone at the slab store.  Partial sums -> slab [slice][tap][Cout][Cin]; disc_wgrad_reduce_kernel adds
// the slices in fixed order and transposes to OIHW.  The centre tap's workgroups of ci block 0 also sum dpre (bias).
__global__ __launch_bounds__(128) void disc_wgrad_kernel(const float* __restrict__ in, const float* __restrict__ dpre, int total_rows,
                                                         int H, int W, int OH, int OW, int Cin, int Cout, int stride,
                                                         int rows_per_slice, float* __restrict__ slab, float* __restrict__ bias_slab)
{
    const int ncib = Cin >> 6, ncob = Cout >> 6;
    int b = blockIdx.x;
    const int cib = b % ncib; b /= ncib;
    const int cob = b % ncob; b /= ncob;
    const int tap = b % 9;
    const int slice = b / 9;
    const int ky = tap / 3, kx = tap % 3;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const int co0 = 64 * cob + 32 * wv, ci0 = 64 * cib;
    f32x4 acc[2][4];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int q = 0; q < 4; ++q) acc[a][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const int r0 = slice * rows_per_slice;
    const int r1 = min(r0 + rows_per_slice, total_rows);
    for (int row = r0; row < r1; ++row) {
        const int img = row / OH, oy = row - img * OH;
        const int iy = oy * stride + ky - 1;
        if ((unsigned)iy >= (unsigned)H) continue;                       // workgroup-uniform
        const float* dr = dpre + (size_t)row * OW * Cout + co0 + 2 * i;
        const float* xr = in + ((size_t)img * H + iy) * W * Cin + ci0 + 4 * i;
        for (int x0 = 0; x0 < OW; x0 += 4) {
            const int ox = x0 + kq;
            const int ix = ox * stride + kx - 1;
            f32x2 av = {0.f, 0.f};
            f32x4 bv = {0.f, 0.f, 0.f, 0.f};
            if (ox < OW) {
                av = *(const f32x2*)(dr + (size_t)ox * Cout);
                if ((unsigned)ix < (unsigned)W) bv = *(const f32x4*)(xr + (size_t)ix * Cin);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                bsum[a] += av[a];
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[a][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[q], acc[a][q], 0, 0, 0);
            }
        }
    }
    // D[row = 4*(lane>>4) + r][col = lane & 15]: row -> co = co0 + 2 row + a, col -> ci = ci0 + 4 col + q
    float* o = slab + ((size_t)slice * 9 + tap) * Cout * Cin;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            *(f32x4*)(o + (size_t)(co0 + 2 * (4 * kq + r) + a) * Cin + ci0 + 4 * i) =
                f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
    if (tap == 4 && cib == 0) {                                          // the centre tap visits every output row
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) bias_slab[(size_t)slice * Cout + co0 + 2 * i + a] = v;
        }
    }
}

// 64 outputs x 4 slice groups per block (as conv3x3_wgrad_reduce_kernel); accumulate != 0 adds to dw / dbias
__global__ __launch_bounds__(256) void disc_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias_slab,
                                                                int slices, int Cin, int Cout, float* __restrict__ dw,
                                                                float* __restrict__ dbias)
{
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t cc = (size_t)Cout * Cin;
    const size_t idx = (size_t)blockIdx.x * 64 + o;                       // over tap * cc + co * Cin + ci; cc % 64 == 0
    const int tap = (int)(idx / cc);
    const size_t rem = idx - (size_t)tap * cc;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sl = g;
    for (; sl + 12 < slices; sl += 16) {
        s0 += slab[((size_t)sl * 9 + tap) * cc + rem];
        s1 += slab[((size_t)(sl + 4) * 9 + tap) * cc + rem];
        s2 += slab[((size_t)(sl + 8) * 9 + tap) * cc + rem];
        s3 += slab[((size_t)(sl + 12) * 9 + tap) * cc + rem];
    }
    for (; sl < slices; sl += 4) s0 += slab[((size_t)sl * 9 + tap) * cc + rem];
    part[g][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0) dw[rem * 9 + tap] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);      // OIHW
    if (dbias) {
        const int c = blockIdx.x * 256 + threadIdx.x;
        if (c < Cout) {
            float s = 0.f;
            for (int k = 0; k < slices; ++k) s += bias_slab[(size_t)k * Cout + c];
            dbias[c] = s;
        }
    }
}

// ---- stem (3 -> 64, Discriminator.py:23): weight gradient and data gradient ------------------------------------------
// dW0[co][ci][tap] = sum_p dpre[p][co] * img[n][ci][y + ky - 1][x + kx - 1]: M = 64 co (lane i loads channels 4i .. 4i+3:
// four row tiles), N = 27 padded to 32 (two column tiles: k = j and 16 + j, k = tap * 3 + ci gathered from the NCHW image),
// K = pixels.  Every wave walks its own pixel range; partial sums -> slab [workgroup][wave][64][32].
constexpr int STEMW_WGS = 512;
__global__ __launch_bounds__(256) void disc_stem_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ dpre, int N, int H,
                                                              int W, float* __restrict__ slab, float* __restrict__ bias_slab)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int i = lane & 15, kq = lane >> 4;
    const long long total = (long long)N * H * W;
    const long long nsteps = (total + 3) / 4;
    const int nwaves = gridDim.x * 4, me = blockIdx.x * 4 + wv;
    const long long per = (nsteps + nwaves - 1) / nwaves;
    const long long s0 = per * me, s1 = s0 + per < nsteps ? s0 + per : nsteps;
    int koff[2], kdy[2], kdx[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int k = 16 * t + i;
        const int tap = k / 3, ci = k - 3 * tap;
        kdy[t] = k < 27 ? tap / 3 - 1 : -100000;
        kdx[t] = tap % 3 - 1;
        koff[t] = (ci * H + tap / 3 - 1) * W + tap % 3 - 1;
    }
    f32x4 acc[4][2];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
    const int hw = H * W;
    // one step = 4 pixels (kq) x 64 channels; the loads of U steps are requested before the first is multiplied (a step a
    // trip was one dependent memory round trip per step: 251 us for 244 steps per wave); products in the same order
    auto fetch = [&](long long st, f32x4& av, float (&bv)[2]) {
        const long long p = st * 4 + kq;
        const bool pv = p < total;
        const long long pc = pv ? p : 0;
        const int n = (int)(pc / hw), r = (int)(pc - (long long)n * hw);
        const int y = r / W, x = r - y * W;
        av = f32x4{0.f, 0.f, 0.f, 0.f};
        if (pv) av = *(const f32x4*)(dpre + pc * 64 + 4 * i);
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int iy = y + kdy[t], ix = x + kdx[t];
            const bool v = pv && iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float tv = img[(size_t)n * 3 * hw + r + (v ? koff[t] : 0)];
            bv[t] = v ? tv : 0.f;
        }
    };
    auto multiply = [&](const f32x4& av, const float (&bv)[2]) {
        bsum += av;
#pragma unroll
        for (int a = 0; a < 4; ++a)
#pragma unroll
            for (int t = 0; t < 2; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
    };
    constexpr int U = 4;
    long long st = s0;
    for (; st + U <= s1; st += U) {
        f32x4 av[U];
        float bv[U][2];
#pragma unroll
        for (int u = 0; u < U; ++u) fetch(st + u, av[u], bv[u]);
#pragma unroll
        for (int u = 0; u < U; ++u) multiply(av[u], bv[u]);
    }
    for (; st < s1; ++st) {
        f32x4 av;
        float bv[2];
        fetch(st, av, bv);
        multiply(av, bv);
    }
    float* o = slab + (size_t)me * 64 * 32;                               // [co 64][k 32], co = 4 * row + a
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)(4 * (4 * kq + r) + a) * 32 + 16 * t + i] = acc[a][t][r];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        float v = bsum[a];
        v += __shfl_xor(v, 16);
        v += __shfl_xor(v, 32);
        if (kq == 0) bias_slab[(size_t)me * 64 + 4 * i + a] = v;
    }
}

__global__ __launch_bounds__(256) void disc_stem_wgrad_reduce_kernel(const float* __restrict__ slab, const float* __restrict__ bias_slab,
                                                                     int parts, float* __restrict__ dw, float* __restrict__ dbias)
{
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;                                  // co * 32 + k  (blocks 0..31), bias: block 32
    const bool is_bias = blockIdx.x == 32;
    const float* src = is_bias ? bias_slab + o : slab + idx;
    const size_t pitch = is_bias ? 64 : 2048;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int p = g;
    for (; p + 12 < parts; p += 16) {
        s0 += src[(size_t)p * pitch];
        s1 += src[(size_t)(p + 4) * pitch];
        s2 += src[(size_t)(p + 8) * pitch];
        s3 += src[(size_t)(p + 12) * pitch];
    }
    for (; p < parts; p += 4) s0 += src[(size_t)p * pitch];
    part[g][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g != 0) return;
    const float v = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
    if (is_bias) { dbias[o] = v; return; }
    const int co = idx >> 5, k = idx & 31;
    if (k < 27) dw[((size_t)co * 3 + k % 3) * 9 + k / 3] = v;
}

// dimg[n][ci][y][x] = sum_{tap, co} dpre[n][y - ky + 1][x - kx + 1][co] * w[co][ci][tap]  (gradient w.r.t. the image the
// discriminator was fed: the path from the adversarial loss into the generator, train.py:168-174).  Workgroup = 8 x 16
// pixels: the 10 x 18 halo of dpre (64 channels, fp32) goes to LDS once (zero outside the image) -- read from memory, the nine
// taps fetched the tensor nine times from L2 (0.34 ms at 64 images for a 256 MB tensor).  Sixteen lanes per pixel: lane q
// owns channels 4q .. 4q + 3 with its 9 x 4 x 3 weights in registers, a tap is one 16-byte LDS read and twelve FMAs per
// lane, the sixteen partial sums meet in four butterfly steps.  Pixel pitch 272 B (17 x 16 B).
constexpr int SD_TH = 8, SD_TW = 16, SD_HR = SD_TH + 2, SD_HC = SD_TW + 2, SD_PITCH = 272;
__global__ __launch_bounds__(256) void disc_stem_dgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ w, int N, int H,
                                                              int W, int tiles_x, int tiles_y, float* __restrict__ dimg)
{
    __shared__ __attribute__((aligned(16))) char tile[SD_HR * SD_HC * SD_PITCH];       // 48,960 B
    const int q = threadIdx.x & 15, ps = threadIdx.x >> 4;
    float wr[9][4][3];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ci = 0; ci < 3; ++ci) wr[tap][e][ci] = w[((4 * q + e) * 3 + ci) * 9 + tap];     // w is (64, 3, 3, 3) OIHW
    int bid = blockIdx.x;
    const int txb = bid % tiles_x;
    bid /= tiles_x;
    const int tyb = bid % tiles_y, n = bid / tiles_y;
    const int y0 = tyb * SD_TH, x0 = txb * SD_TW;
    const float* src = dpre + (size_t)n * H * W * 64 + 4 * q;
#pragma unroll
    for (int k = 0; k < (SD_HR * SD_HC + 15) / 16; ++k) {
        const int px = ps + 16 * k;
        if (px < SD_HR * SD_HC) {
            const int hr = px / SD_HC, hc = px - hr * SD_HC;
            const int gy = y0 - 1 + hr, gx = x0 - 1 + hc;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if ((unsigned)gy < (unsigned)H && (unsigned)gx < (unsigned)W) v = *(const f32x4*)(src + ((size_t)gy * W + gx) * 64);
            *(f32x4*)(tile + px * SD_PITCH + q * 16) = v;
        }
    }
    __syncthreads();
    const int hw = H * W;
#pragma unroll 2
    for (int pass = 0; pass < SD_TH * SD_TW / 16; ++pass) {
        const int p = ps + 16 * pass, ty = p >> 4, tx = p & 15;
        float g0 = 0.f, g1 = 0.f, g2 = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int hr = ty - tap / 3 + 2, hc = tx - tap % 3 + 2;          // source pixel (y - ky + 1, x - kx + 1) in halo coordinates
            const f32x4 dv = *(const f32x4*)(tile + (hr * SD_HC + hc) * SD_PITCH + q * 16);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                g0 = fmaf(dv[e], wr[tap][e][0], g0);
                g1 = fmaf(dv[e], wr[tap][e][1], g1);
                g2 = fmaf(dv[e], wr[tap][e][2], g2);
            }
        }
#pragma unroll
        for (int o = 8; o >= 1; o >>= 1) {
            g0 += __shfl_xor(g0, o);
            g1 += __shfl_xor(g1, o);
            g2 += __shfl_xor(g2, o);
        }
        const int y = y0 + ty, x = x0 + tx;
        if (q == 0 && y < H && x < W) {
            float* o = dimg + (size_t)n * 3 * hw + (size_t)y * W + x;
            o[0] = g0;
            o[hw] = g1;
            o[2 * hw] = g2;
        }
    }
}

// ---- tail (Cin -> 1, no activation, Discriminator.py:37) backward ---------------------------------------------------------
// din[n][y][x][c] = lrelu'(act) * sum_tap dlogit[n][y - ky + 1][x - kx + 1] * w[c][tap]; act = the tail's input (saved).
__global__ __launch_bounds__(256) void disc_tail_dgrad_kernel(const float* __restrict__ dlogit, const float* __restrict__ w,
                                                              const float* __restrict__ act, int H, int W, int Cin, float slope,
                                                              float* __restrict__ din, size_t total)
{
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;           // over (pixel, c)
    if (idx >= total) return;
    const int c = (int)(idx % Cin);
    size_t p = idx / Cin;
    const int x = (int)(p % W); p /= W;
    const int y = (int)(p % H);
    const size_t n = p / H;
    float g = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
        const int sy = y - tap / 3 + 1, sx = x - tap % 3 + 1;
        if ((unsigned)sy < (unsigned)H && (unsigned)sx < (unsigned)W)
            g = fmaf(dlogit[(n * H + sy) * W + sx], w[(size_t)c * 9 + tap], g);
    }
    din[idx] = act[idx] > 0.f ? g : g * slope;
}

// dw[c][tap] = sum_p dlogit[p] * act[p + tap offset][c]; one workgroup per (tap, 64 channels), 16 pixel groups per
// channel joined through LDS in fixed order; db = sum dlogit (workgroup 0)
__global__ __launch_bounds__(1024) void disc_tail_wgrad_kernel(const float* __restrict__ dlogit, const float* __restrict__ act, int N,
                                                               int H, int W, int Cin, float* __restrict__ dw, float* __restrict__ db)
{
    __shared__ float red[16][64];
    const int tap = blockIdx.x % 9, cb = blockIdx.x / 9;
    const int cl = threadIdx.x & 63, q = threadIdx.x >> 6;
    const int c = cb * 64 + cl;
    const int ky = tap / 3, kx = tap % 3;
    const long long total = (long long)N * H * W;
    // eight pixels of a thread's walk at a time, all sixteen loads requested before the first product (one pixel per trip
    // was one dependent L2 round trip per pixel: 285 us for 512 pixels per thread on 36 workgroups); fixed order
    float s = 0.f;
    long long p = q;
    for (; p + 16 * 7 < total; p += 16 * 8) {
        float dl[8], av[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const long long pp = p + 16 * u;
            const int x = (int)(pp % W);
            const int y = (int)((pp / W) % H);
            const int iy = y + ky - 1, ix = x + kx - 1;
            const bool ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
            dl[u] = ok ? dlogit[pp] : 0.f;
            av[u] = ok ? act[(size_t)(pp + (long long)(ky - 1) * W + (kx - 1)) * Cin + c] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) s = fmaf(dl[u], av[u], s);
    }
    for (; p < total; p += 16) {
        const int x = (int)(p % W);
        const int y = (int)((p / W) % H);
        const int iy = y + ky - 1, ix = x + kx - 1;
        if ((unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W)
            s = fmaf(dlogit[p], act[(size_t)(p + (long long)(ky - 1) * W + (kx - 1)) * Cin + c], s);
    }
    red[q][cl] = s;
    __syncthreads();
    if (q == 0) {
        float t = 0.f;
        for (int k = 0; k < 16; ++k) t += red[k][cl];
        dw[(size_t)c * 9 + tap] = t;
    }
    if (blockIdx.x == 0) {
        __syncthreads();
        float t = 0.f;
        for (long long p = threadIdx.x; p < total; p += 1024) t += dlogit[p];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) t += __shfl_xor(t, off);
        if (cl == 0) red[q][0] = t;
        __syncthreads();
        if (threadIdx.x == 0) {
            float u = 0.f;
            for (int k = 0; k < 16; ++k) u += red[k][0];
            db[0] = u;
        }
    }
}

// ---- gradient of the relativistic-average losses (Discriminator.py:60-72) ---------------------------------------------
// L = scale * 0.5 * [mean_i bce(r_i - mean(f), t_r) + mean_j bce(f_j - mean(r), t_f)];  d bce(x, t) / dx = sigmoid(x) - t.
//   dL/dr_i = scale * 0.5 * [(sig(x_i) - t_r) / nr - S_f / nr],  S_f = mean_j (sig(y_j) - t_f)
//   dL/df_j = scale * 0.5 * [(sig(y_j) - t_f) / nf - S_r / nf],  S_r = mean_i (sig(x_i) - t_r)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

__global__ __launch_bounds__(1024) void ragan_grad_kernel(const float* __restrict__ real, const float* __restrict__ fake, size_t nr,
                                                          size_t nf, float t_real, float t_fake, float scale,
                                                          float* __restrict__ dreal, float* __restrict__ dfake)
{
    __shared__ double red[16][2];
    auto block_sum2 = [&](double& a, double& b) {
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); b += __shfl_xor(b, off); }
        __syncthreads();
        if (lane == 0) { red[wv][0] = a; red[wv][1] = b; }
        __syncthreads();
        a = 0.0; b = 0.0;
        for (int k = 0; k < 16; ++k) { a += red[k][0]; b += red[k][1]; }
    };
    double mr = 0.0, mf = 0.0;
    for (size_t i = threadIdx.x; i < nr; i += 1024) mr += (double)real[i];
    for (size_t i = threadIdx.x; i < nf; i += 1024) mf += (double)fake[i];
    block_sum2(mr, mf);
    const float meanr = (float)(mr / (double)nr), meanf = (float)(mf / (double)nf);
    double sr = 0.0, sf = 0.0;
    for (size_t i = threadIdx.x; i < nr; i += 1024) sr += (double)(sigmoidf_(real[i] - meanf) - t_real);
    for (size_t i = threadIdx.x; i < nf; i += 1024) sf += (double)(sigmoidf_(fake[i] - meanr) - t_fake);
    block_sum2(sr, sf);
    const float Sr = (float)(sr / (double)nr), Sf = (float)(sf / (double)nf);
    const float kr = 0.5f * scale / (float)nr, kf = 0.5f * scale / (float)nf;
    if (dreal)
        for (size_t i = threadIdx.x; i < nr; i += 1024) dreal[i] = kr * ((sigmoidf_(real[i] - meanf) - t_real) - Sf);
    for (size_t i = threadIdx.x; i < nf; i += 1024) dfake[i] = kf * ((sigmoidf_(fake[i] - meanr) - t_fake) - Sr);
}

// ---- torch.nn.utils.clip_grad_norm_ over one flat gradient buffer (train.py:175) -------------------------------------------
constexpr int CLIP_BLOCKS = 512;
__global__ __launch_bounds__(256) void sumsq_partial_kernel(const float* __restrict__ g, size_t n, double* __restrict__ partial)
{
    __shared__ double red[4];
    float s = 0.f;
    double acc = 0.0;
    int k = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        s = fmaf(g[i], g[i], s);
        if (++k == 64) { acc += (double)s; s = 0.f; k = 0; }
    }
    acc += (double)s;
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// norm_coef[0] = total norm, norm_coef[1] = min(1, max_norm / (norm + 1e-6)) -- every workgroup recomputes them from
// the partials (512 doubles) and scales its share
__global__ __launch_bounds__(256) void clip_scale_kernel(float* __restrict__ g, size_t n, const double* __restrict__ partial, int nb,
                                                         float max_norm, float* __restrict__ norm_coef)
{
    __shared__ float coef_s;
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < nb; ++k) t += partial[k];
        const float norm = (float)sqrt(t);
        const float c = max_norm / (norm + 1e-6f);
        coef_s = c < 1.f ? c : 1.f;
        if (blockIdx.x == 0) { norm_coef[0] = norm; norm_coef[1] = coef_s; }
    }
    __syncthreads();
    const float c = coef_s;
    if (c >= 1.f) return;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) g[i] *= c;
}

}  // namespace

hipError_t launch_disc_pack(const float* w, float* out, int cin, int cout, int mode, hipStream_t s, bool bf16)
{
    if (mode == 2 || mode == 3) {                                        // 2: stem pack, 3: tail pack
        if (bf16) return hipErrorInvalidValue;                           // (those two layers keep fp32 products)
        const int total = mode == 3 ? 9 * cin : 7 * 4 * 64;
        hipLaunchKernelGGL(disc_pack_small_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out, cin, mode == 3);
        return hipGetLastError();
    }
    if (cin % 64 || cout % 64) return hipErrorInvalidValue;
    if (mode == 4) {                                                     // the four parity-class packs back to back
        size_t off = 0;
        for (int c = 0; c < 4; ++c) {
            const int py = c >> 1, px = c & 1;
            const size_t total = (size_t)(py + 1) * (px + 1) * cin * cout;
            if (bf16)
                hipLaunchKernelGGL(disc_pack_lds_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w,
                                   (bf16_t*)out + off, cin, cout, 2, py, px, total);
            else
                hipLaunchKernelGGL(disc_pack_par_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out + off,
                                   cin, cout, py, px, total);
            off += total;
        }
        return hipGetLastError();
    }
    const size_t total = (size_t)cin * cout * 9;
    if (bf16)
        hipLaunchKernelGGL(disc_pack_lds_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, (bf16_t*)out, cin, cout,
                           mode, 0, 0, total);
    else
        hipLaunchKernelGGL(disc_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, w, out, cin, cout, mode, total);
    return hipGetLastError();
}

static int disc_wgrad_slices(int total_rows, int cin, int cout)
{
    const int tiles = 9 * (cin / 64) * (cout / 64);
    int sl = (4096 + tiles - 1) / tiles;
    if (sl > 512) sl = 512;
    if (sl > total_rows) sl = total_rows;
    return sl < 1 ? 1 : sl;
}

size_t disc_wgrad_scratch_bytes(int n, int oh, int cin, int cout)
{
    const int sl = disc_wgrad_slices(n * oh, cin, cout);
    return ((size_t)sl * 9 * cin * cout + (size_t)sl * cout) * sizeof(float);
}

hipError_t launch_disc_wgrad(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride, float* dw,
                             float* dbias, void* scratch, hipStream_t s)
{
    if (cin % 64 || cout % 64 || (stride != 1 && stride != 2) || n <= 0) return hipErrorInvalidValue;
    const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
    const int total_rows = n * oh;
    const int slices = disc_wgrad_slices(total_rows, cin, cout);
    const int rps = (total_rows + slices - 1) / slices;
    float* slab = (float*)scratch;
    float* bias_slab = slab + (size_t)slices * 9 * cin * cout;
    const long long blocks = (long long)slices * 9 * (cin / 64) * (cout / 64);
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(disc_wgrad_kernel, dim3((unsigned)blocks), dim3(128), 0, s, in, dpre, total_rows, h, w, oh, ow, cin, cout,
                       stride, rps, slab, bias_slab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(disc_wgrad_reduce_kernel, dim3((unsigned)((size_t)9 * cin * cout / 64)), dim3(256), 0, s, slab, bias_slab, slices,
                       cin, cout, dw, dbias);
    return hipGetLastError();
}

hipError_t launch_disc_wgrad_reduce(const float* slab, const float* bias_slab, int slices, int cin, int cout, float* dw, float* dbias,
                                    hipStream_t s)
{
    hipLaunchKernelGGL(disc_wgrad_reduce_kernel, dim3((unsigned)((size_t)9 * cin * cout / 64)), dim3(256), 0, s, slab, bias_slab, slices,
                       cin, cout, dw, dbias);
    return hipGetLastError();
}

size_t disc_stem_wgrad_scratch_bytes() { return ((size_t)STEMW_WGS * 4 * 2048 + (size_t)STEMW_WGS * 4 * 64) * sizeof(float); }

hipError_t launch_disc_stem_wgrad(const float* img, const float* dpre, int n, int h, int w, float* dw, float* dbias, void* scratch,
                                  hipStream_t s)
{
    float* slab = (float*)scratch;
    float* bias_slab = slab + (size_t)STEMW_WGS * 4 * 2048;
    hipLaunchKernelGGL(disc_stem_wgrad_kernel, dim3(STEMW_WGS), dim3(256), 0, s, img, dpre, n, h, w, slab, bias_slab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(disc_stem_wgrad_reduce_kernel, dim3(33), dim3(256), 0, s, slab, bias_slab, STEMW_WGS * 4, dw,
                       dbias);
    return hipGetLastError();
}

hipError_t launch_disc_stem_dgrad(const float* dpre, const float* w, int n, int h, int wd, float* dimg, hipStream_t s)
{
    if (n <= 0 || h <= 0 || wd <= 0) return hipSuccess;
    const int tiles_x = (wd + SD_TW - 1) / SD_TW, tiles_y = (h + SD_TH - 1) / SD_TH;
    const long long blocks = (long long)n * tiles_x * tiles_y;
    if (blocks > 0x7fffffffLL) return hipErrorInvalidValue;
    hipLaunchKernelGGL(disc_stem_dgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, s, dpre, w, n, h, wd, tiles_x, tiles_y, dimg);
    return hipGetLastError();
}

hipError_t launch_disc_tail_bwd(const float* dlogit, const float* w, const float* act, int n, int h, int wd, int cin, float slope,
                                float* din, float* dw, float* db, hipStream_t s)
{
    if (cin % 64) return hipErrorInvalidValue;
    const size_t total = (size_t)n * h * wd * cin;
    hipLaunchKernelGGL(disc_tail_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, dlogit, w, act, h, wd, cin, slope,
                       din, total);
    if (dw) hipLaunchKernelGGL(disc_tail_wgrad_kernel, dim3((unsigned)(9 * (cin / 64))), dim3(1024), 0, s, dlogit, act, n, h, wd, cin, dw, db);
    return hipGetLastError();
}

hipError_t launch_ragan_grad(const float* real, const float* fake, size_t nr, size_t nf, float t_real, float t_fake, float scale,
                             float* dreal, float* dfake, hipStream_t s)
{
    hipLaunchKernelGGL(ragan_grad_kernel, dim3(1), dim3(1024), 0, s, real, fake, nr, nf, t_real, t_fake, scale, dreal, dfake);
    return hipGetLastError();
}

size_t clip_scratch_bytes() { return (size_t)CLIP_BLOCKS * sizeof(double); }

hipError_t launch_clip_grad_norm(float* g, size_t n, float max_norm, float* norm_coef, void* scratch, hipStream_t s)
{
    double* partial = (double*)scratch;
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, s, g, n, partial);
    hipLaunchKernelGGL(clip_scale_kernel, dim3(CLIP_BLOCKS), dim3(256), 0, s, g, n, partial, CLIP_BLOCKS, max_norm, norm_coef);
    return hipGetLastError();
}

}  // namespace srgen
