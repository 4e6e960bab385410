// srgen_kernels_misc.hip -- HBM-bound helpers: layout export for parity taps, the optional
// bilinear resize, and the validate() metrics (PSNR / SSIM / L1).
#include "srgen_internal.h"

#include <mutex>

namespace srgen {

__device__ __forceinline__ float bf16_bits_to_f32(unsigned short b)
{
    return __uint_as_float(((unsigned)b) << 16);
}

// ---- NHWC (fp32 or bf16) -> NCHW fp32, via a 32x32 LDS transpose -----------------------------
__global__ __launch_bounds__(256) void nhwc_to_nchw_kernel(const void* in, int in_is_bf16,
                                                           float* out, int HW, int C)
{
    __shared__ float t[32][33];
    const int n = blockIdx.z, p0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 32 x 8
    for (int r = ty; r < 32; r += 8) {
        const int pix = p0 + r, c = c0 + tx;
        float v = 0.f;
        if (pix < HW && c < C) {
            const size_t idx = ((size_t)n * HW + pix) * C + c;
            v = in_is_bf16 ? bf16_bits_to_f32(((const unsigned short*)in)[idx]) : ((const float*)in)[idx];
        }
        t[r][tx] = v;
    }
    __syncthreads();
    for (int r = ty; r < 32; r += 8) {
        const int c = c0 + r, pix = p0 + tx;
        if (pix < HW && c < C) out[((size_t)n * C + c) * HW + pix] = t[tx][r];
    }
}

hipError_t launch_nhwc_to_nchw_f32(const void* in, int in_is_bf16, float* out, int N, int HW, int C,
                                   hipStream_t s)
{
    dim3 grid((unsigned)((HW + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, grid, dim3(256), 0, s, in, in_is_bf16, out, HW, C);
    return hipGetLastError();
}

// ---- input normalisation: clamp(x / hr_max, 0, 1) -- dataset.py:52-54 -------------------------
// The division is a true IEEE fp32 division (not a multiply by the reciprocal), so the result is
// bit-identical to torch.clamp(x / hr_max, 0, 1).  HBM-bound: 16 bytes per lane in, 16 (fp32) or
// 8 (bf16) bytes out; the tail (numel % 4) is handled by the last lanes one element at a time.
typedef float nf32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float norm1(float v, float hr_max)
{
    // torch.clamp(v / hr_max, 0, 1) = min(max(q, 0), 1); NaN propagates as in ATen
    const float q = __fdiv_rn(v, hr_max);
    const float lo = q < 0.f ? 0.f : q;
    return lo > 1.f ? 1.f : lo;
}

template <bool BF16_OUT>
__global__ __launch_bounds__(256) void normalize_kernel(const float* __restrict__ x, void* __restrict__ y,
                                                        size_t numel, float hr_max)
{
    const size_t n4 = numel / 4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const nf32x4 v = ((const nf32x4*)x)[i];
        const float r0 = norm1(v.x, hr_max), r1 = norm1(v.y, hr_max), r2 = norm1(v.z, hr_max), r3 = norm1(v.w, hr_max);
        if (BF16_OUT) {
            typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
            const bf16x4 o = {(__bf16)r0, (__bf16)r1, (__bf16)r2, (__bf16)r3};
            ((bf16x4*)y)[i] = o;
        } else {
            const nf32x4 o = {r0, r1, r2, r3};
            ((nf32x4*)y)[i] = o;
        }
    }
    const size_t tail = n4 * 4 + (size_t)blockIdx.x * 256 + threadIdx.x;
    if (tail < numel) {
        const float r = norm1(x[tail], hr_max);
        if (BF16_OUT) ((__bf16*)y)[tail] = (__bf16)r;
        else ((float*)y)[tail] = r;
    }
}

hipError_t launch_normalize(const float* x, void* y, size_t numel, float hr_max, int out_bf16, hipStream_t s)
{
    size_t blocks = (numel / 4 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 256 * 32) blocks = 256 * 32;            // grid-stride beyond 32 blocks per CU
    if (out_bf16) hipLaunchKernelGGL(normalize_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, s, x, y, numel, hr_max);
    else hipLaunchKernelGGL(normalize_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, s, x, y, numel, hr_max);
    return hipGetLastError();
}

// ---- bilinear resize, align_corners=False (ATen upsample_bilinear2d semantics) --------------
// src = max(0, (dst + 0.5) * in/out - 0.5); i0 = floor(src); i1 = i0 + (i0 < in-1); l = src - i0.
// A wave owns output rows (plane, oy): the row's source rows and vertical weights are wave-uniform (scalar
// registers, no per-pixel division); lanes walk ox, so stores are coalesced and the 2x2 gathers of neighbouring
// lanes fall in the same 128-byte lines of two source rows.
// A wave takes FOUR output rows at a time: the column terms of a lane's (up to) two pixels are formed once per kernel, the
// sixteen gathers per pixel pair are issued before the first result is stored (one row per trip left a wave with four
// loads in flight and a dependent trip per row: 0.47 of the HBM rate).
constexpr int BIL_ROWS = 4;
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       unsigned rows, int h, int w, int oh, int ow,
                                                       float sh, float sw)
{
    const int lane = threadIdx.x & 63;
    const unsigned wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int oxb = 0; oxb < ow; oxb += 128) {                 // (one trip for rows of up to 128 pixels)
        int xa[2], xb[2];
        float lx[2];
        bool on[2];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int ox = oxb + lane + 64 * k;
            on[k] = ox < ow;
            float fx = ((float)ox + 0.5f) * sw - 0.5f; fx = fx < 0.f ? 0.f : fx;
            const int x0 = min((int)fx, w - 1);
            xa[k] = x0;
            xb[k] = x0 + (x0 < w - 1 ? 1 : 0);
            lx[k] = fx - (float)(int)fx;
        }
        for (unsigned r = wave * BIL_ROWS; r < rows; r += nwaves * BIL_ROWS) {     // r = plane * oh + oy
            float v[BIL_ROWS][2][4], ly[BIL_ROWS];
#pragma unroll
            for (int j = 0; j < BIL_ROWS; ++j) {
                const unsigned rj = min(r + j, rows - 1);
                const unsigned pl = rj / (unsigned)oh;
                const int oy = (int)(rj - pl * (unsigned)oh);
                float fy = ((float)oy + 0.5f) * sh - 0.5f; fy = fy < 0.f ? 0.f : fy;
                const int y0 = (int)fy;
                const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
                ly[j] = fy - (float)y0;
                const float* r0 = x + ((size_t)pl * h + y0) * w;
                const float* r1 = x + ((size_t)pl * h + y1) * w;
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    v[j][k][0] = r0[xa[k]];
                    v[j][k][1] = r0[xb[k]];
                    v[j][k][2] = r1[xa[k]];
                    v[j][k][3] = r1[xb[k]];
                }
            }
#pragma unroll
            for (int j = 0; j < BIL_ROWS; ++j) {
                if (r + j >= rows) break;
                float* o = y + (size_t)(r + j) * ow + oxb + lane;
                const float hy = 1.f - ly[j];
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    const float hx = 1.f - lx[k];
                    if (on[k]) o[64 * k] = hy * (hx * v[j][k][0] + lx[k] * v[j][k][1]) + ly[j] * (hx * v[j][k][2] + lx[k] * v[j][k][3]);
                }
            }
        }
    }
}

hipError_t launch_bilinear(const float* x, float* y, int planes, int h, int w, int oh, int ow, hipStream_t s)
{
    const size_t rows = (size_t)planes * oh;
    if (rows == 0 || ow <= 0) return hipSuccess;
    if (rows > 0xffffffffull) return hipErrorInvalidValue;
    size_t grid = (rows + 4 * BIL_ROWS - 1) / (4 * BIL_ROWS);
    if (grid > 256u * 32u) grid = 256u * 32u;
    hipLaunchKernelGGL(bilinear_kernel, dim3((unsigned)grid), dim3(256), 0, s, x, y, (unsigned)rows, h, w, oh, ow,
                       (float)h / (float)oh, (float)w / (float)ow);
    return hipGetLastError();
}

// ---- PSNR / SSIM / L1 over a batch (utils.py:5-41, train.py:61-68) ---------------------------
// One workgroup per 32x32 output tile of one (n,c) plane: the 42x42 halo tiles of sr and hr go
// to LDS once (zero padded, like F.conv2d(padding=5)), a horizontal 11-tap pass builds the five
// filtered rows in LDS, a vertical pass finishes them; per-workgroup partial sums go to
// `scratch`, and a single-workgroup second kernel adds them in fixed order in fp64.
constexpr int MT = 32, MH = MT + 10;

// NOUT == 4 adds the GAN/pre-train pixel loss weighted_l1 (train.py:18-20): |sr - hr| weighted 10 where
// hr > 0.01 and 1 elsewhere, as a fourth partial sum.
template <int NOUT>
__global__ __launch_bounds__(256) void metrics_tile_kernel(const float* __restrict__ sr,
                                                           const float* __restrict__ hr, int h, int w,
                                                           int tiles_x, int tiles_y, float* scratch)
{
    __shared__ float a[MH][MH + 1], b[MH][MH + 1];
    __shared__ float hq[5][MH][MT + 1];
    __shared__ float red[4][4];
    float g[11];
    {   // utils.py:12-16: fp32 gaussian, sigma 1.5, normalised
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < 11; ++i) { const float d = (float)(i - 5); g[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f)); s += g[i]; }
#pragma unroll
        for (int i = 0; i < 11; ++i) g[i] /= s;
    }
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const size_t plane = bid / tiles_y;
    const float* ps = sr + plane * h * w;
    const float* ph = hr + plane * h * w;
    const int x0 = tx * MT - 5, y0 = ty * MT - 5;
    for (int i = threadIdx.x; i < MH * MH; i += 256) {
        const int r = i / MH, c = i - r * MH;
        const int gy = y0 + r, gx = x0 + c;
        const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
        a[r][c] = ok ? ps[(size_t)gy * w + gx] : 0.f;
        b[r][c] = ok ? ph[(size_t)gy * w + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < MH * MT; i += 256) {
        const int r = i / MT, c = i - r * MT;
        float mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float va = a[r][c + k], vb = b[r][c + k], gk = g[k];
            mx += gk * va; my += gk * vb; xx += gk * va * va; yy += gk * vb * vb; xy += gk * va * vb;
        }
        hq[0][r][c] = mx; hq[1][r][c] = my; hq[2][r][c] = xx; hq[3][r][c] = yy; hq[4][r][c] = xy;
    }
    __syncthreads();
    float se = 0.f, ae = 0.f, ss = 0.f, wl = 0.f;
    for (int i = threadIdx.x; i < MT * MT; i += 256) {
        const int r = i / MT, c = i - r * MT;
        const int gy = ty * MT + r, gx = tx * MT + c;
        if (gy < h && gx < w) {
            float mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float gk = g[k];
                mx += gk * hq[0][r + k][c]; my += gk * hq[1][r + k][c]; xx += gk * hq[2][r + k][c];
                yy += gk * hq[3][r + k][c]; xy += gk * hq[4][r + k][c];
            }
            const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
            const float sxx = xx - mx * mx, syy = yy - my * my, sxy = xy - mx * my;
            const float num = (2.f * mx * my + c1) * (2.f * sxy + c2);
            float den = (mx * mx + my * my + c1) * (sxx + syy + c2);
            den = den < 1e-8f ? 1e-8f : den;
            ss += num / den;
            const float hv = b[r + 5][c + 5];
            const float d = a[r + 5][c + 5] - hv;
            se += d * d; ae += fabsf(d);
            if (NOUT == 4) wl += ((hv > 0.01f ? 1.f : 0.f) * 9.f + 1.f) * fabsf(d);     // train.py:19-20
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        se += __shfl_xor(se, off); ae += __shfl_xor(ae, off); ss += __shfl_xor(ss, off);
        if (NOUT == 4) wl += __shfl_xor(wl, off);
    }
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = se; red[1][wv] = ae; red[2][wv] = ss; red[3][wv] = wl; }
    __syncthreads();
    if (threadIdx.x < NOUT)
        scratch[(size_t)blockIdx.x * NOUT + threadIdx.x] =
            (red[threadIdx.x][0] + red[threadIdx.x][1]) + (red[threadIdx.x][2] + red[threadIdx.x][3]);
}

template <int NOUT>
__global__ __launch_bounds__(256) void metrics_final_kernel(const float* scratch, int nblocks,
                                                            double total, float* out3)
{
    __shared__ double red[NOUT][256];
    double s[NOUT];
    for (int k = 0; k < NOUT; ++k) s[k] = 0.0;
    for (int i = threadIdx.x; i < nblocks; i += 256)
        for (int k = 0; k < NOUT; ++k) s[k] += (double)scratch[(size_t)i * NOUT + k];
    for (int k = 0; k < NOUT; ++k) red[k][threadIdx.x] = s[k];
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st)
            for (int k = 0; k < NOUT; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const double mse = red[0][0] / total;
        out3[0] = (mse == 0.0) ? 100.f : (float)(10.0 * log10(1.0 / mse));   // utils.py:7-9
        out3[1] = (float)(red[2][0] / total);
        out3[2] = (float)(red[1][0] / total);
        if (NOUT == 4) out3[3] = (float)(red[NOUT - 1][0] / total);
    }
}

static inline int metric_blocks(int n, int c, int h, int w)
{
    return n * c * ((h + MT - 1) / MT) * ((w + MT - 1) / MT);
}

size_t metrics_scratch_bytes(int n, int c, int h, int w, int nout)
{
    return (size_t)metric_blocks(n, c, h, w) * nout * sizeof(float);
}

template <int NOUT>
static hipError_t launch_metrics_n(const float* sr, const float* hr, int n, int c, int h, int w, float* out,
                                   void* scratch, hipStream_t s)
{
    const int tiles_x = (w + MT - 1) / MT, tiles_y = (h + MT - 1) / MT;
    const int nb = metric_blocks(n, c, h, w);
    hipLaunchKernelGGL(metrics_tile_kernel<NOUT>, dim3((unsigned)nb), dim3(256), 0, s, sr, hr, h, w, tiles_x,
                       tiles_y, (float*)scratch);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(metrics_final_kernel<NOUT>, dim3(1), dim3(256), 0, s, (const float*)scratch, nb,
                       (double)n * c * h * w, out);
    return hipGetLastError();
}

hipError_t launch_metrics(const float* sr, const float* hr, int n, int c, int h, int w, float* out,
                          void* scratch, int nout, hipStream_t s)
{
    return nout == 4 ? launch_metrics_n<4>(sr, hr, n, c, h, w, out, scratch, s)
                     : launch_metrics_n<3>(sr, hr, n, c, h, w, out, scratch, s);
}

// ---- srgen_metrics_noalloc: the same two kernels with the per-tile partial sums in a file-scope device array ----
// For callers that cannot allocate (HIP graph capture, tight serving loops).  `metrics_static_partials` is a
// host-side stand-in here, NOT a device address: the per-device address comes from hipGetSymbolAddress, once per
// device with that device current (every GPU holds its own copy at its own address), and only that pointer is ever
// handed to a kernel.  Every tile kernel writes its partial before the final kernel reads it, so nothing in the
// array has to start at zero; every call on one GPU shares the array, so calls that overlap on one GPU race.
static __device__ __align__(256) float metrics_static_partials[(size_t)kMetricsStaticMaxTiles * 3];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_static_mu;
float* g_static_ptr[kMaxDevices] = {};
}  // namespace

int metrics_tiles(int n, int c, int h, int w) { return metric_blocks(n, c, h, w); }

hipError_t metrics_static_scratch(float** ptr)
{
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= kMaxDevices) return hipErrorInvalidDevice;
    std::lock_guard<std::mutex> lock(g_static_mu);
    if (!g_static_ptr[dev]) {
        void* p = nullptr;
        e = hipGetSymbolAddress(&p, HIP_SYMBOL(metrics_static_partials));
        if (e != hipSuccess) return e;
        if (!p) return hipErrorInvalidSymbol;
        g_static_ptr[dev] = static_cast<float*>(p);
    }
    *ptr = g_static_ptr[dev];
    return hipSuccess;
}

// ---- train-mode BatchNorm (nn.BatchNorm2d in .train(), Generator.py:35,38,63) over NHWC (P, 96) fp32 ------
// Three HBM-bound passes around the (unfolded) conv: bn_stats (per-channel sum / sum of squares: fp32 inside a
// thread's ~100-pixel run, fp64 across threads and workgroups), bn_final (one workgroup: mean, biased variance,
// the fused scale / shift of the normalisation, and the batch statistics for the caller's running-statistics
// update), bn_apply (normalise in place + PReLU, or residual + factor * normalised).
constexpr int BN_THREADS = 240;                // 10 pixel rows x 24 float4 channel groups

template <typename T>
__global__ __launch_bounds__(256) void bn_stats_kernel(const T* __restrict__ t, size_t npix, double* partial)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V, ROWS = BN_THREADS / CPP;     // 24 x 10 (fp32) or 12 x 20 (bf16)
    __shared__ float red[ROWS][2][96];          // (a thread's sums are fp32; they are added in fp64 below)
    const int tid = threadIdx.x;
    if (tid < BN_THREADS) {
        const int cg = tid % CPP, row = tid / CPP;
        const size_t per = (npix + gridDim.x - 1) / gridDim.x;
        const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
        float sx[V], sq[V];
#pragma unroll
        for (int e = 0; e < V; ++e) sx[e] = sq[e] = 0.f;
        size_t p = p0 + row;
        for (; p + 3 * ROWS < p1; p += 4 * ROWS) {              // four loads in flight per thread
            float v[4][V];
#pragma unroll
            for (int u = 0; u < 4; ++u) ldv(t + (p + ROWS * u) * 96 + cg * V, v[u]);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    sx[e] += v[u][e];
                    sq[e] += v[u][e] * v[u][e];
                }
        }
        for (; p < p1; p += ROWS) {
            float v[V];
            ldv(t + p * 96 + cg * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                sx[e] += v[e];
                sq[e] += v[e] * v[e];
            }
        }
        for (int e = 0; e < V; ++e) {
            red[row][0][cg * V + e] = sx[e];
            red[row][1][cg * V + e] = sq[e];
        }
    }
    __syncthreads();
    if (tid < 192) {
        const int k = tid / 96, c = tid % 96;
        double a = 0.0;
        for (int r = 0; r < ROWS; ++r) a += (double)red[r][k][c];
        partial[(size_t)blockIdx.x * 192 + tid] = a;
    }
}

// grid 24: block b owns channels [4b, 4b + 4): 8 sums x 96 threads, each adding every 96th partial (fixed order).  (Eight
// blocks of 32 adders each walked 32 partials one L2 round trip after the other: 8.4 us per BatchNorm for a sum of 1.5 MB.)
constexpr int BNF_CPB = 4, BNF_BLOCKS = 96 / BNF_CPB, BNF_ADDERS = 768 / (2 * BNF_CPB);
__global__ __launch_bounds__(768) void bn_final_kernel(const double* partial, int nb, double npix,
                                                       const float* gamma, const float* beta, float eps,
                                                       float* scale_shift, float* batch_stats)
{
    __shared__ double part[BNF_ADDERS][2 * BNF_CPB];
    __shared__ double tot[2 * BNF_CPB];
    {
        const int s = threadIdx.x % (2 * BNF_CPB), q = threadIdx.x / (2 * BNF_CPB);
        const int col = (s / BNF_CPB) * 96 + blockIdx.x * BNF_CPB + s % BNF_CPB;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;           // four loads in flight, fixed order
        int b = q;
        for (; b + 3 * BNF_ADDERS < nb; b += 4 * BNF_ADDERS) {
            a0 += partial[(size_t)b * 192 + col];
            a1 += partial[(size_t)(b + BNF_ADDERS) * 192 + col];
            a2 += partial[(size_t)(b + 2 * BNF_ADDERS) * 192 + col];
            a3 += partial[(size_t)(b + 3 * BNF_ADDERS) * 192 + col];
        }
        for (; b < nb; b += BNF_ADDERS) a0 += partial[(size_t)b * 192 + col];
        part[q][s] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (threadIdx.x < 2 * BNF_CPB) {
        double a = 0.0;
        for (int q = 0; q < BNF_ADDERS; ++q) a += part[q][threadIdx.x];
        tot[threadIdx.x] = a;
    }
    __syncthreads();
    if (threadIdx.x < BNF_CPB) {
        const int c = blockIdx.x * BNF_CPB + threadIdx.x;
        const double mean = tot[threadIdx.x] / npix;
        double var = tot[BNF_CPB + threadIdx.x] / npix - mean * mean;   // biased (what the normalisation uses)
        var = var > 0.0 ? var : 0.0;
        const double sc = (double)gamma[c] / sqrt(var + (double)eps);
        scale_shift[c] = (float)sc;
        scale_shift[96 + c] = (float)((double)beta[c] - mean * sc);
        batch_stats[c] = (float)mean;
        batch_stats[96 + c] = (float)var;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* t, T* out, const float* __restrict__ scale_shift,
                                                       const float* __restrict__ slope, const T* __restrict__ res,
                                                       float res_factor, size_t nvec)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CPP) * V;
        float v[V], sc[V], sh[V];
        ldv(t + i * V, v);
        ldf<V>(scale_shift + c, sc);
        ldf<V>(scale_shift + 96 + c, sh);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
        if (slope) {                                             // PReLU(ch), Generator.py:36
            float sl[V];
            ldf<V>(slope + c, sl);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * sl[e];
        }
        if (res) {                                               // x + 0.1 * block(x) (:42) / x1 + mid(x2) (:78)
            float r[V];
            ldv(res + i * V, r);
#pragma unroll
            for (int e = 0; e < V; ++e) v[e] = fmaf(res_factor, v[e], r[e]);
        }
        stv(out + i * V, v);
    }
}

// per-image channel sums in the pool_partial layout of the conv epilogue: [N][parts][96], part = hw/parts pixels
template <typename T>
__global__ __launch_bounds__(256) void pool_partial_kernel(const T* __restrict__ t, int hw, int parts,
                                                           float* pool_partial)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V, ROWS = BN_THREADS / CPP;
    __shared__ float red[ROWS][96];
    const int tid = threadIdx.x;
    const int per = hw / parts;
    const size_t p0 = (size_t)blockIdx.x * per;                  // blockIdx.x = image * parts + part
    if (tid < BN_THREADS) {
        const int cg = tid % CPP, row = tid / CPP;
        float s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = 0.f;
        for (int p = row; p < per; p += ROWS) {
            float v[V];
            ldv(t + (p0 + p) * 96 + cg * V, v);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += v[e];
        }
        for (int e = 0; e < V; ++e) red[row][cg * V + e] = s[e];
    }
    __syncthreads();
    if (tid < 96) {
        float a = 0.f;
        for (int r = 0; r < ROWS; ++r) a += red[r][tid];
        pool_partial[(size_t)blockIdx.x * 96 + tid] = a;
    }
}

size_t bn_scratch_bytes() { return (size_t)kBnBlocks * 192 * sizeof(double) + 2 * 96 * sizeof(float); }

template <typename T>
static hipError_t bn_train_t(T* t, size_t npix, const float* gamma, const float* beta, const float* slope, const T* res,
                             float res_factor, void* scratch, float* batch_stats, hipStream_t s, T* out)
{
    if (!out) out = t;                                   // in place (the forward-values path)
    double* partial = static_cast<double*>(scratch);
    float* ss = reinterpret_cast<float*>(partial + (size_t)kBnBlocks * 192);
    constexpr int ROWS = BN_THREADS / (96 / StoreVec<T>::N);       // pixel rows one workgroup reads at a time
    const int nb = npix < (size_t)kBnBlocks * ROWS ? (int)((npix + ROWS - 1) / ROWS) : kBnBlocks;
    hipLaunchKernelGGL(bn_stats_kernel<T>, dim3(nb), dim3(256), 0, s, t, npix, partial);
    hipLaunchKernelGGL(bn_final_kernel, dim3(BNF_BLOCKS), dim3(768), 0, s, partial, nb, (double)npix, gamma, beta, 1e-5f, ss,
                       batch_stats);
    const size_t nvec = npix * (96 / StoreVec<T>::N);
    const size_t want = (nvec + 255) / 256;
    hipLaunchKernelGGL(bn_apply_kernel<T>, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, s, t, out, ss, slope,
                       res, res_factor, nvec);
    return hipGetLastError();
}

hipError_t launch_bn_train(float* t, size_t npix, const float* gamma, const float* beta, const float* slope,
                           const float* res, float res_factor, void* scratch, float* batch_stats, hipStream_t s, float* out)
{
    return bn_train_t<float>(t, npix, gamma, beta, slope, res, res_factor, scratch, batch_stats, s, out);
}

hipError_t launch_bn_train_bf16(void* t, size_t npix, const float* gamma, const float* beta, const float* slope,
                                const void* res, float res_factor, void* scratch, float* batch_stats, hipStream_t s, void* out)
{
    return bn_train_t<bf16_t>((bf16_t*)t, npix, gamma, beta, slope, (const bf16_t*)res, res_factor, scratch, batch_stats, s,
                              (bf16_t*)out);
}

hipError_t launch_pool_partial(const float* t, int n, int hw, int parts, float* pool_partial, hipStream_t s)
{
    hipLaunchKernelGGL(pool_partial_kernel<float>, dim3((unsigned)(n * parts)), dim3(256), 0, s, t, hw, parts, pool_partial);
    return hipGetLastError();
}

hipError_t launch_pool_partial_bf16(const void* t, int n, int hw, int parts, float* pool_partial, hipStream_t s)
{
    hipLaunchKernelGGL(pool_partial_kernel<bf16_t>, dim3((unsigned)(n * parts)), dim3(256), 0, s, (const bf16_t*)t, hw, parts,
                       pool_partial);
    return hipGetLastError();
}

}  // namespace srgen
