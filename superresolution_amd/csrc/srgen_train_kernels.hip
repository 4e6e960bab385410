// srgen_train_kernels.hip -- backward-pass operators of the pre-train step (SURVEY.md 8(f) row 2;
// /root/reference/train.py:90-97: loss.backward(), optimizer.step() with AdamW, train.py:270), fp32,
// NHWC activations, for the residual trunk of the generator (Generator.py:30-42):
//   * device-side Winograd weight images (forward and data-gradient form) -- the weights change every step,
//     so U = G g Gt is formed on the device, not on the host as srgen_commit_weights does;
//   * conv3x3 weight gradient on v_mfma_f32_16x16x4_f32 (split over pixel slices, fixed-order reduction);
//   * train-mode BatchNorm backward with the PReLU backward fused in front of it;
//   * the gradient of weighted_l1 + w * (1 - ssim) with respect to the generator output (train.py:18-20,93;
//     utils.py:21-41);
//   * AdamW (torch.optim.AdamW semantics) over a flat parameter buffer.
// The data gradient of a 3x3 conv is the forward Winograd kernel (srgen_conv3x3_f32_wino.hip) run on the
// transposed, flipped weight image made here.
#include "srgen_internal.h"

#include <cstdint>
#include <cstring>

namespace srgen {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// ---- Winograd weight image on the device ---------------------------------------------------------------
// Layout of srgen_conv3x3_f32_wino.hip (srgen_api.cpp pack3x3_wino_f32): float4 units
// [cb][nu 4][ci chunk 6][xi 4][ci/4 4][co 96], 4 consecutive ci per unit.
// mode 0 (forward): block b, packed channel co <- original output channel (shuffle ? 4*co + b : 96*b + co).
// mode 1 (data gradient of block `block`): the conv that maps dY (channels = original output channels of the
// block) to dX (channels = original input channels): g'[out = ci0][in = co0][r][c] = w[oc(co0)][ci0][2-r][2-c].
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, float* __restrict__ out, int cb,
                                                        int shuffle, int mode, int block)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cb * 96 * 96) return;
    const int ci = idx % 96, co = (idx / 96) % 96, b = idx / (96 * 96);
    double g[3][3];
    if (mode == 0) {
        const int oc = shuffle ? 4 * co + b : 96 * b + co;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) g[r][c] = (double)w[((size_t)oc * 96 + ci) * 9 + r * 3 + c];
    } else {
        const int oc = shuffle ? 4 * ci + block : 96 * block + ci;       // original output channel = this conv's input
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) g[r][c] = (double)w[((size_t)oc * 96 + co) * 9 + (2 - r) * 3 + (2 - c)];
    }
    const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
    for (int xi = 0; xi < 4; ++xi)
        for (int nu = 0; nu < 4; ++nu) {
            double u = 0;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) u += G[xi][r] * g[r][c] * G[nu][c];
            out[((((((size_t)b * 4 + nu) * 6 + ci / 16) * 4 + xi) * 4 + (ci / 4) % 4) * 96 + co) * 4 + (ci & 3)] = (float)u;
        }
}

hipError_t launch_wino_pack(const float* w, float* out, int cout_total, int mode, int block, hipStream_t s)
{
    const int shuffle = cout_total == 384;
    const int cb = mode == 0 ? cout_total / 96 : 1;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((cb * 96 * 96 + 255) / 256)), dim3(256), 0, s, w, out, cb, shuffle,
                       mode, block);
    return hipGetLastError();
}

// ---- split-f16 weight image on the device (layout of srgen_conv3x3_f32s.hip / srgen_api.cpp pack3x3_split_f16) ----
// 8-half units [cb][chunk 6][ky 3][kx 3][nt 3][hi | lo][lane 64]; lane l of a unit: co = 32 nt + (l & 31),
// ci = 16 chunk + 8 (l >> 5) + e; then a tail of 32 partial maxima of |w| over the whole weight, by whose exponent the
// weights are scaled before the split (srgen_internal.h: split_exp).  Modes as wino_pack_kernel.

// max |x| of a tensor as kAmaxSlots per-workgroup maxima (the split conv kernel reduces them itself: no atomics, nothing
// to zero, one launch).  NaNs are ignored (fmaxf).
__device__ __forceinline__ float block_max_256(float m)
{
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    return fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
}

__global__ __launch_bounds__(256) void absmax_kernel(const float* __restrict__ x, size_t n4, size_t n, float* out)
{
    float m = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = ((const f32x4*)x)[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) m = fmaxf(m, fabsf(x[i]));
    m = block_max_256(m);
    if (threadIdx.x == 0) out[blockIdx.x] = m;
}

// per image: one workgroup of 1024 threads each; out[i] = max |x[i]|, zero_out[i] = 0
__global__ __launch_bounds__(1024) void absmax_per_image_kernel(const float* __restrict__ x, size_t per_image, float* out,
                                                                float* zero_out)
{
    const f32x4* src = (const f32x4*)(x + (size_t)blockIdx.x * per_image);
    float m = 0.f;
    for (size_t i = threadIdx.x; i < per_image / 4; i += 1024) {
        const f32x4 v = src[i];
        m = fmaxf(fmaxf(m, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
#pragma unroll
    for (int o = 32; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    __shared__ float part[16];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int i = 1; i < 16; ++i) m = fmaxf(m, part[i]);
        out[blockIdx.x] = m;
        zero_out[blockIdx.x] = 0.f;
    }
}

hipError_t launch_absmax_per_image(const float* x, int n, size_t per_image, float* out, float* zero_out, hipStream_t s)
{
    if (per_image % 4 || ((uintptr_t)x & 15)) return hipErrorInvalidValue;
    hipLaunchKernelGGL(absmax_per_image_kernel, dim3((unsigned)n), dim3(1024), 0, s, x, per_image, out, zero_out);
    return hipGetLastError();
}

hipError_t launch_absmax(const float* x, size_t n, float* out, hipStream_t s)
{
    const size_t n4 = ((uintptr_t)x & 15) ? 0 : n / 4;
    hipLaunchKernelGGL(absmax_kernel, dim3(kAmaxSlots), dim3(256), 0, s, x, n4, n, out);
    return hipGetLastError();
}

// 32 partial maxima of |w| into the image's tail (its own small launch: 32 workgroups), then the pack
__device__ __forceinline__ void weight_absmax_body(const float* __restrict__ w, int n, float* tail)
{
    float m = 0.f;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += 32 * 256) m = fmaxf(m, fabsf(w[i]));   // (4-byte aligned views)
    m = block_max_256(m);
    if (threadIdx.x == 0) tail[blockIdx.x] = m;
}

__global__ __launch_bounds__(256) void weight_absmax_kernel(const float* __restrict__ w, int n, float* tail)
{
    weight_absmax_body(w, n, tail);
}

__device__ __forceinline__ void split_pack_body(const float* __restrict__ w, _Float16* __restrict__ out, int cb, int shuffle,
                                                int mode, int block, const float* tail)
{
    const float sc = pow2f(split_tail_exp(tail, threadIdx.x & 63));
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= cb * 96 * 96) return;
    const int ci = idx % 96, co = (idx / 96) % 96, b = idx / (96 * 96);
    const int lane = (co & 31) + 32 * ((ci >> 3) & 1), e = ci & 7;
    for (int ky = 0; ky < 3; ++ky)
        for (int kx = 0; kx < 3; ++kx) {
            float v;
            if (mode == 0) {
                const int oc = shuffle ? 4 * co + b : 96 * b + co;
                v = w[((size_t)oc * 96 + ci) * 9 + ky * 3 + kx];
            } else {
                const int oc = shuffle ? 4 * ci + block : 96 * block + ci;
                v = w[((size_t)oc * 96 + co) * 9 + (2 - ky) * 3 + (2 - kx)];
            }
            v *= sc;
            const _Float16 hi = (_Float16)v;
            const _Float16 lo = (_Float16)(v - (float)hi);
            const size_t unit = ((((((size_t)b * 6 + ci / 16) * 3 + ky) * 3 + kx) * 3 + co / 32) * 2) * 64 + lane;
            out[unit * 8 + e] = hi;
            out[(unit + 64) * 8 + e] = lo;
        }
}

__global__ __launch_bounds__(256) void split_pack_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int cb,
                                                         int shuffle, int mode, int block, const float* tail)
{
    split_pack_body(w, out, cb, shuffle, mode, block, tail);
}

// ---- many images in two launches (a training step packs 73 images of unchanged layout: 146 launches of ~5 us) ----------
__global__ __launch_bounds__(256) void weight_absmax_many_kernel(SplitPackJobs jobs)
{
    const SplitPackJob j = jobs.j[blockIdx.y];
    const int cb = j.mode == 0 ? j.cout_total / 96 : 1;
    weight_absmax_body(j.w, j.cout_total * 864, (float*)((char*)j.out + (size_t)cb * kSplitImageBytes));
}

__global__ __launch_bounds__(256) void split_pack_many_kernel(SplitPackJobs jobs)
{
    const SplitPackJob j = jobs.j[blockIdx.y];
    const int cb = j.mode == 0 ? j.cout_total / 96 : 1;
    if ((int)blockIdx.x * 256 >= cb * 96 * 96) return;
    split_pack_body(j.w, (_Float16*)j.out, cb, j.cout_total == 384, j.mode, j.block,
                    (const float*)((const char*)j.out + (size_t)cb * kSplitImageBytes));
}

hipError_t launch_split_pack_many(const SplitPackJobs& jobs, hipStream_t s)
{
    if (jobs.count <= 0 || jobs.count > kSplitPackMaxJobs) return hipErrorInvalidValue;
    int max_cb = 1;
    for (int i = 0; i < jobs.count; ++i)
        if (jobs.j[i].mode == 0 && jobs.j[i].cout_total / 96 > max_cb) max_cb = jobs.j[i].cout_total / 96;
    hipLaunchKernelGGL(weight_absmax_many_kernel, dim3(32, (unsigned)jobs.count), dim3(256), 0, s, jobs);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(split_pack_many_kernel, dim3((unsigned)(max_cb * 36), (unsigned)jobs.count), dim3(256), 0, s, jobs);
    return hipGetLastError();
}

hipError_t launch_split_pack(const float* w, void* out, int cout_total, int mode, int block, hipStream_t s)
{
    const int shuffle = cout_total == 384;
    const int cb = mode == 0 ? cout_total / 96 : 1;
    float* tail = (float*)((char*)out + (size_t)cb * kSplitImageBytes);
    hipLaunchKernelGGL(weight_absmax_kernel, dim3(32), dim3(256), 0, s, w, cout_total * 864, tail);   // whole weight: a bound
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(split_pack_kernel, dim3((unsigned)((cb * 96 * 96 + 255) / 256)), dim3(256), 0, s, w, (_Float16*)out, cb,
                       shuffle, mode, block, tail);
    return hipGetLastError();
}

// ---- bf16 weight images of the 16-bit step (layout of srgen_conv3x3_bf16.hip: 8-element units [cb][tap][ci/8][co 96]),
// from the live fp32 OIHW weights, many images in one launch; modes as split_pack_body ----
__global__ __launch_bounds__(256) void bf16_pack_many_kernel(SplitPackJobs jobs)
{
    const SplitPackJob j = jobs.j[blockIdx.y];
    const int cb = j.mode == 0 ? j.cout_total / 96 : 1;
    const int shuffle = j.cout_total == 384;
    const int idx = blockIdx.x * 256 + threadIdx.x;               // over (b, ci / 8, co): one 16-byte unit per tap
    if (idx >= cb * 12 * 96) return;
    const int co = idx % 96, c8 = (idx / 96) % 12, b = idx / (96 * 12);
    const float* __restrict__ w = j.w;
    for (int tap = 0; tap < 9; ++tap) {
        unsigned u[4];
#pragma unroll
        for (int e2 = 0; e2 < 4; ++e2) {
            float v[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int ci = 8 * c8 + 2 * e2 + k;
                if (j.mode == 0) {
                    const int oc = shuffle ? 4 * co + b : 96 * b + co;
                    v[k] = w[((size_t)oc * 96 + ci) * 9 + tap];
                } else {
                    const int oc = shuffle ? 4 * ci + j.block : 96 * j.block + ci;
                    v[k] = w[((size_t)oc * 96 + co) * 9 + (8 - tap)];
                }
            }
            typedef __bf16 b2 __attribute__((ext_vector_type(2)));
            const b2 t = {(__bf16)v[0], (__bf16)v[1]};
            u[e2] = __builtin_bit_cast(unsigned, t);
        }
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        *(u4*)((char*)j.out + ((((size_t)b * 9 + tap) * 12 + c8) * 96 + co) * 16) = u4{u[0], u[1], u[2], u[3]};
    }
}

hipError_t launch_bf16_pack_many(const SplitPackJobs& jobs, hipStream_t s)
{
    if (jobs.count <= 0 || jobs.count > kSplitPackMaxJobs) return hipErrorInvalidValue;
    int max_cb = 1;
    for (int i = 0; i < jobs.count; ++i)
        if (jobs.j[i].mode == 0 && jobs.j[i].cout_total / 96 > max_cb) max_cb = jobs.j[i].cout_total / 96;
    hipLaunchKernelGGL(bf16_pack_many_kernel, dim3((unsigned)((max_cb * 12 * 96 + 255) / 256), (unsigned)jobs.count), dim3(256), 0,
                       s, jobs);
    return hipGetLastError();
}

// ---- conv3x3 weight gradient ------------------------------------------------------------------------------
// dW[co][ci][ky][kx] = sum over pixels dY[n][y][x][co] * X[n][y+ky-1][x+kx-1][ci] (zero outside the image).
// A GEMM with M = co, N = ci, K = pixels per tap.  Workgroup = (tap, slice of image rows), 3 waves, wave w owning
// co [32w, 32w+32) x all 96 ci as 2 x 6 tiles of v_mfma_f32_16x16x4_f32.  Operands come straight from the NHWC
// tensors with one wide load per lane: A = dY^T, lane (i, kq) loads dY[pixel kq][32w + 2i .. +1] (8 bytes; tile a
// has row i = channel 32w + 2i + a), B = X, lane (kq, j) loads X[shifted pixel kq][4j .. 4j+3] and [64 + 2j .. +1]
// (tiles 0-3: column j = channel 4j + b, tiles 4-5: 64 + 2j + b - 4).  The channel permutation is undone when the
// partial sums are stored to the slab [slice][tap][co][ci]; a second kernel adds the slices in fixed order
// (bitwise reproducible) and transposes to OIHW.  The nine taps of a slice are consecutive workgroups of ONE XCD
// (blockIdx % 8), so X and dY come from HBM once and from that XCD's L2 eight more times.  The centre-tap
// workgroups also produce the bias gradient sum_pixels dY.
__global__ __launch_bounds__(192) void conv3x3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            int total_rows, int H, int W, int rows_per_slice, int slices,
                                                            float* __restrict__ slab, float* __restrict__ bias_slab)
{
    int tap, slice;
    if ((slices & 7) == 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tap = j % 9;
        slice = (j / 9) * 8 + xcd;
    } else {
        tap = blockIdx.x % 9;
        slice = blockIdx.x / 9;
    }
    const int ky = tap / 3, kx = tap % 3;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int i = lane & 15, kq = lane >> 4;
    f32x4 acc[2][6];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 6; ++b) acc[a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[2] = {0.f, 0.f};
    const int r0 = slice * rows_per_slice;
    const int r1 = min(r0 + rows_per_slice, total_rows);
    for (int row = r0; row < r1; ++row) {
        const int yy = row % H;                       // row = n * H + y
        const int ys = yy + ky - 1;
        if ((unsigned)ys >= (unsigned)H) continue;    // the shifted row lies outside the image: contributes zero
        const float* dyr = dy + (size_t)row * W * 96 + 32 * wv + 2 * i;
        const float* xr = x + ((size_t)(row + ky - 1) * W + (kx - 1)) * 96;
        for (int x0 = 0; x0 < W; x0 += 4) {
            const int px = x0 + kq;
            const int xs = px + kx - 1;
            const bool ok = (unsigned)xs < (unsigned)W;
            const f32x2 av = *(const f32x2*)(dyr + (size_t)px * 96);
            f32x4 b4 = f32x4{0.f, 0.f, 0.f, 0.f};
            f32x2 b2 = f32x2{0.f, 0.f};
            if (ok) {
                b4 = *(const f32x4*)(xr + (size_t)px * 96 + 4 * i);
                b2 = *(const f32x2*)(xr + (size_t)px * 96 + 64 + 2 * i);
            }
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                bsum[a] += av[a];
#pragma unroll
                for (int b = 0; b < 4; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], b4[b], acc[a][b], 0, 0, 0);
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[a][4 + b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], b2[b], acc[a][4 + b], 0, 0, 0);
            }
        }
    }
    // D[row = 4*(lane>>4) + r][col = lane & 15]; row -> co = 32w + 2 row + a, col -> ci as above
    float* o = slab + ((size_t)slice * 9 + tap) * 96 * 96;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            float* orow = o + (size_t)(32 * wv + 2 * (4 * kq + r) + a) * 96;
            *(f32x4*)(orow + 4 * i) = f32x4{acc[a][0][r], acc[a][1][r], acc[a][2][r], acc[a][3][r]};
            *(f32x2*)(orow + 64 + 2 * i) = f32x2{acc[a][4][r], acc[a][5][r]};
        }
    if (tap == 4) {                                   // every row is visited by the centre tap (ys = yy)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) bias_slab[(size_t)slice * 96 + 32 * wv + 2 * i + a] = v;
        }
    }
}

// dbias[c] = sum over slices of bias_slab[slice][c], by one block of 1024 threads: ten groups of 96 threads take every
// tenth slice (a single thread walking all slices waited one memory round trip per slice: 20 us of a 23 us kernel),
// `red` (>= 960 floats of LDS, free after a barrier) joins them in fixed order
__device__ __forceinline__ void bias_slab_sum(const float* __restrict__ bias_slab, int slices, float* __restrict__ dbias,
                                              float* red)
{
    const int t = threadIdx.x, c = t % 96, q = t / 96;
    __syncthreads();
    if (t < 960) {
        float s = 0.f;
        for (int sl = q; sl < slices; sl += 10) s += bias_slab[(size_t)sl * 96 + c];
        red[t] = s;
    }
    __syncthreads();
    if (t < 96) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 10; ++k) s += red[k * 96 + t];
        dbias[t] = s;
    }
}

// 64 outputs x 16 slice groups per block; group g adds slices g, g+16, ... (up to four loads in flight per thread, all of a
// thread's loads within one or two round trips), LDS joins the groups in fixed order.  (Four groups: 23 us at 80 slices)
__global__ __launch_bounds__(1024) void conv3x3_wgrad_reduce_kernel(const float* __restrict__ slab,
                                                                    const float* __restrict__ bias_slab, int slices,
                                                                    float* __restrict__ dw, float* __restrict__ dbias)
{
    __shared__ float part[16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;                     // over tap * 9216 + co * 96 + ci  (9 * 9216 = 1296 * 64)
    const int tap = idx / 9216, rem = idx % 9216;
    {
        // up to 16 partials per thread (slices <= 256), ALL requested before the first is added: the loop of four loads in
        // flight walked 168 slices in five dependent L2 round trips (17 us per gradient for 56 MB that sit in the cache)
        float v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int sl = g + 16 * k;
            v[k] = sl < slices ? slab[((size_t)sl * 9 + tap) * 9216 + rem] : 0.f;
        }
        float s = 0.f;
        for (int sl = g + 256; sl < slices; sl += 16) s += slab[((size_t)sl * 9 + tap) * 9216 + rem];      // (never with today's slice counts)
#pragma unroll
        for (int k = 0; k < 16; k += 4) s += (v[k] + v[k + 1]) + (v[k + 2] + v[k + 3]);
        part[g][o] = s;
    }
    __syncthreads();
    if (g == 0) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][o];
        dw[(size_t)rem * 9 + tap] = s;                       // OIHW
    }
    if (dbias && blockIdx.x == 0) bias_slab_sum(bias_slab, slices, dbias, &part[0][0]);
}

// ---- conv3x3 weight gradient in the Winograd domain (the default) ---------------------------------------------------
// dW is a correlation of dY with X, so the minimal-filtering identity applies with the roles of F(2x2,3x3) exchanged --
// F(3x3, 2x2): per 2x2 tile of dY (g) and the 4x4 patch P of X around it (rows / columns 2t-1 .. 2t+2, zero outside
// the image: the forward kernel's input tile),
//     dW += A'^T [ (G' g G'^T) (.) (B^T P B) ] A',      B^T as in the forward,  G' = [[1,0],[.5,.5],[.5,-.5],[0,1]],
//     A'^T = [[1,1,1,0],[0,1,-1,0],[0,1,1,-1]],
// and the sum over tiles commutes with A': 16 GEMMs  M[xi][nu] (co x ci) = sum_tiles Yh[xi][nu][tile][co] * Xh[xi][nu][tile][ci]
// with K = tiles -- 16 multiplies per (tile, co, ci) instead of 36 -- and ONE 4x4 -> 3x3 output transform at the end.
// The two 1/2 of G' are left out in the kernel and applied to M in the final pass.
// Partial M -> slab [slice][xi*4+nu][co][ci] (+ 96 bias partials per slice); wino_wgrad_sum_kernel adds the slices in
// fixed order, wino_wgrad_final_kernel applies the scaling and A' and writes OIHW.  sum_tiles Yh[1][1] (unscaled) is the
// bias gradient.  (A first version without LDS -- every wave loading its own 8-byte pieces -- is kept as
// tools/conv3x3_wgrad_wino_reg.hip: 0.35 ms against 0.28, bound by vector-memory instructions.)
constexpr size_t kWgSliceFloats = 147456 + 128;               // per slice: 16 x 96 x 96 partial M | 96 bias partials | pad
// A workgroup of four waves = (slice, nu) stages what ONE k-step (4 tiles) needs -- the 5 or 8 patch columns this nu reads x 4 rows
// of X, the 4 or 8 pixels x 2 rows of dY, all 96 channels -- with 16-byte loads (2.6 / 4.5 per thread and k-step),
// zero-filled outside the image, into a double-buffered LDS tile (pixel pitch 100 floats); wave (ch, cih) owns
// 48 co x 48 ci x the four xi = 36 tiles (144 accumulators), reads its fragments with ds_read_b64 / b32
// (ci = 48 cih + 2i + b | 48 cih + 32 + i, co likewise) and issues 36 MFMAs per k-step.  Global loads run two k-steps
// ahead in registers (G0 / G1), the LDS write of step s+1 follows the MFMAs of step s, one barrier per k-step; two
// workgroups per CU.
#ifndef WGL_SPLIT
#define WGL_SPLIT 0
#endif
constexpr int WGL_PITCH = 100;                                // LDS floats per staged pixel
constexpr int WGL_BUF = 48 * WGL_PITCH;                       // floats per buffer: at most 32 X + 16 dY pixels

template <int NU>
__device__ __forceinline__ void wgrad_wino_lds_body(const float* __restrict__ x, const float* __restrict__ dy, int total_trows, int H,
                                                    int W, int rows_per_slice, float* __restrict__ slab, int slice, float* lds)
{
    constexpr bool EDGE = NU == 0 || NU == 3;
    constexpr int PX = EDGE ? 5 : 8, PY = EDGE ? 4 : 8;       // staged pixels per row of X / dY
    constexpr int NX = 4 * PX * 24, NE = NX + 2 * PY * 24;    // float4 units per stage
    constexpr int NL = (NE + 255) / 256;                      // per thread
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int ch = wv & 1, cih = wv >> 1, i = lane & 15, kq = lane >> 4;
    const int TH = H >> 1, TW = W >> 1, spr = TW >> 2;
    const int r0 = slice * rows_per_slice;
    const int r1 = min(r0 + rows_per_slice, total_trows);
    const int nst = (r1 > r0 ? r1 - r0 : 0) * spr;

    // what this thread stages: unit e = tid + 256 k, packed as row+1 | (col+1) << 2 | c4 << 7 | isx << 12 | on << 13
    int s_pk[NL], s_lds[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int e = tid + 256 * k;
        const bool on = e < NE, isx = e < NX;
        const int u = isx ? e : (on ? e - NX : 0);
        const int c4 = u % 24, slot = (u / 24) % (isx ? PX : PY), rr = u / (24 * (isx ? PX : PY));
        const int col = isx ? (NU == 0 ? 2 * slot : NU == 3 ? 2 * slot + 1 : slot + 1) - 1        // column - 2 tx0
                            : (NU == 0 ? 2 * slot : NU == 3 ? 2 * slot + 1 : slot);
        const int row = isx ? rr - 1 : rr;                                                      // row - 2 ty
        s_pk[k] = (row + 1) | (col + 1) << 2 | c4 << 7 | (int)isx << 12 | (int)on << 13;
        s_lds[k] = ((isx ? rr * PX : 4 * PX + rr * PY) + slot) * WGL_PITCH + 4 * c4;
    }
    f32x4 G0[NL], G1[NL];
    unsigned m0 = 0, m1 = 0;                                  // bit k: unit k of that buffer lies inside the image
    // float offset of unit k from pixel (2 ty - 1, 2 tx0 - 1) of its tensor: all an interior k-step needs per load
    unsigned s_off[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int pk = s_pk[k];
        s_off[k] = (unsigned)(((pk & 3) * W + ((pk >> 2) & 31)) * 96 + 4 * ((pk >> 7) & 31));
    }
    auto issue = [&](f32x4 (&G)[NL], unsigned& mask, int sg) {
        const int trow = r0 + sg / spr;
        const int n = trow / TH, ty = trow - n * TH, tx0 = (sg % spr) * 4;
        if (ty > 0 && ty < TH - 1 && tx0 > 0 && tx0 < TW - 4) {          // wave-uniform: nothing can leave the image
            const size_t uni = (((size_t)n * H + 2 * ty - 1) * W + 2 * tx0 - 1) * 96;
            const float* bx = x + uni;
            const float* by = dy + uni;
#pragma unroll
            for (int k = 0; k < NL; ++k) G[k] = *(const f32x4*)(((s_pk[k] >> 12) & 1 ? bx : by) + s_off[k]);
            mask = ~0u;
            return;
        }
        mask = 0;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int pk = s_pk[k];
            const int y = 2 * ty + (pk & 3) - 1, c = 2 * tx0 + ((pk >> 2) & 31) - 1;
            const bool v = (pk >> 13) && (unsigned)y < (unsigned)H && (unsigned)c < (unsigned)W;
            const float* base = (pk >> 12) & 1 ? x : dy;
            G[k] = *(const f32x4*)(base + (((size_t)n * H + (v ? y : 0)) * W + (v ? c : 0)) * 96 + 4 * ((pk >> 7) & 31));
            mask |= (unsigned)v << k;
        }
    };
    // zero fill outside the image happens HERE (the consumer side of the loads), not in issue()
    auto commit = [&](const f32x4 (&G)[NL], unsigned mask, float* buf) {
#pragma unroll
        for (int k = 0; k < NL; ++k)
            if (s_pk[k] >> 13) *(f32x4*)(buf + s_lds[k]) = G[k] * (((mask >> k) & 1u) ? 1.f : 0.f);
    };

    f32x4 acc[4][3][3];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 3; ++b) acc[xi][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[3] = {0.f, 0.f, 0.f};

    // fragment addresses of tile kq: X slots of the two columns, dY slots
    const int sa = EDGE ? kq : (NU == 1 ? 2 * kq : 2 * kq + 1), sb = EDGE ? kq + 1 : (NU == 1 ? 2 * kq + 1 : 2 * kq);
    const int xfrag = 48 * cih + 2 * i, xfrag2 = 48 * cih + 32 + i;
    const int yfrag = 48 * ch + 2 * i, yfrag2 = 48 * ch + 32 + i;
    // raw fragments of one k-step
    f32x2 Fa2[4], Fb2[4], Fg2[2], Fh2[2];
    float Fa1[4], Fb1[4], Fg1[2], Fh1[2];
    auto read_frag = [&](const float* buf) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const float* pa = buf + (r * PX + sa) * WGL_PITCH;
            const float* pb = buf + (r * PX + sb) * WGL_PITCH;
            Fa2[r] = *(const f32x2*)(pa + xfrag); Fb2[r] = *(const f32x2*)(pb + xfrag);
            Fa1[r] = pa[xfrag2]; Fb1[r] = pb[xfrag2];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            const float* q0 = buf + (4 * PX + a * PY + (EDGE ? kq : 2 * kq)) * WGL_PITCH;
            Fg2[a] = *(const f32x2*)(q0 + yfrag); Fg1[a] = q0[yfrag2];
            if (!EDGE) { Fh2[a] = *(const f32x2*)(q0 + WGL_PITCH + yfrag); Fh1[a] = q0[WGL_PITCH + yfrag2]; }
        }
    };
    float xh[4][3], yh[4][3];
    auto transform = [&]() {
        float w[4][3];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            if (NU == 1) { w[r][0] = Fa2[r][0] + Fb2[r][0]; w[r][1] = Fa2[r][1] + Fb2[r][1]; w[r][2] = Fa1[r] + Fb1[r]; }
            else { w[r][0] = Fa2[r][0] - Fb2[r][0]; w[r][1] = Fa2[r][1] - Fb2[r][1]; w[r][2] = Fa1[r] - Fb1[r]; }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            xh[0][e] = w[0][e] - w[2][e]; xh[1][e] = w[1][e] + w[2][e]; xh[2][e] = w[2][e] - w[1][e]; xh[3][e] = w[1][e] - w[3][e];
        }
        float c[2][3];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            if (EDGE) { c[a][0] = Fg2[a][0]; c[a][1] = Fg2[a][1]; c[a][2] = Fg1[a]; }
            else if (NU == 1) { c[a][0] = Fg2[a][0] + Fh2[a][0]; c[a][1] = Fg2[a][1] + Fh2[a][1]; c[a][2] = Fg1[a] + Fh1[a]; }
            else { c[a][0] = Fg2[a][0] - Fh2[a][0]; c[a][1] = Fg2[a][1] - Fh2[a][1]; c[a][2] = Fg1[a] - Fh1[a]; }
        }
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            yh[0][e] = c[0][e]; yh[1][e] = c[0][e] + c[1][e]; yh[2][e] = c[0][e] - c[1][e]; yh[3][e] = c[1][e];
        }
    };
    auto mfmas = [&]() {
        if (NU == 1) {
#pragma unroll
            for (int e = 0; e < 3; ++e) bsum[e] += yh[1][e];
        }
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 3; ++b)
                    acc[xi][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(yh[xi][a], xh[xi][b], acc[xi][a][b], 0, 0, 0);
    };

    // LDS-only barrier: __syncthreads() also waits vmcnt(0), which would bring the prefetched global loads home at every
    // k-step.  lgkmcnt(0) = this wave's ds_writes are done (and its ds_reads have returned).
    auto lds_barrier = [] {
        __builtin_amdgcn_s_waitcnt(0xC07F);                   // vmcnt(63) expcnt(7) lgkmcnt(0)
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    // Per k-step: global loads two steps ahead -> registers; fragments of this step from LDS, transforms, 36 MFMAs; the
    // next step's staged registers -> the other LDS buffer; barrier.  (A three-stage variant that reads the next step's
    // fragments under the MFMAs needs 36 more registers: it spilled at two workgroups per CU and, at one, ran 407 us
    // against 292.)
    float* buf0 = lds;
    float* buf1 = lds + WGL_BUF;
    auto clampst = [&](int sg) { return sg < nst ? sg : nst - 1; };
    if (nst > 0) {                                            // nst is even (W % 16 == 0), uniform over the workgroup
        issue(G0, m0, 0);
        issue(G1, m1, 1);
        commit(G0, m0, buf0);
        lds_barrier();
        for (int sg = 0; sg < nst; sg += 2) {
            issue(G0, m0, clampst(sg + 2));
            __builtin_amdgcn_sched_barrier(0);
            read_frag(buf0);
            transform();
            if (WGL_SPLIT) __builtin_amdgcn_sched_barrier(0);
            mfmas();
            __builtin_amdgcn_sched_barrier(0);
            commit(G1, m1, buf1);
            lds_barrier();
            issue(G1, m1, clampst(sg + 3));
            __builtin_amdgcn_sched_barrier(0);
            read_frag(buf1);
            transform();
            if (WGL_SPLIT) __builtin_amdgcn_sched_barrier(0);
            mfmas();
            __builtin_amdgcn_sched_barrier(0);
            commit(G0, m0, buf0);
            lds_barrier();
        }
    }
    // D[row = 4*kq + r][col = i]: row -> co = 48 ch + 2 row + a (a < 2) | 48 ch + 32 + row; col -> ci likewise with cih
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        float* o = slab + (size_t)slice * kWgSliceFloats + (size_t)(xi * 4 + NU) * 9216;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * kq + r;
                float* orow = o + (size_t)(48 * ch + (a < 2 ? 2 * row + a : 32 + row)) * 96;
                *(f32x2*)(orow + xfrag) = f32x2{acc[xi][a][0][r], acc[xi][a][1][r]};
                orow[xfrag2] = acc[xi][a][2][r];
            }
    }
    if (NU == 1 && cih == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) slab[(size_t)slice * kWgSliceFloats + 147456 + (a < 2 ? yfrag + a : yfrag2)] = v;
        }
    }
}

__global__ __launch_bounds__(256, 2) void conv3x3_wgrad_wino_lds_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                       int total_trows, int H, int W, int rows_per_slice,
                                                                       float* __restrict__ slab)
{
    __shared__ __attribute__((aligned(16))) float lds[2 * WGL_BUF];
    // the four nu of a slice are consecutive workgroups of one XCD (blockIdx % 8)
    int b = blockIdx.x, slice;
    if ((gridDim.x / 4) % 8 == 0) {
        const int xcd = b & 7, j = b >> 3;
        slice = (j / 4) * 8 + xcd;
        b = j % 4;
    } else {
        slice = b / 4;
        b = b % 4;
    }
    switch (b) {                                              // workgroup-uniform
    case 0: wgrad_wino_lds_body<0>(x, dy, total_trows, H, W, rows_per_slice, slab, slice, lds); break;
    case 1: wgrad_wino_lds_body<1>(x, dy, total_trows, H, W, rows_per_slice, slab, slice, lds); break;
    case 2: wgrad_wino_lds_body<2>(x, dy, total_trows, H, W, rows_per_slice, slab, slice, lds); break;
    default: wgrad_wino_lds_body<3>(x, dy, total_trows, H, W, rows_per_slice, slab, slice, lds); break;
    }
}

// M[pos][co][ci] (and the bias partials behind them) = sum over slices, fixed order: 64 entries x 4 slice groups per
// block (grid: kWgSliceFloats / 64 blocks; the pad entries are summed too and never read)
__global__ __launch_bounds__(256) void wino_wgrad_sum_kernel(const float* __restrict__ slab, int slices, float* __restrict__ m)
{
    __shared__ float part[4][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const size_t idx = (size_t)blockIdx.x * 64 + o;           // pos * 9216 + co * 96 + ci | 147456 + bias channel (| pad)
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int sl = g;
    for (; sl + 12 < slices; sl += 16) {
        s0 += slab[(size_t)sl * kWgSliceFloats + idx];
        s1 += slab[(size_t)(sl + 4) * kWgSliceFloats + idx];
        s2 += slab[(size_t)(sl + 8) * kWgSliceFloats + idx];
        s3 += slab[(size_t)(sl + 12) * kWgSliceFloats + idx];
    }
    for (; sl < slices; sl += 4) s0 += slab[(size_t)sl * kWgSliceFloats + idx];
    part[g][o] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0) m[idx] = (part[0][o] + part[1][o]) + (part[2][o] + part[3][o]);
}

// dW[co][ci][ky][kx] = sum_{xi,nu} A'[ky][xi] A'[kx][nu] s_xi s_nu M[xi][nu][co][ci],  s = (1, 1/2, 1/2, 1)
__global__ __launch_bounds__(256) void wino_wgrad_final_kernel(const float* __restrict__ m, float* __restrict__ dw,
                                                               float* __restrict__ dbias)
{
    const int idx = blockIdx.x * 256 + threadIdx.x;           // co * 96 + ci
    if (idx < 9216) {
        const float sc[4] = {1.f, 0.5f, 0.5f, 1.f};
        float t[3][4];                                        // rows transformed: t[ky][nu]
        float mm[4][4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int nu = 0; nu < 4; ++nu) mm[xi][nu] = m[(size_t)(xi * 4 + nu) * 9216 + idx] * (sc[xi] * sc[nu]);
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
            t[0][nu] = mm[0][nu] + mm[1][nu] + mm[2][nu];
            t[1][nu] = mm[1][nu] - mm[2][nu];
            t[2][nu] = mm[1][nu] + mm[2][nu] - mm[3][nu];
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            dw[(size_t)idx * 9 + ky * 3 + 0] = t[ky][0] + t[ky][1] + t[ky][2];
            dw[(size_t)idx * 9 + ky * 3 + 1] = t[ky][1] - t[ky][2];
            dw[(size_t)idx * 9 + ky * 3 + 2] = t[ky][1] + t[ky][2] - t[ky][3];
        }
    }
    if (dbias && idx < 96) dbias[idx] = m[147456 + idx];
}

static int wgrad_slices(int total_rows)
{
    constexpr int big = 128;
    return total_rows < 4 * big ? (total_rows + 3) / 4 : big;
}

static bool wgrad_direct()
{
    const char* e = getenv("SRGEN_WGRAD");            // read per call: tests switch it inside one process
    return e && !strcmp(e, "direct");
}

static int wgrad_lds_slices(int total_trows)
{
    constexpr int big = 128;
    return total_trows < 2 * big ? (total_trows + 1) / 2 : big;
}

size_t conv3x3_wgrad_scratch_bytes(int n, int h)
{
    const int slices = wgrad_slices(n * h);
    const size_t direct = ((size_t)slices * 9 * 9216 + (size_t)slices * 96) * sizeof(float);
    const int ws = wgrad_lds_slices(n * (h / 2));
    const size_t wino = ((size_t)ws + 1) * kWgSliceFloats * sizeof(float);
    const size_t split = (h & 63) ? 0 : conv3x3_wgrad_split_scratch_bytes(n, h);
    const size_t m = direct > wino ? direct : wino;
    return m > split ? m : split;
}

hipError_t launch_conv3x3_wgrad_split(const void* x, const void* dy, int n, int h, const float* dy_amax, float* dw,
                                      float* dbias, void* scratch, hipStream_t s, int terms, bool bf16)
{
    int slices = 0;
    float* slab = (float*)scratch;
    // (the bias partials sit behind the largest slab the slice count can ask for)
    float* bias_slab = slab + (size_t)kWgradSplitMaxSlices * 9 * 9216;
    hipError_t e = launch_conv3x3_wgrad_split_partial(x, dy, n, h, dy_amax, slab, bias_slab, &slices, s, terms, bf16);
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(9 * 9216 / 64), dim3(1024), 0, s, slab, bias_slab, slices, dw, dbias);
    return hipGetLastError();
}

hipError_t launch_conv3x3_wgrad(const float* x, const float* dy, int n, int h, float* dw, float* dbias, void* scratch,
                                hipStream_t s)
{
    if (!wgrad_direct() && h % 16 == 0) {
        const int total_trows = n * (h / 2);
        const int slices = wgrad_lds_slices(total_trows);
        const int rps = (total_trows + slices - 1) / slices;
        float* slab = (float*)scratch;
        float* m = slab + (size_t)slices * kWgSliceFloats;
        hipLaunchKernelGGL(conv3x3_wgrad_wino_lds_kernel, dim3((unsigned)(slices * 4)), dim3(256), 0, s, x, dy, total_trows, h, h,
                           rps, slab);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(wino_wgrad_sum_kernel, dim3((unsigned)(kWgSliceFloats / 64)), dim3(256), 0, s, slab, slices, m);
        hipLaunchKernelGGL(wino_wgrad_final_kernel, dim3(36), dim3(256), 0, s, m, dw, dbias);
        return hipGetLastError();
    }
    const int total_rows = n * h;
    const int slices = wgrad_slices(total_rows);
    const int rps = (total_rows + slices - 1) / slices;
    float* slab = (float*)scratch;
    float* bias_slab = slab + (size_t)slices * 9 * 9216;
    hipLaunchKernelGGL(conv3x3_wgrad_kernel, dim3((unsigned)(slices * 9)), dim3(192), 0, s, x, dy, total_rows, h, h, rps,
                       slices, slab, bias_slab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(conv3x3_wgrad_reduce_kernel, dim3(9 * 9216 / 64), dim3(1024), 0, s, slab, bias_slab, slices,
                       dw, dbias);
    return hipGetLastError();
}

// ---- train-mode BatchNorm backward (+ PReLU backward in front of it) ------------------------------------------
// Forward (Generator.py:35-36 / :38,42): z = gamma * xhat + beta, xhat = (c - mean) * rstd; t = PReLU(z) or z.
// Given dt (times dout_scale: the 0.1 of the residual branch), dz = dt * (z >= 0 ? 1 : slope[ch]);
//   dgamma = sum dz * xhat, dbeta = sum dz, dslope = sum dt * z * [z < 0],
//   dc = gamma * rstd * (dz - dbeta / m - xhat * dgamma / m)                    (m = pixels per channel).
// Pass 1: per-channel sums (fp32 inside a thread's run, fp64 across threads and workgroups, fixed order);
// pass 2 (one workgroup): coefficients; pass 3: apply.  xhat and z are recomputed from the saved conv output.
constexpr int BNB_THREADS = 240, BNB_BLOCKS = 1024;

// per-channel constants shared by the three passes: z = c * p + q (BatchNorm output), xhat = c * rstd - mr
__device__ __forceinline__ void bn_consts(const float* stats, const float* gamma, const float* beta, int ch, float& rstd,
                                          float& mr, float& p, float& q)
{
    rstd = 1.0f / sqrtf(stats[96 + ch] + 1e-5f);
    mr = stats[ch] * rstd;
    p = gamma[ch] * rstd;
    q = beta[ch] - stats[ch] * p;
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_stats_kernel(const T* __restrict__ c, const T* __restrict__ dt,
                                                           const float* __restrict__ stats, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, const float* __restrict__ slope,
                                                           float dscale, size_t npix, double* partial)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V, ROWS = BNB_THREADS / CPP;   // 24 x 10 (fp32) or 12 x 20 (bf16)
    constexpr int STEP = ROWS * 96;
    __shared__ float red[ROWS][3][96];          // (a thread's sums are fp32; they are added in fp64 below)
    const int tid = threadIdx.x;
    if (tid < BNB_THREADS) {
        const int cg = tid % CPP, row = tid / CPP;
        float rstd[V], mr[V], pp[V], qq[V], sl[V];
        {                                                        // bn_consts for this thread's channels, from 16-byte loads
            float mean[V], var[V], ga[V], be[V];
            ldf<V>(stats + cg * V, mean);
            ldf<V>(stats + 96 + cg * V, var);
            ldf<V>(gamma + cg * V, ga);
            ldf<V>(beta + cg * V, be);
            if (slope) ldf<V>(slope + cg * V, sl);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                rstd[e] = 1.0f / sqrtf(var[e] + 1e-5f);
                mr[e] = mean[e] * rstd[e];
                pp[e] = ga[e] * rstd[e];
                qq[e] = be[e] - mean[e] * pp[e];
                if (!slope) sl[e] = 1.f;
            }
        }
        const size_t per = (npix + gridDim.x - 1) / gridDim.x;
        const size_t p0 = (size_t)blockIdx.x * per, p1 = p0 + per < npix ? p0 + per : npix;
        float s_dz[V], s_dzx[V], s_sl[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s_dz[e] = s_dzx[e] = s_sl[e] = 0.f;
        const T* cp = c + (p0 + row) * 96 + cg * V;
        const T* dp = dt + (p0 + row) * 96 + cg * V;
        const int iters = p1 > p0 + row ? (int)((p1 - p0 - row + ROWS - 1) / ROWS) : 0;     // pixels of this thread
        int it = 0;
#define BNB_ADD(cv, dv)                                                     \
            _Pragma("unroll") for (int e = 0; e < V; ++e) {                 \
                const float xh = fmaf(cv[e], rstd[e], -mr[e]);              \
                const float z = fmaf(cv[e], pp[e], qq[e]);                  \
                const float d = dv[e] * dscale;                             \
                const bool neg = slope && z < 0.f;                          \
                const float dz = neg ? d * sl[e] : d;                       \
                s_dz[e] += dz;                                              \
                s_dzx[e] += dz * xh;                                        \
                s_sl[e] += neg ? d * z : 0.f;                               \
            }
        for (; it + 4 <= iters; it += 4) {                       // eight loads in flight per thread
            float cv0[V], cv1[V], cv2[V], cv3[V], dv0[V], dv1[V], dv2[V], dv3[V];
            ldv(cp, cv0); ldv(cp + STEP, cv1); ldv(cp + 2 * STEP, cv2); ldv(cp + 3 * STEP, cv3);
            ldv(dp, dv0); ldv(dp + STEP, dv1); ldv(dp + 2 * STEP, dv2); ldv(dp + 3 * STEP, dv3);
            cp += 4 * STEP; dp += 4 * STEP;
            BNB_ADD(cv0, dv0) BNB_ADD(cv1, dv1) BNB_ADD(cv2, dv2) BNB_ADD(cv3, dv3)
        }
        for (; it < iters; ++it) {
            float cv[V], dv[V];
            ldv(cp, cv);
            ldv(dp, dv);
            cp += STEP; dp += STEP;
            BNB_ADD(cv, dv)
        }
#undef BNB_ADD
        for (int e = 0; e < V; ++e) {
            red[row][0][cg * V + e] = s_dz[e];
            red[row][1][cg * V + e] = s_dzx[e];
            red[row][2][cg * V + e] = s_sl[e];
        }
    }
    __syncthreads();
    for (int k = tid; k < 288; k += 256) {
        const int which = k / 96, ch = k % 96;
        double a = 0.0;
        for (int r = 0; r < ROWS; ++r) a += (double)red[r][which][ch];
        partial[(size_t)blockIdx.x * 288 + k] = a;
    }
}

// coef [5][96]: dc = A * dz - C - B * c with z = c * p + q deciding the PReLU branch:  {A, B, C, p, q}.
// Block b owns channels [4b, 4b + 4): 12 sums x 85 threads, each adding every 85th partial (fixed order; eight blocks of 28
// adders took 11.5 us per BatchNorm: nine dependent L2 round trips).
constexpr int BNBF_CPB = 4, BNBF_BLOCKS = 96 / BNBF_CPB, BNBF_ADDERS = 1024 / (3 * BNBF_CPB);
__global__ __launch_bounds__(1024) void bn_bwd_final_kernel(const double* partial, int nb, double npix,
                                                            const float* stats, const float* gamma, const float* beta, float* coef,
                                                            float* dgamma, float* dbeta, float* dslope, float* amax)
{
    if (amax && blockIdx.x == 0) amax[threadIdx.x] = 0.f;       // kAmaxSlots = 1024 = this workgroup (see bn_bwd_apply_kernel)
    __shared__ double part[BNBF_ADDERS][3 * BNBF_CPB];
    __shared__ double tot[3 * BNBF_CPB];
    const int t = threadIdx.x;
    if (t < BNBF_ADDERS * 3 * BNBF_CPB) {
        const int s = t % (3 * BNBF_CPB), q = t / (3 * BNBF_CPB);
        const int k = (s / BNBF_CPB) * 96 + blockIdx.x * BNBF_CPB + s % BNBF_CPB;
        double a0 = 0.0, a1 = 0.0, a2 = 0.0, a3 = 0.0;
        int b = q;
        for (; b + 3 * BNBF_ADDERS < nb; b += 4 * BNBF_ADDERS) {
            a0 += partial[(size_t)b * 288 + k];
            a1 += partial[(size_t)(b + BNBF_ADDERS) * 288 + k];
            a2 += partial[(size_t)(b + 2 * BNBF_ADDERS) * 288 + k];
            a3 += partial[(size_t)(b + 3 * BNBF_ADDERS) * 288 + k];
        }
        for (; b < nb; b += BNBF_ADDERS) a0 += partial[(size_t)b * 288 + k];
        part[q][s] = (a0 + a1) + (a2 + a3);
    }
    __syncthreads();
    if (t < 3 * BNBF_CPB) {
        double a = 0.0;
        for (int q = 0; q < BNBF_ADDERS; ++q) a += part[q][t];
        tot[t] = a;
    }
    __syncthreads();
    if (t < BNBF_CPB) {
        const int ch = blockIdx.x * BNBF_CPB + t;
        const double rstd = 1.0 / sqrt((double)stats[96 + ch] + 1e-5);
        const double mean = (double)stats[ch];
        const double k1 = (double)gamma[ch] * rstd;
        const double k2 = k1 * tot[t] / npix, k3 = k1 * tot[BNBF_CPB + t] / npix;      // dc = k1 dz - k2 - k3 xhat
        coef[ch] = (float)k1;
        coef[96 + ch] = (float)(k3 * rstd);
        coef[192 + ch] = (float)(k2 - k3 * mean * rstd);
        coef[288 + ch] = (float)k1;                                             // p = gamma * rstd
        coef[384 + ch] = (float)((double)beta[ch] - mean * k1);                 // q
        dbeta[ch] = (float)tot[t];
        dgamma[ch] = (float)tot[BNBF_CPB + t];
        if (dslope) dslope[ch] = (float)tot[2 * BNBF_CPB + t];
    }
}

// grid * 256 is a multiple of 24 (and of 12), so a thread meets the same channels at every stride: constants in registers
template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ c, const T* __restrict__ dt,
                                                           const float* __restrict__ slope, const float* __restrict__ coef,
                                                           float dscale, T* __restrict__ dc, size_t nvec, float* amax)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V;
    const size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x;
    float m = 0.f;
    const int ch = (int)(i0 % CPP) * V;
    float A[V], B[V], C[V], P[V], Q[V], S[V];
    ldf<V>(coef + ch, A);
    ldf<V>(coef + 96 + ch, B);
    ldf<V>(coef + 192 + ch, C);
    ldf<V>(coef + 288 + ch, P);
    ldf<V>(coef + 384 + ch, Q);
    if (slope) ldf<V>(slope + ch, S);
#pragma unroll
    for (int e = 0; e < V; ++e) {
        A[e] *= dscale;
        if (!slope) S[e] = 1.f;
    }
    for (size_t i = i0; i < nvec; i += (size_t)gridDim.x * 256) {
        float cv[V], dv[V], o[V];
        ldv(c + i * V, cv);
        ldv(dt + i * V, dv);
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float z = fmaf(cv[e], P[e], Q[e]);
            const float dz = (slope && z < 0.f) ? dv[e] * S[e] : dv[e];
            o[e] = fmaf(A[e], dz, -C[e]) - B[e] * cv[e];
            m = fmaxf(m, fabsf(o[e]));
        }
        stv(dc + i * V, o);
    }
    // max |dc| for the split-f16 data-gradient conv that reads dc next (its in_amax): slot = workgroup mod kAmaxSlots,
    // zeroed by bn_bwd_final_kernel; non-negative floats order like their bit patterns
    if (amax) {
        m = block_max_256(m);
        if (threadIdx.x == 0 && m == m) atomicMax((unsigned*)amax + (blockIdx.x & (kAmaxSlots - 1)), __float_as_uint(m));
    }
}

size_t bn_bwd_scratch_bytes() { return (size_t)BNB_BLOCKS * 288 * sizeof(double) + 480 * sizeof(float); }

template <typename T>
static hipError_t bn_train_bwd_t(const T* c, const T* dt, const float* stats, const float* gamma, const float* beta,
                                 const float* slope, float dscale, T* dc, float* dgamma, float* dbeta, float* dslope,
                                 size_t npix, void* scratch, hipStream_t s, float* dc_amax)
{
    static_assert(kAmaxSlots == 1024, "bn_bwd_final_kernel zeroes one slot per thread");
    double* partial = (double*)scratch;
    float* coef = (float*)(partial + (size_t)BNB_BLOCKS * 288);
    constexpr int ROWS = BNB_THREADS / (96 / StoreVec<T>::N);
    const int nb = npix < (size_t)BNB_BLOCKS * ROWS ? (int)((npix + ROWS - 1) / ROWS) : BNB_BLOCKS;
    hipLaunchKernelGGL(bn_bwd_stats_kernel<T>, dim3(nb), dim3(256), 0, s, c, dt, stats, gamma, beta, slope, dscale, npix, partial);
    hipLaunchKernelGGL(bn_bwd_final_kernel, dim3(BNBF_BLOCKS), dim3(1024), 0, s, partial, nb, (double)npix, stats, gamma, beta, coef, dgamma,
                       dbeta, dslope, dc_amax);
    const size_t nvec = npix * (96 / StoreVec<T>::N);
    size_t want = (nvec + 255) / 256;
    const size_t cap = 6144 * 4 / StoreVec<T>::N;                // (the per-thread constants are loaded once: >= 4 units each)
    if (want > cap) want = cap;
    want = (want + 2) / 3 * 3;                                   // (grid * 256) % 24 == 0
    hipLaunchKernelGGL(bn_bwd_apply_kernel<T>, dim3((unsigned)want), dim3(256), 0, s, c, dt, slope, coef, dscale, dc, nvec, dc_amax);
    return hipGetLastError();
}

hipError_t launch_bn_train_bwd(const float* c, const float* dt, const float* stats, const float* gamma, const float* beta,
                               const float* slope, float dscale, float* dc, float* dgamma, float* dbeta, float* dslope,
                               size_t npix, void* scratch, hipStream_t s, float* dc_amax)
{
    return bn_train_bwd_t<float>(c, dt, stats, gamma, beta, slope, dscale, dc, dgamma, dbeta, dslope, npix, scratch, s, dc_amax);
}

hipError_t launch_bn_train_bwd_bf16(const void* c, const void* dt, const float* stats, const float* gamma, const float* beta,
                                    const float* slope, float dscale, void* dc, float* dgamma, float* dbeta, float* dslope,
                                    size_t npix, void* scratch, hipStream_t s)
{
    return bn_train_bwd_t<bf16_t>((const bf16_t*)c, (const bf16_t*)dt, stats, gamma, beta, slope, dscale, (bf16_t*)dc, dgamma,
                                  dbeta, dslope, npix, scratch, s, nullptr);
}

// ---- gradient of the pixel losses: weighted_l1(sr, hr) + w_ssim * (1 - ssim(sr, hr)) with respect to sr -------------
// ssim map S = A1 A2 / (B1 B2) over the five gaussian-filtered maps a = mu_x, b = mu_y, e = E[x^2], f = E[y^2],
// g = E[xy] (utils.py:26-41; zero-padded 11x11, sigma 1.5; den clamped at 1e-8: no gradient where clamped).
//   dS/da = (2b (A2 - A1)) / (B1 B2) - A1 A2 (2a (B2 - B1)) / (B1 B2)^2,  dS/de = -A1 A2 / (B1 B2^2),  dS/dg = 2 A1 / (B1 B2)
//   d ssim / d sr = (1/M) [ G*Da + 2 sr . (G*De) + hr . (G*Dg) ]            (G*: the same zero-padded filter)
// Kernel 1 (tile kernel, same staging as metrics_tile_kernel) writes the three derivative maps; kernel 2 filters
// them and adds the weighted-L1 term: (9 [hr > 0.01] + 1) sign(sr - hr) / M  (train.py:18-20).
constexpr int LT = 32, LH = LT + 10;

__device__ __forceinline__ void gauss11(float* g)
{
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 11; ++i) { const float d = (float)(i - 5); g[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f)); s += g[i]; }
#pragma unroll
    for (int i = 0; i < 11; ++i) g[i] /= s;
}

__global__ __launch_bounds__(256) void ssim_dmaps_kernel(const float* __restrict__ sr, const float* __restrict__ hr, int h, int w,
                                                         int tiles_x, int tiles_y, float* __restrict__ dmaps /*[3][planes][h][w]*/,
                                                         size_t plane_stride_total)
{
    __shared__ float a[LH][LH + 1], b[LH][LH + 1];
    __shared__ float hq[5][LH][LT + 1];
    float g[11];
    gauss11(g);
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const size_t plane = bid / tiles_y;
    const float* ps = sr + plane * h * w;
    const float* ph = hr + plane * h * w;
    const int x0 = tx * LT - 5, y0 = ty * LT - 5;
    for (int i = threadIdx.x; i < LH * LH; i += 256) {
        const int r = i / LH, c = i - r * LH;
        const int gy = y0 + r, gx = x0 + c;
        const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
        a[r][c] = ok ? ps[(size_t)gy * w + gx] : 0.f;
        b[r][c] = ok ? ph[(size_t)gy * w + gx] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LH * LT; i += 256) {
        const int r = i / LT, c = i - r * LT;
        float mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) {
            const float va = a[r][c + k], vb = b[r][c + k], gk = g[k];
            mx += gk * va; my += gk * vb; xx += gk * va * va; yy += gk * vb * vb; xy += gk * va * vb;
        }
        hq[0][r][c] = mx; hq[1][r][c] = my; hq[2][r][c] = xx; hq[3][r][c] = yy; hq[4][r][c] = xy;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LT * LT; i += 256) {
        const int r = i / LT, c = i - r * LT;
        const int gy = ty * LT + r, gx = tx * LT + c;
        if (gy < h && gx < w) {
            float mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) {
                const float gk = g[k];
                mx += gk * hq[0][r + k][c]; my += gk * hq[1][r + k][c]; xx += gk * hq[2][r + k][c];
                yy += gk * hq[3][r + k][c]; xy += gk * hq[4][r + k][c];
            }
            const float c1 = 0.01f * 0.01f, c2 = 0.03f * 0.03f;
            const float A1 = 2.f * mx * my + c1, A2 = 2.f * (xy - mx * my) + c2;
            const float B1 = mx * mx + my * my + c1, B2 = (xx - mx * mx) + (yy - my * my) + c2;
            const float den = B1 * B2;
            float da = 0.f, de = 0.f, dg = 0.f;
            if (den >= 1e-8f) {                                   // clamp(min=1e-8): no gradient below it
                const float inv = 1.f / den;
                da = 2.f * my * (A2 - A1) * inv - A1 * A2 * (2.f * mx * (B2 - B1)) * inv * inv;
                de = -A1 * A2 * inv / B2;
                dg = 2.f * A1 * inv;
            }
            const size_t o = plane * h * w + (size_t)gy * w + gx;
            dmaps[o] = da;
            dmaps[plane_stride_total + o] = de;
            dmaps[2 * plane_stride_total + o] = dg;
        }
    }
}

__global__ __launch_bounds__(256) void pixel_loss_grad_kernel(const float* __restrict__ sr, const float* __restrict__ hr,
                                                              const float* __restrict__ dmaps, size_t plane_stride_total, int h,
                                                              int w, int tiles_x, int tiles_y, float w_l1, float w_ssim,
                                                              float inv_m, float* __restrict__ dsr)
{
    __shared__ float d[3][LH][LH + 1];
    __shared__ float hq[3][LH][LT + 1];
    float g[11];
    gauss11(g);
    int bid = blockIdx.x;
    const int tx = bid % tiles_x; bid /= tiles_x;
    const int ty = bid % tiles_y;
    const size_t plane = bid / tiles_y;
    const int x0 = tx * LT - 5, y0 = ty * LT - 5;
    for (int i = threadIdx.x; i < LH * LH; i += 256) {
        const int r = i / LH, c = i - r * LH;
        const int gy = y0 + r, gx = x0 + c;
        const bool ok = (unsigned)gy < (unsigned)h && (unsigned)gx < (unsigned)w;
        const size_t o = plane * h * w + (size_t)gy * w + gx;
#pragma unroll
        for (int k = 0; k < 3; ++k) d[k][r][c] = ok ? dmaps[k * plane_stride_total + o] : 0.f;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LH * LT; i += 256) {
        const int r = i / LT, c = i - r * LT;
        float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
        for (int k = 0; k < 11; ++k) { s0 += g[k] * d[0][r][c + k]; s1 += g[k] * d[1][r][c + k]; s2 += g[k] * d[2][r][c + k]; }
        hq[0][r][c] = s0; hq[1][r][c] = s1; hq[2][r][c] = s2;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < LT * LT; i += 256) {
        const int r = i / LT, c = i - r * LT;
        const int gy = ty * LT + r, gx = tx * LT + c;
        if (gy < h && gx < w) {
            float s0 = 0, s1 = 0, s2 = 0;
#pragma unroll
            for (int k = 0; k < 11; ++k) { s0 += g[k] * hq[0][r + k][c]; s1 += g[k] * hq[1][r + k][c]; s2 += g[k] * hq[2][r + k][c]; }
            const size_t o = plane * h * w + (size_t)gy * w + gx;
            const float xs = sr[o], ys = hr[o];
            const float dssim = (s0 + 2.f * xs * s1 + ys * s2) * inv_m;
            const float df = xs - ys;
            const float sgn = df > 0.f ? 1.f : (df < 0.f ? -1.f : 0.f);
            const float dl1 = ((ys > 0.01f ? 1.f : 0.f) * 9.f + 1.f) * sgn * inv_m;
            dsr[o] = w_l1 * dl1 - w_ssim * dssim;
        }
    }
}

hipError_t launch_pixel_loss_grad(const float* sr, const float* hr, int n, int c, int h, int w, float w_l1, float w_ssim,
                                  float* dsr, void* scratch, hipStream_t s)
{
    const int tiles_x = (w + LT - 1) / LT, tiles_y = (h + LT - 1) / LT;
    const int nb = n * c * tiles_x * tiles_y;
    const size_t total = (size_t)n * c * h * w;
    float* dmaps = (float*)scratch;
    hipLaunchKernelGGL(ssim_dmaps_kernel, dim3((unsigned)nb), dim3(256), 0, s, sr, hr, h, w, tiles_x, tiles_y, dmaps, total);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    hipLaunchKernelGGL(pixel_loss_grad_kernel, dim3((unsigned)nb), dim3(256), 0, s, sr, hr, dmaps, total, h, w, tiles_x, tiles_y,
                       w_l1, w_ssim, 1.0f / (float)total, dsr);
    return hipGetLastError();
}

// ---- AdamW (torch.optim.AdamW: decoupled weight decay, bias-corrected moments) over a flat buffer ----------------
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                    float* __restrict__ v, size_t n, float lr, float beta1, float beta2, float eps,
                                                    float wd, float bc1, float bc2_sqrt)
{
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = g[i];
        float pi = p[i] * (1.f - lr * wd);                       // param.mul_(1 - lr * weight_decay)
        const float mi = beta1 * m[i] + (1.f - beta1) * gi;      // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = beta2 * v[i] + (1.f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        pi -= (lr / bc1) * (mi / denom);
        p[i] = pi;
    }
}

hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                        float wd, int step, hipStream_t s)
{
    const double bc1 = 1.0 - pow((double)beta1, step), bc2 = 1.0 - pow((double)beta2, step);
    size_t blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) return hipSuccess;
    hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, s, p, g, m, v, n, lr, beta1, beta2, eps, wd, (float)bc1,
                       (float)sqrt(bc2));
    return hipGetLastError();
}


// ---- scalar-slope PReLU (Generator.py:51, 9, 70), elementwise over any fp32 tensor -----------------------------------
template <typename T>
__global__ __launch_bounds__(256) void prelu_scalar_fwd_kernel(const T* __restrict__ x, const float* __restrict__ slope,
                                                               T* __restrict__ y, size_t nvec)
{
    constexpr int V = StoreVec<T>::N;
    const float a = slope[0];
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float v[V];
        ldv(x + i * V, v);
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * a;
        stv(y + i * V, v);
    }
}

// dx = dy * (x >= 0 ? 1 : a), dy = dy1 (+ dy2: a second consumer of the PReLU output, e.g. the long skip);
// per-workgroup partial of da = sum dy * x * [x < 0] (fp64 across threads)
// UNSH (the upsample block's PReLU, Generator.py:8-9 backward): dx is written in the layout the PixelShuffle's backward
// wants -- (2, 2, n, full/2, full/2, 96): plane (y & 1, x & 1) holds the gradient of conv-output block 2 (y & 1) + (x & 1) --
// so the four data-gradient / weight-gradient convs below read contiguous tensors and no permute copy runs (129 us per
// 64-image step for a pass this kernel makes anyway).
template <typename T, bool UNSH = false>
__global__ __launch_bounds__(256) void prelu_scalar_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                               const T* __restrict__ dy2, const float* __restrict__ slope,
                                                               T* __restrict__ dx, size_t nvec, double* __restrict__ partial,
                                                               unsigned nimg = 0, unsigned full = 0)
{
    constexpr int V = StoreVec<T>::N;
    __shared__ double red[256];
    const float a = slope[0];
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        float v[V], d[V];
        ldv(x + i * V, v);
        ldv(dy + i * V, d);
        if (dy2) {
            float d2[V];
            ldv(dy2 + i * V, d2);
#pragma unroll
            for (int e = 0; e < V; ++e) d[e] += d2[e];
        }
#pragma unroll
        for (int e = 0; e < V; ++e)
            if (v[e] < 0.f) { acc += d[e] * v[e]; d[e] *= a; }
        size_t j = i;
        if (UNSH) {
            constexpr unsigned CPP = 96 / V;
            const size_t pixel = i / CPP;
            const unsigned c = (unsigned)(i - pixel * CPP);
            const unsigned xx = (unsigned)(pixel % full), t = (unsigned)(pixel / full);
            const unsigned yy = t % full, n = t / full, half = full >> 1;
            j = ((((size_t)((yy & 1u) * 2u + (xx & 1u)) * nimg + n) * half + (yy >> 1)) * half + (xx >> 1)) * CPP + c;
        }
        stv(dx + j * V, d);
    }
    red[threadIdx.x] = (double)acc;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void sum_partials_kernel(const double* __restrict__ partial, int nb, float* __restrict__ out)
{
    __shared__ double red[256];
    double a = 0.0;
    for (int i = threadIdx.x; i < nb; i += 256) a += partial[i];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = (float)red[0];
}

constexpr int PRELU_BLOCKS = 2048;
size_t prelu_scratch_bytes() { return PRELU_BLOCKS * sizeof(double); }

template <typename T>
static hipError_t prelu_fwd_t(const T* x, const float* slope, T* y, size_t numel, hipStream_t s)
{
    const size_t n4 = numel / StoreVec<T>::N;
    size_t nb = (n4 + 255) / 256;
    if (nb > 16384) nb = 16384;
    hipLaunchKernelGGL(prelu_scalar_fwd_kernel<T>, dim3((unsigned)nb), dim3(256), 0, s, x, slope, y, n4);
    return hipGetLastError();
}

template <typename T>
static hipError_t prelu_bwd_t(const T* x, const T* dy, const T* dy2, const float* slope, T* dx, float* dslope, size_t numel,
                              void* scratch, hipStream_t s, int unshuffle_n = 0, int full = 0)
{
    const size_t n4 = numel / StoreVec<T>::N;
    size_t nb = (n4 + 255) / 256;
    if (nb > PRELU_BLOCKS) nb = PRELU_BLOCKS;
    if (unshuffle_n > 0)
        hipLaunchKernelGGL((prelu_scalar_bwd_kernel<T, true>), dim3((unsigned)nb), dim3(256), 0, s, x, dy, dy2, slope, dx, n4,
                           (double*)scratch, (unsigned)unshuffle_n, (unsigned)full);
    else
        hipLaunchKernelGGL((prelu_scalar_bwd_kernel<T, false>), dim3((unsigned)nb), dim3(256), 0, s, x, dy, dy2, slope, dx, n4,
                           (double*)scratch, 0u, 0u);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(256), 0, s, (const double*)scratch, (int)nb, dslope);
    return hipGetLastError();
}

hipError_t launch_prelu_scalar_fwd(const float* x, const float* slope, float* y, size_t numel, hipStream_t s)
{
    return prelu_fwd_t<float>(x, slope, y, numel, s);
}

hipError_t launch_prelu_scalar_bwd(const float* x, const float* dy, const float* dy2, const float* slope, float* dx, float* dslope,
                                   size_t numel, void* scratch, hipStream_t s)
{
    return prelu_bwd_t<float>(x, dy, dy2, slope, dx, dslope, numel, scratch, s);
}

// x, dy (n, full, full, 96) -> dx (2, 2, n, full/2, full/2, 96)
hipError_t launch_prelu_scalar_bwd_unshuffle(const void* x, const void* dy, const float* slope, void* dx, float* dslope, int n, int full,
                                             void* scratch, hipStream_t s, bool bf16)
{
    const size_t numel = (size_t)n * full * full * 96;
    if (bf16)
        return prelu_bwd_t<bf16_t>((const bf16_t*)x, (const bf16_t*)dy, nullptr, slope, (bf16_t*)dx, dslope, numel, scratch, s, n, full);
    return prelu_bwd_t<float>((const float*)x, (const float*)dy, nullptr, slope, (float*)dx, dslope, numel, scratch, s, n, full);
}

hipError_t launch_prelu_scalar_fwd_bf16(const void* x, const float* slope, void* y, size_t numel, hipStream_t s)
{
    return prelu_fwd_t<bf16_t>((const bf16_t*)x, slope, (bf16_t*)y, numel, s);
}

hipError_t launch_prelu_scalar_bwd_bf16(const void* x, const void* dy, const void* dy2, const float* slope, void* dx,
                                        float* dslope, size_t numel, void* scratch, hipStream_t s)
{
    return prelu_bwd_t<bf16_t>((const bf16_t*)x, (const bf16_t*)dy, (const bf16_t*)dy2, slope, (bf16_t*)dx, dslope, numel, scratch, s);
}

// ---- 9x9 convs against a 3-channel NCHW image (stem Generator.py:50; head Generator.py:71 transposed) -------------------
// One GEMM shape serves four uses: K or N = j = (c3, ky, kx), 243 taps (padded to 244 / 256), the other side = 96 channels.
//   tap of output / activation pixel (y, x):  img[n][c3][y + s*(ky-4)][x + s*(kx-4)],  s = +1 (flip = 0) or -1 (flip = 1),
//   zero outside the ih x iw image.
//   conv9x9_c3_kernel   : out[n][y][x][c] = bias[c] + sum_j tap_j * wm[j][c]
//        flip 0: the stem forward (img = lr input, wm[j][co] = w[co][ci][ky][kx]);
//        flip 1: the head's data gradient (img = d out, 125 x 125, wm[(co,ky,kx)][ci] = w[co][ci][ky][kx], out 128 x 128).
//   conv9x9_c3_wgrad    : r[c][j] = sum over pixels act[n][y][x][c] * tap_j
//        flip 0: the stem's weight gradient (act = d x1, img = lr input): r = dw[co][ci][ky][kx];
//        flip 1: the head's weight gradient (act = head input, img = d out): r[ci][(co,ky,kx)] = dw[co][ci][ky][kx].
// Both on v_mfma_f32_16x16x4_f32 with operands straight from memory (the image is tiny and cache resident).
__device__ __forceinline__ void tap_split(int j, int& c3, int& ky, int& kx)
{
    c3 = j / 81;
    const int r = j - c3 * 81;
    ky = r / 9;
    kx = r - ky * 9;
}

__global__ __launch_bounds__(256) void conv9x9_c3_kernel(const float* __restrict__ img, const float* __restrict__ wm,
                                                         const float* __restrict__ bias, float* __restrict__ out, int ih, int iw,
                                                         int oh, int ow, int flip, long long total_pix)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const long long p0 = ((long long)blockIdx.x * 4 + wv) * 64;          // this wave's 64 consecutive output pixels
    if (p0 >= total_pix) return;
    // per pixel tile t: pixel p0 + 16 t + col (a row never straddles a tile: ow is a multiple of 16)
    int py[4], px[4];
    const float* ib[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const long long p = p0 + 16 * t + col;
        const long long n = p / ((long long)oh * ow);
        const int rem = (int)(p - n * (long long)oh * ow);
        py[t] = rem / ow;
        px[t] = rem - py[t] * ow;
        ib[t] = img + (size_t)n * 3 * ih * iw;
    }
    f32x4 acc[6][4];
#pragma unroll
    for (int a = 0; a < 6; ++a) {
        f32x4 b4 = {0.f, 0.f, 0.f, 0.f};
        if (bias) b4 = *(const f32x4*)(bias + 16 * a + 4 * kq);          // D rows 4*kq + r = channels 16a + 4kq + r
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[a][t] = b4;
    }
    const int sg = flip ? -1 : 1;
    for (int s = 0; s < 61; ++s) {
        const int j = 4 * s + kq;
        int c3, ky, kx;
        tap_split(j, c3, ky, kx);
        const bool jv = j < 243;
        float av[6], bv[4];
#pragma unroll
        for (int a = 0; a < 6; ++a) av[a] = jv ? wm[(size_t)j * 96 + 16 * a + col] : 0.f;      // A[i = channel][k = tap]
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int yy = py[t] + sg * (ky - 4), xx = px[t] + sg * (kx - 4);
            const bool ok = jv && (unsigned)yy < (unsigned)ih && (unsigned)xx < (unsigned)iw;
            bv[t] = ok ? ib[t][((size_t)c3 * ih + yy) * iw + xx] : 0.f;                         // B[k = tap][j = pixel]
        }
#pragma unroll
        for (int a = 0; a < 6; ++a)
#pragma unroll
            for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        float* o = out + (size_t)(p0 + 16 * t + col) * 96 + 4 * kq;       // D[row = channel 4kq + r][col = pixel]
#pragma unroll
        for (int a = 0; a < 6; ++a) *(f32x4*)(o + 16 * a) = acc[a][t];
    }
}

hipError_t launch_conv9x9_c3(const float* img, const float* wm, const float* bias, float* out, int n, int ih, int iw, int oh,
                             int ow, int flip, hipStream_t s)
{
    const long long total = (long long)n * oh * ow;
    hipLaunchKernelGGL(conv9x9_c3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, img, wm, bias, out, ih, iw, oh,
                       ow, flip, total);
    return hipGetLastError();
}

// slab [slice][96][256]; wave wv owns taps j in [64 wv, 64 wv + 64) x all 96 channels
template <typename T>
__global__ __launch_bounds__(256) void conv9x9_c3_wgrad_kernel(const T* __restrict__ act, const float* __restrict__ img,
                                                               int total_rows, int ah, int aw, int ih, int iw, int flip,
                                                               int rows_per_slice, float* __restrict__ slab,
                                                               float* __restrict__ bias_slab)
{
    const int slice = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int sg = flip ? -1 : 1;
    int toff[4], tdy[4], tdx[4];
    bool tv[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int j = 64 * wv + 16 * t + col;
        int c3, ky, kx;
        tap_split(j, c3, ky, kx);
        tv[t] = j < 243;
        tdy[t] = sg * (ky - 4);
        tdx[t] = sg * (kx - 4);
        toff[t] = (c3 * ih + tdy[t]) * iw + tdx[t];
    }
    f32x4 acc[6][4];
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[a][t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int r0 = slice * rows_per_slice, r1 = min(r0 + rows_per_slice, total_rows);
    for (int row = r0; row < r1; ++row) {
        const int n = row / ah, y = row - n * ah;
        const T* ar = act + (size_t)row * aw * 96 + col;
        const float* ibase = img + (size_t)n * 3 * ih * iw + (size_t)y * iw;
        for (int x0 = 0; x0 < aw; x0 += 4) {
            const int x = x0 + kq;
            float av[6], bv[4];
#pragma unroll
            for (int a = 0; a < 6; ++a) av[a] = ld1(ar + (size_t)x * 96 + 16 * a);               // A[i = channel][k = pixel]
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const bool ok = tv[t] && (unsigned)(y + tdy[t]) < (unsigned)ih && (unsigned)(x + tdx[t]) < (unsigned)iw;
                bv[t] = ok ? ibase[toff[t] + x] : 0.f;                                           // B[k = pixel][j = tap]
            }
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                if (wv == 0) bsum[a] += av[a];
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[a][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv[t], acc[a][t], 0, 0, 0);
            }
        }
    }
    float* o = slab + (size_t)slice * 96 * 256;
#pragma unroll
    for (int a = 0; a < 6; ++a)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) o[(size_t)(16 * a + 4 * kq + r) * 256 + 64 * wv + 16 * t + col] = acc[a][t][r];
    if (wv == 0) {
#pragma unroll
        for (int a = 0; a < 6; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) bias_slab[(size_t)slice * 96 + 16 * a + col] = v;
        }
    }
}

// 64 outputs x 16 slice groups per block (as conv3x3_wgrad_reduce_kernel; one thread per output over 256 slices: 63 us)
__global__ __launch_bounds__(1024) void conv9x9_c3_wgrad_reduce_kernel(const float* __restrict__ slab,
                                                                       const float* __restrict__ bias_slab, int slices,
                                                                       float* __restrict__ r, float* __restrict__ dbias)
{
    __shared__ float part[16][64];
    const int o = threadIdx.x & 63, g = threadIdx.x >> 6;
    const int idx = blockIdx.x * 64 + o;                         // over c * 256 + j  (96 * 256 = 384 * 64)
    {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
        int sl = g;
        for (; sl + 48 < slices; sl += 64) {
            s0 += slab[(size_t)sl * 96 * 256 + idx];
            s1 += slab[(size_t)(sl + 16) * 96 * 256 + idx];
            s2 += slab[(size_t)(sl + 32) * 96 * 256 + idx];
            s3 += slab[(size_t)(sl + 48) * 96 * 256 + idx];
        }
        for (; sl < slices; sl += 16) s0 += slab[(size_t)sl * 96 * 256 + idx];
        part[g][o] = (s0 + s1) + (s2 + s3);
    }
    __syncthreads();
    const int c = idx >> 8, j = idx & 255;
    if (g == 0 && j < 243) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) s += part[k][o];
        r[c * 243 + j] = s;
    }
    if (dbias && blockIdx.x == 0) bias_slab_sum(bias_slab, slices, dbias, &part[0][0]);
}

// two workgroups per CU once there are rows enough (one: 1.0 ms for the head's gradient at 64 images)
static int wgrad9_slices(int total_rows)
{
    constexpr int cap = 512;
    return total_rows < 4 * cap ? (total_rows + 3) / 4 : cap;
}

size_t conv9x9_c3_wgrad_scratch_bytes(int n, int ah)
{
    const int slices = wgrad9_slices(n * ah);
    return ((size_t)slices * 96 * 256 + (size_t)slices * 96) * sizeof(float);
}

template <typename T>
static hipError_t conv9x9_c3_wgrad_t(const T* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                     float* dbias, void* scratch, hipStream_t s)
{
    const int total_rows = n * ah;
    const int slices = wgrad9_slices(total_rows);
    const int rps = (total_rows + slices - 1) / slices;
    float* slab = (float*)scratch;
    float* bias_slab = slab + (size_t)slices * 96 * 256;
    hipLaunchKernelGGL(conv9x9_c3_wgrad_kernel<T>, dim3((unsigned)slices), dim3(256), 0, s, act, img, total_rows, ah, aw, ih, iw, flip,
                       rps, slab, bias_slab);
    hipLaunchKernelGGL(conv9x9_c3_wgrad_reduce_kernel, dim3(96 * 256 / 64), dim3(1024), 0, s, slab, bias_slab, slices, r,
                       dbias);
    return hipGetLastError();
}

hipError_t launch_conv9x9_c3_wgrad(const float* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                   float* dbias, void* scratch, hipStream_t s)
{
    return conv9x9_c3_wgrad_t<float>(act, img, n, ah, aw, ih, iw, flip, r, dbias, scratch, s);
}

hipError_t launch_conv9x9_c3_wgrad_reduce(const float* slab, const float* bias_slab, int slices, float* r, float* dbias, hipStream_t s)
{
    hipLaunchKernelGGL(conv9x9_c3_wgrad_reduce_kernel, dim3(96 * 256 / 64), dim3(1024), 0, s, slab, bias_slab, slices, r, dbias);
    return hipGetLastError();
}

hipError_t launch_conv9x9_c3_wgrad_bf16(const void* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                        float* dbias, void* scratch, hipStream_t s)
{
    // rows of 64 or 128 pixels (the generator's stem and head): the bf16 matrix instruction (srgen_wgrad9_bf16.hip), one
    // workgroup per CU; any other width: the fp32 matrix instruction on the operands as they are
    if (aw == 64 || aw == 128) {
        const int total_rows = n * ah;
        const int cap = wgrad9_slices(total_rows);                       // what the scratch was sized for
        const int slices = cap < 256 ? cap : 256;                        // (one workgroup per CU)
        return launch_conv9x9_c3_wgrad_bf16_mfma(act, img, n, ah, aw, ih, iw, flip, r, dbias, scratch, slices, s);
    }
    return conv9x9_c3_wgrad_t<bf16_t>((const bf16_t*)act, img, n, ah, aw, ih, iw, flip, r, dbias, scratch, s);
}

// ---- ChannelAttention backward (Generator.py:15-27): y = x * s[n][c], s = sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2) ------
// pass 1: ds[n][c] = sum_hw dy * x (per image, fixed order); pass 2 (one workgroup): the two tiny FC layers backward over
// all images (weight / bias gradients summed over the batch in image order) and dpool[n][c]; pass 3:
// dx = dy * s + dpool / hw.
template <typename T>
__global__ __launch_bounds__(256) void ca_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dy, int hw,
                                                            int parts, float* __restrict__ ds_partial)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V, ROWS = 240 / CPP;
    __shared__ float red[ROWS][96];
    const int tid = threadIdx.x;
    const int per = hw / parts;
    const size_t p0 = (size_t)blockIdx.x * per;                   // blockIdx.x = image * parts + part
    if (tid < 240) {
        const int cg = tid % CPP, row = tid / CPP;
        float s[V];
#pragma unroll
        for (int e = 0; e < V; ++e) s[e] = 0.f;
        for (int p = row; p < per; p += ROWS) {
            float v[V], d[V];
            ldv(x + (p0 + p) * 96 + cg * V, v);
            ldv(dy + (p0 + p) * 96 + cg * V, d);
#pragma unroll
            for (int e = 0; e < V; ++e) s[e] += v[e] * d[e];
        }
        for (int e = 0; e < V; ++e) red[row][cg * V + e] = s[e];
    }
    __syncthreads();
    if (tid < 96) {
        float a = 0.f;
        for (int r = 0; r < ROWS; ++r) a += red[r][tid];
        ds_partial[(size_t)blockIdx.x * 96 + tid] = a;
    }
}

// one workgroup per image: the small vectors of the two Linear layers.  vec [n][216] = {pool 96, dq 96, hid 12, dh 12}
__global__ __launch_bounds__(128) void ca_bwd_fc_kernel(const float* __restrict__ pool_partial, const float* __restrict__ ds_partial,
                                                        int parts, int hw, const float* __restrict__ w1,
                                                        const float* __restrict__ b1, const float* __restrict__ w2,
                                                        const float* __restrict__ scale, float* __restrict__ dpool,
                                                        float* __restrict__ vec)
{
    __shared__ float pool[96], dq[96], dh[12];
    const int tid = threadIdx.x, img = blockIdx.x;
    float* v = vec + (size_t)img * 216;
    if (tid < 96) {
        float p = 0.f, d = 0.f;
        for (int t = 0; t < parts; ++t) {
            p += pool_partial[((size_t)img * parts + t) * 96 + tid];
            d += ds_partial[((size_t)img * parts + t) * 96 + tid];
        }
        const float sc = scale[(size_t)img * 96 + tid];
        pool[tid] = p / (float)hw;
        dq[tid] = d * sc * (1.f - sc);                            // through the sigmoid
        v[tid] = pool[tid];
        v[96 + tid] = dq[tid];
    }
    __syncthreads();
    if (tid < 12) {
        float a = b1[tid];
        for (int c = 0; c < 96; ++c) a += w1[tid * 96 + c] * pool[c];
        float g = 0.f;
        for (int c = 0; c < 96; ++c) g += w2[c * 12 + tid] * dq[c];
        dh[tid] = a > 0.f ? g : 0.f;                              // through the ReLU
        v[192 + tid] = fmaxf(a, 0.f);
        v[204 + tid] = dh[tid];
    }
    __syncthreads();
    if (tid < 96) {
        float a = 0.f;
        for (int r = 0; r < 12; ++r) a += w1[r * 96 + tid] * dh[r];
        dpool[(size_t)img * 96 + tid] = a;
    }
}

// gradients of the Linear weights: entry i of dw2[c][r] / dw1[r][c] per thread, images added in index order
__global__ __launch_bounds__(256) void ca_bwd_wgrad_kernel(const float* __restrict__ vec, int n, float* __restrict__ dw1,
                                                           float* __restrict__ db1, float* __restrict__ dw2, float* __restrict__ db2)
{
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 1152) {
        float g2 = 0.f, g1 = 0.f;
        for (int img = 0; img < n; ++img) {
            const float* v = vec + (size_t)img * 216;
            g2 = fmaf(v[96 + i / 12], v[192 + i % 12], g2);       // dq[c] * hid[r]
            g1 = fmaf(v[204 + i / 96], v[i % 96], g1);            // dh[r] * pool[c]
        }
        dw2[i] = g2;
        dw1[i] = g1;
    } else if (i < 1152 + 96) {
        const int c = i - 1152;
        float g = 0.f;
        for (int img = 0; img < n; ++img) g += vec[(size_t)img * 216 + 96 + c];
        db2[c] = g;
    } else if (i < 1152 + 108) {
        const int r = i - 1248;
        float g = 0.f;
        for (int img = 0; img < n; ++img) g += vec[(size_t)img * 216 + 204 + r];
        db1[r] = g;
    }
}

template <typename T>
__global__ __launch_bounds__(256) void ca_bwd_apply_kernel(const T* __restrict__ dy, const float* __restrict__ scale,
                                                           const float* __restrict__ dpool, int hw, T* __restrict__ dx, size_t nvec)
{
    constexpr int V = StoreVec<T>::N, CPP = 96 / V;
    const float inv = 1.f / (float)hw;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CPP) * V;
        const size_t img = i / ((size_t)hw * CPP);
        float d[V], sc[V], dp[V];
        ldv(dy + i * V, d);
        ldf<V>(scale + img * 96 + c, sc);
        ldf<V>(dpool + img * 96 + c, dp);
#pragma unroll
        for (int e = 0; e < V; ++e) d[e] = fmaf(d[e], sc[e], dp[e] * inv);
        stv(dx + i * V, d);
    }
}

template <typename T>
static hipError_t chan_attn_bwd_t(const T* x, const T* dy, const float* pool_partial, const float* scale, int n, int hw,
                                  const float* w1, const float* b1, const float* w2, const float* b2, T* dx, float* dw1, float* db1,
                                  float* dw2, float* db2, void* scratch, hipStream_t s)
{
    constexpr int parts = 16;
    float* ds_partial = (float*)scratch;                          // [n][16][96]
    float* dpool = ds_partial + (size_t)n * parts * 96;           // [n][96]
    hipLaunchKernelGGL(ca_bwd_reduce_kernel<T>, dim3((unsigned)(n * parts)), dim3(256), 0, s, x, dy, hw, parts, ds_partial);
    float* vec = dpool + (size_t)n * 96;                          // [n][216]
    hipLaunchKernelGGL(ca_bwd_fc_kernel, dim3((unsigned)n), dim3(128), 0, s, pool_partial, ds_partial, parts, hw, w1, b1, w2, scale,
                       dpool, vec);
    hipLaunchKernelGGL(ca_bwd_wgrad_kernel, dim3(5), dim3(256), 0, s, vec, n, dw1, db1, dw2, db2);
    const size_t n4 = (size_t)n * hw * (96 / StoreVec<T>::N);
    const size_t want = (n4 + 255) / 256;
    hipLaunchKernelGGL(ca_bwd_apply_kernel<T>, dim3((unsigned)(want < 16384 ? want : 16384)), dim3(256), 0, s, dy, scale, dpool, hw, dx, n4);
    return hipGetLastError();
}

hipError_t launch_chan_attn_bwd(const float* x, const float* dy, const float* pool_partial, const float* scale, int n, int hw,
                                const float* w1, const float* b1, const float* w2, const float* b2, float* dx, float* dw1, float* db1,
                                float* dw2, float* db2, void* scratch, hipStream_t s)
{
    return chan_attn_bwd_t<float>(x, dy, pool_partial, scale, n, hw, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, scratch, s);
}

hipError_t launch_chan_attn_bwd_bf16(const void* x, const void* dy, const float* pool_partial, const float* scale, int n, int hw,
                                     const float* w1, const float* b1, const float* w2, const float* b2, void* dx, float* dw1,
                                     float* db1, float* dw2, float* db2, void* scratch, hipStream_t s)
{
    return chan_attn_bwd_t<bf16_t>((const bf16_t*)x, (const bf16_t*)dy, pool_partial, scale, n, hw, w1, b1, w2, b2, (bf16_t*)dx,
                                   dw1, db1, dw2, db2, scratch, s);
}

size_t chan_attn_bwd_scratch_bytes(int n) { return ((size_t)n * 16 * 96 + (size_t)n * 96 + (size_t)n * 216) * sizeof(float); }

}  // namespace srgen
