// srgen_train_api.cpp -- C ABI of the training-step operators (include/srgen.h, "pre-train step" section):
// what autograd + torch.optim.AdamW do for the reference at /root/reference/train.py:90-97,270, for the residual
// trunk of the generator (Generator.py:30-42) and the pixel losses (train.py:18-20,93; utils.py:21-41).
#include "../../include/srgen.h"
#include "srgen_internal.h"

#include <mutex>
#include <vector>

using namespace srgen;

namespace {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(SRGEN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

// Per-device constants the Winograd conv kernel wants beside its operands: a zero page (source of out-of-image halo
// rows, and the bias of a bias-free conv) followed by 384 ones (identity PReLU slope: the raw conv output).
// Allocated once per device on first use and kept until the process ends.
constexpr size_t kOnesOff = kWinoZeroBytes;
constexpr size_t kConstBytes = kWinoZeroBytes + 384 * sizeof(float);
std::mutex g_mu;
char* g_consts[64] = {};

int train_consts(const char** out)
{
    int dev = -1;
    HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return fail(SRGEN_E_UNSUPPORTED, "device index %d outside [0, 64)", dev);
    std::lock_guard<std::mutex> lock(g_mu);
    if (!g_consts[dev]) {
        char* p = nullptr;
        HIP_TRY(hipMalloc((void**)&p, kConstBytes));
        std::vector<float> host(kConstBytes / sizeof(float), 0.f);
        for (size_t i = kOnesOff / sizeof(float); i < host.size(); ++i) host[i] = 1.f;
        hipError_t e = hipMemcpy(p, host.data(), kConstBytes, hipMemcpyHostToDevice);
        if (e != hipSuccess) {
            (void)hipFree(p);
            return fail(SRGEN_E_HIP, "hipMemcpy of the train constants failed: %s", hipGetErrorString(e));
        }
        g_consts[dev] = p;
    }
    *out = g_consts[dev];
    return SRGEN_OK;
}

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

namespace srgen {
int train_zero_page(const float** out)
{
    const char* c = nullptr;
    if (int rc = train_consts(&c)) return rc;
    *out = reinterpret_cast<const float*>(c);
    return SRGEN_OK;
}
}  // namespace srgen

extern "C" {

int srgen_wino_pack(const float* w_oihw, int cout_total, int mode, int block, float* out, void* stream)
{
    if (!w_oihw || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (cout_total != 96 && cout_total != 384) return fail(SRGEN_E_UNSUPPORTED, "cout %d: 96 or 384", cout_total);
    if (mode != 0 && mode != 1) return fail(SRGEN_E_INVALID, "mode %d: 0 forward, 1 data gradient", mode);
    if (block < 0 || block >= cout_total / 96) return fail(SRGEN_E_INVALID, "block %d outside [0, %d)", block, cout_total / 96);
    HIP_TRY(launch_wino_pack(w_oihw, out, cout_total, mode, block, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_split_pack(const float* w_oihw, int cout_total, int mode, int block, void* out, void* stream)
{
    if (!w_oihw || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (cout_total != 96 && cout_total != 384) return fail(SRGEN_E_UNSUPPORTED, "cout %d: 96 or 384", cout_total);
    if (mode != 0 && mode != 1) return fail(SRGEN_E_INVALID, "mode %d: 0 forward, 1 data gradient", mode);
    if (block < 0 || block >= cout_total / 96) return fail(SRGEN_E_INVALID, "block %d outside [0, %d)", block, cout_total / 96);
    HIP_TRY(launch_split_pack(w_oihw, out, cout_total, mode, block, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_split_pack_many(int count, const float* const* w_oihw, const int* cout_total, const int* mode, const int* block,
                          void* const* out, void* stream)
{
    if (count <= 0 || count > kSplitPackMaxJobs) return fail(SRGEN_E_INVALID, "count %d: 1 .. %d", count, kSplitPackMaxJobs);
    if (!w_oihw || !cout_total || !mode || !block || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    SplitPackJobs jobs;
    jobs.count = count;
    for (int i = 0; i < count; ++i) {
        if (!w_oihw[i] || !out[i] || !aligned16(out[i])) return fail(SRGEN_E_INVALID, "job %d: NULL or misaligned pointer", i);
        if (cout_total[i] != 96 && cout_total[i] != 384) return fail(SRGEN_E_UNSUPPORTED, "job %d: cout %d", i, cout_total[i]);
        if ((mode[i] != 0 && mode[i] != 1) || block[i] < 0 || block[i] >= cout_total[i] / 96)
            return fail(SRGEN_E_INVALID, "job %d: bad mode / block", i);
        jobs.j[i] = SplitPackJob{w_oihw[i], out[i], cout_total[i], mode[i], block[i], 0};
    }
    HIP_TRY(launch_split_pack_many(jobs, (hipStream_t)stream));
    return SRGEN_OK;
}

static int conv3x3_nhwc_any(const float* x, const void* wpack, const float* bias, const float* res, float* out, int n, int hw,
                            int cout_blocks, int shuffle, void* stream, bool split, const float* in_amax, int terms = 3);

int srgen_absmax(const float* x, size_t numel, float* out, void* stream)
{
    if (!x || !out || !numel) return fail(SRGEN_E_INVALID, "NULL or empty argument");
    if (!aligned16(out)) return fail(SRGEN_E_INVALID, "out must be 16-byte aligned");
    HIP_TRY(launch_absmax(x, numel, out, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv3x3_nhwc(const float* x, const float* wpack, const float* bias, const float* res, float* out, int n, int hw,
                       int cout_blocks, int shuffle, void* stream)
{
    return conv3x3_nhwc_any(x, wpack, bias, res, out, n, hw, cout_blocks, shuffle, stream, false, nullptr);
}

int srgen_conv3x3_nhwc_split(const float* x, const void* wpack, const float* bias, const float* res, float* out, int n, int hw,
                             int cout_blocks, int shuffle, const float* in_amax, void* stream)
{
    return conv3x3_nhwc_any(x, wpack, bias, res, out, n, hw, cout_blocks, shuffle, stream, true, in_amax);
}

int srgen_conv3x3_nhwc_f16(const float* x, const void* wpack, const float* bias, const float* res, float* out, int n, int hw,
                           int cout_blocks, int shuffle, const float* in_amax, void* stream)
{
    return conv3x3_nhwc_any(x, wpack, bias, res, out, n, hw, cout_blocks, shuffle, stream, true, in_amax, 1);
}

static int conv3x3_nhwc_any(const float* x, const void* wpack, const float* bias, const float* res, float* out, int n, int hw,
                            int cout_blocks, int shuffle, void* stream, bool split, const float* in_amax, int terms)
{
    if (!x || !wpack || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (!((hw == kLR && cout_blocks == 1 && !shuffle) || (hw == kLR && cout_blocks == 4 && shuffle && !res) ||
          (hw == kHR && cout_blocks == 1 && !shuffle && !res)))
        return fail(SRGEN_E_UNSUPPORTED, "conv3x3 shape (hw %d, cout blocks %d, shuffle %d, residual %d) has no kernel", hw,
                    cout_blocks, shuffle, res != nullptr);
    if (!aligned16(x) || !aligned16(wpack) || !aligned16(out) || !aligned16(res))
        return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    const char* consts = nullptr;
    if (int rc = train_consts(&consts)) return rc;
    Conv3x3Args a;
    a.in = x; a.out = out; a.wpack = wpack;
    a.zeros = consts;
    a.bias = bias ? bias : (const float*)consts;                          // bias-free: the zero page
    a.slope = res ? nullptr : (const float*)(consts + kOnesOff);          // identity PReLU = the raw conv output
    a.res = res; a.pool_partial = nullptr;
    if (in_amax && !aligned16(in_amax)) return fail(SRGEN_E_INVALID, "in_amax must be 16-byte aligned");
    a.in_amax = in_amax;
    a.terms = terms;
    a.N = n; a.H = hw; a.W = hw; a.cout_blocks = cout_blocks; a.shuffle = shuffle;
    HIP_TRY(split ? launch_conv3x3_f32_split(a, (hipStream_t)stream) : launch_conv3x3_f32_wino(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv3x3_wgrad_scratch_bytes(int n, int hw, size_t* bytes)
{
    if (!bytes || n <= 0 || hw <= 0) return fail(SRGEN_E_INVALID, "bad argument");
    *bytes = (conv3x3_wgrad_scratch_bytes(n, hw) + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srgen_conv3x3_wgrad(const float* x, const float* dy, int n, int hw, float* dw, float* dbias, void* scratch,
                        size_t scratch_bytes, void* stream)
{
    if (!x || !dy || !dw || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || hw <= 0 || (hw & 3)) return fail(SRGEN_E_INVALID, "bad shape (n %d, hw %d: a multiple of 4)", n, hw);
    if (scratch_bytes < conv3x3_wgrad_scratch_bytes(n, hw)) return fail(SRGEN_E_WORKSPACE, "wgrad scratch too small");
    HIP_TRY(launch_conv3x3_wgrad(x, dy, n, hw, dw, dbias, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

static int wgrad_split_any(const float* x, const float* dy, int n, int hw, const float* dy_amax, float* dw, float* dbias,
                           void* scratch, size_t scratch_bytes, void* stream, int terms);

int srgen_conv3x3_wgrad_split(const float* x, const float* dy, int n, int hw, const float* dy_amax, float* dw, float* dbias,
                              void* scratch, size_t scratch_bytes, void* stream)
{
    return wgrad_split_any(x, dy, n, hw, dy_amax, dw, dbias, scratch, scratch_bytes, stream, 3);
}

int srgen_conv3x3_wgrad_f16(const float* x, const float* dy, int n, int hw, const float* dy_amax, float* dw, float* dbias,
                            void* scratch, size_t scratch_bytes, void* stream)
{
    return wgrad_split_any(x, dy, n, hw, dy_amax, dw, dbias, scratch, scratch_bytes, stream, 1);
}

static int wgrad_split_any(const float* x, const float* dy, int n, int hw, const float* dy_amax, float* dw, float* dbias,
                           void* scratch, size_t scratch_bytes, void* stream, int terms)
{
    if (!x || !dy || !dw || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || hw <= 0 || (hw & 63)) return fail(SRGEN_E_INVALID, "bad shape (n %d, hw %d: a multiple of 64)", n, hw);
    if (scratch_bytes < conv3x3_wgrad_scratch_bytes(n, hw)) return fail(SRGEN_E_WORKSPACE, "wgrad scratch too small");
    if (!aligned16(x) || !aligned16(dy) || (dy_amax && !aligned16(dy_amax)))
        return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HIP_TRY(launch_conv3x3_wgrad_split(x, dy, n, hw, dy_amax, dw, dbias, scratch, (hipStream_t)stream, terms));
    return SRGEN_OK;
}

int srgen_bn_train_scratch_bytes(size_t* bytes)
{
    if (!bytes) return fail(SRGEN_E_INVALID, "bytes is NULL");
    const size_t a = bn_scratch_bytes(), b = bn_bwd_scratch_bytes();
    *bytes = ((a > b ? a : b) + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srgen_bn_train_fwd(const float* c, float* out, const float* gamma, const float* beta, const float* slope,
                       const float* res, float res_factor, float* batch_stats, size_t npix, void* scratch,
                       size_t scratch_bytes, void* stream)
{
    if (!c || !out || !gamma || !beta || !batch_stats || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (npix == 0) return fail(SRGEN_E_INVALID, "empty batch");
    if (scratch_bytes < bn_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "BatchNorm scratch too small");
    if (!aligned16(c) || !aligned16(out) || !aligned16(res)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HIP_TRY(launch_bn_train(const_cast<float*>(c), npix, gamma, beta, slope, res, res_factor, scratch, batch_stats,
                            (hipStream_t)stream, out));
    return SRGEN_OK;
}

int srgen_bn_train_bwd(const float* c, const float* dout, const float* batch_stats, const float* gamma, const float* beta,
                       const float* slope, float dout_scale, float* dc, float* dgamma, float* dbeta, float* dslope,
                       size_t npix, void* scratch, size_t scratch_bytes, float* dc_amax, void* stream)
{
    if (!c || !dout || !batch_stats || !gamma || !beta || !dc || !dgamma || !dbeta || !scratch)
        return fail(SRGEN_E_INVALID, "NULL argument");
    if (slope && !dslope) return fail(SRGEN_E_INVALID, "slope without dslope");
    if (npix == 0) return fail(SRGEN_E_INVALID, "empty batch");
    if (scratch_bytes < bn_bwd_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "BatchNorm scratch too small");
    if (!aligned16(c) || !aligned16(dout) || !aligned16(dc)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HIP_TRY(launch_bn_train_bwd(c, dout, batch_stats, gamma, beta, slope, dout_scale, dc, dgamma, dbeta, slope ? dslope : nullptr,
                                npix, scratch, (hipStream_t)stream, dc_amax));
    return SRGEN_OK;
}

int srgen_pixel_loss_grad(const float* sr, const float* hr, int n, int c, int hgt, int wid, float w_l1, float w_ssim,
                          float* dsr, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!sr || !hr || !dsr || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || c <= 0 || hgt <= 0 || wid <= 0) return fail(SRGEN_E_INVALID, "bad shape");
    if (scratch_bytes < (size_t)3 * n * c * hgt * wid * sizeof(float))
        return fail(SRGEN_E_WORKSPACE, "loss-gradient scratch too small: 3 * n*c*h*w floats");
    HIP_TRY(launch_pixel_loss_grad(sr, hr, n, c, hgt, wid, w_l1, w_ssim, dsr, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_adamw(float* p, const float* g, float* m, float* v, size_t numel, float lr, float beta1, float beta2, float eps,
                float weight_decay, int step, void* stream)
{
    if (numel == 0) return SRGEN_OK;
    if (!p || !g || !m || !v) return fail(SRGEN_E_INVALID, "NULL argument");
    if (step < 1) return fail(SRGEN_E_INVALID, "step %d: counts from 1", step);
    HIP_TRY(launch_adamw(p, g, m, v, numel, lr, beta1, beta2, eps, weight_decay, step, (hipStream_t)stream));
    return SRGEN_OK;
}


int srgen_prelu_scratch_bytes(size_t* bytes)
{
    if (!bytes) return fail(SRGEN_E_INVALID, "bytes is NULL");
    *bytes = (prelu_scratch_bytes() + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srgen_prelu_scalar_fwd(const float* x, const float* slope, float* y, size_t numel, void* stream)
{
    if (numel == 0) return SRGEN_OK;
    if (!x || !slope || !y) return fail(SRGEN_E_INVALID, "NULL argument");
    if ((numel & 3) || !aligned16(x) || !aligned16(y)) return fail(SRGEN_E_INVALID, "numel must be a multiple of 4, operands 16-byte aligned");
    HIP_TRY(launch_prelu_scalar_fwd(x, slope, y, numel, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_prelu_scalar_bwd(const float* x, const float* dy, const float* dy2, const float* slope, float* dx, float* dslope,
                           size_t numel, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!x || !dy || !slope || !dx || !dslope || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (numel == 0 || (numel & 3) || !aligned16(x) || !aligned16(dy) || !aligned16(dy2) || !aligned16(dx))
        return fail(SRGEN_E_INVALID, "numel must be a positive multiple of 4, operands 16-byte aligned");
    if (scratch_bytes < prelu_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "PReLU scratch too small");
    HIP_TRY(launch_prelu_scalar_bwd(x, dy, dy2, slope, dx, dslope, numel, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

static int prelu_bwd_unshuffle_any(const void* x, const void* dy, const float* slope, void* dx, float* dslope, int n, int full,
                                   void* scratch, size_t scratch_bytes, void* stream, bool bf16)
{
    if (!x || !dy || !slope || !dx || !dslope || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || full <= 0 || (full & 1)) return fail(SRGEN_E_INVALID, "n %d, full size %d (even)", n, full);
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    if (scratch_bytes < prelu_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "PReLU scratch too small");
    HIP_TRY(launch_prelu_scalar_bwd_unshuffle(x, dy, slope, dx, dslope, n, full, scratch, (hipStream_t)stream, bf16));
    return SRGEN_OK;
}

int srgen_prelu_scalar_bwd_unshuffle(const float* x, const float* dy, const float* slope, float* dx, float* dslope, int n, int full,
                                     void* scratch, size_t scratch_bytes, void* stream)
{
    return prelu_bwd_unshuffle_any(x, dy, slope, dx, dslope, n, full, scratch, scratch_bytes, stream, false);
}

int srgen_prelu_scalar_bwd_unshuffle_bf16(const void* x, const void* dy, const float* slope, void* dx, float* dslope, int n, int full,
                                          void* scratch, size_t scratch_bytes, void* stream)
{
    return prelu_bwd_unshuffle_any(x, dy, slope, dx, dslope, n, full, scratch, scratch_bytes, stream, true);
}

int srgen_conv9x9_c3_nhwc(const float* img, const float* wm, const float* bias, float* out, int n, int ih, int iw, int oh, int ow,
                          int flip, void* stream)
{
    if (!img || !wm || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return fail(SRGEN_E_INVALID, "bad shape");
    if ((ow & 15) || (((long long)n * oh * ow) & 63)) return fail(SRGEN_E_UNSUPPORTED, "output width must be a multiple of 16 and n*oh*ow of 64");
    if (!aligned16(out) || !aligned16(bias)) return fail(SRGEN_E_INVALID, "out / bias must be 16-byte aligned");
    HIP_TRY(launch_conv9x9_c3(img, wm, bias, out, n, ih, iw, oh, ow, flip ? 1 : 0, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv9x9_c3_nhwc_split(const float* img, const float* wm, const float* bias, float* out, int n, int ih, int iw, int oh,
                                int ow, int flip, void* wpack_scratch, size_t wpack_scratch_bytes, void* stream)
{
    if (!img || !wm || !out || !wpack_scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return fail(SRGEN_E_INVALID, "bad shape");
    if ((oh & 3) || (ow & 63)) return fail(SRGEN_E_UNSUPPORTED, "output must be a multiple of 4 rows x 64 columns, got %d x %d", oh, ow);
    if (wpack_scratch_bytes < kStemSplitHalfs * 2 + kSplitTailBytes) return fail(SRGEN_E_WORKSPACE, "weight-image scratch too small");
    if (!aligned16(out) || !aligned16(bias) || !aligned16(wpack_scratch)) return fail(SRGEN_E_INVALID, "out / bias / scratch must be 16-byte aligned");
    HIP_TRY(launch_stem_split_pack(wm, flip ? 1 : 0, wpack_scratch, (hipStream_t)stream));
    Stem9Args a{img, out, wpack_scratch, bias, nullptr, n, ih, iw, oh, ow, nullptr, 0};
    HIP_TRY(launch_conv9x9_c3_split(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv9x9_c3_wgrad_scratch_bytes(int n, int ah, size_t* bytes)
{
    if (!bytes || n <= 0 || ah <= 0) return fail(SRGEN_E_INVALID, "bad argument");
    *bytes = (conv9x9_c3_wgrad_scratch_bytes(n, ah) + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srgen_conv9x9_c3_wgrad(const float* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                           float* dbias, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!act || !img || !r || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || ah <= 0 || aw <= 0 || (aw & 3) || ih <= 0 || iw <= 0) return fail(SRGEN_E_INVALID, "bad shape");
    if (scratch_bytes < conv9x9_c3_wgrad_scratch_bytes(n, ah)) return fail(SRGEN_E_WORKSPACE, "9x9 wgrad scratch too small");
    HIP_TRY(launch_conv9x9_c3_wgrad(act, img, n, ah, aw, ih, iw, flip ? 1 : 0, r, dbias, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_head_nhwc(const float* v, const float* wpack, const float* bias, float* y, int n, void* stream)
{
    if (!v || !wpack || !bias || !y) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    HeadArgs a{v, y, wpack, bias, n};
    HIP_TRY(launch_head_f32(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_head_nhwc_split(const float* v, const void* wpack, const float* bias, float* y, int n, void* stream)
{
    if (!v || !wpack || !bias || !y) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (!aligned16(v) || !aligned16(wpack)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HeadArgs a{v, y, wpack, bias, n};
    HIP_TRY(launch_head_f32_split(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_chan_attn_nhwc(float* x, const float* w1, const float* b1, const float* w2, const float* b2, float* pool_partial,
                         float* scale, int n, void* stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !pool_partial || !scale) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    HIP_TRY(launch_pool_partial(x, n, kLR * kLR, 16, pool_partial, (hipStream_t)stream));
    ChanAttnArgs a{x, pool_partial, 16, w1, b1, w2, b2, scale, n, kLR * kLR};
    HIP_TRY(launch_chan_attn_f32(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_chan_attn_bwd_scratch_bytes(int n, size_t* bytes)
{
    if (!bytes || n <= 0) return fail(SRGEN_E_INVALID, "bad argument");
    *bytes = (chan_attn_bwd_scratch_bytes(n) + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srgen_chan_attn_bwd(const float* x, const float* dy, const float* pool_partial, const float* scale, int n, const float* w1,
                        const float* b1, const float* w2, const float* b2, float* dx, float* dw1, float* db1, float* dw2,
                        float* db2, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!x || !dy || !pool_partial || !scale || !w1 || !b1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !scratch)
        return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (scratch_bytes < chan_attn_bwd_scratch_bytes(n)) return fail(SRGEN_E_WORKSPACE, "attention scratch too small");
    HIP_TRY(launch_chan_attn_bwd(x, dy, pool_partial, scale, n, kLR * kLR, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, scratch,
                                 (hipStream_t)stream));
    return SRGEN_OK;
}

}  // extern "C"
