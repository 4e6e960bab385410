// srgen_train_bf16_api.cpp -- C ABI of the 16-bit training step (include/srgen.h, "16-bit step" section): the operators of
// srgen_train_api.cpp over bf16 activation and gradient tensors -- the storage dtype the reference's own training step has
// under torch.autocast(dtype=torch.bfloat16) (/root/reference/train.py:91,148,166).  Parameters, statistics, parameter
// gradients, the input image and the output image stay fp32; every kernel accumulates in fp32.
#include "../../include/srgen.h"
#include "srgen_internal.h"

using namespace srgen;

namespace {

#define HIP_TRY(expr)                                                                      \
    do {                                                                                   \
        hipError_t e_ = (expr);                                                            \
        if (e_ != hipSuccess)                                                              \
            return fail(SRGEN_E_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                               \
    } while (0)

inline bool aligned16(const void* p) { return ((uintptr_t)p & 15) == 0; }

}  // namespace

extern "C" {

int srgen_bf16_pack_many(int count, const float* const* w_oihw, const int* cout_total, const int* mode, const int* block,
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
    HIP_TRY(launch_bf16_pack_many(jobs, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv3x3_nhwc_bf16(const void* x, const void* wpack, const float* bias, const void* res, void* out, int n, int hw,
                            int cout_blocks, int shuffle, void* stream)
{
    if (!x || !wpack || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (!((hw == kLR && cout_blocks == 1 && !shuffle) || (hw == kLR && cout_blocks == 4 && shuffle && !res) ||
          (hw == kHR && cout_blocks == 1 && !shuffle && !res)))
        return fail(SRGEN_E_UNSUPPORTED, "conv3x3 shape (hw %d, cout blocks %d, shuffle %d, residual %d) has no kernel", hw,
                    cout_blocks, shuffle, res != nullptr);
    if (!aligned16(x) || !aligned16(wpack) || !aligned16(out) || !aligned16(res))
        return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    const float* zeros = nullptr;
    if (int rc = train_zero_page(&zeros)) return rc;
    Conv3x3Args a;
    a.in = x; a.out = out; a.wpack = wpack;
    a.bias = bias ? bias : zeros;                      // bias-free (data gradient): the zero page (>= 384 floats)
    a.slope = nullptr;                                 // the raw conv output
    a.res = res; a.pool_partial = nullptr;
    a.N = n; a.H = hw; a.W = hw; a.cout_blocks = cout_blocks; a.shuffle = shuffle;
    HIP_TRY(launch_conv3x3_bf16(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv3x3_wgrad_bf16(const void* x, const void* dy, int n, int hw, float* dw, float* dbias, void* scratch,
                             size_t scratch_bytes, void* stream)
{
    if (!x || !dy || !dw || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || hw <= 0 || (hw & 63)) return fail(SRGEN_E_INVALID, "bad shape (n %d, hw %d: a multiple of 64)", n, hw);
    if (scratch_bytes < conv3x3_wgrad_scratch_bytes(n, hw)) return fail(SRGEN_E_WORKSPACE, "wgrad scratch too small");
    if (((uintptr_t)x & 7) || ((uintptr_t)dy & 7)) return fail(SRGEN_E_INVALID, "operands must be 8-byte aligned");
    HIP_TRY(launch_conv3x3_wgrad_split(x, dy, n, hw, nullptr, dw, dbias, scratch, (hipStream_t)stream, 1, true));
    return SRGEN_OK;
}

int srgen_bn_train_fwd_bf16(const void* c, void* out, const float* gamma, const float* beta, const float* slope, const void* res,
                            float res_factor, float* batch_stats, size_t npix, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!c || !out || !gamma || !beta || !batch_stats || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (npix == 0) return fail(SRGEN_E_INVALID, "empty batch");
    if (scratch_bytes < bn_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "BatchNorm scratch too small");
    if (!aligned16(c) || !aligned16(out) || !aligned16(res)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HIP_TRY(launch_bn_train_bf16(const_cast<void*>(c), npix, gamma, beta, slope, res, res_factor, scratch, batch_stats,
                                 (hipStream_t)stream, out));
    return SRGEN_OK;
}

int srgen_bn_train_bwd_bf16(const void* c, const void* dout, const float* batch_stats, const float* gamma, const float* beta,
                            const float* slope, float dout_scale, void* dc, float* dgamma, float* dbeta, float* dslope,
                            size_t npix, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!c || !dout || !batch_stats || !gamma || !beta || !dc || !dgamma || !dbeta || !scratch)
        return fail(SRGEN_E_INVALID, "NULL argument");
    if (slope && !dslope) return fail(SRGEN_E_INVALID, "slope without dslope");
    if (npix == 0) return fail(SRGEN_E_INVALID, "empty batch");
    if (scratch_bytes < bn_bwd_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "BatchNorm scratch too small");
    if (!aligned16(c) || !aligned16(dout) || !aligned16(dc)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HIP_TRY(launch_bn_train_bwd_bf16(c, dout, batch_stats, gamma, beta, slope, dout_scale, dc, dgamma, dbeta,
                                     slope ? dslope : nullptr, npix, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_prelu_scalar_fwd_bf16(const void* x, const float* slope, void* y, size_t numel, void* stream)
{
    if (numel == 0) return SRGEN_OK;
    if (!x || !slope || !y) return fail(SRGEN_E_INVALID, "NULL argument");
    if ((numel & 7) || !aligned16(x) || !aligned16(y))
        return fail(SRGEN_E_INVALID, "numel must be a multiple of 8, operands 16-byte aligned");
    HIP_TRY(launch_prelu_scalar_fwd_bf16(x, slope, y, numel, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_prelu_scalar_bwd_bf16(const void* x, const void* dy, const void* dy2, const float* slope, void* dx, float* dslope,
                                size_t numel, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!x || !dy || !slope || !dx || !dslope || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (numel == 0 || (numel & 7) || !aligned16(x) || !aligned16(dy) || !aligned16(dy2) || !aligned16(dx))
        return fail(SRGEN_E_INVALID, "numel must be a positive multiple of 8, operands 16-byte aligned");
    if (scratch_bytes < prelu_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "PReLU scratch too small");
    HIP_TRY(launch_prelu_scalar_bwd_bf16(x, dy, dy2, slope, dx, dslope, numel, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv9x9_c3_nhwc_split_bf16(const float* img, const float* wm, const float* bias, void* out, int n, int ih, int iw, int oh,
                                     int ow, int flip, void* wpack_scratch, size_t wpack_scratch_bytes, void* stream)
{
    if (!img || !wm || !out || !wpack_scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || ih <= 0 || iw <= 0 || oh <= 0 || ow <= 0) return fail(SRGEN_E_INVALID, "bad shape");
    if ((oh & 3) || (ow & 63)) return fail(SRGEN_E_UNSUPPORTED, "output must be a multiple of 4 rows x 64 columns, got %d x %d", oh, ow);
    if (wpack_scratch_bytes < kStemSplitHalfs * 2 + kSplitTailBytes) return fail(SRGEN_E_WORKSPACE, "weight-image scratch too small");
    if (((uintptr_t)out & 7) || !aligned16(bias) || !aligned16(wpack_scratch))
        return fail(SRGEN_E_INVALID, "out must be 8-byte, bias / scratch 16-byte aligned");
    HIP_TRY(launch_stem_split_pack(wm, flip ? 1 : 0, wpack_scratch, (hipStream_t)stream));
    Stem9Args a{img, (float*)out, wpack_scratch, bias, nullptr, n, ih, iw, oh, ow, nullptr, 1};
    HIP_TRY(launch_conv9x9_c3_split(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_conv9x9_c3_wgrad_bf16(const void* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                float* dbias, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!act || !img || !r || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0 || ah <= 0 || aw <= 0 || (aw & 3) || ih <= 0 || iw <= 0) return fail(SRGEN_E_INVALID, "bad shape");
    if (scratch_bytes < conv9x9_c3_wgrad_scratch_bytes(n, ah)) return fail(SRGEN_E_WORKSPACE, "9x9 wgrad scratch too small");
    HIP_TRY(launch_conv9x9_c3_wgrad_bf16(act, img, n, ah, aw, ih, iw, flip ? 1 : 0, r, dbias, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_head_nhwc_bf16(const void* v, const void* wpack, const float* bias, float* y, int n, void* stream)
{
    if (!v || !wpack || !bias || !y) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (!aligned16(v) || !aligned16(wpack)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HeadArgs a{v, y, wpack, bias, n};
    a.y_f32 = 1;
    HIP_TRY(launch_head_bf16(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_chan_attn_nhwc_bf16(void* x, const float* w1, const float* b1, const float* w2, const float* b2, float* pool_partial,
                              float* scale, int n, void* stream)
{
    if (!x || !w1 || !b1 || !w2 || !b2 || !pool_partial || !scale) return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (!aligned16(x)) return fail(SRGEN_E_INVALID, "x must be 16-byte aligned");
    HIP_TRY(launch_pool_partial_bf16(x, n, kLR * kLR, 16, pool_partial, (hipStream_t)stream));
    ChanAttnArgs a{x, pool_partial, 16, w1, b1, w2, b2, scale, n, kLR * kLR};
    HIP_TRY(launch_chan_attn_bf16(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_chan_attn_bwd_bf16(const void* x, const void* dy, const float* pool_partial, const float* scale, int n, const float* w1,
                             const float* b1, const float* w2, const float* b2, void* dx, float* dw1, float* db1, float* dw2,
                             float* db2, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!x || !dy || !pool_partial || !scale || !w1 || !b1 || !w2 || !b2 || !dx || !dw1 || !db1 || !dw2 || !db2 || !scratch)
        return fail(SRGEN_E_INVALID, "NULL argument");
    if (n <= 0) return fail(SRGEN_E_INVALID, "batch %d", n);
    if (scratch_bytes < chan_attn_bwd_scratch_bytes(n)) return fail(SRGEN_E_WORKSPACE, "attention scratch too small");
    if (!aligned16(x) || !aligned16(dy) || !aligned16(dx)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    HIP_TRY(launch_chan_attn_bwd_bf16(x, dy, pool_partial, scale, n, kLR * kLR, w1, b1, w2, b2, dx, dw1, db1, dw2, db2, scratch,
                                      (hipStream_t)stream));
    return SRGEN_OK;
}

}  // extern "C"
