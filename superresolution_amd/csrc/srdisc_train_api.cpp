// srdisc_train_api.cpp -- C ABI of the GAN-phase training operators (include/srgen.h, "GAN step operators"):
// what autograd does for the reference at /root/reference/train.py:145-176 on the discriminator
// (Discriminator.py:17-50), the relativistic-average losses (Discriminator.py:53-72) and clip_grad_norm_ (train.py:175).
// One entry per operator; device pointers, caller-owned scratch, nothing synchronises, nothing is allocated here
// (the zero page out-of-image taps read is the per-device constant block of srgen_train_api.cpp).
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
inline bool pow2(int v) { return v > 0 && !(v & (v - 1)); }

int check_img(int n, int h, int w)
{
    if (n <= 0 || h < 1 || w < 1) return fail(SRGEN_E_INVALID, "bad shape n=%d h=%d w=%d", n, h, w);
    if ((long long)n * h * w > 0x7fffffffLL) return fail(SRGEN_E_INVALID, "%d x %d x %d exceeds the kernels' 2^31 pixel indexing", n, h, w);
    return SRGEN_OK;
}

int check_channels(int cin, int cout)
{
    if (cin % 64 || cout % 64 || !pow2(cin / 64) || cin > kDiscZeroFloats || cin <= 0 || cout <= 0)
        return fail(SRGEN_E_UNSUPPORTED, "conv %d -> %d: channels must be multiples of 64, cin / 64 a power of two <= %d",
                    cin, cout, kDiscZeroFloats / 64);
    return SRGEN_OK;
}

}  // namespace

extern "C" {

static int disc_pack_any(const float* w_oihw, int cin, int cout, int mode, void* out, void* stream, bool bf16);

int srdisc_pack(const float* w_oihw, int cin, int cout, int mode, float* out, void* stream)
{
    return disc_pack_any(w_oihw, cin, cout, mode, out, stream, false);
}

int srdisc_pack_bf16(const float* w_oihw, int cin, int cout, int mode, void* out, void* stream)
{
    if (mode == 2 || mode == 3) return fail(SRGEN_E_UNSUPPORTED, "the stem and tail layers keep fp32 products: no bf16 pack (mode %d)", mode);
    return disc_pack_any(w_oihw, cin, cout, mode, out, stream, true);
}

static int disc_pack_any(const float* w_oihw, int cin, int cout, int mode, void* out, void* stream, bool bf16)
{
    if (!w_oihw || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (mode < 0 || mode > 4)
        return fail(SRGEN_E_INVALID, "mode %d: 0 forward, 1 data gradient, 2 stem, 3 tail, 4 stride-2 data gradient", mode);
    if (mode <= 1 || mode == 4) { if (int rc = check_channels(mode ? cout : cin, mode ? cin : cout)) return rc; }
    else if (mode == 2 && (cin != 3 || cout != 64)) return fail(SRGEN_E_UNSUPPORTED, "stem pack is 3 -> 64");
    else if (mode == 3 && (cin % 4 || cout != 1)) return fail(SRGEN_E_UNSUPPORTED, "tail pack is cin -> 1, cin %% 4 == 0");
    HIP_TRY(launch_disc_pack(w_oihw, (float*)out, cin, cout, mode, (hipStream_t)stream, bf16));
    return SRGEN_OK;
}

int srdisc_stem_nhwc(const float* x_nchw, const float* wpack, const float* bias, float* out, int n, int h, int w, float slope,
                     void* stream)
{
    if (!x_nchw || !wpack || !bias || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (!aligned16(out) || !aligned16(bias)) return fail(SRGEN_E_INVALID, "out / bias must be 16-byte aligned");
    DiscStemArgs a{x_nchw, out, wpack, bias, n, h, w, slope};
    HIP_TRY(launch_disc_stem(a, (hipStream_t)stream));
    return SRGEN_OK;
}

static int disc_conv_any(const float* in, const void* wpack, const float* bias, const float* mask, float* out, int n, int h, int w,
                         int cin, int cout, int stride, float slope, void* stream, int bf16);

int srdisc_conv_nhwc(const float* in, const float* wpack, const float* bias, const float* mask, float* out, int n, int h, int w,
                     int cin, int cout, int stride, float slope, void* stream)
{
    return disc_conv_any(in, wpack, bias, mask, out, n, h, w, cin, cout, stride, slope, stream, 0);
}

int srdisc_conv_nhwc_bf16p(const float* in, const void* wpack_bf16, const float* bias, const float* mask, float* out, int n, int h,
                           int w, int cin, int cout, int stride, float slope, void* stream)
{
    return disc_conv_any(in, wpack_bf16, bias, mask, out, n, h, w, cin, cout, stride, slope, stream, 1);
}

static int disc_conv_any(const float* in, const void* wpack, const float* bias, const float* mask, float* out, int n, int h, int w,
                         int cin, int cout, int stride, float slope, void* stream, int bf16)
{
    if (!in || !wpack || !out || (!bias && !mask)) return fail(SRGEN_E_INVALID, "NULL argument (bias and mask cannot both be NULL)");
    if (int rc = check_img(n, h, w)) return rc;
    if (int rc = check_channels(cin, cout)) return rc;
    if (stride != 1 && stride != 2) return fail(SRGEN_E_INVALID, "stride %d", stride);
    if (!aligned16(in) || !aligned16(wpack) || !aligned16(out) || !aligned16(bias) || !aligned16(mask))
        return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    const float* zeros = nullptr;
    if (int rc = train_zero_page(&zeros)) return rc;
    DiscConvArgs a{in, out, (const float*)wpack, bias, n, h, w, cin, cout, stride, slope, zeros};
    a.mask = mask;
    a.bf16 = bf16;
    HIP_TRY(launch_disc_conv(a, (hipStream_t)stream));
    return SRGEN_OK;
}

static int disc_dgrad2_any(const float* dpre, const void* wpack4, const float* mask, float* out, int n, int h, int w, int cin, int cout,
                           float slope, void* stream, int bf16);

int srdisc_conv_dgrad2(const float* dpre, const float* wpack4, const float* mask, float* out, int n, int h, int w, int cin, int cout,
                       float slope, void* stream)
{
    return disc_dgrad2_any(dpre, wpack4, mask, out, n, h, w, cin, cout, slope, stream, 0);
}

int srdisc_conv_dgrad2_bf16p(const float* dpre, const void* wpack4_bf16, const float* mask, float* out, int n, int h, int w, int cin,
                             int cout, float slope, void* stream)
{
    return disc_dgrad2_any(dpre, wpack4_bf16, mask, out, n, h, w, cin, cout, slope, stream, 1);
}

static int disc_dgrad2_any(const float* dpre, const void* wpack4, const float* mask, float* out, int n, int h, int w, int cin, int cout,
                           float slope, void* stream, int bf16)
{
    if (!dpre || !wpack4 || !mask || !out) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (int rc = check_channels(cout, cin)) return rc;              // the gradient conv runs cout -> cin channels
    if (!aligned16(dpre) || !aligned16(wpack4) || !aligned16(mask) || !aligned16(out))
        return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    const float* zeros = nullptr;
    if (int rc = train_zero_page(&zeros)) return rc;
    const int oh = (h - 1) / 2 + 1, ow = (w - 1) / 2 + 1;
    DiscConvArgs a{dpre, out, (const float*)wpack4, nullptr, n, oh, ow, cout, cin, 1, slope, zeros};
    a.mask = mask;
    a.bf16 = bf16;
    HIP_TRY(launch_disc_dgrad2(a, (const float*)wpack4, h, w, (hipStream_t)stream));
    return SRGEN_OK;
}

int srdisc_tail(const float* in, const float* wpack, const float* bias, float* logits, int n, int h, int w, int cin, void* stream)
{
    if (!in || !wpack || !bias || !logits) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (cin % 4 || !aligned16(in) || !aligned16(wpack)) return fail(SRGEN_E_INVALID, "cin %% 4 and 16-byte alignment required");
    DiscTailArgs a{in, logits, wpack, bias, n, h, w, cin};
    HIP_TRY(launch_disc_tail(a, (hipStream_t)stream));
    return SRGEN_OK;
}

int srdisc_conv_wgrad_scratch_bytes(int n, int h, int cin, int cout, int stride, size_t* bytes)
{
    if (!bytes || n <= 0 || h < 1 || (stride != 1 && stride != 2)) return fail(SRGEN_E_INVALID, "bad argument");
    if (int rc = check_channels(cin, cout)) return rc;
    const size_t a = disc_wgrad_scratch_bytes(n, (h - 1) / stride + 1, cin, cout), b = disc_wgrad_bf16_scratch_bytes(cin, cout);
    *bytes = ((a > b ? a : b) + 255) & ~(size_t)255;         // serves srdisc_conv_wgrad and srdisc_conv_wgrad_bf16p
    return SRGEN_OK;
}

int srdisc_conv_wgrad(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride, float* dw, float* dbias,
                      void* scratch, size_t scratch_bytes, void* stream)
{
    if (!in || !dpre || !dw || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (int rc = check_channels(cin, cout)) return rc;
    if (stride != 1 && stride != 2) return fail(SRGEN_E_INVALID, "stride %d", stride);
    if (!aligned16(in) || !aligned16(dpre) || !aligned16(scratch)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    const size_t need = disc_wgrad_scratch_bytes(n, (h - 1) / stride + 1, cin, cout);
    if (scratch_bytes < need) return fail(SRGEN_E_WORKSPACE, "scratch of %zu bytes, %zu needed", scratch_bytes, need);
    HIP_TRY(launch_disc_wgrad(in, dpre, n, h, w, cin, cout, stride, dw, dbias, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srdisc_conv_wgrad_bf16p(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride, float* dw, float* dbias,
                            void* scratch, size_t scratch_bytes, void* stream)
{
    if (!in || !dpre || !dw || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (int rc = check_channels(cin, cout)) return rc;
    if (stride != 1 && stride != 2) return fail(SRGEN_E_INVALID, "stride %d", stride);
    if (!aligned16(in) || !aligned16(dpre) || !aligned16(scratch)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    const size_t need = disc_wgrad_bf16_scratch_bytes(cin, cout);
    if (scratch_bytes < need) return fail(SRGEN_E_WORKSPACE, "scratch of %zu bytes, %zu needed", scratch_bytes, need);
    HIP_TRY(launch_disc_wgrad_bf16(in, dpre, n, h, w, cin, cout, stride, dw, dbias, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srdisc_stem_wgrad_scratch_bytes(size_t* bytes)
{
    if (!bytes) return fail(SRGEN_E_INVALID, "bytes is NULL");
    *bytes = (disc_stem_wgrad_scratch_bytes() + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srdisc_stem_wgrad(const float* x_nchw, const float* dpre, int n, int h, int w, float* dw, float* dbias, void* scratch,
                      size_t scratch_bytes, void* stream)
{
    if (!x_nchw || !dpre || !dw || !dbias || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (!aligned16(dpre) || !aligned16(scratch)) return fail(SRGEN_E_INVALID, "operands must be 16-byte aligned");
    if (scratch_bytes < disc_stem_wgrad_scratch_bytes())
        return fail(SRGEN_E_WORKSPACE, "scratch of %zu bytes, %zu needed", scratch_bytes, disc_stem_wgrad_scratch_bytes());
    HIP_TRY(launch_disc_stem_wgrad(x_nchw, dpre, n, h, w, dw, dbias, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

int srdisc_stem_dgrad(const float* dpre, const float* w_oihw, int n, int h, int w, float* dimg_nchw, void* stream)
{
    if (!dpre || !w_oihw || !dimg_nchw) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (!aligned16(dpre)) return fail(SRGEN_E_INVALID, "dpre must be 16-byte aligned");
    HIP_TRY(launch_disc_stem_dgrad(dpre, w_oihw, n, h, w, dimg_nchw, (hipStream_t)stream));
    return SRGEN_OK;
}

int srdisc_tail_bwd(const float* dlogits, const float* w_oihw, const float* act, int n, int h, int w, int cin, float slope, float* din,
                    float* dw, float* dbias, void* stream)
{
    if (!dlogits || !w_oihw || !act || !din || (dw && !dbias)) return fail(SRGEN_E_INVALID, "NULL argument");
    if (int rc = check_img(n, h, w)) return rc;
    if (cin % 64) return fail(SRGEN_E_UNSUPPORTED, "cin %d: multiple of 64", cin);
    HIP_TRY(launch_disc_tail_bwd(dlogits, w_oihw, act, n, h, w, cin, slope, din, dw, dbias, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_ragan_loss_grad(const float* real, const float* fake, size_t n_real, size_t n_fake, float target_real, float target_fake,
                          float scale, float* dreal, float* dfake, void* stream)
{
    if (!real || !fake || !dfake) return fail(SRGEN_E_INVALID, "real, fake or dfake is NULL");
    if (n_real == 0 || n_fake == 0) return fail(SRGEN_E_INVALID, "empty logits (n_real=%zu, n_fake=%zu)", n_real, n_fake);
    HIP_TRY(launch_ragan_grad(real, fake, n_real, n_fake, target_real, target_fake, scale, dreal, dfake, (hipStream_t)stream));
    return SRGEN_OK;
}

int srgen_clip_grad_norm_scratch_bytes(size_t* bytes)
{
    if (!bytes) return fail(SRGEN_E_INVALID, "bytes is NULL");
    *bytes = (clip_scratch_bytes() + 255) & ~(size_t)255;
    return SRGEN_OK;
}

int srgen_clip_grad_norm(float* grad, size_t numel, float max_norm, float* norm_coef2, void* scratch, size_t scratch_bytes, void* stream)
{
    if (!grad || !norm_coef2 || !scratch) return fail(SRGEN_E_INVALID, "NULL argument");
    if (scratch_bytes < clip_scratch_bytes()) return fail(SRGEN_E_WORKSPACE, "scratch of %zu bytes, %zu needed", scratch_bytes, clip_scratch_bytes());
    if (((uintptr_t)scratch & 7)) return fail(SRGEN_E_INVALID, "scratch must be 8-byte aligned");
    HIP_TRY(launch_clip_grad_norm(grad, numel, max_norm, norm_coef2, scratch, (hipStream_t)stream));
    return SRGEN_OK;
}

}  // extern "C"
