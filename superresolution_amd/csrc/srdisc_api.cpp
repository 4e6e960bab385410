// srdisc_api.cpp -- C ABI of the discriminator forward and the relativistic-average loss
// (include/srgen.h, srdisc_* and srgen_ragan_loss).
//
// Reference semantics followed here (file:line into /root/reference):
//   module tree / key names ......... Discriminator.py:22-38 (backbone.0, backbone.{2..8}.block.0,
//                                     head.0, head.2)
//   LeakyReLU(0.2) after every conv but the last ... Discriminator.py:10,24,36
//   forward order ................... Discriminator.py:49-50
//   loss ............................ Discriminator.py:53-72
#include "../../include/srgen.h"
#include "srgen_internal.h"

#include <cstring>
#include <map>
#include <new>
#include <string>
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

inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

constexpr float kSlope = 0.2f;          // nn.LeakyReLU(0.2)

struct Layer {
    std::string key;                    // state_dict prefix ("backbone.2.block.0")
    int cin, cout, stride;
    size_t w = 0, bias = 0;             // byte offsets into the device blob
};

std::vector<Layer> layer_table(const srdisc_config& c)
{
    const int f = c.base_ch;
    return {
        {"backbone.0", c.in_ch, f, 1},                 // Discriminator.py:23
        {"backbone.2.block.0", f, f, 2},               // :25
        {"backbone.3.block.0", f, 2 * f, 1},           // :26
        {"backbone.4.block.0", 2 * f, 2 * f, 2},       // :27
        {"backbone.5.block.0", 2 * f, 4 * f, 1},       // :28
        {"backbone.6.block.0", 4 * f, 4 * f, 2},       // :29
        {"backbone.7.block.0", 4 * f, 8 * f, 1},       // :30
        {"backbone.8.block.0", 8 * f, 8 * f, 2},       // :31
        {"head.0", 8 * f, 4 * f, 1},                   // :35
        {"head.2", 4 * f, 1, 1},                       // :37
    };
}

// stem: [j 7][s 4][lane 64]; lane (m = lane & 15, kq = lane >> 4) of row tile s in k-step j holds
// w[co = 16*(m >> 2) + 4*s + (m & 3)][k = 4*j + kq], k = (ky*3 + kx)*3 + ci, zero for the pad k == 27
// (same channel permutation as pack_disc_conv: a lane ends up with 16 consecutive channels)
std::vector<float> pack_disc_stem(const std::vector<float>& w)
{
    std::vector<float> p((size_t)7 * 4 * 64, 0.f);
    for (int j = 0; j < 7; ++j)
        for (int s = 0; s < 4; ++s)
            for (int lane = 0; lane < 64; ++lane) {
                const int m = lane & 15, kq = lane >> 4;
                const int co = 16 * (m >> 2) + 4 * s + (m & 3);
                const int k = 4 * j + kq;
                if (k >= 27) continue;
                const int t = k / 3, ci = k % 3;
                p[((size_t)j * 4 + s) * 64 + lane] = w[((size_t)co * 3 + ci) * 9 + t];
            }
    return p;
}

// MFMA conv: float4 units [cb][it = tap*nch + ch][s 4][lane 64].  Lane (m = lane & 15, kq = lane >> 4)
// of row tile s holds, for the four k-steps j of the chunk, w[co][ci = 16*ch + 4*kq + j][tap] with
// co = 64*cb + 16*(m >> 2) + 4*s + (m & 3): MFMA output row m = 4*kq' + r then lands on channel
// 64*cb + 16*kq' + 4*s + r, i.e. a lane's four accumulators cover 16 consecutive channels.
std::vector<float> pack_disc_conv(const std::vector<float>& w, int cin, int cout)
{
    const int nch = cin / 16, ncb = cout / 64;
    std::vector<float> p((size_t)cout * cin * 9);
    for (int cb = 0; cb < ncb; ++cb)
        for (int tap = 0; tap < 9; ++tap)
            for (int ch = 0; ch < nch; ++ch)
                for (int s = 0; s < 4; ++s)
                    for (int lane = 0; lane < 64; ++lane) {
                        const int m = lane & 15, kq = lane >> 4;
                        const int co = 64 * cb + 16 * (m >> 2) + 4 * s + (m & 3);
                        const size_t unit = ((((size_t)cb * 9 + tap) * nch + ch) * 4 + s) * 64 + lane;
                        for (int j = 0; j < 4; ++j) {
                            const int ci = 16 * ch + 4 * kq + j;
                            p[unit * 4 + j] = w[((size_t)co * cin + ci) * 9 + tap];
                        }
                    }
    return p;
}

// tail: [tap 9][ci]
std::vector<float> pack_disc_tail(const std::vector<float>& w, int cin)
{
    std::vector<float> p((size_t)9 * cin);
    for (int ci = 0; ci < cin; ++ci)
        for (int t = 0; t < 9; ++t) p[(size_t)t * cin + ci] = w[(size_t)ci * 9 + t];
    return p;
}

inline int down(int v) { return (v - 1) / 2 + 1; }      // 3x3, stride 2, pad 1

struct Plan {
    int h[10], w[10];            // output height / width of each conv
    size_t a_bytes = 0, b_bytes = 0, total = 0;   // ping-pong NHWC buffers: even layers -> a, odd -> b
};

Plan make_plan(const std::vector<Layer>& L, int n, int hgt, int wid)
{
    Plan p;
    int ch = hgt, cw = wid;
    for (int i = 0; i < 10; ++i) {
        if (L[i].stride == 2) { ch = down(ch); cw = down(cw); }
        p.h[i] = ch;
        p.w[i] = cw;
        if (i == 9) break;
        const size_t bytes = (size_t)n * ch * cw * L[i].cout * sizeof(float);
        size_t& dst = (i & 1) ? p.b_bytes : p.a_bytes;
        if (bytes > dst) dst = bytes;
    }
    p.a_bytes = align256(p.a_bytes);
    p.b_bytes = align256(p.b_bytes);
    p.total = p.a_bytes + p.b_bytes;
    return p;
}

}  // namespace

struct srdisc_handle {
    srdisc_config cfg;
    int device = 0;
    std::vector<Layer> layers;
    std::map<std::string, std::vector<float>> tensors;
    bool committed = false;
    char* blob = nullptr;
    size_t blob_bytes = 0;
    size_t zeros = 0;                   // kDiscZeroFloats zero floats in the blob
};

extern "C" {

int srdisc_create(const srdisc_config* cfg, srdisc_handle** out)
{
    if (!out) return fail(SRGEN_E_INVALID, "out is NULL");
    *out = nullptr;
    if (!cfg) return fail(SRGEN_E_INVALID, "cfg is NULL");
    if (cfg->in_ch != 3 || cfg->base_ch != 64)
        return fail(SRGEN_E_UNSUPPORTED, "Discriminator(in_ch=%d, base_ch=%d): this build has kernels for "
                    "in_ch=3, base_ch=64 only", cfg->in_ch, cfg->base_ch);
    int dev = 0;
    HIP_TRY(hipGetDevice(&dev));
    srdisc_handle* h = new (std::nothrow) srdisc_handle();
    if (!h) return fail(SRGEN_E_INVALID, "out of host memory");
    h->cfg = *cfg;
    h->device = dev;
    h->layers = layer_table(*cfg);
    *out = h;
    return SRGEN_OK;
}

int srdisc_destroy(srdisc_handle* h)
{
    if (!h) return SRGEN_OK;
    if (h->blob) (void)hipFree(h->blob);
    delete h;
    return SRGEN_OK;
}

int srdisc_set_tensor(srdisc_handle* h, const char* key, const float* host_data, size_t numel)
{
    if (!h || !key) return fail(SRGEN_E_INVALID, "handle or key is NULL");
    if (!host_data) return fail(SRGEN_E_INVALID, "data for '%s' is NULL", key);
    const std::string k(key);
    for (const Layer& l : h->layers) {
        size_t want = 0;
        if (k == l.key + ".weight") want = (size_t)l.cout * l.cin * 9;
        else if (k == l.key + ".bias") want = (size_t)l.cout;
        else continue;
        if (want != numel)
            return fail(SRGEN_E_KEY, "size mismatch for %s: expected %zu elements, got %zu", key, want, numel);
        h->tensors[k].assign(host_data, host_data + numel);
        h->committed = false;
        return SRGEN_OK;
    }
    return fail(SRGEN_E_KEY, "unexpected key in state_dict: %s", key);
}

int srdisc_commit_weights(srdisc_handle* h, void* stream)
{
    if (!h) return fail(SRGEN_E_INVALID, "handle is NULL");
    std::string missing;
    int nmiss = 0;
    for (const Layer& l : h->layers)
        for (const char* suf : {".weight", ".bias"})
            if (!h->tensors.count(l.key + suf))
                if (nmiss++ < 4) missing += (missing.empty() ? "" : ", ") + l.key + suf;
    if (nmiss) return fail(SRGEN_E_KEY, "missing %d key(s) in state_dict: %s%s", nmiss, missing.c_str(),
                           nmiss > 4 ? ", ..." : "");

    std::vector<char> bytes;
    auto put = [&](const std::vector<float>& v) {
        const size_t off = align256(bytes.size());
        bytes.resize(off + v.size() * sizeof(float));
        memcpy(bytes.data() + off, v.data(), v.size() * sizeof(float));
        return off;
    };
    h->zeros = put(std::vector<float>(kDiscZeroFloats, 0.f));
    for (size_t i = 0; i < h->layers.size(); ++i) {
        Layer& l = h->layers[i];
        const std::vector<float>& w = h->tensors.at(l.key + ".weight");
        l.w = put(i == 0 ? pack_disc_stem(w) : i == 9 ? pack_disc_tail(w, l.cin) : pack_disc_conv(w, l.cin, l.cout));
        l.bias = put(h->tensors.at(l.key + ".bias"));
    }

    hipStream_t s = (hipStream_t)stream;
    DeviceScope on_device(h->device);
    HIP_TRY(on_device.err);
    HIP_TRY(hipStreamSynchronize(s));                 // earlier forwards may still read the old blob
    if (h->blob_bytes < bytes.size()) {
        if (h->blob) HIP_TRY(hipFree(h->blob));
        h->blob = nullptr;
        h->blob_bytes = 0;
        HIP_TRY(hipMalloc((void**)&h->blob, bytes.size()));
        h->blob_bytes = bytes.size();
    }
    HIP_TRY(hipMemcpyAsync(h->blob, bytes.data(), bytes.size(), hipMemcpyHostToDevice, s));
    HIP_TRY(hipStreamSynchronize(s));                 // host staging buffer dies with this scope
    h->committed = true;
    return SRGEN_OK;
}

static int check_shape(int n, int hgt, int wid)
{
    if (n < 0) return fail(SRGEN_E_INVALID, "negative batch %d", n);
    if (hgt < 1 || wid < 1) return fail(SRGEN_E_INVALID, "bad image size %d x %d", hgt, wid);
    if ((long long)(n > 0 ? n : 1) * hgt * wid > 0x7fffffffLL)
        return fail(SRGEN_E_INVALID, "batch of %d images of %d x %d exceeds the kernels' 2^31 pixel indexing",
                    n, hgt, wid);
    return SRGEN_OK;
}

int srdisc_workspace_bytes(const srdisc_handle* h, int n, int hgt, int wid, size_t* bytes)
{
    if (!h || !bytes) return fail(SRGEN_E_INVALID, "handle or bytes is NULL");
    if (int rc = check_shape(n, hgt, wid)) return rc;
    *bytes = make_plan(h->layers, n > 0 ? n : 1, hgt, wid).total;
    return SRGEN_OK;
}

int srdisc_forward(srdisc_handle* h, const float* x, float* logits, void* workspace, size_t workspace_bytes,
                   int n, int hgt, int wid, void* const* taps, void* stream)
{
    if (!h) return fail(SRGEN_E_INVALID, "handle is NULL");
    if (!h->committed) return fail(SRGEN_E_STATE, "srdisc_forward before srdisc_commit_weights");
    if (int rc = check_shape(n, hgt, wid)) return rc;
    if (n == 0) return SRGEN_OK;
    if (!x || !logits) return fail(SRGEN_E_INVALID, "x or logits is NULL");
    if (((uintptr_t)x | (uintptr_t)logits) & 3) return fail(SRGEN_E_INVALID, "x / logits not 4-byte aligned");
    const Plan p = make_plan(h->layers, n, hgt, wid);
    if (!workspace || ((uintptr_t)workspace & 255))
        return fail(SRGEN_E_WORKSPACE, "workspace is NULL or not 256-byte aligned");
    if (workspace_bytes < p.total)
        return fail(SRGEN_E_WORKSPACE, "workspace of %zu bytes, %zu needed", workspace_bytes, p.total);

    hipStream_t s = (hipStream_t)stream;
    float* buf[2] = {reinterpret_cast<float*>(workspace),
                     reinterpret_cast<float*>(static_cast<char*>(workspace) + p.a_bytes)};
    auto dev = [&](size_t off) { return reinterpret_cast<const float*>(h->blob + off); };
    auto tap = [&](int i, const float* nhwc, int c) -> hipError_t {
        if (!taps || !taps[i]) return hipSuccess;
        return launch_nhwc_to_nchw_f32(nhwc, 0, static_cast<float*>(taps[i]), n, p.h[i] * p.w[i], c, s);
    };

    const std::vector<Layer>& L = h->layers;
    {
        DiscStemArgs a{x, buf[0], dev(L[0].w), dev(L[0].bias), n, hgt, wid, kSlope};
        HIP_TRY(launch_disc_stem(a, s));
        HIP_TRY(tap(0, buf[0], L[0].cout));
    }
    int ih = hgt, iw = wid;
    for (int i = 1; i <= 8; ++i) {
        DiscConvArgs a{buf[(i - 1) & 1], buf[i & 1], dev(L[i].w), dev(L[i].bias), n, ih, iw,
                       L[i].cin, L[i].cout, L[i].stride, kSlope, dev(h->zeros)};
        HIP_TRY(launch_disc_conv(a, s));
        HIP_TRY(tap(i, buf[i & 1], L[i].cout));
        ih = p.h[i];
        iw = p.w[i];
    }
    {
        DiscTailArgs a{buf[0], logits, dev(L[9].w), dev(L[9].bias), n, ih, iw, L[9].cin};
        HIP_TRY(launch_disc_tail(a, s));
        if (taps && taps[9])
            HIP_TRY(hipMemcpyAsync(taps[9], logits, (size_t)n * ih * iw * sizeof(float), hipMemcpyDeviceToDevice, s));
    }
    return SRGEN_OK;
}

int srgen_ragan_loss(const float* real, const float* fake, size_t n_real, size_t n_fake, float real_label,
                     float fake_label, float* out2, void* stream)
{
    if (!real || !fake || !out2) return fail(SRGEN_E_INVALID, "real, fake or out2 is NULL");
    if (n_real == 0 || n_fake == 0) return fail(SRGEN_E_INVALID, "empty logits (n_real=%zu, n_fake=%zu)", n_real, n_fake);
    HIP_TRY(launch_ragan_loss(real, fake, n_real, n_fake, real_label, fake_label, out2, (hipStream_t)stream));
    return SRGEN_OK;
}

}  // extern "C"
