/*
 * oracle/srgen_oracle.c -- TEST INFRASTRUCTURE ONLY.
 *
 * Plain-C CPU restatement of the arithmetic that the reference's SRGAN generator forward
 * dispatches to torch.nn (reference: /root/reference/Generator.py).  It exists so that
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg have an on-box truth
 * to check the HIP path against; nothing in superresolution_amd/ may import, link or call
 * it.  It is deliberately the UNFUSED chain (conv -> BN -> PReLU -> ... exactly in the
 * reference's order, no weight folding), NCHW in / NCHW out, fp32 storage with fp64
 * accumulation inside each op, so that it checks the product's fold algebra rather than
 * sharing it.
 *
 * Parity pin: the reference has no tests/golden vectors of its own,
 * so this restatement is pinned against outputs of the real reference generated in the
 * build container by tests/golden/make_golden.py (fixtures under tests/golden/).
 *
 * Build: make -C oracle   (gcc -O3 -mavx2 -mfma -fopenmp -shared -fPIC)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define EXPORT __attribute__((visibility("default")))

/* ---- nn.Conv2d(cin, cout, k, padding=k/2), stride 1 --------------------------------
 * Reference call sites: Generator.py:7 (3x3, 96->384), :34/:37 (3x3, 96->96),
 * :50 (9x9, 3->96), :62 (3x3), :69 (3x3), :71 (9x9, 96->3), :20/:22 (1x1 on 1x1 maps).
 * x: (n,cin,h,w)  w: (cout,cin,k,k)  b: (cout) or NULL  y: (n,cout,h,w).  Zero padding.
 */
EXPORT int oracle_conv2d(const float* x, const float* w, const float* b, float* y,
                         int n, int cin, int cout, int h, int wd, int k)
{
    const int pad = k / 2;
    const int hp = h + 2 * pad, wp = wd + 2 * pad;
    /* weights -> [ky][kx][ci][co] in double so the inner loop is a pure fp64 FMA over co */
    double* wt = (double*)malloc(sizeof(double) * (size_t)k * k * cin * cout);
    if (!wt) return -1;
    for (int co = 0; co < cout; ++co)
        for (int ci = 0; ci < cin; ++ci)
            for (int ky = 0; ky < k; ++ky)
                for (int kx = 0; kx < k; ++kx)
                    wt[(((size_t)ky * k + kx) * cin + ci) * cout + co] =
                        (double)w[(((size_t)co * cin + ci) * k + ky) * k + kx];
    int rc = 0;
    for (int img = 0; img < n; ++img) {
        /* zero-padded HWC copy of this image */
        double* xp = (double*)calloc((size_t)hp * wp * cin, sizeof(double));
        if (!xp) { rc = -1; break; }
        for (int ci = 0; ci < cin; ++ci)
            for (int yy = 0; yy < h; ++yy)
                for (int xx = 0; xx < wd; ++xx)
                    xp[((size_t)(yy + pad) * wp + (xx + pad)) * cin + ci] =
                        (double)x[(((size_t)img * cin + ci) * h + yy) * wd + xx];
#pragma omp parallel for schedule(static)
        for (int yy = 0; yy < h; ++yy) {
            double* acc = (double*)malloc(sizeof(double) * cout);
            for (int xx = 0; xx < wd; ++xx) {
                for (int co = 0; co < cout; ++co) acc[co] = b ? (double)b[co] : 0.0;
                for (int ky = 0; ky < k; ++ky)
                    for (int kx = 0; kx < k; ++kx) {
                        const double* xr = xp + ((size_t)(yy + ky) * wp + (xx + kx)) * cin;
                        const double* wr = wt + ((size_t)ky * k + kx) * cin * cout;
                        for (int ci = 0; ci < cin; ++ci) {
                            const double a = xr[ci];
                            const double* wc = wr + (size_t)ci * cout;
                            for (int co = 0; co < cout; ++co) acc[co] += a * wc[co];
                        }
                    }
                for (int co = 0; co < cout; ++co)
                    y[(((size_t)img * cout + co) * h + yy) * wd + xx] = (float)acc[co];
            }
            free(acc);
        }
        free(xp);
    }
    free(wt);
    return rc;
}

/* ---- nn.BatchNorm2d in eval mode (Generator.py:35,38,63) ----------------------------
 * y = (x - running_mean) / sqrt(running_var + eps) * weight + bias, eps = 1e-5.
 * In place allowed (y == x).
 */
EXPORT void oracle_batchnorm_eval(const float* x, float* y, const float* gamma, const float* beta,
                                  const float* mean, const float* var, double eps,
                                  int n, int c, int hw)
{
    for (int img = 0; img < n; ++img)
        for (int ch = 0; ch < c; ++ch) {
            const double inv = 1.0 / sqrt((double)var[ch] + eps);
            const double g = gamma[ch], bt = beta[ch], mu = mean[ch];
            const size_t o = ((size_t)img * c + ch) * hw;
            for (int i = 0; i < hw; ++i)
                y[o + i] = (float)(((double)x[o + i] - mu) * inv * g + bt);
        }
}

/* ---- nn.PReLU (Generator.py:9,36,51,70) ----------------------------------------------
 * nslope == 1: one shared slope; nslope == c: per-channel.  y = x>=0 ? x : a*x.
 */
EXPORT void oracle_prelu(const float* x, float* y, const float* slope, int nslope,
                         int n, int c, int hw)
{
    for (int img = 0; img < n; ++img)
        for (int ch = 0; ch < c; ++ch) {
            const float a = slope[nslope == 1 ? 0 : ch];
            const size_t o = ((size_t)img * c + ch) * hw;
            for (int i = 0; i < hw; ++i) {
                const float v = x[o + i];
                y[o + i] = v >= 0.0f ? v : a * v;
            }
        }
}

/* ---- y = a + s*b elementwise (ResBlock.forward Generator.py:42, long skip :78) ------- */
EXPORT void oracle_axpy(const float* a, const float* b, float s, float* y, size_t count)
{
    for (size_t i = 0; i < count; ++i) y[i] = (float)((double)a[i] + (double)s * (double)b[i]);
}

/* ---- nn.PixelShuffle(r) (Generator.py:8): pure permutation --------------------------
 * x: (n, c*r*r, h, w) -> y: (n, c, h*r, w*r);  y[n,c,h*r+i,w*r+j] = x[n, c*r*r + i*r + j, h, w]
 */
EXPORT void oracle_pixel_shuffle(const float* x, float* y, int n, int c, int h, int w, int r)
{
    const int H = h * r, W = w * r;
    for (int img = 0; img < n; ++img)
        for (int ch = 0; ch < c; ++ch)
            for (int yy = 0; yy < H; ++yy)
                for (int xx = 0; xx < W; ++xx) {
                    const int i = yy % r, j = xx % r;
                    const int cs = ch * r * r + i * r + j;
                    y[(((size_t)img * c + ch) * H + yy) * W + xx] =
                        x[(((size_t)img * c * r * r + cs) * h + yy / r) * w + xx / r];
                }
}

/* ---- ChannelAttention (Generator.py:15-27) -------------------------------------------
 * s = sigmoid(W2 * relu(W1 * mean_hw(x) + b1) + b2);  y = x * s[n,c].
 * w1: (cr, c)  b1: (cr)  w2: (c, cr)  b2: (c).  scale_out (n*c) optional.
 */
EXPORT int oracle_channel_attention(const float* x, float* y, const float* w1, const float* b1,
                                    const float* w2, const float* b2, float* scale_out,
                                    int n, int c, int cr, int hw)
{
    double* pool = (double*)malloc(sizeof(double) * c);
    double* hid = (double*)malloc(sizeof(double) * cr);
    if (!pool || !hid) { free(pool); free(hid); return -1; }
    for (int img = 0; img < n; ++img) {
        for (int ch = 0; ch < c; ++ch) {
            const size_t o = ((size_t)img * c + ch) * hw;
            double s = 0.0;
            for (int i = 0; i < hw; ++i) s += (double)x[o + i];
            pool[ch] = (double)(float)(s / hw);           /* pool output is an fp32 tensor */
        }
        for (int r = 0; r < cr; ++r) {
            double s = b1[r];
            for (int ch = 0; ch < c; ++ch) s += (double)w1[(size_t)r * c + ch] * pool[ch];
            const float f = (float)s;
            hid[r] = f > 0.0f ? (double)f : 0.0;
        }
        for (int ch = 0; ch < c; ++ch) {
            double s = b2[ch];
            for (int r = 0; r < cr; ++r) s += (double)w2[(size_t)ch * cr + r] * hid[r];
            const float z = (float)s;
            const float sc = (float)(1.0 / (1.0 + exp(-(double)z)));
            if (scale_out) scale_out[(size_t)img * c + ch] = sc;
            const size_t o = ((size_t)img * c + ch) * hw;
            for (int i = 0; i < hw; ++i) y[o + i] = x[o + i] * sc;
        }
    }
    free(pool); free(hid);
    return 0;
}

/* ---- crop out[:, :, :oh, :ow] (Generator.py:80), written contiguous ------------------ */
EXPORT void oracle_crop(const float* x, float* y, int n, int c, int h, int w, int oh, int ow)
{
    for (int img = 0; img < n; ++img)
        for (int ch = 0; ch < c; ++ch)
            for (int yy = 0; yy < oh; ++yy)
                memcpy(y + (((size_t)img * c + ch) * oh + yy) * ow,
                       x + (((size_t)img * c + ch) * h + yy) * w, sizeof(float) * ow);
}

/* ---- F.interpolate(x, (oh,ow), mode='bilinear', align_corners=False) -----------------
 * Not on the reference's forward path (Generator.py:80 crops; README.md:52 only *says*
 * bilinear) -- restated from ATen's upsample_bilinear2d (area_pixel_compute_source_index:
 * src = max(0, (dst + 0.5) * in/out - 0.5), neighbours clamped to in-1) for the optional
 * resize op BASELINE.json's north_star names.
 */
EXPORT void oracle_bilinear_resize(const float* x, float* y, int n, int c, int h, int w,
                                   int oh, int ow)
{
    const float sh = (float)h / (float)oh, sw = (float)w / (float)ow;
    for (int p = 0; p < n * c; ++p) {
        const float* xs = x + (size_t)p * h * w;
        float* ys = y + (size_t)p * oh * ow;
        for (int oy = 0; oy < oh; ++oy) {
            float fy = ((float)oy + 0.5f) * sh - 0.5f; if (fy < 0.f) fy = 0.f;
            int y0 = (int)fy; const int y1 = y0 + (y0 < h - 1 ? 1 : 0);
            const float ly = fy - (float)y0, hy = 1.f - ly;
            for (int ox = 0; ox < ow; ++ox) {
                float fx = ((float)ox + 0.5f) * sw - 0.5f; if (fx < 0.f) fx = 0.f;
                int x0 = (int)fx; const int x1 = x0 + (x0 < w - 1 ? 1 : 0);
                const float lx = fx - (float)x0, hx = 1.f - lx;
                ys[(size_t)oy * ow + ox] =
                    hy * (hx * xs[(size_t)y0 * w + x0] + lx * xs[(size_t)y0 * w + x1]) +
                    ly * (hx * xs[(size_t)y1 * w + x0] + lx * xs[(size_t)y1 * w + x1]);
            }
        }
    }
}

/* ---- metrics (utils.py:5-41, train.py:61,68) ------------------------------------------
 * psnr: 10*log10(max_val^2 / mean((sr-hr)^2)), exactly 100.0 when mse == 0 (utils.py:7-8).
 * l1:   mean(|sr-hr|)  (train.py:61 nn.L1Loss -- the value validate() calls "mse").
 * ssim: 11x11 gaussian (sigma 1.5, normalised 1-D outer product), depthwise, ZERO padding 5,
 *       c1=0.01^2, c2=0.03^2, den clamped at 1e-8, mean over the whole (n,c,h,w) map.
 * All three return the scalar over the whole batch, like the reference.
 */
EXPORT void oracle_metrics(const float* sr, const float* hr, int n, int c, int h, int w,
                           double* psnr_out, double* ssim_out, double* l1_out)
{
    const size_t total = (size_t)n * c * h * w;
    double se = 0.0, ae = 0.0;
    for (size_t i = 0; i < total; ++i) {
        const double d = (double)sr[i] - (double)hr[i];
        se += d * d; ae += fabs(d);
    }
    const double mse = se / (double)total;
    *psnr_out = (mse == 0.0) ? 100.0 : 10.0 * log10(1.0 / mse);
    *l1_out = ae / (double)total;

    float g[11]; { float s = 0.f; for (int i = 0; i < 11; ++i) { float d = (float)(i - 5);
        g[i] = expf(-(d * d) / (2.f * 1.5f * 1.5f)); s += g[i]; } for (int i = 0; i < 11; ++i) g[i] /= s; }
    double k2[11][11];
    for (int i = 0; i < 11; ++i) for (int j = 0; j < 11; ++j) k2[i][j] = (double)(g[i] * g[j]);
    const double c1 = 0.01 * 0.01, c2 = 0.03 * 0.03;
    double acc = 0.0;
#pragma omp parallel for reduction(+:acc) schedule(static)
    for (int p = 0; p < n * c; ++p) {
        const float* a = sr + (size_t)p * h * w;
        const float* b = hr + (size_t)p * h * w;
        for (int y = 0; y < h; ++y)
            for (int x = 0; x < w; ++x) {
                double mx = 0, my = 0, xx = 0, yy = 0, xy = 0;
                for (int i = 0; i < 11; ++i) {
                    const int sy = y + i - 5; if (sy < 0 || sy >= h) continue;
                    for (int j = 0; j < 11; ++j) {
                        const int sx = x + j - 5; if (sx < 0 || sx >= w) continue;
                        const double kv = k2[i][j];
                        const double va = a[(size_t)sy * w + sx], vb = b[(size_t)sy * w + sx];
                        mx += kv * va; my += kv * vb;
                        xx += kv * va * va; yy += kv * vb * vb; xy += kv * va * vb;
                    }
                }
                const double sxx = xx - mx * mx, syy = yy - my * my, sxy = xy - mx * my;
                const double num = (2 * mx * my + c1) * (2 * sxy + c2);
                double den = (mx * mx + my * my + c1) * (sxx + syy + c2);
                if (den < 1e-8) den = 1e-8;
                acc += num / den;
            }
    }
    *ssim_out = acc / (double)total;
}
