/*
 * srgen.h -- C ABI of libsrgen.so: the MI355X (gfx950) SRGAN-generator forward.
 *
 * The reference (Charan6924/SuperResolution) has no FFI/plugin interface of its own: its
 * boundary for this path is the Python torch.nn.Module API of `Generator`
 * (/root/reference/Generator.py:45-80), exercised at train.py:37,65,92,149,167 (calls),
 * train.py:247,250 (construction / .to(device)) and train.py:119,214,268 (state_dict /
 * load_state_dict).  Each entry point below states which part of that Python surface it
 * replaces; superresolution_amd/generator.py binds them with ctypes.
 *
 * Conventions
 *   - every function returns 0 on success and a negative SRGEN_E_* code on failure; nothing
 *     throws or aborts across the ABI; srgen_last_error() gives the message (thread-local).
 *   - plain pointers and sizes only: device pointers are HIP device addresses on the
 *     handle's device, host pointers are ordinary host memory; `stream` is a hipStream_t
 *     passed as void* (NULL = the default stream).
 *   - srgen_forward* allocate nothing and never synchronise: input, output and workspace
 *     are the caller's (PyTorch: torch.empty on the current device); all work is enqueued
 *     on `stream`.  Device memory owned by the library is only the packed weights, which
 *     srgen_commit_weights (re)allocates.
 *   - a handle is bound to the device that was current at srgen_create; it is not safe to
 *     call concurrently on ONE handle, different handles are independent.
 *   - tensors cross the boundary in the reference's layout: NCHW, contiguous.
 */
#ifndef SRGEN_H
#define SRGEN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRGEN_ABI_VERSION 1

#if defined(__GNUC__)
#define SRGEN_API __attribute__((visibility("default")))
#else
#define SRGEN_API
#endif

enum {
    SRGEN_OK = 0,
    SRGEN_E_INVALID = -1,      /* bad argument (NULL, shape, dtype, alignment)          */
    SRGEN_E_KEY = -2,          /* unknown / missing / wrongly sized state_dict entry    */
    SRGEN_E_STATE = -3,        /* call order (forward before commit, ...)               */
    SRGEN_E_HIP = -4,          /* a HIP runtime call failed                             */
    SRGEN_E_WORKSPACE = -5,    /* workspace too small or misaligned                     */
    SRGEN_E_UNSUPPORTED = -6   /* configuration this build has no kernel for            */
};

/* arithmetic / storage type of the activations between layers (accumulation is fp32) */
enum { SRGEN_F32 = 0, SRGEN_BF16 = 1 };

typedef struct srgen_handle srgen_handle;

/* Generator(num_blocks=16, in_ch=3, base_ch=96) -- Generator.py:46.  ChannelAttention is
 * inserted after residual blocks 7 and 15 (Generator.py:57).  This build has kernels for
 * in_ch == 3, base_ch == 96, num_blocks in [1, 64]; anything else -> SRGEN_E_UNSUPPORTED. */
typedef struct srgen_config {
    int32_t num_blocks;
    int32_t in_ch;
    int32_t base_ch;
} srgen_config;

/* Replaces: Generator.__init__ + .to(device) (Generator.py:45-72, train.py:247). */
SRGEN_API int srgen_create(const srgen_config* cfg, srgen_handle** out);
SRGEN_API int srgen_destroy(srgen_handle* h);

/* Replaces: Module.load_state_dict (train.py:268).  Feed every floating-point entry of the
 * reference state_dict by its exact key ("input.0.weight", "blocks.3.block.1.running_var",
 * "blocks.8.fc.2.bias", ...), fp32, contiguous, host memory; `numel` must match the shape
 * Generator.py implies.  "*.num_batches_tracked" keys are accepted and ignored.  Then
 * srgen_commit_weights folds eval-mode BatchNorm (Generator.py:35,38,63; eps 1e-5) and the
 * 0.1 residual scale (Generator.py:42) into the conv weights in fp64, packs them for the
 * MFMA kernels (fp32 and bf16 images) and uploads them.  Missing keys -> SRGEN_E_KEY.
 * Re-callable: a later set/commit replaces the weights (ordered after prior work on
 * `stream`). */
SRGEN_API int srgen_set_tensor(srgen_handle* h, const char* key, const float* host_data, size_t numel);
SRGEN_API int srgen_commit_weights(srgen_handle* h, void* stream);

/* Bytes of caller-provided device workspace srgen_forward needs for a batch of n images.
 * 256-byte alignment required. */
SRGEN_API int srgen_workspace_bytes(const srgen_handle* h, int n, int dtype, size_t* bytes);

/* Replaces: Generator.forward in eval mode (Generator.py:74-80; train.py:37,65).
 *   x  : device, (n, 3, 64, 64)   NCHW contiguous, fp32 (SRGEN_F32) or bf16 (SRGEN_BF16)
 *   y  : device, (n, 3, 125, 125) NCHW contiguous, same dtype as x.  The reference returns
 *        the [:125,:125] crop of a 128x128 map as a view (Generator.py:80); here the crop is
 *        folded into the last store and y is dense.  Not clamped (callers clamp, train.py:38).
 */
SRGEN_API int srgen_forward(srgen_handle* h, const void* x, void* y, void* workspace,
                  size_t workspace_bytes, int n, int dtype, void* stream);

/* Replaces: Generator.forward in TRAIN mode, forward values only (Generator.py:74-80 after .train(), as at
 * train.py:92,149,167 minus autograd): every BatchNorm2d (Generator.py:35,38,63) normalises with the batch
 * mean and the biased batch variance over (n, 64, 64).  fp32.  x, y, workspace as for srgen_forward
 * (srgen_workspace_bytes with SRGEN_F32).
 *   bn_batch_stats : device, [2 * num_blocks + 1][2][96] floats, in module order (blocks.<i>.block.1,
 *                    blocks.<i>.block.4, ..., mid.1): {batch mean, biased batch variance} of each BatchNorm.
 *                    The running-statistics update (momentum 0.1, unbiased variance = biased * m / (m - 1),
 *                    m = n * 4096) is the caller's: superresolution_amd.Generator applies it to its buffers.
 * The first call after srgen_commit_weights packs and uploads the unfolded conv weights (synchronises once).
 * The backward pass of the training step is the "Pre-train step operators" section below; the Python mirror runs its
 * train mode on those (superresolution_amd/training.py, autograd.py) and keeps this entry for other generator shapes. */
SRGEN_API int srgen_forward_train(srgen_handle* h, const void* x, void* y, void* workspace,
                                  size_t workspace_bytes, int n, float* bn_batch_stats, void* stream);

/* Layer taps for parity tests: same as srgen_forward, and additionally copies intermediate
 * activations out as NCHW fp32.  taps[i] is a device pointer or NULL (skip):
 *   0 stem (n,96,64,64)   1 block0   2 block7   3 ca8   4 block16   5 ca17   6 skip (x1+x3)
 *   7 upsample (n,96,128,128)   8 output1 (n,96,128,128)
 *   9 ca8_scale (n,96)   10 ca17_scale (n,96)
 * (module outputs of Generator.py:75, blocks[0], [7], [8], [16], [17], :78, :12, output[1],
 *  and the sigmoid outputs of :19-24.) */
#define SRGEN_NUM_TAPS 11
SRGEN_API int srgen_forward_taps(srgen_handle* h, const void* x, void* y, void* workspace,
                       size_t workspace_bytes, int n, int dtype, void* const* taps,
                       void* stream);

/* Measurement aid for bench.py's roofline leg; no counterpart in the reference.  Runs one
 * forward with a HIP event pair around every kernel launch on `stream`, synchronises the
 * stream, and returns per kernel class the summed device time (ms) and the launch count.
 * Creates/destroys events: not for the serving path. */
enum {
    SRGEN_K_STEM = 0,           /* stem_*_kernel                          1 launch            */
    SRGEN_K_CONV3X3_TRUNK = 1,  /* conv3x3_*_kernel<64,false>             2*num_blocks + 1    */
    SRGEN_K_CONV3X3_UP = 2,     /* conv3x3_*_kernel<64,true>  (96->384)   1                   */
    SRGEN_K_CONV3X3_HR = 3,     /* conv3x3_*_kernel<128,false>            1                   */
    SRGEN_K_CHAN_ATTN = 4,      /* chan_attn_*_kernel                     up to 2             */
    SRGEN_K_HEAD = 5,           /* head_*_kernel                          1                   */
    SRGEN_NUM_KCLASS = 6
};
SRGEN_API int srgen_forward_profile(srgen_handle* h, const void* x, void* y, void* workspace,
                          size_t workspace_bytes, int n, int dtype, void* stream,
                          float* ms_per_class, int* launches_per_class);

/* Optional op named by BASELINE.json's north_star, NOT on the reference's forward path
 * (Generator.py:80 crops; only README.md:52 mentions bilinear): coalesced bilinear resize
 * with torch.nn.functional.interpolate(mode='bilinear', align_corners=False) semantics.
 *   x: device (planes, h, w) fp32   y: device (planes, oh, ow) fp32 */
SRGEN_API int srgen_bilinear_resize(const float* x, float* y, int planes, int h, int w, int oh, int ow,
                          void* stream);

/* Replaces: the input normalisation of JetImageDataset.__iter__ (dataset.py:52-54),
 * torch.clamp(t / hr_max, 0, 1) with hr_max = normalization_stats.pt['hr_p995'] (train.py:234),
 * moved from the DataLoader workers' CPU to the device: raw chunk -> H2D -> this -> generator.
 * x: device fp32, numel elements; y: device, fp32 (SRGEN_F32) or bf16 (SRGEN_BF16), may alias x
 * only for SRGEN_F32.  True fp32 division: bit-identical to the reference expression. */
SRGEN_API int srgen_normalize(const float* x, void* y, size_t numel, float hr_max, int out_dtype, void* stream);

/* Replaces: utils.psnr / utils.ssim / nn.L1Loss as called per batch by validate()
 * (train.py:66-68; utils.py:5-41).  sr, hr: device (n, 3, hgt, wid) NCHW fp32.
 * out3: device, 3 floats {psnr, ssim, l1} for the whole batch -- stays on the device so the
 * caller can accumulate without a .item() sync.  scratch: device, srgen_metrics_scratch_bytes(). */
SRGEN_API int srgen_metrics_scratch_bytes(int n, int c, int hgt, int wid, size_t* bytes);
SRGEN_API int srgen_metrics(const float* sr, const float* hr, int n, int c, int hgt, int wid,
                  float* out3, void* scratch, size_t scratch_bytes, void* stream);

/* No counterpart in the reference.  srgen_metrics for callers that cannot allocate device memory (code under HIP
 * graph capture, tight serving loops, harnesses that forbid hipMalloc): identical kernels and results, but the
 * per-tile partial sums live in a file-scope __device__ array inside the library (768 KiB per GPU, held until the
 * process ends) instead of a caller-provided scratch buffer.  Runs on the CURRENT device; calls no hipMalloc /
 * hipFree.  Limit: n * c * ceil(hgt/32) * ceil(wid/32) <= SRGEN_METRICS_NOALLOC_MAX_TILES (a 256 x 3 x 125 x 125
 * batch is 12,288), larger shapes -> SRGEN_E_WORKSPACE.  Every call on one GPU shares that array: calls that
 * overlap on one GPU (two streams, two threads) race -- serialise them on one stream. */
#define SRGEN_METRICS_NOALLOC_MAX_TILES 65536
SRGEN_API int srgen_metrics_noalloc(const float* sr, const float* hr, int n, int c, int hgt, int wid,
                                    float* out3, void* stream);

/* Replaces: the pixel terms of the training losses (train.py:94 and :170-172): weighted_l1(sr, hr)
 * (train.py:18-20: mean of |sr - hr| weighted 10 where hr > 0.01, 1 elsewhere) next to ssim(sr, hr).
 * Same pass, arguments and scratch as srgen_metrics; out4 = {psnr, ssim, l1, weighted_l1}.
 * Forward values only (the reference differentiates through them). */
SRGEN_API int srgen_pixel_losses(const float* sr, const float* hr, int n, int c, int hgt, int wid,
                                 float* out4, void* scratch, size_t scratch_bytes, void* stream);

/* ---- Pre-train step operators (backward pass + optimizer) -------------------------------------------------
 * SURVEY.md 8(f) "next" row 2: what autograd and torch.optim.AdamW do for the reference's pre-train step
 * (train.py:90-97: loss.backward(), optimizer.step(); train.py:270: AdamW(lr 1e-4, default weight decay 0.01)),
 * as operators on fp32 NHWC activations (n, hw, hw, 96), one per layer type of the generator (Generator.py:45-80);
 * superresolution_amd/training.py composes them into forward + backward + optimizer step.  All pointers are device pointers; nothing synchronises. */

/* Replaces: the weight side of F.conv2d(x, w, padding=1) and of its data gradient.  Forms the Winograd weight image
 * the 3x3 kernel reads (U = G g Gt, fp64 on the device, rounded once) from OIHW fp32 weights that change every step.
 *   mode 0: forward image of a (cout_total, 96, 3, 3) weight; cout_total = 96, or 384 with the PixelShuffle channel
 *           permutation of Generator.py:7-8 (out: cout_total/96 * 16*96*96 floats)
 *   mode 1: image of the data-gradient conv of cout block `block` (transposed, flipped taps; out: 16*96*96 floats) */
SRGEN_API int srgen_wino_pack(const float* w_oihw, int cout_total, int mode, int block, float* out, void* stream);

/* Replaces: F.conv2d(x, w, b, padding=1) on NHWC tensors (Generator.py:34,37,62 in train mode, where no BatchNorm can
 * be folded), and -- with a mode-1 image and bias NULL -- its data gradient.  out = conv(x) + bias [+ res].
 * hw 64 (cout_blocks 1, or 4 with shuffle = 1: the PixelShuffle(2) store) or hw 128 (cout_blocks 1).
 * The first call on a device allocates 26 KB of constants (hipMalloc) that stay until the process ends. */
SRGEN_API int srgen_conv3x3_nhwc(const float* x, const float* wpack, const float* bias, const float* res, float* out,
                                 int n, int hw, int cout_blocks, int shuffle, void* stream);

/* The same two operators for the split path (srgen_conv3x3_f32s.hip: fp32 tensors and accumulation, every product from
 * two-term f16 splits of both operands on the f16 MFMA; the default fp32 path of srgen_forward).  Replaces the same
 * reference sites: F.conv2d(x, w, b, padding=1) of Generator.py:34,37,62 and its data gradient.
 * f16 keeps 11 significant bits down to 6.1e-5 only, so both operands are scaled by powers of two into its normal range
 * before they are split (exact) and the sums scaled back: the weights by their own maximum when the image is packed, the
 * input by *in_amax.
 *   srgen_split_pack: modes and `block` as srgen_wino_pack; out: cout_total/96 (mode 0) or 1 (mode 1) images of 331,776
 *                     bytes, then a 128-byte tail (32 partial maxima of |w|); 16-byte aligned
 *   srgen_absmax:     out[SRGEN_AMAX_SLOTS] = partial maxima of |x| over numel floats (their maximum is max |x|; what
 *                     in_amax wants: one launch, nothing to zero; a NaN in x is ignored); out 16-byte aligned
 *   srgen_conv3x3_nhwc_split: arguments as srgen_conv3x3_nhwc with an image from srgen_split_pack; in_amax:
 *                     SRGEN_AMAX_SLOTS device floats whose maximum bounds max |x| within a few binades (e.g. from
 *                     srgen_absmax), or NULL for inputs of order 1
 *                     (activations): unscaled, inputs below 2^-3 then carry an absolute error of 2^-25, and |x| must
 *                     stay below the f16 maximum 65,504. */
#define SRGEN_AMAX_SLOTS 1024
SRGEN_API int srgen_split_pack(const float* w_oihw, int cout_total, int mode, int block, void* out, void* stream);
/* srgen_split_pack for `count` (<= 96) weights in two launches: what a training step does for the images of all its 3x3
 * convs (forward and data-gradient images of the live weights, train.py:91-95), where one launch pair per image is launch
 * latency only.  The arrays are HOST arrays of `count` entries: device pointers w_oihw[i] / out[i] and the arguments of
 * srgen_split_pack for job i. */
SRGEN_API int srgen_split_pack_many(int count, const float* const* w_oihw, const int* cout_total, const int* mode,
                                    const int* block, void* const* out, void* stream);
SRGEN_API int srgen_absmax(const float* x, size_t numel, float* out, void* stream);
SRGEN_API int srgen_conv3x3_nhwc_split(const float* x, const void* wpack, const float* bias, const float* res, float* out,
                                       int n, int hw, int cout_blocks, int shuffle, const float* in_amax, void* stream);
/* Mixed-precision variant of the operator above: the counterpart of F.conv2d inside the reference's
 * `with torch.autocast(device_type='cuda', dtype=torch.bfloat16)` blocks (train.py:91-92,148-149,166-167), where the conv
 * runs on operands rounded to a 16-bit type with fp32 accumulation.  Same arguments, same weight image (its high parts
 * alone are read), same power-of-two range scaling; operands are rounded to f16 (11 significant bits against bf16's 8),
 * ONE f16 MFMA per product instead of three; tensors stay fp32. */
SRGEN_API int srgen_conv3x3_nhwc_f16(const float* x, const void* wpack, const float* bias, const float* res, float* out,
                                     int n, int hw, int cout_blocks, int shuffle, const float* in_amax, void* stream);

/* Replaces: the weight / bias gradient of F.conv2d(x, w, b, padding=1): dw (96, 96, 3, 3) OIHW =
 * sum over pixels dy[n,y,x,co] * x[n,y+ky-1,x+kx-1,ci]; dbias [96] = sum dy (NULL to skip).  Fixed summation order. */
SRGEN_API int srgen_conv3x3_wgrad_scratch_bytes(int n, int hw, size_t* bytes);
SRGEN_API int srgen_conv3x3_wgrad(const float* x, const float* dy, int n, int hw, float* dw, float* dbias,
                                  void* scratch, size_t scratch_bytes, void* stream);
/* The same gradient on the split arithmetic (srgen_wgrad_f32s.hip; hw a multiple of 64; same scratch size).  dy_amax:
 * SRGEN_AMAX_SLOTS device floats whose maximum bounds max |dy| (srgen_absmax / the dc_amax of srgen_bn_train_bwd), or
 * NULL for a dy of order 1: gradients are scaled by a power of two into f16's range before they are split. */
SRGEN_API int srgen_conv3x3_wgrad_split(const float* x, const float* dy, int n, int hw, const float* dy_amax, float* dw,
                                        float* dbias, void* scratch, size_t scratch_bytes, void* stream);
/* Mixed-precision variant (see srgen_conv3x3_nhwc_f16): x and the scaled dy rounded to f16, one MFMA per product, fp32
 * accumulation and fp32 dw / dbias; the weight gradient of a conv that ran under the reference's autocast. */
SRGEN_API int srgen_conv3x3_wgrad_f16(const float* x, const float* dy, int n, int hw, const float* dy_amax, float* dw,
                                      float* dbias, void* scratch, size_t scratch_bytes, void* stream);

/* Replaces: nn.BatchNorm2d(96) in train mode (+ PReLU(96), + the residual add) forward and backward
 * (Generator.py:35-36,38,42).  c: conv output (npix, 96), kept for the backward.
 *   fwd: out = res + res_factor * PReLU_slope(bn(c))   (slope / res may be NULL; out may alias c);
 *        batch_stats [2][96] = {batch mean, biased batch variance} (the caller applies the running-stat update)
 *   bwd: given dout = d/d(PReLU(bn(c))), scaled by dout_scale (the 0.1 of the residual branch):
 *        dc, dgamma [96], dbeta [96], dslope [96] (iff slope); dc_amax: NULL, or SRGEN_AMAX_SLOTS floats that receive
 *        partial maxima of |dc| (what srgen_conv3x3_nhwc_split takes as in_amax for the data-gradient conv that reads
 *        dc next: saves the srgen_absmax pass) */
SRGEN_API int srgen_bn_train_scratch_bytes(size_t* bytes);
SRGEN_API int srgen_bn_train_fwd(const float* c, float* out, const float* gamma, const float* beta, const float* slope,
                                 const float* res, float res_factor, float* batch_stats, size_t npix, void* scratch,
                                 size_t scratch_bytes, void* stream);
SRGEN_API int srgen_bn_train_bwd(const float* c, const float* dout, const float* batch_stats, const float* gamma,
                                 const float* beta, const float* slope, float dout_scale, float* dc, float* dgamma,
                                 float* dbeta, float* dslope, size_t npix, void* scratch, size_t scratch_bytes,
                                 float* dc_amax, void* stream);

/* Replaces: nn.PReLU() with ONE slope (Generator.py:51, 9, 70) forward and backward, elementwise over numel floats
 * (a multiple of 4).  slope / dslope: one device float.  y may alias x; dx may alias dy.  dy2 (may be NULL): a second
 * gradient added to dy first -- the PReLU output has two consumers where the long skip branches off (Generator.py:75,78). */
SRGEN_API int srgen_prelu_scratch_bytes(size_t* bytes);
SRGEN_API int srgen_prelu_scalar_fwd(const float* x, const float* slope, float* y, size_t numel, void* stream);
SRGEN_API int srgen_prelu_scalar_bwd(const float* x, const float* dy, const float* dy2, const float* slope, float* dx,
                                     float* dslope, size_t numel, void* scratch, size_t scratch_bytes, void* stream);
/* Replaces: the same backward for the upsample block (Generator.py:8-9: PixelShuffle, then PReLU) together with the
 * PixelShuffle's backward: x, dy (n, full, full, 96) NHWC; dx (2, 2, n, full/2, full/2, 96): plane (i, j) holds the gradient
 * at pixels (2y + i, 2x + j), i.e. of conv-output block 2i + j -- what the four data- / weight-gradient convs of the
 * 96 -> 384 conv read, without a permute copy.  _bf16: bf16 x, dy, dx (the 16-bit step). */
SRGEN_API int srgen_prelu_scalar_bwd_unshuffle(const float* x, const float* dy, const float* slope, float* dx, float* dslope,
                                               int n, int full, void* scratch, size_t scratch_bytes, void* stream);
SRGEN_API int srgen_prelu_scalar_bwd_unshuffle_bf16(const void* x, const void* dy, const float* slope, void* dx, float* dslope,
                                                    int n, int full, void* scratch, size_t scratch_bytes, void* stream);

/* Replaces: the 9x9 convs between a 3-channel NCHW image and a 96-channel NHWC tensor, in the four forms the step needs
 * (Generator.py:50 input[0], :71 output[2]).  j = (c3, ky, kx) runs over 243 taps; a tap of pixel (y, x) reads
 * img[n][c3][y + s (ky - 4)][x + s (kx - 4)], s = +1 (flip 0) / -1 (flip 1), zero outside the ih x iw image.
 *   srgen_conv9x9_c3_nhwc : out[n][y][x][c] = bias[c] + sum_j tap_j * wm[j][c]      (wm: [243][96]; bias may be NULL)
 *       flip 0: stem forward, F.conv2d(lr, w, b, padding=4) with wm[(ci,ky,kx)][co] = w[co][ci][ky][kx];
 *       flip 1: data gradient of the head conv: img = d out (n, 3, 125, 125), out = (n, 128, 128, 96),
 *               wm[(co,ky,kx)][ci] = w[co][ci][ky][kx]
 *   srgen_conv9x9_c3_wgrad: r[c][j] = sum over pixels act[n][y][x][c] * tap_j        (r: [96][243])
 *       flip 0: weight gradient of the stem (act = d x1, img = lr): r = dw (96, 3, 9, 9); dbias [96] = sum act
 *       flip 1: weight gradient of the head (act = the head's input, img = d out): r[ci][(co,ky,kx)] = dw[co][ci][ky][kx] */
SRGEN_API int srgen_conv9x9_c3_nhwc(const float* img, const float* wm, const float* bias, float* out, int n, int ih,
                                    int iw, int oh, int ow, int flip, void* stream);
/* srgen_conv9x9_c3_nhwc on the split arithmetic (fp32 products from two-term f16 splits, srgen_stem_f32s.hip: the kernel
 * of the eval forward's stem): same arguments and result to fp32 rounding; oh a multiple of 4, ow of 64; the weight
 * image is made on the device from wm on every call into wpack_scratch (>= 165,888 + 128 bytes, 16-byte aligned). */
SRGEN_API int srgen_conv9x9_c3_nhwc_split(const float* img, const float* wm, const float* bias, float* out, int n, int ih,
                                          int iw, int oh, int ow, int flip, void* wpack_scratch, size_t wpack_scratch_bytes,
                                          void* stream);
SRGEN_API int srgen_conv9x9_c3_wgrad_scratch_bytes(int n, int ah, size_t* bytes);
SRGEN_API int srgen_conv9x9_c3_wgrad(const float* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip,
                                     float* r, float* dbias, void* scratch, size_t scratch_bytes, void* stream);

/* Replaces: F.conv2d(v, w, b, padding=4)[:, :, :125, :125] of Generator.py:71,80 on an NHWC (n, 128, 128, 96) input with
 * weights that change every step: wpack = the head kernel's weight image, float2 units [ky 9][c8 12][u 2][h 2][n 32] with
 * ci = 8 c8 + 2 (2u + h) + e, n = 3 kx + co (27 used) -- superresolution_amd/training.py forms it on the device.
 * y: (n, 3, 125, 125) NCHW. */
SRGEN_API int srgen_head_nhwc(const float* v, const float* wpack, const float* bias, float* y, int n, void* stream);
/* The same on the split path (srgen_head_f32s.hip).  wpack: f16 units of 8 [chunk 6][ky 9][hi | lo][lane 64] with lane =
 * 32 h + n, ci = 16 chunk + 8 h + e, n = 3 kx + co, of the weight times 2^k (k: srgen_internal.h split_exp of max |w|),
 * then a 128-byte tail whose first float is max |w| -- superresolution_amd/training.py forms it on the device. */
SRGEN_API int srgen_head_nhwc_split(const float* v, const void* wpack, const float* bias, float* y, int n, void* stream);

/* Replaces: ChannelAttention.forward and its backward (Generator.py:15-27) on an NHWC (n, 64, 64, 96) tensor.
 *   fwd: x is scaled IN PLACE by s = sigmoid(W2 relu(W1 mean_hw(x) + b1) + b2); pool_partial [n][16][96] (channel sums of
 *        16 row bands) and scale [n][96] are returned for the backward
 *   bwd: x = the UNSCALED input; dx = dy * s + dpool / 4096; dw1 (12, 96), db1 (12), dw2 (96, 12), db2 (96) summed over the
 *        batch in image order */
SRGEN_API int srgen_chan_attn_nhwc(float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                   float* pool_partial, float* scale, int n, void* stream);
SRGEN_API int srgen_chan_attn_bwd_scratch_bytes(int n, size_t* bytes);
SRGEN_API int srgen_chan_attn_bwd(const float* x, const float* dy, const float* pool_partial, const float* scale, int n,
                                  const float* w1, const float* b1, const float* w2, const float* b2, float* dx, float* dw1,
                                  float* db1, float* dw2, float* db2, void* scratch, size_t scratch_bytes, void* stream);

/* Replaces: autograd through the pre-train loss (train.py:93): dsr = d/dsr [w_l1 * weighted_l1(sr, hr) +
 * w_ssim * (1 - ssim(sr, hr))] (train.py:18-20; utils.py:21-41), sr / hr / dsr (n, c, hgt, wid) NCHW fp32.
 * scratch: 3 * n*c*hgt*wid floats. */
SRGEN_API int srgen_pixel_loss_grad(const float* sr, const float* hr, int n, int c, int hgt, int wid, float w_l1,
                                    float w_ssim, float* dsr, void* scratch, size_t scratch_bytes, void* stream);

/* Replaces: torch.optim.AdamW.step() (train.py:97,270) over one flat fp32 buffer: decoupled weight decay,
 * bias-corrected first / second moments; step counts from 1. */
SRGEN_API int srgen_adamw(float* p, const float* g, float* m, float* v, size_t numel, float lr, float beta1,
                          float beta2, float eps, float weight_decay, int step, void* stream);

/* ---- 16-bit pre-train step: the operators above over bf16 activation and gradient tensors ---------------------------------
 * Replaces: the same sites as the section above when the step runs as the reference runs it, inside
 * torch.autocast(device_type="cuda", dtype=torch.bfloat16) (/root/reference/train.py:91,148,166): autocast keeps the
 * conv / BatchNorm / PReLU outputs and their gradients in bf16 and multiplies bf16 operands with fp32 accumulation; so do
 * these.  Every (n, h, w, 96) NHWC tensor argument typed `void*` is bf16; parameters, batch statistics, parameter
 * gradients, the 3-channel input image and the output image (fp32 NCHW) are as in the fp32 entries; arithmetic inside a
 * kernel is fp32 and a result is rounded once (to nearest even) when stored.  Shapes, scratch sizes and error behaviour as
 * the fp32 entry of the same name.  Host mirror: superresolution_amd/training.py (`precision="bf16"`).
 *   srgen_bf16_pack_many       weight images for srgen_conv3x3_nhwc_bf16 from the live fp32 OIHW weights, all of a step's
 *                              in one launch: out[i] = 165,888 bytes per 96-channel cout block; mode / block as
 *                              srgen_split_pack
 *   srgen_conv3x3_nhwc_bf16    conv forward and data gradient (the eval forward's bf16 MFMA kernel with identity epilogue)
 *   srgen_conv3x3_wgrad_bf16   weight + bias gradient from bf16 x and dy (v_mfma_f32_32x32x16_bf16), fp32 result
 *   srgen_bn_train_fwd_bf16 / srgen_bn_train_bwd_bf16, srgen_prelu_scalar_fwd_bf16 / _bwd_bf16,
 *   srgen_chan_attn_nhwc_bf16 / srgen_chan_attn_bwd_bf16: as their fp32 namesakes
 *   srgen_conv9x9_c3_nhwc_split_bf16   fp32 3-channel image -> bf16 NHWC (stem forward, head data gradient)
 *   srgen_conv9x9_c3_wgrad_bf16        bf16 activation x fp32 image -> fp32 weight gradient.  Rows of 64 or 128 pixels (the
 *                              generator's stem and head): v_mfma_f32_32x32x16_bf16, the image rounded to bf16 where it is
 *                              multiplied (inside autocast the conv's image operand is bf16 too), fp32 accumulation; any
 *                              other width: the fp32 matrix instruction on the operands as they are
 *   srgen_head_nhwc_bf16       bf16 (n,128,128,96) -> fp32 (n,3,125,125); wpack: bf16 8-element units [ky 9][g 6][h 2][n 32],
 *                              ci = 16 g + 8 h + e, n = 3 kx + co (27 of 32 used)
 */
SRGEN_API int srgen_bf16_pack_many(int count, const float* const* w_oihw, const int* cout_total, const int* mode,
                                   const int* block, void* const* out, void* stream);
SRGEN_API int srgen_conv3x3_nhwc_bf16(const void* x, const void* wpack, const float* bias, const void* res, void* out, int n,
                                      int hw, int cout_blocks, int shuffle, void* stream);
SRGEN_API int srgen_conv3x3_wgrad_bf16(const void* x, const void* dy, int n, int hw, float* dw, float* dbias, void* scratch,
                                       size_t scratch_bytes, void* stream);
SRGEN_API int srgen_bn_train_fwd_bf16(const void* c, void* out, const float* gamma, const float* beta, const float* slope,
                                      const void* res, float res_factor, float* batch_stats, size_t npix, void* scratch,
                                      size_t scratch_bytes, void* stream);
SRGEN_API int srgen_bn_train_bwd_bf16(const void* c, const void* dout, const float* batch_stats, const float* gamma,
                                      const float* beta, const float* slope, float dout_scale, void* dc, float* dgamma,
                                      float* dbeta, float* dslope, size_t npix, void* scratch, size_t scratch_bytes,
                                      void* stream);
SRGEN_API int srgen_prelu_scalar_fwd_bf16(const void* x, const float* slope, void* y, size_t numel, void* stream);
SRGEN_API int srgen_prelu_scalar_bwd_bf16(const void* x, const void* dy, const void* dy2, const float* slope, void* dx,
                                          float* dslope, size_t numel, void* scratch, size_t scratch_bytes, void* stream);
SRGEN_API int srgen_conv9x9_c3_nhwc_split_bf16(const float* img, const float* wm, const float* bias, void* out, int n, int ih,
                                               int iw, int oh, int ow, int flip, void* wpack_scratch,
                                               size_t wpack_scratch_bytes, void* stream);
SRGEN_API int srgen_conv9x9_c3_wgrad_bf16(const void* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip,
                                          float* r, float* dbias, void* scratch, size_t scratch_bytes, void* stream);
SRGEN_API int srgen_head_nhwc_bf16(const void* v, const void* wpack, const float* bias, float* y, int n, void* stream);
SRGEN_API int srgen_chan_attn_nhwc_bf16(void* x, const float* w1, const float* b1, const float* w2, const float* b2,
                                        float* pool_partial, float* scale, int n, void* stream);
SRGEN_API int srgen_chan_attn_bwd_bf16(const void* x, const void* dy, const float* pool_partial, const float* scale, int n,
                                       const float* w1, const float* b1, const float* w2, const float* b2, void* dx,
                                       float* dw1, float* db1, float* dw2, float* db2, void* scratch, size_t scratch_bytes,
                                       void* stream);

/* ---- Discriminator + relativistic-average GAN loss (forward values) ------------------------
 * SURVEY.md 8(f) "next" row 3: the PatchGAN critic the reference's GAN phase evaluates on hr and on
 * generator output (train.py:145-176).  Same conventions as above; fp32 only.  These entry points
 * compute forward VALUES; the backward pass of the GAN step is the "GAN step operators" section below. */
typedef struct srdisc_handle srdisc_handle;

/* Discriminator(in_ch=3, base_ch=64) -- Discriminator.py:18.  This build has kernels for
 * in_ch == 3, base_ch == 64; anything else -> SRGEN_E_UNSUPPORTED. */
typedef struct srdisc_config {
    int32_t in_ch;
    int32_t base_ch;
} srdisc_config;

/* Replaces: Discriminator.__init__ + .to(device) (Discriminator.py:17-40, train.py:248,251). */
SRGEN_API int srdisc_create(const srdisc_config* cfg, srdisc_handle** out);
SRGEN_API int srdisc_destroy(srdisc_handle* h);

/* Replaces: Module.load_state_dict for the discriminator.  The 20 reference keys
 * ("backbone.0.weight", "backbone.2.block.0.bias", ..., "head.2.weight", "head.2.bias";
 * Discriminator.py:22-38), fp32, contiguous, host memory.  srdisc_commit_weights packs them for
 * the MFMA kernel and uploads them; missing keys -> SRGEN_E_KEY. */
SRGEN_API int srdisc_set_tensor(srdisc_handle* h, const char* key, const float* host_data, size_t numel);
SRGEN_API int srdisc_commit_weights(srdisc_handle* h, void* stream);

/* Bytes of caller-provided device workspace (256-byte aligned) for n images of hgt x wid. */
SRGEN_API int srdisc_workspace_bytes(const srdisc_handle* h, int n, int hgt, int wid, size_t* bytes);

/* Replaces: Discriminator.forward (Discriminator.py:49-50; train.py:152-153,168-169).
 *   x      : device, (n, 3, hgt, wid) NCHW contiguous fp32 (the reference feeds 125 x 125)
 *   logits : device, (n, 1, oh, ow) fp32, oh = s(s(s(s(hgt)))), s(v) = (v - 1) / 2 + 1: the four
 *            stride-2 convs (125 -> 63 -> 32 -> 16 -> 8)
 *   taps   : NULL, or SRDISC_NUM_TAPS device pointers (NULL entries are skipped) receiving the
 *            NCHW fp32 output of each of the ten convs after its LeakyReLU (parity tests). */
#define SRDISC_NUM_TAPS 10
SRGEN_API int srdisc_forward(srdisc_handle* h, const float* x, float* logits, void* workspace,
                             size_t workspace_bytes, int n, int hgt, int wid, void* const* taps,
                             void* stream);

/* Replaces: RelativisticAverageLoss.discriminator_loss / .generator_loss (Discriminator.py:53-72):
 * BCE-with-logits on real - mean(fake) and fake - mean(real), targets (real_label, fake_label) for
 * the discriminator loss and (0, 1) for the generator loss.  real: n_real floats, fake: n_fake
 * floats (device).  out2: device, {discriminator_loss, generator_loss}; stays on the device. */
SRGEN_API int srgen_ragan_loss(const float* real, const float* fake, size_t n_real, size_t n_fake,
                               float real_label, float fake_label, float* out2, void* stream);

/* ---- GAN step operators (discriminator backward, loss gradient, gradient clipping) ---------------------------------
 * SURVEY.md 8(f) "next" row 3, backward part: what autograd does for the reference's GAN step (train.py:145-176:
 * d_loss.backward() through both discriminator passes, g_loss.backward() through the discriminator into the
 * generator, clip_grad_norm_).  Operators on fp32 NHWC activations, one per layer type of Discriminator.py:17-50;
 * superresolution_amd/gan_training.py composes them.  The forward saves each layer's output out_i = lrelu(pre_i);
 * every backward operator takes / produces dpre_i = dout_i * (out_i > 0 ? 1 : slope).  Device pointers, caller-owned
 * scratch, nothing synchronises. */

/* Replaces: the weight side of F.conv2d in Discriminator.py:9,23,35,37 for weights that change every step.
 *   mode 0: forward pack of a (cout, cin, 3, 3) weight (cin * cout * 9 floats)
 *   mode 1: pack of its data-gradient conv (cout -> cin channels, flipped taps; same size)
 *   mode 2: stem pack (cin 3, cout 64; 1792 floats)      mode 3: tail pack (cout 1; 9 * cin floats)
 *   mode 4: the four output-parity packs of a STRIDE-2 layer's data gradient, back to back (1 + 2 + 2 + 4 taps:
 *           cin * cout * 9 floats), for srdisc_conv_dgrad2 */
SRGEN_API int srdisc_pack(const float* w_oihw, int cin, int cout, int mode, float* out, void* stream);

/* Replaces: Conv2d(3, 64, 3, padding=1) + LeakyReLU (Discriminator.py:23-24) on an NCHW image, NHWC (n, h, w, 64) out. */
SRGEN_API int srdisc_stem_nhwc(const float* x_nchw, const float* wpack, const float* bias, float* out, int n, int h,
                               int w, float slope, void* stream);

/* Replaces: DiscBlock / head.0 forward (Discriminator.py:8-11,35-36): out = lrelu(conv(in) + bias), stride 1 | 2 -- and,
 * with a mode-1 pack, mask != NULL and stride 1, the data gradient of a STRIDE-1 layer: out = conv(in) * (mask > 0 ? 1 :
 * slope), where in is dpre of that layer (a stride-2 layer: srdisc_conv_dgrad2), mask the saved activation below
 * (the shape of out), bias unused.  cin, cout: multiples of 64, cin / 64 a power of two <= 8. */
SRGEN_API int srdisc_conv_nhwc(const float* in, const float* wpack, const float* bias, const float* mask, float* out,
                               int n, int h, int w, int cin, int cout, int stride, float slope, void* stream);

/* Replaces: the data gradient of a stride-2 DiscBlock (Discriminator.py:25,27,29,31) + the LeakyReLU derivative of the
 * layer below: out (n, h, w, cin) from dpre (n, oh, ow, cout), oh = (h - 1) / 2 + 1; one launch per output parity
 * (y mod 2, x mod 2), each running only the 1, 2 or 4 taps that reach it.  wpack4: srdisc_pack mode 4; mask: the saved
 * activation (n, h, w, cin). */
SRGEN_API int srdisc_conv_dgrad2(const float* dpre, const float* wpack4, const float* mask, float* out, int n, int h, int w,
                                 int cin, int cout, float slope, void* stream);

/* Replaces: Conv2d(256, 1, 3, padding=1) (Discriminator.py:37): logits (n, 1, h, w). */
SRGEN_API int srdisc_tail(const float* in, const float* wpack, const float* bias, float* logits, int n, int h, int w,
                          int cin, void* stream);

/* Replaces: weight / bias gradient of a Cin -> Cout layer: dw (cout, cin, 3, 3) OIHW, dbias [cout] (NULL to skip);
 * in (n, h, w, cin), dpre (n, oh, ow, cout).  Fixed summation order. */
SRGEN_API int srdisc_conv_wgrad_scratch_bytes(int n, int h, int cin, int cout, int stride, size_t* bytes);
SRGEN_API int srdisc_conv_wgrad(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride,
                                float* dw, float* dbias, void* scratch, size_t scratch_bytes, void* stream);

/* Stem: weight / bias gradient (dw (64, 3, 3, 3), dbias [64]) and the gradient w.r.t. the NCHW image the discriminator
 * was fed (the path of the adversarial loss into the generator, train.py:168-174). */
SRGEN_API int srdisc_stem_wgrad_scratch_bytes(size_t* bytes);
SRGEN_API int srdisc_stem_wgrad(const float* x_nchw, const float* dpre, int n, int h, int w, float* dw, float* dbias,
                                void* scratch, size_t scratch_bytes, void* stream);
SRGEN_API int srdisc_stem_dgrad(const float* dpre, const float* w_oihw, int n, int h, int w, float* dimg_nchw, void* stream);

/* Tail backward: din (n, h, w, cin) = dpre of the layer below (act = its saved output); dw (1, cin, 3, 3) and
 * dbias [1] when dw != NULL. */
SRGEN_API int srdisc_tail_bwd(const float* dlogits, const float* w_oihw, const float* act, int n, int h, int w, int cin,
                              float slope, float* din, float* dw, float* dbias, void* stream);

/* 16-bit products for the discriminator's 64..512-channel convs (forward and data gradient): what torch.autocast(bfloat16)
 * gives them at /root/reference/train.py:150,168.  Tensors stay fp32 in memory; an activation fragment is rounded to bf16
 * when it is multiplied, the weights come from a bf16 pack (srdisc_pack_bf16: modes 0, 1, 4 of srdisc_pack, 2 bytes per
 * weight), accumulation is fp32.  The 3-channel stem and the 1-channel tail keep fp32 products.
 * Arguments as srdisc_pack / srdisc_conv_nhwc / srdisc_conv_dgrad2. */
SRGEN_API int srdisc_pack_bf16(const float* w_oihw, int cin, int cout, int mode, void* out, void* stream);
SRGEN_API int srdisc_conv_nhwc_bf16p(const float* in, const void* wpack_bf16, const float* bias, const float* mask, float* out,
                                     int n, int h, int w, int cin, int cout, int stride, float slope, void* stream);
SRGEN_API int srdisc_conv_dgrad2_bf16p(const float* dpre, const void* wpack4_bf16, const float* mask, float* out, int n, int h,
                                       int w, int cin, int cout, float slope, void* stream);
/* The weight gradient of such a layer on bf16 products (what autograd computes for these convs inside the autocast of
 * /root/reference/train.py:148-152: both operands of the gradient GEMM rounded to bf16 where they are multiplied, fp32
 * accumulation, fp32 tensors and results).  Arguments, result layout and summation discipline (fixed order) as
 * srdisc_conv_wgrad; scratch: srdisc_conv_wgrad_scratch_bytes serves both. */
SRGEN_API int srdisc_conv_wgrad_bf16p(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride,
                                      float* dw, float* dbias, void* scratch, size_t scratch_bytes, void* stream);
/* Replaces: the gradient of RelativisticAverageLoss (Discriminator.py:60-72) w.r.t. its logits, times `scale`:
 * targets (real_label, fake_label) = discriminator_loss, (0, 1) = generator_loss.  dreal may be NULL (the generator
 * step detaches the real logits, train.py:169). */
SRGEN_API int srgen_ragan_loss_grad(const float* real, const float* fake, size_t n_real, size_t n_fake, float target_real,
                                    float target_fake, float scale, float* dreal, float* dfake, void* stream);

/* Replaces: torch.nn.utils.clip_grad_norm_(parameters, max_norm) (train.py:175) over one flat gradient buffer:
 * norm_coef2 = {total L2 norm, min(1, max_norm / (norm + 1e-6))}, grad *= coefficient. */
SRGEN_API int srgen_clip_grad_norm_scratch_bytes(size_t* bytes);
SRGEN_API int srgen_clip_grad_norm(float* grad, size_t numel, float max_norm, float* norm_coef2, void* scratch,
                                   size_t scratch_bytes, void* stream);

SRGEN_API const char* srgen_last_error(void);
SRGEN_API int srgen_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* SRGEN_H */
