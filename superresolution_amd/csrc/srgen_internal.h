// srgen_internal.h -- kernel launch interface between srgen_api.cpp (host: fold, pack, plan)
// and the gfx950 kernel translation units.  Not part of the public ABI (include/srgen.h).
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace srgen {

// records the message srgen_last_error() returns (thread-local) and returns `code`
int fail(int code, const char* fmt, ...) __attribute__((format(printf, 2, 3)));

// Makes `device` current for a scope and puts the caller's device back afterwards (the weight uploads
// allocate on the handle's device; the caller may be working on another one).
struct DeviceScope {
    int prev = -1;
    hipError_t err;
    explicit DeviceScope(int device)
    {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != device) err = hipSetDevice(device);
        else if (err == hipSuccess) prev = -1;          // already current: nothing to restore
    }
    ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

constexpr int kC = 96;          // trunk width (base_ch)
constexpr int kLR = 64;         // low-res side
constexpr int kHR = 128;        // after PixelShuffle(2)
constexpr int kOut = 125;       // cropped output side (Generator.py:80)

// Internal activation layout: NHWC, C = 96, element = float (F32) or bf16 (BF16).

// ---- 3x3 conv, 96 -> 96*cout_blocks, fused epilogue ------------------------------------
struct Conv3x3Args {
    const void* in;        // (N,H,W,96)
    void* out;             // (N,H,W,96)  or (N,2H,2W,96) when shuffle
    const void* wpack;     // packed weights, per dtype (see pack_conv3x3_*)
    const float* bias;     // [96*cout_blocks] folded bias (fp32)
    const float* slope;    // [96*cout_blocks] PReLU slope per output channel, or nullptr
    const void* res;       // residual (N,H,W,96) added after activation, or nullptr
    float* pool_partial;   // [N][tiles_per_image][96] per-tile channel sums, or nullptr
    int N, H, W;
    int cout_blocks;       // 1, or 4 for the upsample conv
    int shuffle;           // 1: PixelShuffle(2) store (cout block b -> sub-pixel (b>>1, b&1))
    const void* zeros = nullptr;   // kWinoZeroBytes zero bytes in device memory (Winograd kernel: source
                                   // of out-of-image halo rows for global_load_lds), else unused
    float* out_amax = nullptr;        // split kernel only: [N] device floats, zero on entry: max |out| per image (atomic
                                      // max over the waves that store parts of the image) -- the next conv's in_amax
    float* zero_amax = nullptr;       // split kernel only: [N] device floats this launch sets to zero (what the launch
                                      // after the next accumulates into)
    int in_amax_per_image = 0;        // 0: in_amax is ONE bound for the whole tensor (kAmaxSlots partial maxima: training
                                      // gradients); 1: in_amax is [N], one bound per image, and every image is scaled
                                      // by its own (eval forward: an image's result does not depend on its batch)
    int terms = 3;                    // split kernel only: 3 = hi*hi + hi*lo + lo*hi (fp32 products), 1 = hi*hi alone (f16
                                      // products, fp32 accumulation: the mixed-precision training convs)
    const float* in_scale = nullptr;  // bf16 and split kernels (eval forward): [N][96] channel-attention scales the INPUT is multiplied
                                      // by while its halo tile is staged (products rounded to bf16: what the attention's
                                      // in-place pass used to store), so that pass over the tensor is not run
    const float* res_scale = nullptr; // the same for the RESIDUAL (the block behind an attention reads the
                                      // unscaled tensor twice: as conv1's input and as conv2's residual)
    const float* in_amax = nullptr;   // split kernel only: kAmaxSlots device floats whose maximum is >= max |in| (any
                                      // bound within a few binades): the input is scaled by a power of two into f16's
                                      // range before it is split.  nullptr: no scaling (activations of order 1)
};

// ---- split path (f16 two-term splits): power-of-two scaling into f16's normal range ---------------------------
// x = hi + lo is good to 2^-22 |x| only while lo is a normal f16 number, i.e. |x| >= 2^-3; below that the error is the
// f16 subnormal spacing 2^-25, absolute.  Weights (order 1e-2) and gradients (order 1e-6) are therefore multiplied by
// 2^k with amax * 2^k in [2^13, 2^14) before the split (exact in fp32) and the sums by 2^-k after: the absolute error
// floor becomes 2^-38 amax.  The weight image carries its amax in a tail of 32 partial maxima (kSplitTailBytes).
constexpr size_t kSplitImageBytes = (size_t)18 * 1152 * 16;     // one 96 -> 96 cout block
constexpr size_t kSplitTailBytes = 128;                         // 32 floats whose maximum is max |w|
constexpr int kAmaxSlots = 1024;                                // partial maxima a whole-tensor in_amax points at
__host__ __device__ inline int split_exp(float amax)
{
    const unsigned u = __builtin_bit_cast(unsigned, amax);
    const int e = (int)((u >> 23) & 255u);
    if (e == 0 || e == 255) return 0;                  // zero, subnormal, inf, nan: leave the tensor as it is
    const int k = 13 - (e - 127);
    return k < -60 ? -60 : (k > 60 ? 60 : k);
}
__host__ __device__ inline float pow2f(int k) { return __builtin_bit_cast(float, (unsigned)(k + 127) << 23); }
#ifdef __HIPCC__
// exponent for a weight image: every lane of the wave returns split_exp(max of the 32 tail floats)
__device__ inline int split_tail_exp(const void* tail, int lane)
{
    float m = lane < 32 ? ((const float*)tail)[lane] : 0.f;
#pragma unroll
    for (int o = 16; o >= 1; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    return split_exp(__shfl(m, 0));
}
#endif

// ---- 9x9 stem, NCHW 3ch in -> NHWC 96 out, bias + PReLU ---------------------------------
struct StemArgs {
    const void* x;         // (N,3,64,64) NCHW, dtype of the forward
    void* out;             // (N,64,64,96) NHWC
    const void* wpack;
    const float* bias;     // [96]
    const float* slope;    // [96]
    int N;
    float* out_amax = nullptr;   // split stem only: [N] device floats, zero on entry: max |out| per image (atomic max)
};

// ---- storage types of the training operators' activation / gradient tensors ------------------------------------
// float (the fp32 and f16-product steps) or bf16 (the 16-bit step: the reference's autocast dtype, train.py:91).  The
// elementwise and reduction kernels are templates over the storage type and compute in fp32: a thread's unit is one
// 16-byte access = four (fp32) or eight (bf16) consecutive channels.
struct bf16_t { unsigned short bits; };
// elements of one 16-byte access: a thread's unit in the elementwise kernels
template <typename T> struct StoreVec { static constexpr int N = 4; };
template <> struct StoreVec<bf16_t> { static constexpr int N = 8; };
__device__ __forceinline__ void ldv(const float* p, float (&o)[4])
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    const f4 v = *(const f4*)p;
    o[0] = v.x; o[1] = v.y; o[2] = v.z; o[3] = v.w;
}
__device__ __forceinline__ void ldv(const bf16_t* p, float (&o)[8])
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 v = *(const u4*)p;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        o[2 * k] = __builtin_bit_cast(float, v[k] << 16);
        o[2 * k + 1] = __builtin_bit_cast(float, v[k] & 0xffff0000u);
    }
}
__device__ __forceinline__ void stv(float* p, const float (&o)[4])
{
    typedef float f4 __attribute__((ext_vector_type(4)));
    *(f4*)p = f4{o[0], o[1], o[2], o[3]};
}
__device__ __forceinline__ unsigned pack2_bf16(float lo, float hi)         // round to nearest even
{
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    const b2 t = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, t);
}
__device__ __forceinline__ void stv(bf16_t* p, const float (&o)[8])
{
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    *(u4*)p = u4{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3]), pack2_bf16(o[4], o[5]), pack2_bf16(o[6], o[7])};
}
// V consecutive fp32 values (per-channel constants) with 16-byte loads
template <int V> __device__ __forceinline__ void ldf(const float* p, float (&o)[V])
{
    typedef float f4 __attribute__((ext_vector_type(4)));
#pragma unroll
    for (int k = 0; k < V / 4; ++k) {
        const f4 v = *(const f4*)(p + 4 * k);
        o[4 * k] = v.x; o[4 * k + 1] = v.y; o[4 * k + 2] = v.z; o[4 * k + 3] = v.w;
    }
}
// four consecutive channels (8-byte access for bf16): kernels whose lane layout is fixed by an MFMA tile
__device__ __forceinline__ void st4(float* p, const float (&o)[4]) { stv(p, o); }
__device__ __forceinline__ void st4(bf16_t* p, const float (&o)[4])
{
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    *(u2*)p = u2{pack2_bf16(o[0], o[1]), pack2_bf16(o[2], o[3])};
}
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16_t* p) { return __builtin_bit_cast(float, (unsigned)p->bits << 16); }

// ---- channel attention: pool finish + FC + sigmoid + in-place scale ---------------------
struct ChanAttnArgs {
    void* x;               // (N,64,64,96) NHWC, scaled in place
    const float* pool_partial;   // [N][tiles][96]
    int tiles;             // partial sums per image
    const float* w1;       // [12][96]
    const float* b1;       // [12]
    const float* w2;       // [96][12]
    const float* b2;       // [96]
    float* scale_out;      // [N][96]
    int N;
    int hw;                // pixels per image (4096)
    int scale_only = 0;    // 1 = compute scale_out (and refresh amax) and leave x alone (the convs behind apply it: Conv3x3Args::in_scale)
    float* amax = nullptr; // split path, eval forward (fp32 kernel only): [N] per-image bounds of max |x| left by the conv
                           // in front; multiplied by the image's largest channel scale, so the conv behind scales by a
                           // bound of what it really reads (a strongly damping attention would leave the old one far too high)
};

// ---- 9x9 head, NHWC 96 in (128x128) -> NCHW 3ch out, cropped to 125x125 -----------------
struct HeadArgs {
    const void* in;        // (N,128,128,96) NHWC
    void* y;               // (N,3,125,125) NCHW, dtype of the forward
    const void* wpack;
    const float* bias;     // [3]
    int N;
    const float* in_amax = nullptr;   // split head only: [N] bounds of max |in| per image (as Conv3x3Args, per image)
    int y_f32 = 0;                    // bf16 head only: 1 = y is fp32 (the 16-bit training step)
};

// fp32 kernels (srgen_kernels_f32.hip)
hipError_t launch_stem_f32(const StemArgs& a, hipStream_t s);
hipError_t launch_conv3x3_f32(const Conv3x3Args& a, hipStream_t s);
// fp32 tensors, f16 matrix instruction on two-term splits of both operands (srgen_conv3x3_f32s.hip)
hipError_t launch_conv3x3_f32_split(const Conv3x3Args& a, hipStream_t s);
// Winograd F(2x2,3x3) form of the same conv (srgen_conv3x3_f32_wino.hip); wpack = U = G g Gt packed
// as float4 units [cb][nu 4][ci chunk 6][xi 4][ci/4 4][co 96]; needs Conv3x3Args::zeros; same
// pool_partial tiling as the direct kernel
hipError_t launch_conv3x3_f32_wino(const Conv3x3Args& a, hipStream_t s);
constexpr size_t kWinoZeroBytes = 64 * 96 * 4;        // one 64-pixel row of 96 channels
hipError_t launch_chan_attn_f32(const ChanAttnArgs& a, hipStream_t s);
// the stem on the split path (srgen_stem_f32s.hip); wpack = f16 units [ky 9][s 3][nt 3][hi | lo][lane 64] + amax tail
hipError_t launch_stem_f32_split(const StemArgs& a, hipStream_t s);
constexpr size_t kStemSplitHalfs = (size_t)9 * 3 * 3 * 2 * 64 * 8;
// the same kernel for any 3-channel NCHW image (ih x iw, zero outside) and an NHWC output of oh x ow (multiples of 4 / 64);
// bias, slope, out_amax may be nullptr
struct Stem9Args {
    const float* x;
    float* out;
    const void* wpack;
    const float* bias;
    const float* slope;
    int N, ih, iw, oh, ow;
    float* out_amax;
    int out_bf16 = 0;      // 1: out is a bf16 tensor
};
hipError_t launch_conv9x9_c3_split(const Stem9Args& a, hipStream_t s);
// weight image (kStemSplitHalfs halfs + kSplitTailBytes) from the tap matrix wm[243][96] on the device
hipError_t launch_stem_split_pack(const float* wm, int flip, void* out, hipStream_t s);
hipError_t launch_head_f32(const HeadArgs& a, hipStream_t s);
// the head on the split path (srgen_head_f32s.hip); wpack = f16 units [chunk 6][ky 9][hi | lo][lane 64]
hipError_t launch_head_f32_split(const HeadArgs& a, hipStream_t s);
hipError_t launch_head_f32_split_stream(const HeadArgs& a, hipStream_t s);   // srgen_head_f32s_stream.hip
// bf16 kernels (srgen_kernels_bf16.hip)
hipError_t launch_stem_bf16(const StemArgs& a, hipStream_t s);
hipError_t launch_conv3x3_bf16(const Conv3x3Args& a, hipStream_t s);
hipError_t launch_chan_attn_bf16(const ChanAttnArgs& a, hipStream_t s);
hipError_t launch_head_bf16(const HeadArgs& a, hipStream_t s);
hipError_t launch_head_bf16_stream(const HeadArgs& a, hipStream_t s);      // srgen_head_bf16_stream.hip

// misc (srgen_kernels_misc.hip)
hipError_t launch_nhwc_to_nchw_f32(const void* in, int in_is_bf16, float* out, int N, int HW,
                                   int C, hipStream_t s);
hipError_t launch_bilinear(const float* x, float* y, int planes, int h, int w, int oh, int ow,
                           hipStream_t s);
hipError_t launch_normalize(const float* x, void* y, size_t numel, float hr_max, int out_bf16, hipStream_t s);
// nout = 3: {psnr, ssim, l1}; nout = 4: additionally weighted_l1 (train.py:18-20)
size_t metrics_scratch_bytes(int n, int c, int h, int w, int nout);
hipError_t launch_metrics(const float* sr, const float* hr, int n, int c, int h, int w,
                          float* out, void* scratch, int nout, hipStream_t s);

// srgen_metrics_noalloc: per-tile partial sums in a file-scope __device__ array of the library (srgen_kernels_misc.hip)
constexpr int kMetricsStaticMaxTiles = 65536;        // 32x32 tiles over all (n, c) planes: 768 KiB of partials
int metrics_tiles(int n, int c, int h, int w);
// device address of that array on the CURRENT device (hipGetSymbolAddress once per device); allocates nothing
hipError_t metrics_static_scratch(float** ptr);

// ---- discriminator (Discriminator.py:17-50): NHWC fp32 activations (srdisc_kernels.hip) ---------
struct DiscStemArgs {          // conv 3 -> 64, stride 1, pad 1, + LeakyReLU
    const float* x;            // (N,3,H,W) NCHW
    float* out;                // (N,H,W,64) NHWC
    const float* w;            // [j 7][s 4][lane 64]: w[co = 16*(m>>2) + 4*s + (m&3)][k = 4*j + (lane>>4)],
                               // m = lane & 15, k = (ky*3 + kx)*3 + ci, k == 27 zero
    const float* bias;         // [64]
    int N, H, W;
    float slope;
};
struct DiscConvArgs {          // conv Cin -> Cout (Cin % 16 == 0, Cout % 64 == 0), stride 1|2, pad 1
    const float* in;           // (N,H,W,Cin)
    float* out;                // (N,OH,OW,Cout), OH = (H-1)/stride + 1
    const float* wpack;        // float4 units [cout/64][tap 9][cin/16][s 4][lane 64], see pack_disc_conv
    const float* bias;         // [Cout]
    int N, H, W, Cin, Cout, stride;
    float slope;               // LeakyReLU slope
    const float* zeros;        // kDiscZeroFloats zeros in device memory: what out-of-image taps read
    const float* mask = nullptr;   // data-gradient use: (N,OH,OW,Cout) saved activation; out = acc * (mask > 0 ? 1 : slope)
    // stride-2 data gradient, one output parity class (disc_conv_kernel<.., PAR = true>)
    unsigned long long taps = 0;   // 4-bit tap ids (ky*3 + kx), real taps first, 15 = padding
    int ntaps = 0, nreal = 0, py = 0, px = 0, FH = 0, FW = 0;
    int bf16 = 0;                  // 1: wpack is the bf16 pack of launch_disc_pack(.., bf16 = true) (the layout of
                                   // srdisc_conv_bf16.hip) and the products are bf16 x bf16 with fp32 accumulation; tensors stay fp32
};
constexpr int kDiscZeroFloats = 512;
struct DiscTailArgs {          // conv Cin -> 1, stride 1, pad 1, no activation
    const float* in;           // (N,H,W,Cin), Cin % 4 == 0
    float* y;                  // (N,1,H,W)
    const float* w;            // [tap 9][Cin]
    const float* bias;         // [1]
    int N, H, W, Cin;
};
hipError_t launch_disc_stem(const DiscStemArgs& a, hipStream_t s);
hipError_t launch_disc_conv(const DiscConvArgs& a, hipStream_t s);
hipError_t launch_disc_dgrad2(const DiscConvArgs& a0, const float* wpack4, int fh, int fw, hipStream_t s);
// the a.bf16 convs (srdisc_conv_bf16.hip): OH x OW = the layer's output grid; the stride-2 data gradient's four parity classes in one launch
hipError_t launch_disc_conv_bf16(const DiscConvArgs& a, int OH, int OW, hipStream_t s);
hipError_t launch_disc_dgrad2_bf16(const DiscConvArgs& a0, const void* wpack4, int fh, int fw, hipStream_t s);
hipError_t launch_disc_tail(const DiscTailArgs& a, hipStream_t s);
hipError_t launch_ragan_loss(const float* real, const float* fake, size_t n_real, size_t n_fake,
                             float real_label, float fake_label, float* out2, hipStream_t s);

// ---- GAN-phase backward operators (srdisc_train_kernels.hip) ---------------------------------------------------------
// packs from live OIHW weights on the device: mode 0 forward conv, 1 data-gradient conv, 2 stem, 3 tail
hipError_t launch_disc_pack(const float* w, float* out, int cin, int cout, int mode, hipStream_t s, bool bf16 = false);
size_t disc_wgrad_scratch_bytes(int n, int oh, int cin, int cout);
hipError_t launch_disc_wgrad_reduce(const float* slab, const float* bias_slab, int slices, int cin, int cout, float* dw, float* dbias,
                                    hipStream_t s);
// bf16 products (srdisc_wgrad_bf16.hip): same operands and results, its own slice count
size_t disc_wgrad_bf16_scratch_bytes(int cin, int cout);
hipError_t launch_disc_wgrad_bf16(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride, float* dw,
                                  float* dbias, void* scratch, hipStream_t s);
hipError_t launch_disc_wgrad(const float* in, const float* dpre, int n, int h, int w, int cin, int cout, int stride, float* dw,
                             float* dbias, void* scratch, hipStream_t s);
size_t disc_stem_wgrad_scratch_bytes();
hipError_t launch_disc_stem_wgrad(const float* img, const float* dpre, int n, int h, int w, float* dw, float* dbias, void* scratch,
                                  hipStream_t s);
hipError_t launch_disc_stem_dgrad(const float* dpre, const float* w, int n, int h, int wd, float* dimg, hipStream_t s);
hipError_t launch_disc_tail_bwd(const float* dlogit, const float* w, const float* act, int n, int h, int wd, int cin, float slope,
                                float* din, float* dw, float* db, hipStream_t s);
hipError_t launch_ragan_grad(const float* real, const float* fake, size_t nr, size_t nf, float t_real, float t_fake, float scale,
                             float* dreal, float* dfake, hipStream_t s);
size_t clip_scratch_bytes();
hipError_t launch_clip_grad_norm(float* g, size_t n, float max_norm, float* norm_coef, void* scratch, hipStream_t s);
// per-device page of zeros (>= kDiscZeroFloats floats) kept by srgen_train_api.cpp
int train_zero_page(const float** out);

// ---- train-mode BatchNorm over an NHWC (npix, 96) fp32 tensor, in place (srgen_kernels_misc.hip) ------------
// t <- slope ? PReLU(bn(t)) : bn(t);  then, with res: t <- res + res_factor * t.  bn uses the batch mean and
// the biased batch variance (eps 1e-5); batch_stats receives {mean[96], biased var[96]}.
constexpr int kBnBlocks = 1024;
size_t bn_scratch_bytes();
// out == nullptr: in place; else t stays (the saved conv output of the training step) and out receives the result
hipError_t launch_bn_train(float* t, size_t npix, const float* gamma, const float* beta, const float* slope,
                           const float* res, float res_factor, void* scratch, float* batch_stats, hipStream_t s,
                           float* out = nullptr);
// the same over bf16 tensors (t, res, out); statistics and arithmetic in fp32
hipError_t launch_bn_train_bf16(void* t, size_t npix, const float* gamma, const float* beta, const float* slope,
                                const void* res, float res_factor, void* scratch, float* batch_stats, hipStream_t s,
                                void* out = nullptr);

// ---- backward-pass operators of the pre-train step (srgen_train_kernels.hip) --------------------------------------
// Winograd weight image formed on the device from OIHW fp32 weights: mode 0 forward (cout_total 96 or 384, the
// latter with the PixelShuffle channel permutation), mode 1 data gradient of cout block `block`
hipError_t launch_wino_pack(const float* w, float* out, int cout_total, int mode, int block, hipStream_t s);
// the same for the split-f16 kernel: out = kSplitImageBytes per cout block + kSplitTailBytes
hipError_t launch_split_pack(const float* w, void* out, int cout_total, int mode, int block, hipStream_t s);
// many images in two launches; job i: arguments as launch_split_pack
constexpr int kSplitPackMaxJobs = 96;
struct SplitPackJob { const float* w; void* out; int cout_total, mode, block, pad; };
struct SplitPackJobs { SplitPackJob j[kSplitPackMaxJobs]; int count; };
hipError_t launch_split_pack_many(const SplitPackJobs& jobs, hipStream_t s);
// bf16 images of the 16-bit step (the eval bf16 conv kernel's layout), kConv3x3Bf16Bytes per cout block, same jobs
constexpr size_t kConv3x3Bf16Bytes = (size_t)9 * 96 * 96 * 2;
hipError_t launch_bf16_pack_many(const SplitPackJobs& jobs, hipStream_t s);
// out[kAmaxSlots] = per-workgroup maxima of |x| (their maximum is max |x|)
hipError_t launch_absmax(const float* x, size_t n, float* out, hipStream_t s);
// out[i] = max |x[i]| per image (per_image floats each, a multiple of 4), zero_out[i] = 0
hipError_t launch_absmax_per_image(const float* x, int n, size_t per_image, float* out, float* zero_out, hipStream_t s);
size_t conv3x3_wgrad_scratch_bytes(int n, int h);
// dw: (96, 96, 3, 3) OIHW; dbias: [96] or nullptr; x, dy: (n, h, h, 96) NHWC
hipError_t launch_conv3x3_wgrad(const float* x, const float* dy, int n, int h, float* dw, float* dbias, void* scratch,
                                hipStream_t s);
// the same on the split arithmetic (srgen_wgrad_f32s.hip; h a multiple of 64); dy_amax: kAmaxSlots partial maxima of |dy|
// or nullptr (dy of order 1); scratch of conv3x3_wgrad_scratch_bytes
// terms: 3 (split products) or 1 (f16 products), as Conv3x3Args::terms
// bf16: x and dy are bf16 tensors (terms = 1, no dy_amax), multiplied on the bf16 matrix instruction
hipError_t launch_conv3x3_wgrad_split(const void* x, const void* dy, int n, int h, const float* dy_amax, float* dw,
                                      float* dbias, void* scratch, hipStream_t s, int terms = 3, bool bf16 = false);
constexpr int kWgradSplitMaxSlices = 168;         // slab = [slices][9][96][96] floats, then [slices][96] bias partials
size_t conv3x3_wgrad_split_scratch_bytes(int n, int h);
hipError_t launch_conv3x3_wgrad_split_partial(const void* x, const void* dy, int n, int h, const float* dy_amax,
                                              float* slab, float* bias_slab, int* slices_out, hipStream_t s, int terms = 3,
                                              bool bf16 = false);
size_t bn_bwd_scratch_bytes();
// c: saved conv output; dt: gradient of the BatchNorm(+PReLU) output (scaled by dscale); stats: {mean, biased var}
hipError_t launch_bn_train_bwd(const float* c, const float* dt, const float* stats, const float* gamma, const float* beta,
                               const float* slope, float dscale, float* dc, float* dgamma, float* dbeta, float* dslope,
                               size_t npix, void* scratch, hipStream_t s, float* dc_amax = nullptr);
// dsr = d/dsr [w_l1 * weighted_l1(sr, hr) + w_ssim * (1 - ssim(sr, hr))]; scratch: 3 * n*c*h*w floats
hipError_t launch_pixel_loss_grad(const float* sr, const float* hr, int n, int c, int h, int w, float w_l1, float w_ssim,
                                  float* dsr, void* scratch, hipStream_t s);
size_t prelu_scratch_bytes();
hipError_t launch_prelu_scalar_fwd(const float* x, const float* slope, float* y, size_t numel, hipStream_t s);
hipError_t launch_prelu_scalar_bwd(const float* x, const float* dy, const float* dy2, const float* slope, float* dx, float* dslope,
                                   size_t numel, void* scratch, hipStream_t s);
// 9x9 convs against a 3-channel NCHW image (see srgen_train_kernels.hip): wm [243][96]; r [96][243]
hipError_t launch_conv9x9_c3(const float* img, const float* wm, const float* bias, float* out, int n, int ih, int iw, int oh,
                             int ow, int flip, hipStream_t s);
size_t conv9x9_c3_wgrad_scratch_bytes(int n, int ah);
hipError_t launch_conv9x9_c3_wgrad(const float* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                   float* dbias, void* scratch, hipStream_t s);
// the same operators over bf16 activation / gradient tensors (the 16-bit training step); parameters, statistics and
// parameter gradients stay fp32
hipError_t launch_bn_train_bwd_bf16(const void* c, const void* dt, const float* stats, const float* gamma, const float* beta,
                                    const float* slope, float dscale, void* dc, float* dgamma, float* dbeta, float* dslope,
                                    size_t npix, void* scratch, hipStream_t s);
hipError_t launch_prelu_scalar_fwd_bf16(const void* x, const float* slope, void* y, size_t numel, hipStream_t s);
hipError_t launch_prelu_scalar_bwd_bf16(const void* x, const void* dy, const void* dy2, const float* slope, void* dx,
                                        float* dslope, size_t numel, void* scratch, hipStream_t s);
hipError_t launch_prelu_scalar_bwd_unshuffle(const void* x, const void* dy, const float* slope, void* dx, float* dslope, int n, int full,
                                             void* scratch, hipStream_t s, bool bf16);
hipError_t launch_conv9x9_c3_wgrad_bf16(const void* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                        float* dbias, void* scratch, hipStream_t s);
hipError_t launch_conv9x9_c3_wgrad_reduce(const float* slab, const float* bias_slab, int slices, float* r, float* dbias, hipStream_t s);
// rows of 64 | 128 pixels on v_mfma_f32_32x32x16_bf16 (srgen_wgrad9_bf16.hip); scratch as conv9x9_c3_wgrad_scratch_bytes, slices <= its count
hipError_t launch_conv9x9_c3_wgrad_bf16_mfma(const void* act, const float* img, int n, int ah, int aw, int ih, int iw, int flip, float* r,
                                             float* dbias, void* scratch, int slices, hipStream_t s);
hipError_t launch_chan_attn_bwd_bf16(const void* x, const void* dy, const float* pool_partial, const float* scale, int n, int hw,
                                     const float* w1, const float* b1, const float* w2, const float* b2, void* dx, float* dw1,
                                     float* db1, float* dw2, float* db2, void* scratch, hipStream_t s);
size_t chan_attn_bwd_scratch_bytes(int n);
hipError_t launch_chan_attn_bwd(const float* x, const float* dy, const float* pool_partial, const float* scale, int n, int hw,
                                const float* w1, const float* b1, const float* w2, const float* b2, float* dx, float* dw1, float* db1,
                                float* dw2, float* db2, void* scratch, hipStream_t s);
hipError_t launch_adamw(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2, float eps,
                        float wd, int step, hipStream_t s);
hipError_t launch_pool_partial(const float* t, int n, int hw, int parts, float* pool_partial, hipStream_t s);
hipError_t launch_pool_partial_bf16(const void* t, int n, int hw, int parts, float* pool_partial, hipStream_t s);

// ---- packed-weight geometry shared by the packers (host) and the kernels ---------------
// fp32 3x3: float4 units  [cout_block][tap 9][ci4 24][co 96] -> 4 consecutive ci
constexpr size_t kConv3x3F32Floats = (size_t)9 * 96 * 96;             // per cout block
// fp32 stem: [k 256][co 96], k = ky*28 + (kx*3+ci), rows >= 252 and kx*3+ci == 27 are zero
constexpr int kStemK = 256;
constexpr size_t kStemF32Floats = (size_t)kStemK * 96;
// fp32 head: float2 units [ky 9][c8 12][u 2][h 2][n 32] -> ci = 8*c8 + 2*(2u+h) + e,
//            n = kx*3 + co (27 used, rest zero)
constexpr size_t kHeadF32Floats = (size_t)9 * 12 * 2 * 2 * 32 * 2;

}  // namespace srgen
