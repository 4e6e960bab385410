// tools/conv3x3_wgrad_wino_reg.hip -- round-2 experiment, NOT built into libsrgen.so: the Winograd-domain weight gradient
// with every wave loading its own operand pieces straight from the NHWC tensors (no LDS).  Parity-correct (same tests as
// the product kernel) and slower: 0.35 ms per launch against 0.28 for the LDS-staged kernel of
// superresolution_amd/csrc/srgen_train_kernels.hip -- 11 vector-memory instructions per 24 MFMAs at ~30 cycles each per CU
// are the bound (PMC: 62 % of wave time in s_waitcnt).  Kept as the record of that measurement.
// Needs the typedefs (f32x2, f32x4) and includes of srgen_train_kernels.hip to compile.
constexpr size_t kWgSliceFloats = 147456 + 128;               // per slice: 16 x 96 x 96 partial M | 96 bias partials | pad
struct __attribute__((aligned(4))) F3 { float v[3]; };       // three consecutive channels: one 12-byte load
struct WgRaw {
    f32x2 pa[4], pb[4];          // patch rows r = 0..3, the two columns of this nu: channels ci0 + 2i, +1
    F3 g[2][2];                  // dY tile rows a = 0..1, columns b = 0..1: channels co0 + 3i .. +2
    float frow[4];               // 1 / 0: patch row r lies inside the image (wave-uniform)
    float fcol;                  // 1 / 0: this nu's outermost patch column lies inside the image (nu = 0: left, 3: right)
};

template <int NU>
__device__ __forceinline__ void wg_load(WgRaw& R, const float* __restrict__ x, const float* __restrict__ dy, int n, int ty, int tx,
                                        int H, int W, int ci_off, int co_off)
{
    constexpr int XA = NU == 0 ? 0 : NU == 2 ? 2 : 1;
    constexpr int XB = NU == 0 ? 2 : NU == 1 ? 2 : NU == 2 ? 1 : 3;
    const int ca = 2 * tx - 1 + XA, cb = 2 * tx - 1 + XB;
    const bool va = XA != 0 || ca >= 0, vb = XB != 3 || cb < W;           // only the outermost columns can leave the image
    const unsigned offa = (unsigned)(va ? ca : 0) * 96u + ci_off, offb = (unsigned)(vb ? cb : 0) * 96u + ci_off;
    R.fcol = (va && vb) ? 1.f : 0.f;
    // out-of-image rows / columns are loaded from a clamped address and multiplied by 0 when the patch is transformed:
    // nothing here consumes a loaded value, so the loads stay in flight across the MFMA blocks
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int y = 2 * ty - 1 + r;                                     // wave-uniform
        const bool vy = (unsigned)y < (unsigned)H;
        const float* row = x + ((size_t)n * H + (vy ? y : 0)) * W * 96;
        R.pa[r] = *(const f32x2*)(row + offa);
        R.pb[r] = *(const f32x2*)(row + offb);
        R.frow[r] = vy ? 1.f : 0.f;
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        const float* row = dy + (((size_t)n * H + 2 * ty + a) * W) * 96;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            if ((NU == 0 && b == 1) || (NU == 3 && b == 0)) continue;     // this nu does not read that column
            R.g[a][b] = *(const F3*)(row + (unsigned)(2 * tx + b) * 96u + co_off);
        }
    }
}

// one wave = (slice, nu, 32 ci, 48 co) x the four xi: 4 x 3 x 2 MFMA tiles (96 accumulators, two waves per SIMD)
template <int NU>
__device__ __forceinline__ void wgrad_wino_body(const float* __restrict__ x, const float* __restrict__ dy, int total_trows, int H,
                                                int W, int rows_per_slice, float* __restrict__ slab, int cg, int ch,
                                                int slice)
{
    const int lane = threadIdx.x, i = lane & 15, kq = lane >> 4;
    const int TH = H >> 1, TW = W >> 1;
    const int ci_off = 32 * cg + 2 * i, co_off = 48 * ch + 3 * i;
    f32x4 acc[4][3][2];
#pragma unroll
    for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[xi][a][b] = f32x4{0.f, 0.f, 0.f, 0.f};
    float bsum[3] = {0.f, 0.f, 0.f};
    const int r0 = slice * rows_per_slice;
    const int r1 = min(r0 + rows_per_slice, total_trows);
    const int spr = TW >> 2;                                  // k-steps per tile row
    const int nsteps = (r1 > r0 ? r1 - r0 : 0) * spr;
    auto load_step = [&](WgRaw& R, int st) {
        const int trow = r0 + st / spr;
        wg_load<NU>(R, x, dy, trow / TH, trow % TH, (st % spr) * 4 + kq, H, W, ci_off, co_off);
    };
    auto do_step = [&](WgRaw& R, int st) {
        // ---- transforms of step st ----
        f32x2 w[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x2 pa = NU == 0 ? R.pa[r] * R.fcol : R.pa[r], pb = NU == 3 ? R.pb[r] * R.fcol : R.pb[r];
            w[r] = (NU == 1 ? pa + pb : pa - pb) * R.frow[r];
        }
        const f32x2 xh[4] = {w[0] - w[2], w[1] + w[2], w[2] - w[1], w[1] - w[3]};
        float yh[4][3];
#pragma unroll
        for (int e = 0; e < 3; ++e) {
            float c[2];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                c[a] = NU == 0 ? R.g[a][0].v[e] : NU == 1 ? R.g[a][0].v[e] + R.g[a][1].v[e]
                     : NU == 2 ? R.g[a][0].v[e] - R.g[a][1].v[e] : R.g[a][1].v[e];
            yh[0][e] = c[0]; yh[1][e] = c[0] + c[1]; yh[2][e] = c[0] - c[1]; yh[3][e] = c[1];
            if (NU == 1) bsum[e] += yh[1][e];
        }
        // ---- the raw loads of step st + 2 go into the buffer just consumed and fly under two steps of MFMAs.  No branch
        // around them (past the end the last step is simply loaded again): a conditional load makes the compiler's
        // counted vmcnt waits collapse to vmcnt(0) at the join, which serialises load and MFMA phases ----
        // sched_barrier: without them the compiler sinks the next step's transforms (the loads' consumers) to the end of
        // this one and waits vmcnt(0) there
        __builtin_amdgcn_sched_barrier(0);
        load_step(R, st + 2 < nsteps ? st + 2 : nsteps - 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
#pragma unroll
            for (int a = 0; a < 3; ++a)
#pragma unroll
                for (int b = 0; b < 2; ++b)
                    acc[xi][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(yh[xi][a], xh[xi][b], acc[xi][a][b], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    };
    WgRaw R0, R1;
    if (nsteps > 0) load_step(R0, 0);
    if (nsteps > 1) load_step(R1, 1);
    for (int st = 0; st < nsteps; st += 2) {                  // nsteps is even (W % 16 == 0)
        do_step(R0, st);
        do_step(R1, st + 1);
    }
    // D[row = 4*kq + r][col = i]: row -> co = 48 ch + 3 row + a; col -> ci = 32 cg + 2 i + b
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
        float* o = slab + (size_t)slice * kWgSliceFloats + (size_t)(xi * 4 + NU) * 9216 + ci_off;
#pragma unroll
        for (int a = 0; a < 3; ++a)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                *(f32x2*)(o + (size_t)(48 * ch + 3 * (4 * kq + r) + a) * 96) = f32x2{acc[xi][a][0][r], acc[xi][a][1][r]};
    }
    if (NU == 1 && cg == 0) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            float v = bsum[a];
            v += __shfl_xor(v, 16);
            v += __shfl_xor(v, 32);
            if (kq == 0) slab[(size_t)slice * kWgSliceFloats + 147456 + co_off + a] = v;
        }
    }
}

__global__ __launch_bounds__(64, 2) void conv3x3_wgrad_wino_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                                  int total_trows, int H, int W, int rows_per_slice,
                                                                  float* __restrict__ slab)
{
    // the 24 waves of a slice (4 nu x 2 co halves x 3 ci thirds) are consecutive workgroups of ONE XCD (blockIdx % 8):
    // X and dY come over the fabric once and from that XCD's L2 for the other 23
    int b = blockIdx.x, slice;
    if ((gridDim.x / 24) % 8 == 0) {
        const int xcd = b & 7, j = b >> 3;
        slice = (j / 24) * 8 + xcd;
        b = j % 24;
    } else {
        slice = b / 24;
        b = b % 24;
    }
    const int cg = b % 3, ch = (b / 3) & 1, nu = b / 6;
    switch (nu) {                                             // workgroup-uniform
    case 0: wgrad_wino_body<0>(x, dy, total_trows, H, W, rows_per_slice, slab, cg, ch, slice); break;
    case 1: wgrad_wino_body<1>(x, dy, total_trows, H, W, rows_per_slice, slab, cg, ch, slice); break;
    case 2: wgrad_wino_body<2>(x, dy, total_trows, H, W, rows_per_slice, slab, cg, ch, slice); break;
    default: wgrad_wino_body<3>(x, dy, total_trows, H, W, rows_per_slice, slab, cg, ch, slice); break;
    }
}

