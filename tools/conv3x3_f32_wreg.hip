// srgen_conv3x3_f32_wreg.hip -- fp32 3x3 conv 96 -> 96 as Winograd F(2x2,3x3) with every operand in
// registers: no LDS, no barriers.
//
// Same arithmetic as srgen_conv3x3_f32_wino.hip (U = G g Gt packed on the host, V = Bt d B on the fly, 16
// GEMMs M[xi][nu] = U[xi][nu] V[xi][nu] on v_mfma_f32_16x16x4_f32, fold y = At M A), different data path:
// the MFMA's B operand of a lane is "its tile x 4 consecutive input channels", and the input transform
// Bt d B mixes pixels of ONE tile and nothing else, so a lane loads the eight patch pixels a Winograd
// column nu needs as 16-byte pieces straight from the NHWC activations, transforms them in registers and
// feeds the matrix pipe; the U fragments come as 16-byte loads of a lane-ordered pack, shared by the
// workgroup's four waves through L1.  Out-of-image patch pixels read a page of zeros.
//
// Workgroup = 4 waves = one item of 4 rows x 64 pixels x 96 channels (same items and pool tiling as the
// LDS kernel); wave = 16 tiles (one tile row of 32 pixels x 2 rows) x 96 channels: M[4 xi][6 co tiles] and
// y[2][2][6] are 96 + 96 accumulator registers.  One workgroup (one wave per SIMD) per CU, persistent.
// k order per item: nu (4) > 16-channel chunk (6) > xi (4): a "slice" (nu, chunk, xi) is 6 U fragment
// loads and 24 MFMAs; U fragments run in a ring of four slices (three ahead), the patch pixels of the next
// chunk are loaded a chunk ahead.
#include "srgen_internal.h"

namespace srgen {
namespace c3r {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSlices = 4 * 6 * 4;           // per item

template <int HW, bool RES, bool ACT, bool POOL>
__global__ __launch_bounds__(256, 1) void conv3x3_f32_wreg_kernel(Conv3x3Args a, int nitems)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int l16 = lane & 15, kq = lane >> 4;
    const int tr = wv >> 1;                              // tile row inside the item (rows 2tr, 2tr+1)
    const int txi = 16 * (wv & 1) + l16;                 // tile column inside the item
    constexpr int XB = HW / 64, RB = HW / 4;
    const float* __restrict__ in = static_cast<const float*>(a.in);
    float* __restrict__ out = static_cast<float*>(a.out);
    const float* zero = static_cast<const float*>(a.zeros) + kq * 4;
    // U pack: float4 units [nu 4][chunk 6][xi 4][s 6][lane 64]

#ifndef WREG_RING
#define WREG_RING 4
#endif
    constexpr int R = WREG_RING;                         // U ring depth in slices (divides 24), R - 1 ahead
    f32x4 ub[R][6];                                      // slice q of a column lives in ub[q % R]
    // prefetch pointer: the slice three ahead of the one being multiplied; wave-uniform, so the loads take
    // the scalar-base form (base in SGPRs, lane * 16 in one VGPR) and advancing it costs no VALU
    const f32x4* const ubeg = static_cast<const f32x4*>(a.wpack);
    const f32x4* const uend = ubeg + (size_t)kSlices * 6 * 64;
    const f32x4* upf = ubeg;
    auto load_u = [&](int slot) {
#pragma unroll
        for (int s = 0; s < 6; ++s) ub[slot][s] = upf[s * 64 + lane];
        upf += 6 * 64;
        if (upf == uend) upf = ubeg;                     // U is the same for every item: the ring wraps
    };

    // ring prologue: slices 0..2
#pragma unroll
    for (int q = 0; q < R - 1; ++q) load_u(q);

    // tile geometry of an item
    struct Geo {
        const float* base;       // patch (1,1) = the tile's top-left output pixel, channel 4*kq
        bool rv0, rv3;           // patch rows 0 / 3 inside the image (wave-uniform)
        bool cv0, cv3;           // patch columns 0 / 3 inside the image (per lane)
        int img, oy, ox;
    };
    auto geometry = [&](int it) {
        Geo g;
        const int xb = it % XB, rb = (it / XB) % RB;
        g.img = it / (XB * RB);
        g.oy = rb * 4 + 2 * tr;
        g.ox = xb * 64 + 2 * txi;
        g.rv0 = g.oy > 0;
        g.rv3 = g.oy + 2 < HW;
        g.cv0 = g.ox > 0;
        g.cv3 = g.ox + 2 < HW;
        g.base = in + (((size_t)g.img * HW + g.oy) * HW + g.ox) * kC + kq * 4;
        return g;
    };
    // patch pixel pointers of Winograd column nu (runtime): V column nu = sa*d[.][ca] + sb*d[.][cb] with
    //   nu 0: d0 - d2   nu 1: d1 + d2   nu 2: d2 - d1 (= -d1 + d2)   nu 3: d1 - d3
    const float* px[4][2];
    auto set_nu = [&](const Geo& g, int nu) {
        const int ca = nu == 0 ? 0 : 1, cbn = nu == 3 ? 3 : 2;
        const bool va = nu != 0 || g.cv0, vb = nu != 3 || g.cv3;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const bool rv = r == 0 ? g.rv0 : (r == 3 ? g.rv3 : true);
            const float* row = g.base + ((r - 1) * HW - 1) * kC;
            px[r][0] = (rv && va) ? row + ca * kC : zero;
            px[r][1] = (rv && vb) ? row + cbn * kC : zero;
        }
    };
    f32x4 xr[2][4][2];                                   // raw patch pixels of chunk c in xr[c & 1]
    auto load_x = [&](int buf, int chunk) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            xr[buf][r][0] = *reinterpret_cast<const f32x4*>(px[r][0] + chunk * 16);
            xr[buf][r][1] = *reinterpret_cast<const f32x4*>(px[r][1] + chunk * 16);
        }
    };

    Geo G = geometry(blockIdx.x);
    set_nu(G, 0);
    load_x(0, 0);

    for (int it = blockIdx.x; it < nitems; it += gridDim.x) {
        const int itn = it + gridDim.x < nitems ? it + gridDim.x : it;      // past the end: harmless re-loads
        const Geo GN = geometry(itn);
        const int img = G.img, oy = G.oy, ox = G.ox;

        f32x4 y[2][2][6];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int s = 0; s < 6; ++s) y[i][j][s] = f32x4{0.f, 0.f, 0.f, 0.f};

        // one trip = one Winograd column: 6 chunks x 4 slices of straight-line code (no control flow, so
        // every load stays where it is issued); nu is a run-time value, the signs are data
#pragma unroll 1
        for (int nu = 0; nu < 4; ++nu) {
            const float sa = nu == 2 ? -1.f : 1.f, sb = (nu == 1 || nu == 2) ? 1.f : -1.f;
            f32x4 m[4][6];
#pragma unroll
            for (int chunk = 0; chunk < 6; ++chunk) {
                const int buf = chunk & 1;
                f32x4 dc[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) dc[r] = sa * xr[buf][r][0] + sb * xr[buf][r][1];
                // row transform: xi 0: r0 - r2   1: r1 + r2   2: r2 - r1   3: r1 - r3
                f32x4 v[4];
                v[0] = dc[0] - dc[2];
                v[1] = dc[1] + dc[2];
                v[2] = dc[2] - dc[1];
                v[3] = dc[1] - dc[3];
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    const int q = chunk * 4 + xi;
                    load_u((q + R - 1) % R);
                    if (xi == 0) {                        // next chunk's pixels, a chunk ahead
                        if (chunk == 5) {                 // ... of the next column, or of the next item
                            const bool last = nu == 3;
                            Geo g;
                            g.base = last ? GN.base : G.base;
                            g.rv0 = last ? GN.rv0 : G.rv0;
                            g.rv3 = last ? GN.rv3 : G.rv3;
                            g.cv0 = last ? GN.cv0 : G.cv0;
                            g.cv3 = last ? GN.cv3 : G.cv3;
                            set_nu(g, (nu + 1) & 3);
                            load_x(0, 0);
                        } else {
                            load_x(buf ^ 1, chunk + 1);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int s = 0; s < 6; ++s) {
                            const f32x4 c = (chunk == 0 && j == 0) ? f32x4{0.f, 0.f, 0.f, 0.f} : m[xi][s];
                            m[xi][s] = __builtin_amdgcn_mfma_f32_16x16x4f32(ub[q % R][s][j], v[xi][j], c, 0, 0, 0);
                        }
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
            // fold: T_i = sum_xi At[i][xi] M[xi], At = [[1,1,1,0],[0,1,-1,-1]]; y[i][j] += At[j][nu] T_i
            const float f0 = nu <= 2 ? 1.f : 0.f, f1 = nu == 1 ? 1.f : (nu >= 2 ? -1.f : 0.f);
#pragma unroll
            for (int s = 0; s < 6; ++s) {
                const f32x4 t0 = m[0][s] + m[1][s] + m[2][s];
                const f32x4 t1 = m[1][s] - m[2][s] - m[3][s];
                y[0][0][s] += f0 * t0;
                y[1][0][s] += f0 * t1;
                y[0][1][s] += f1 * t0;
                y[1][1][s] += f1 * t1;
            }
        }
        G = GN;

        // epilogue: lane holds channels kq*24 + s*4 + r of the tile's four pixels
        const int c0 = kq * 24;
        f32x4 bias[6], slope[6];
#pragma unroll
        for (int s = 0; s < 6; ++s) {
            bias[s] = *reinterpret_cast<const f32x4*>(a.bias + c0 + s * 4);
            if (ACT) slope[s] = *reinterpret_cast<const f32x4*>(a.slope + c0 + s * 4);
        }
        f32x4 psum[6];
        if (POOL) {
#pragma unroll
            for (int s = 0; s < 6; ++s) psum[s] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const size_t o = (((size_t)img * HW + oy + i) * HW + ox + j) * kC + c0;
#pragma unroll
                for (int s = 0; s < 6; ++s) {
                    f32x4 r = y[i][j][s] + bias[s];
                    if (ACT) {
#pragma unroll
                        for (int e = 0; e < 4; ++e) r[e] = r[e] >= 0.f ? r[e] : r[e] * slope[s][e];
                    }
                    if (POOL) psum[s] += r;               // pooled before the residual (block output t)
                    if (RES) r += *reinterpret_cast<const f32x4*>(static_cast<const float*>(a.res) + o + s * 4);
                    *reinterpret_cast<f32x4*>(out + o + s * 4) = r;
                }
            }
        (void)psum;
    }
}

template <int HW, bool RES, bool ACT, bool POOL>
hipError_t launch_one(const Conv3x3Args& a, hipStream_t s)
{
    const int nitems = a.N * (HW / 4) * (HW / 64);
    int dev = 0, cus = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    if (e != hipSuccess) return e;
    const int grid = nitems < cus ? nitems : cus;
    hipLaunchKernelGGL((conv3x3_f32_wreg_kernel<HW, RES, ACT, POOL>), dim3(grid), dim3(256), 0, s, a, nitems);
    return hipGetLastError();
}

}  // namespace c3r

hipError_t launch_conv3x3_f32_wreg(const Conv3x3Args& a, hipStream_t s)
{
    using namespace c3r;
    if (a.H != a.W || a.N <= 0 || !a.zeros || a.shuffle || a.cout_blocks != 1 || a.pool_partial)
        return hipErrorInvalidValue;
    const bool res = a.res != nullptr, act = a.slope != nullptr;
    if (a.H == kLR && !res && act) return launch_one<kLR, false, true, false>(a, s);
    if (a.H == kLR && res && !act) return launch_one<kLR, true, false, false>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace srgen
