// srgen_conv3x3_f32_wino.hip -- gfx950 fp32 3x3 conv 96 -> 96 (x cout_blocks) as Winograd
// F(2x2, 3x3) on the exact-fp32 matrix instruction v_mfma_f32_16x16x4_f32.
//
// Reference sites (file:line into /root/reference): Generator.py:34,37 (ResBlock convs; eval
// BatchNorm :35,38 and the 0.1 of :42 folded into w, b on the host), :62 (mid), :7 (upsample
// conv, PixelShuffle :8 folded into the store address), :69 (output[0]); PReLU :36,:9,:70 and
// the residual adds :42,:78 are the epilogue.
//
// Why: the fp32 direct kernel (srgen_conv3x3_f32.hip) already runs at 93 % of what the matrix
// pipe delivers on this part, so the only lever left is fewer matrix FLOPs.  For a 2x2 output
// tile, Y = At [ (G g Gt) . (Bt d B) ] A needs 16 multiplies per (ci, co) instead of 36: the 16
// element-wise products, summed over ci, are 16 independent GEMMs
//     M[xi][nu][co][tile] = sum_ci U[xi][nu][co][ci] * V[xi][nu][ci][tile]
// with U = G g Gt precomputed on the host (fp64, rounded once) and V = Bt d B formed on the fly.
// All transform coefficients are 0, +-1, +-1/2: no precision-critical constants.  Error against
// the fp64 oracle stays ~1e-6 of max|ref| per conv (tests: <= 1e-4 end to end).
//
// Mapping (one persistent workgroup of 4 waves per CU, one wave per SIMD):
//   * work item = 4 output rows x 64 pixels = 2 tile-rows x 32 tiles, x 96 co; wave w owns 16
//     tiles: tile-row (w >> 1), tiles 16*(w & 1) + (lane & 15).  The MFMA's A operand is U
//     (rows = co), its B operand is V (columns = tiles), so accumulator register r of lane l is
//     C[co = 16*mt + 4*(l >> 4) + r][tile = l & 15]: 4 consecutive channels of one tile, and the
//     output transform + epilogue are registers only;
//   * loop nest: nu (4) > ci chunk of 16 (6) > xi (4) > 24 MFMAs (4 k-steps x 6 co blocks).  For
//     a fixed nu only the four M[xi][nu] (96 registers) are live; after the last chunk they are
//     folded into the 2x2 output accumulators y[i][j] (96 registers): T_i = sum_xi At[i][xi]
//     M[xi][nu], y[i][j] += At[j][nu] T_i.  The input chunks are therefore re-staged once per nu
//     (4x the L2 -> LDS traffic of the direct kernel: cheap at fp32 matrix rates);
//   * input halo tile (6 rows x 66 cols x 16 ci) and the U slab of (nu, chunk) (24 KB) go global
//     -> LDS with global_load_lds into double buffers (the next chunk lands while this one is
//     multiplied); even / odd halo columns are de-interleaved so the 16 tiles of a lane group
//     read consecutive 64-byte entries (2-way bank conflict on 8 reads per chunk: irrelevant);
//     out-of-image pieces are sourced from a zero page;
//   * V for the next chunk is formed (8 LDS reads, 32 adds) under the current chunk's last 24
//     MFMAs; U fragments are read from LDS one xi ahead (conflict-free).
#include "srgen_internal.h"

#include <type_traits>

namespace srgen {
namespace c3w {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define MFMA16_F32(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

constexpr int IN_ROWS = 6;                        // halo rows of an item (4 output rows + 2)
constexpr int CK = 16;                            // ci per chunk
constexpr int ENTRY = 64;                         // bytes per halo entry (one pixel, 16 ci)
// Halo buffer: row r at r*ROWB holds the item's 64 interior columns, de-interleaved: entry
// (par*32 + idx) = column 2*idx + par -- exactly the 256 16-byte pieces one glds round of the
// workgroup writes, from ONE global row segment (validity depends on the row only: wave-uniform).
// The two edge columns (-1 and 64) of the 6 rows sit behind the rows: entry EDGE_OFF + (r*2 + LR).
constexpr int ROWB = 64 * ENTRY;                                    // 4,096
constexpr int EDGE_OFF = IN_ROWS * ROWB;                            // 24,576
constexpr int BUF_BYTES = EDGE_OFF + 1024;                          // 25,600 (edge piece = one wave piece, 768 B used)
constexpr int USLAB_BYTES = 4 * 4 * 96 * 16;                        // 24,576: U of one (nu, chunk): [xi][ci/4][co] float4
constexpr int NBUF = 3;                                             // chunk c+2 is staged while chunk c is multiplied
constexpr int U_OFF = NBUF * BUF_BYTES;                             // the U slabs sit behind the halo buffers
constexpr int BS_OFF = U_OFF + NBUF * USLAB_BYTES;                     // bias[4][96], slope[4][96]
constexpr int LDS_BYTES = BS_OFF + 2 * 4 * 96 * 4;                  // 153,600 (one workgroup per CU)
constexpr int MAX_WGS = 256;                      // persistent grid: one workgroup per CU

__device__ __forceinline__ int opaque(int v)
{
    asm volatile("" : "+v"(v));
    return v;
}

__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base)
{
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)gsrc,
                                     (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

struct Item { int n, ty, tx, cb; };

// acc += a (x) b on v_mfma_f32_16x16x4_f32, accumulator in place by construction: through the builtin
// the register allocator is free to rotate loop-carried accumulators through copies.
__device__ __forceinline__ void mfma_acc(f32x4& acc, float a, float b)
{
    asm volatile("v_mfma_f32_16x16x4_f32 %0, %1, %2, %0" : "+a"(acc) : "v"(a), "v"(b));
}

// The 2x2 output accumulators are touched only by the four folds and the epilogue of an item, but
// are live across all of its 2,304 MFMAs; left to the register allocator they compete with the
// pipeline registers for the 256 architectural VGPRs and spill.  Pin them to the accumulator half
// of the register file (vector ALU instructions cannot address it, hence the explicit moves).
__device__ __forceinline__ float acc_get(float a)
{
    float v;
    asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a));
    return v;
}
__device__ __forceinline__ void acc_set(float& a, float v) { asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); }

#ifdef SRGEN_C3W_STAMPS      // development only (tools/convbench_f32): per-item phase timestamps
__device__ long long* g_wstamps = nullptr;
#define C3W_STAMP(k)                                                                      \
    do {                                                                                  \
        if (g_wstamps && tid == 0) g_wstamps[(size_t)item * 16 + (k)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define C3W_STAMP(k) do { } while (0)
#endif

template <int HW, bool SHUFFLE, bool POOL, bool RES, bool ACT>
__global__ __launch_bounds__(256, 1) void conv3x3_f32_wino_kernel(Conv3x3Args a, int nitems)
{
    extern __shared__ __attribute__((aligned(16))) char lds[];
    const int tid = threadIdx.x, lane = tid & 63, kq = lane >> 4, c16 = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tr = w >> 1, tt = 16 * (w & 1) + c16;
    constexpr int tiles_x = HW / 64, tiles_y = HW / 4, tiles_img = tiles_x * tiles_y;
    constexpr int CB = SHUFFLE ? 4 : 1;

    // persistent item walk: XCD x owns items [x*per, (x+1)*per) (see srgen_conv3x3_bf16.hip)
    const int per_xcd = nitems >> 3;
    const int wg_per_xcd = (int)gridDim.x >> 3;
    const int xcd = blockIdx.x & 7;
    const int item_end = (xcd + 1) * per_xcd;
    int item = xcd * per_xcd + ((int)blockIdx.x >> 3);
    auto decode = [&](int it) {
        Item r;
        r.cb = it % CB;
        const int tile = it / CB;
        r.tx = tile % tiles_x;
        r.ty = (tile / tiles_x) % tiles_y;
        r.n = tile / tiles_img;
        return r;
    };

    float* const ldsBias = (float*)(lds + BS_OFF);
    float* const ldsSlope = ldsBias + 4 * 96;
    for (int i = tid; i < CB * 96; i += 256) {
        ldsBias[i] = a.bias[i];
        if (ACT) ldsSlope[i] = a.slope[i];
    }

    // ---- staging ---------------------------------------------------------------------------
    // stage(item, nu, chunk, buf): halo chunk -> halo buffer buf, and the U slab of (cb, nu, chunk)
    // -> U buffer buf.  Packed U: float4 units [cb][nu 4][chunk 6][xi 4][ci/4 4][co 96].  A thread's
    // piece of halo row k is its own fixed 16 bytes of that row's 64-pixel segment (byte offset
    // row_off); rows outside the image come from the zero page (kWinoZeroBytes) at the same offset.
    constexpr bool EDGES = tiles_x > 1;            // with one tile per image row both edge columns are padding
    if (!EDGES && tid < 64 * NBUF) {               // ... zeroed once, never staged
        const f32x4 z = {0.f, 0.f, 0.f, 0.f};
        *(f32x4*)(lds + (tid >> 6) * BUF_BYTES + EDGE_OFF + (tid & 63) * 16) = z;
    }
    const int row_off = ((2 * ((tid >> 2) & 31) + ((tid >> 7) & 1)) * kC + (tid & 3) * 4) * 4;
    // part 0 / 1 / 2: thirds of the pieces, issued before the MFMA blocks of xi = 0 / 1 / 2; -1: all
    auto stage = [&](const Item& it, int nu, int chunk, int buf, int part) {
        const char* in = (const char*)((const float*)a.in + ((size_t)it.n * HW * HW + it.tx * 64) * kC + chunk * CK);
        char* dst = lds + buf * BUF_BYTES;
        const int y0 = it.ty * 4 - 1;
#pragma unroll
        for (int k = 0; k < IN_ROWS; ++k) {
            if (part >= 0 && part != k / 4) continue;
            const int gy = y0 + k;
            const char* rowp = (unsigned)gy < (unsigned)HW ? in + (size_t)gy * (HW * kC * 4) : (const char*)a.zeros;
            glds16(rowp + opaque(row_off), dst + k * ROWB + w * 1024);
        }
        if (EDGES && w == 0 && (part < 0 || part == 1)) {       // edge columns: lane l -> row l/8, side (l/4)&1, 4 ci l&3
            const int l = opaque(lane);
            const int gy = y0 + (l >> 3), dx = ((l >> 2) & 1) ? 64 : -1, gx = it.tx * 64 + dx;
            const bool ok = l < 48 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const char* src = ok ? in + ((ptrdiff_t)gy * HW + dx) * (kC * 4) + (l & 3) * 16 : (const char*)a.zeros;
            glds16(src, dst + EDGE_OFF);
        }
        const char* usrc = (const char*)a.wpack + (size_t)((it.cb * 4 + nu) * 6 + chunk) * USLAB_BYTES + opaque(tid) * 16;
        char* udst = lds + U_OFF + buf * USLAB_BYTES;
#pragma unroll
        for (int k = 0; k < USLAB_BYTES / 4096; ++k)
            if (part < 0 || part == (k < 2 ? 1 : 2)) glds16(usrc + k * 4096, udst + (k * 256 + w * 64) * 16);
    };

    // V[xi] (xi = 0..3) for column nu of a chunk, in two halves so the LDS latency sits under
    // MFMAs: read_raw issues the 8 reads, combine does the adds.
    //   e[a] = sum_b d[a][b] Bt[nu][b]:  nu0: d0 - d2   nu1: d1 + d2   nu2: d2 - d1   nu3: d1 - d3
    //   V[xi] = sum_a Bt[xi][a] e[a]:    e0 - e2, e1 + e2, e2 - e1, e1 - e3
    // Input column b (0..3) of tile tt is image column 2*tt - 1 + b: b = 1, 2 -> (par 0 / 1, idx tt);
    // b = 0 -> (par 1, idx tt - 1) or the left edge entry; b = 3 -> (par 0, idx tt + 1) or the right
    // edge entry.  Per lane: byte offset of halo row 2*tr and the row stride, for each b.
    int rb[4], rs[4];
    rb[1] = (2 * tr) * ROWB + tt * ENTRY + 16 * kq;
    rb[2] = (2 * tr) * ROWB + (32 + tt) * ENTRY + 16 * kq;
    rb[0] = tt > 0 ? (2 * tr) * ROWB + (32 + tt - 1) * ENTRY + 16 * kq : EDGE_OFF + (2 * tr) * 2 * ENTRY + 16 * kq;
    rb[3] = tt < 31 ? (2 * tr) * ROWB + (tt + 1) * ENTRY + 16 * kq : EDGE_OFF + ((2 * tr) * 2 + 1) * ENTRY + 16 * kq;
    rs[1] = rs[2] = ROWB;
    rs[0] = tt > 0 ? ROWB : 2 * ENTRY;
    rs[3] = tt < 31 ? ROWB : 2 * ENTRY;
    auto read_raw = [&](f32x4 (&R)[8], int nu, int buf) {
        const int b0 = nu == 0 ? 0 : 1, b1 = nu == 3 ? 3 : 2;
        const char* base = lds + buf * BUF_BYTES;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            R[2 * r] = *(const f32x4*)(base + opaque(rb[b0]) + r * (b0 == 1 ? ROWB : rs[b0]));
            R[2 * r + 1] = *(const f32x4*)(base + opaque(rb[b1]) + r * (b1 == 2 ? ROWB : rs[b1]));
        }
    };
    auto combine = [&](f32x4 (&V)[4], const f32x4 (&R)[8], int nu) {
        f32x4 e[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            e[r] = nu == 1 ? R[2 * r] + R[2 * r + 1] : nu == 2 ? R[2 * r + 1] - R[2 * r] : R[2 * r] - R[2 * r + 1];
        V[0] = e[0] - e[2];
        V[1] = e[1] + e[2];
        V[2] = e[2] - e[1];
        V[3] = e[1] - e[3];
    };
    struct UF { f32x4 u[6]; };
    auto load_u = [&](UF& u, int buf, int xi) {
        const f32x4* p = (const f32x4*)(lds + U_OFF + buf * USLAB_BYTES) + (xi * 4 + kq) * 96 + opaque(c16);
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) u.u[mt] = p[16 * mt];
    };

    // In the main loop the staging of chunk s+2 is issued as three clusters of four wave pieces in front
    // of the MFMA blocks of xi = 0, 1, 2.  A global_load_lds costs ~30 cycles of an idle matrix pipe by
    // itself (tools/issuebench.cpp) and as much again for the scalar address arithmetic in front of it,
    // so that arithmetic is done one block earlier, in MFMA gaps, into four piece descriptors
    // (source row / slab pointer and LDS destination, both wave-uniform); a cluster is then nothing
    // but 4 x {s_mov m0, global_load_lds}.  Pieces g = 0..5: halo rows, 6..11: the U slab.
    struct Stage { const char* in; const char* usrc; int dst, udst, y0, x0; };
    auto make_stage = [&](const Item& it, int nu, int chunk, int buf) {
        Stage st;
        st.in = (const char*)((const float*)a.in + ((size_t)it.n * HW * HW + it.tx * 64) * kC + chunk * CK);
        st.usrc = (const char*)a.wpack + (size_t)((it.cb * 4 + nu) * 6 + chunk) * USLAB_BYTES;
        st.dst = buf * BUF_BYTES + w * 1024;
        st.udst = U_OFF + buf * USLAB_BYTES + w * 1024;
        st.y0 = it.ty * 4 - 1;
        st.x0 = it.tx * 64;
        return st;
    };
    const char* d_src[4];
    int d_dst[4];
    auto make_desc = [&](const Stage& st, int g, int j) {
        const char* sp;
        int dm;
        if (g < IN_ROWS) {
            const int gy = st.y0 + g;
            sp = (unsigned)gy < (unsigned)HW ? st.in + (size_t)gy * (HW * kC * 4) : (const char*)a.zeros;
            dm = st.dst + g * ROWB;
        } else {
            sp = st.usrc + (g - IN_ROWS) * 4096;
            dm = st.udst + (g - IN_ROWS) * 4096;
        }
        asm volatile("" : "+s"(sp), "+s"(dm));       // computed HERE (an MFMA gap), not at the use
        d_src[j] = sp;
        d_dst[j] = dm;
    };
    auto issue_piece = [&](int g, int j) {
        const unsigned voff = (unsigned)opaque(g < IN_ROWS ? row_off : tid * 16);
        glds16(d_src[j] + voff, lds + d_dst[j]);
    };
    auto issue_edge = [&](const Stage& st) {        // edge columns (tiles_x > 1 only), wave 0
        if (EDGES && w == 0) {
            const int l = opaque(lane);
            const int gy = st.y0 + (l >> 3), dx = ((l >> 2) & 1) ? 64 : -1, gx = st.x0 + dx;
            const bool ok = l < 48 && (unsigned)gy < (unsigned)HW && (unsigned)gx < (unsigned)HW;
            const char* src = ok ? st.in + ((ptrdiff_t)gy * HW + dx) * (kC * 4) + (l & 3) * 16 : (const char*)a.zeros;
            glds16(src, lds + st.dst + EDGE_OFF);     // st.dst includes w*1024 = 0
        }
    };

    Item cur = decode(item);
    stage(cur, 0, 0, 0, -1);
    stage(cur, 0, 1, 1, -1);
    __syncthreads();
    Stage st = make_stage(cur, 0, 2, 2);           // staging target of the first chunk iteration
#pragma unroll
    for (int j = 0; j < 4; ++j) make_desc(st, j, j);

    // pipeline registers: V of the current chunk, raw halo reads of the next, U fragments of the
    // current and the next xi
    f32x4 V[4], R[8];
    UF ua, ub;
    read_raw(R, 0, 0);
    load_u(ua, 0, 0);
    combine(V, R, 0);

#pragma unroll 1
    for (;;) {
        const int nxt = item + wg_per_xcd;
        const bool has_next = nxt < item_end;
        const Item nx = decode(has_next ? nxt : item);
        // y[i][j][mt]: output pixel (i, j) of the lane's tile, channels 16*mt + 4*kq .. +3
        float y[2][2][6][4];             // (accumulator registers: acc_get / acc_set)
#pragma unroll
        for (int mt = 0; mt < 6; ++mt) {
            const f32x4 b4 = *(const f32x4*)(ldsBias + cur.cb * 96 + 16 * mt + 4 * kq);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc_set(y[0][0][mt][r], b4[r]);
                acc_set(y[0][1][mt][r], b4[r]);
                acc_set(y[1][0][mt][r], b4[r]);
                acc_set(y[1][1][mt][r], b4[r]);
            }
        }

        // ---- main loop: nu > chunk > xi, one flat software pipeline -----------------------------
        // Invariant at the top of a chunk iteration: V holds this chunk's V, ua the U fragments of
        // its xi = 0, and this chunk's LDS buffers have landed (barrier of the previous chunk).
        // The residual of the lane's 4 pixels (24 16-byte pieces): loaded in MFMA gaps of the first two
        // chunks of the nu = 3 pass, one every second or third gap -- a wave load here touches 16 cache lines, and 24
        // of them issued back to back stall the wave in the texture path for ~5k cycles.  Wave-uniform
        // base + one per-lane offset + immediates: no address arithmetic between the MFMAs.
        const int oy = cur.ty * 4 + 2 * tr, ox = cur.tx * 64 + 2 * tt;
        f32x4 rr[RES ? 24 : 1];
        const char* const rbase = (const char*)((const float*)a.res + (((size_t)cur.n * HW + cur.ty * 4) * HW + cur.tx * 64) * kC);
        const unsigned roff = (unsigned)opaque(((2 * tr * HW + 2 * tt) * kC + 4 * kq) * 4);
        auto load_res = [&](int q) {
            rr[RES ? q : 0] = *(const f32x4*)(rbase + (q / 12) * (HW * kC * 4) + roff + (((q / 6) & 1) * kC + 16 * (q % 6)) * 4);
        };
        auto run_nu = [&](auto NU) {
            constexpr int nu = decltype(NU)::value;
            constexpr int nu_next = (nu + 1) & 3;
            f32x4 M[4][6];
#pragma unroll
            for (int xi = 0; xi < 4; ++xi)
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) M[xi][mt] = f32x4{0.f, 0.f, 0.f, 0.f};
            // (PEEL: the chunk is one of the two peeled ones of a residual conv's nu = 3 pass, which carry
            // the residual loads -- as straight-line code, so that the loads are not inside a loop, where
            // the compiler would wait for each with vmcnt(0))
            auto do_chunk = [&](const int chunk, auto PEEL) {
                constexpr bool peel = decltype(PEEL)::value;
                // chunk sequence number s = nu*6 + chunk lives in buffer s % 3 = chunk % 3; s + 1 is being
                // read ahead at xi = 3, s + 2 is staged (this nu's, the next nu's, or the next item's: nx is
                // this item again after the last one, a harmless re-stage that keeps the vmcnt arithmetic)
                const int buf = chunk >= 3 ? chunk - 3 : chunk;
                const int bufn = buf == 2 ? 0 : buf + 1;
                // LDS addresses of this chunk's reads, formed here so that nothing but the reads
                // themselves (address register + immediate offset) sits between the MFMAs below
                const int u_lane = (kq * 96 + c16) * 16;
                const int ua_cur = opaque(U_OFF + buf * USLAB_BYTES + u_lane);
                const int ua_nxt = opaque(U_OFF + bufn * USLAB_BYTES + u_lane);
                int va[8];                                                  // va[2r + which], chunk read ahead at xi = 3
                auto set_va = [&](int nuc) {
                    const int b0 = nuc == 0 ? 0 : 1, b1 = nuc == 3 ? 3 : 2;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        va[2 * r] = opaque(bufn * BUF_BYTES + rb[b0] + r * (b0 == 1 ? ROWB : rs[b0]));
                        va[2 * r + 1] = opaque(bufn * BUF_BYTES + rb[b1] + r * (b1 == 2 ? ROWB : rs[b1]));
                    }
                };
                if (chunk < 5) set_va(nu);
                else set_va(nu_next);
#pragma unroll
                for (int xi = 0; xi < 4; ++xi) {
                    if (xi < 3) {
                        // chunk s+2 -> the third buffer pair
#pragma unroll
                        for (int j = 0; j < 4; ++j) issue_piece(4 * xi + j, j);
                        if (xi == 1) issue_edge(st);
                    } else {
                        // chunk s+1 was staged a whole chunk ago: wait for it only -- the 12 pieces this wave
                        // issued for s+2 during this chunk stay in flight (vmcnt counts in order) -- and
                        // for this wave's LDS reads of the buffer that s+3 will overwrite.
                        asm volatile("" ::: "memory");
                        // Residual convs, nu = 3: the loads issued behind the last cluster of chunks 0 and 1 stay
                        // in flight too -- chunk 0: 12 + 6, chunk 1: 12 + 12 + 6, chunk 2: 12 + 12.
                        if (peel) {
                            if (chunk == 0) __builtin_amdgcn_s_waitcnt(0x4072);               // vmcnt(18)
                            else __builtin_amdgcn_s_waitcnt(0x407e);                          // vmcnt(30)
                        } else if (RES && nu == 3 && chunk == 2) __builtin_amdgcn_s_waitcnt(0x4078);          // vmcnt(24)
                        else __builtin_amdgcn_s_waitcnt(0x0070 | 12);     // vmcnt(12) lgkmcnt(0)
                        __builtin_amdgcn_s_barrier();
                        asm volatile("" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    const UF& u = (xi & 1) ? ub : ua;
                    UF& un = (xi & 1) ? ua : ub;
                    // 24 MFMAs with ONE LDS read in every third gap: there a ds_read_b128 issues for
                    // free, in a cluster in front of the block each costs ~30 cycles of an idle matrix
                    // pipe (tools/issuebench.cpp).  xi 0..2: the U fragments of xi+1; xi 3: the raw halo
                    // reads and the first U fragments of the next chunk.
#pragma unroll
                    for (int sl = 0; sl < 24; ++sl) {
                        const int e = sl / 6, mt = sl % 6;
                        mfma_acc(M[xi][mt], u.u[mt][e], V[xi][e]);
                        if (xi < 3) {
                            // (with LDS-DMA in the loop the compiler waits for ALL outstanding LDS reads at the
                            // next block's first MFMA: the last read is issued 7 gaps before that)
                            if (sl % 3 == 1 && sl / 3 < 6) un.u[sl / 3] = *(const f32x4*)(lds + ua_cur + ((xi + 1) * 4 * 96 + 16 * (sl / 3)) * 16);
                            // descriptors of the next cluster (scalar work, free in these gaps)
                            if (xi < 2 && sl % 6 == 2) make_desc(st, 4 * (xi + 1) + sl / 6, sl / 6);
                            // residual pieces: AFTER the chunk's last staging cluster, so that (vmcnt counts in
                            // order) no barrier forces them home for the next two chunks
                            if (peel && xi == 2 && sl % 3 == 0 && sl != 12 && sl < 21) load_res(chunk * 12 + (sl < 12 ? sl / 3 : sl / 3 - 1));
                            if (xi == 2 && sl == 2) {
                                // the next chunk iteration stages ITS s+2 into the buffer this one is reading
                                if (chunk < 3) st = make_stage(cur, nu, chunk + 3, buf);
                                else if (chunk < 5) st = nu < 3 ? make_stage(cur, nu_next, chunk - 3, buf) : make_stage(nx, 0, chunk - 3, buf);
                                else st = nu < 3 ? make_stage(cur, nu_next, 2, buf) : make_stage(nx, 0, 2, buf);
                            }
                            if (xi == 2 && sl >= 8 && sl % 4 == 0 && sl / 4 - 2 < 4) make_desc(st, sl / 4 - 2, sl / 4 - 2);
                        } else {
                            if (sl < 16 && sl % 2 == 1) R[sl / 2] = *(const f32x4*)(lds + va[sl / 2]);
                            if (peel && sl < 12 && sl % 2 == 0) load_res(chunk * 12 + 6 + sl / 2);
                            if (sl >= 16 && sl < 22) ua.u[sl - 16] = *(const f32x4*)(lds + ua_nxt + 16 * (sl - 16) * 16);
                        }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    if (xi == 3) {
                        if (chunk < 5) combine(V, R, nu);
                        else combine(V, R, nu_next);
                    }
                }
            };
            if (RES && nu == 3) {
                do_chunk(0, std::true_type{});
                do_chunk(1, std::true_type{});
#pragma unroll 1
                for (int chunk = 2; chunk < 6; ++chunk) do_chunk(chunk, std::false_type{});
            } else {
#pragma unroll 1
                for (int chunk = 0; chunk < 6; ++chunk) do_chunk(chunk, std::false_type{});
            }
            // the asm MFMAs are opaque to the hazard recognizer: let the last ones retire before M is read
            asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
            C3W_STAMP(1 + 2 * nu);
            // fold into the outputs: T_i = sum_xi At[i][xi] M[xi],  At = [[1,1,1,0],[0,1,-1,-1]];
            // y[i][0] += T_i for nu = 0,1,2;  y[i][1] += T_i (nu 1), -= T_i (nu 2, 3).
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                const f32x4 T0 = M[0][mt] + M[1][mt] + M[2][mt];
                const f32x4 T1 = M[1][mt] - M[2][mt] - M[3][mt];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (nu <= 2) {
                        acc_set(y[0][0][mt][r], acc_get(y[0][0][mt][r]) + T0[r]);
                        acc_set(y[1][0][mt][r], acc_get(y[1][0][mt][r]) + T1[r]);
                    }
                    if (nu == 1) {
                        acc_set(y[0][1][mt][r], acc_get(y[0][1][mt][r]) + T0[r]);
                        acc_set(y[1][1][mt][r], acc_get(y[1][1][mt][r]) + T1[r]);
                    }
                    if (nu >= 2) {
                        acc_set(y[0][1][mt][r], acc_get(y[0][1][mt][r]) - T0[r]);
                        acc_set(y[1][1][mt][r], acc_get(y[1][1][mt][r]) - T1[r]);
                    }
                }
            }
            C3W_STAMP(2 + 2 * nu);
        };
        C3W_STAMP(0);
        run_nu(std::integral_constant<int, 0>{});
        run_nu(std::integral_constant<int, 1>{});
        run_nu(std::integral_constant<int, 2>{});
        run_nu(std::integral_constant<int, 3>{});

        // ---- epilogue: PReLU -> (+ residual) -> 16-byte NHWC stores ------------------------------
        f32x4 psum[6];
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                float* out = (float*)a.out;
                if (SHUFFLE)
                    out += (((size_t)cur.n * 2 * HW + 2 * (oy + i) + (cur.cb >> 1)) * 2 * HW + 2 * (ox + j) + (cur.cb & 1)) * kC + 4 * kq;
                else
                    out += (((size_t)cur.n * HW + oy + i) * HW + ox + j) * kC + 4 * kq;
#pragma unroll
                for (int mt = 0; mt < 6; ++mt) {
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = acc_get(y[i][j][mt][r]);
                    if (ACT) {
                        const f32x4 s4 = *(const f32x4*)(ldsSlope + cur.cb * 96 + 16 * mt + 4 * kq);
#pragma unroll
                        for (int e = 0; e < 4; ++e) v[e] = v[e] >= 0.f ? v[e] : v[e] * s4[e];
                    }
                    if (RES) v += rr[RES ? (i * 2 + j) * 6 + mt : 0];
                    if (POOL) psum[mt] = (i == 0 && j == 0) ? v : psum[mt] + v;
                    *(f32x4*)(out + 16 * mt) = v;
                }
            }
        if (POOL) {
            // per-item channel sums (ChannelAttention's pool), fixed order: xor butterfly over the 16
            // tile lanes of each kq group, then the 4 waves through LDS.  P sits in halo buffer 2: the
            // item's last chunk (sequence 23) used it and every wave is past that chunk's barrier and
            // V reads; buffers 0 and 1 hold / receive the next item's first chunks.
            float* P = (float*)(lds + 2 * BUF_BYTES);
            __syncthreads();
#pragma unroll
            for (int mt = 0; mt < 6; ++mt) {
                f32x4 s = psum[mt];
#pragma unroll
                for (int b = 1; b < 16; b <<= 1)
#pragma unroll
                    for (int e = 0; e < 4; ++e) s[e] += __shfl_xor(s[e], b);
                if (c16 == 0) *(f32x4*)(P + w * kC + 16 * mt + 4 * kq) = s;
            }
            __syncthreads();
            if (tid < kC) {
                const float s = (P[tid] + P[kC + tid]) + (P[2 * kC + tid] + P[3 * kC + tid]);
                a.pool_partial[((size_t)cur.n * tiles_img + cur.ty * tiles_x + cur.tx) * kC + tid] = s;
            }
            __syncthreads();
        }
        C3W_STAMP(9);
        if (!has_next) break;
        item = nxt;
        cur = nx;
    }
    __syncthreads();        // the last chunks' (unused) staging must land before the LDS is released
}

template <int HW, bool SHUFFLE, bool POOL, bool RES, bool ACT>
hipError_t launch_one(const Conv3x3Args& a, hipStream_t s)
{
    static bool attr_set[64] = {};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    if (dev < 0 || dev >= 64 || !attr_set[dev]) {
        e = hipFuncSetAttribute((const void*)conv3x3_f32_wino_kernel<HW, SHUFFLE, POOL, RES, ACT>,
                                hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64) attr_set[dev] = true;
    }
    const int nitems = a.N * (HW / 4) * (HW / 64) * (SHUFFLE ? 4 : 1);     // multiple of 16
    const int grid = nitems < MAX_WGS ? nitems & ~7 : MAX_WGS;
    hipLaunchKernelGGL((conv3x3_f32_wino_kernel<HW, SHUFFLE, POOL, RES, ACT>), dim3((unsigned)grid), dim3(256), LDS_BYTES,
                       s, a, nitems);
    return hipGetLastError();
}

}  // namespace c3w

hipError_t launch_conv3x3_f32_wino(const Conv3x3Args& a, hipStream_t s)
{
    using namespace c3w;
    if (a.H != a.W || a.N <= 0 || !a.zeros) return hipErrorInvalidValue;       // zeros: kWinoZeroBytes
    const bool res = a.res != nullptr, pool = a.pool_partial != nullptr, act = a.slope != nullptr;
    // the generator only ever pairs (PReLU, no residual) or (residual, no PReLU)
    if (a.H == kLR && !a.shuffle && a.cout_blocks == 1) {
        if (!pool && !res && act) return launch_one<kLR, false, false, false, true>(a, s);
        if (!pool && res && !act) return launch_one<kLR, false, false, true, false>(a, s);
        if (pool && res && !act) return launch_one<kLR, false, true, true, false>(a, s);
    }
    if (a.H == kLR && a.shuffle && a.cout_blocks == 4 && !pool && !res && act) return launch_one<kLR, true, false, false, true>(a, s);
    if (a.H == kHR && !a.shuffle && a.cout_blocks == 1 && !pool && !res && act) return launch_one<kHR, false, false, false, true>(a, s);
    return hipErrorInvalidValue;
}

}  // namespace srgen
