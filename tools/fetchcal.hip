// tools/fetchcal.hip -- calibrates rocprofv3's FETCH_SIZE for the access patterns of the trunk 3x3 conv (the guide: "other
// access widths are uncalibrated: calibrate on a known byte count in your own access pattern").  Every kernel reads a
// (256, 64, 64, 96) fp32 tensor (402,653,184 B) in a known way and nothing else:
//   cal_stream        every byte once, 16 B per lane, fully coalesced (the guide's case: FETCH_SIZE = 1/2 of the bytes)
//   cal_chunks<4>     the conv's pixel staging (per 4-row work item, six 16-channel chunks: 4 lanes cover the 64 B of a
//                     pixel's chunk, lane groups 384 B apart, 66 columns with the border columns clamped), WITHOUT the halo
//                     rows: every byte of the tensor once
//   cal_chunks<6>     the same with the halo rows (rows y0 - 1 .. y0 + 4): what the conv issues; 1.5 x the bytes if no halo row
//                     is shared in L2, 1.0 x if all are
//   cal_residual      the conv's epilogue read of the residual tensor: 16 B per lane, lanes = 32 pixels x 2 halves of 32 B
// Items are walked as the conv walks them (XCD x owns a contiguous range, its 64 workgroups take consecutive items).
//   hipcc -O3 --offload-arch=gfx950 tools/fetchcal.hip -o tools/fetchcal
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- tools/fetchcal
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int N = 256, HW = 64, C = 96, TH = 4, HWP = 66;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void cal_stream(const float* in, size_t nvec, float* sink)
{
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) s += ((const f32x4*)in)[i];
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = 1.f;
}

template <int ROWS>
__global__ __launch_bounds__(256, 2) void cal_chunks(const float* in, int nitems, float* sink)
{
    constexpr int ITEMS = ROWS * HWP * 4, REGS = (ITEMS + 255) / 256, R0 = ROWS == 6 ? -1 : 0;
    const int tid = threadIdx.x;
    const int per_xcd = nitems >> 3, wg_per_xcd = (int)gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int item_end = (xcd + 1) * per_xcd;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int item = xcd * per_xcd + ((int)blockIdx.x >> 3); item < item_end; item += wg_per_xcd) {
        const int ty = item % (HW / TH), n = item / (HW / TH);
        const float* img = in + (size_t)n * HW * HW * C;
        for (int c = 0; c < 6; ++c) {
            f32x4 v[REGS];
#pragma unroll
            for (int k = 0; k < REGS; ++k) {
                const int i = min(tid + k * 256, ITEMS - 1);
                const int pix = i >> 2, q = i & 3, r = pix / HWP, cc = pix - r * HWP;
                const int gy = min(max(ty * TH + R0 + r, 0), HW - 1), gx = min(max(cc - 1, 0), HW - 1);
                v[k] = *(const f32x4*)(img + (size_t)(gy * HW + gx) * C + c * 16 + q * 4);
            }
#pragma unroll
            for (int k = 0; k < REGS; ++k) s += v[k];
            __syncthreads();                                      // (the conv's chunks are a barrier apart)
        }
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = 1.f;
}

__global__ __launch_bounds__(256, 2) void cal_residual(const float* in, int nitems, float* sink)
{
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, h = lane >> 5, p = lane & 31;
    const int per_xcd = nitems >> 3, wg_per_xcd = (int)gridDim.x >> 3, xcd = blockIdx.x & 7;
    const int item_end = (xcd + 1) * per_xcd;
    f32x4 s = {0.f, 0.f, 0.f, 0.f};
    for (int item = xcd * per_xcd + ((int)blockIdx.x >> 3); item < item_end; item += wg_per_xcd) {
        const int ty = item % (HW / TH), n = item / (HW / TH);
        const float* res = in + (((size_t)n * HW + ty * TH + w) * HW + p) * C + 4 * h;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int cot = 0; cot < 3; ++cot)
#pragma unroll
                for (int G4 = 0; G4 < 4; ++G4) s += *(const f32x4*)(res + (size_t)(32 * mt) * C + 32 * cot + 8 * G4);
    }
    if (s.x + s.y + s.z + s.w == 12345.678f) sink[0] = 1.f;
}

int main()
{
    const size_t elems = (size_t)N * HW * HW * C;
    float *in, *sink;
    CHECK(hipMalloc(&in, elems * 4));
    CHECK(hipMalloc(&sink, 256));
    CHECK(hipMemset(in, 0, elems * 4));
    const int nitems = N * (HW / TH);
    for (int rep = 0; rep < 3; ++rep) {
        hipLaunchKernelGGL(cal_stream, dim3(2048), dim3(256), 0, 0, in, elems / 4, sink);
        hipLaunchKernelGGL(cal_chunks<4>, dim3(512), dim3(256), 0, 0, in, nitems, sink);
        hipLaunchKernelGGL(cal_chunks<6>, dim3(512), dim3(256), 0, 0, in, nitems, sink);
        hipLaunchKernelGGL(cal_residual, dim3(512), dim3(256), 0, 0, in, nitems, sink);
    }
    CHECK(hipDeviceSynchronize());
    printf("tensor bytes %zu\n", elems * 4);
    return 0;
}
