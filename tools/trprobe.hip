// tools/trprobe.hip -- what ds_read_b64_tr_b16 delivers (gfx950): LDS image [16 rows][32 cols] of f16 = row * 32 + col,
// pitch 64 B; lane l = 16 g + 4 q + p supplies the address of row 8 (g >> 1) + q, columns 16 (g & 1) + 4 p .. + 3.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s4v __attribute__((__vector_size__(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
__global__ void k(float* out)
{
    __shared__ __attribute__((aligned(16))) _Float16 img[16 * 32];
    for (int i = threadIdx.x; i < 512; i += 64) img[i] = (_Float16)(float)i;
    __syncthreads();
    const int l = threadIdx.x, g = l >> 4, q = (l >> 2) & 3, p = l & 3;
    const _Float16* a = img + (8 * (g >> 1) + q) * 32 + 16 * (g & 1) + 4 * p;
    s4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s4v*)a);
    h4 v = __builtin_bit_cast(h4, r);
    for (int e = 0; e < 4; ++e) out[l * 4 + e] = (float)v[e];
}
int main()
{
    float* d; float h[256];
    hipMalloc(&d, sizeof h);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) {
        printf("lane %2d:", l);
        for (int e = 0; e < 4; ++e) printf(" (r%d,c%d)", (int)h[l * 4 + e] / 32, (int)h[l * 4 + e] % 32);
        printf("\n");
    }
    return 0;
}
