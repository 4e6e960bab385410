// tools/f16denorm.cpp -- does v_mfma_f32_32x32x16_f16 keep f16 subnormal inputs?  (dev probe, not part of the product)
//   hipcc --offload-arch=gfx950 -O2 tools/f16denorm.cpp -o tools/bin/f16denorm && tools/bin/f16denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float a_val, float b_val)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)a_val; b[i] = (_Float16)b_val; }
    f16v c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main()
{
    float* d; hipMalloc(&d, 4);
    const float vals[] = {1.0f, 1e-4f, 3e-5f, 1e-6f, 6e-8f};
    for (float v : vals) {
        k<<<1, 64>>>(d, v, 1.0f);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a = %g (f16 %s): sum over k=16 of a*1 = %g (expected %g)\n", v, v < 6.1e-5f ? "subnormal" : "normal", h,
               16.0 * (double)(float)(_Float16)v);
    }
    // subnormal x subnormal-free large: 3e-5 * 1024
    k<<<1, 64>>>(d, 3e-5f, 1024.f);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("3e-5 * 1024 * 16 = %g (expected %g)\n", h, 16.0 * 1024 * (double)(float)(_Float16)3e-5f);
    return 0;
}
