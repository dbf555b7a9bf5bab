// Does v_mfma_f32_32x32x16_f16 keep fp16 subnormal inputs?  A = 2^-20 (subnormal in fp16), B = 2^10 -> 16 products of 2^-10 each.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f16v __attribute__((ext_vector_type(16)));
__global__ void k(float *out, float av, float bv)
{
    h8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)av; b[i] = (_Float16)bv; }
    f16v c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; }
}
int main()
{
    float *d, h[2];
    hipMalloc(&d, 8);
    const float av[] = {9.5367431640625e-07f /* 2^-20 */, 5.9604644775390625e-08f /* 2^-24 */, 6.103515625e-05f /* 2^-14 */};
    for (float a : av) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, a, 1024.f);
        hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
        printf("a=%g (as fp16 %g) x 1024 x 16 terms -> %g (expected %g)\n", a, h[1], h[0], a * 1024.f * 16.f);
    }
    return 0;
}
