// Does v_mfma_f32_32x32x16_f16 honour f16 subnormal inputs on gfx950, and how is its result rounded?
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float *out, float a_val, float b_val) {
    f16x8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = 0; b[i] = 0; }
    // k index 0 of lanes 0..31 only
    if (threadIdx.x < 32) { a[0] = (_Float16)a_val; b[0] = (_Float16)b_val; }
    f32x16 c = {0};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}
int main() {
    float *d; hipMalloc(&d, 4);
    const float tests[][2] = {{9.5367431640625e-07f /*2^-20 subnormal*/, 1024.f}, {5.9604645e-08f /*2^-24 min subnormal*/, 16384.f},
                              {3.0517578125e-05f /*2^-15 subnormal*/, 2.f}, {1.0f, 9.5367431640625e-07f}, {0.5f, 0.5f}};
    for (auto &t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, t[0], t[1]);
        float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
        printf("a=%g b=%g -> %g (expect %g)\n", t[0], t[1], h, t[0] * t[1]);
    }
    return 0;
}
