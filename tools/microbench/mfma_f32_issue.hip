// Microbenchmark: how many VALU / LDS-read issue cycles hide under v_mfma_f32_32x32x2_f32 on gfx950?
// One wave per SIMD (256 threads per block, one block per CU).  Prints cycles per MFMA for N filler
// instructions per group of 4 independent MFMAs.  Development tool (not part of libsrx).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int NL, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += blockDim.x) lds[i] = i * 1e-6f;
    __syncthreads();
    f32x16 a0 = {0}, a1 = {0}, a2 = {0}, a3 = {0};
    float x = threadIdx.x * 1e-3f, y = 1.0f, f0 = 0.5f, f1 = 0.25f, f2 = 0.125f, f3 = 0.0625f;
    const unsigned paddr = (unsigned)(size_t)(lds) + (threadIdx.x & 63) * 16;
    float4 q0 = {0, 0, 0, 0};
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int rep = 0; rep < 4; rep++) {
            a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, a3, 0, 0, 0);
#pragma unroll
            for (int v = 0; v < NV; v++) {
                if ((v & 3) == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f0));
                if ((v & 3) == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f1));
                if ((v & 3) == 2) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f2));
                if ((v & 3) == 3) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(f3));
            }
#pragma unroll
            for (int l = 0; l < NL; l++) {
                f32x4 q;
                asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(q) : "v"(paddr), "n"(l * 1024));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)");
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = f0 + f1 + f2 + f3 + q0.x;
    for (int i = 0; i < 16; i++) s += a0[i] + a1[i] + a2[i] + a3[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int NV, int NL, int WAVES>
void run(float *out, unsigned long long *cyc) {
    const int iters = 2000;
    hipLaunchKernelGGL((k<NV, NL, WAVES>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long c;
    hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    printf("waves/SIMD %d  VALU/4mfma %2d  ds_read_b128/4mfma %d : %.1f cycles per MFMA (per wave)\n", WAVES / 4, NV, NL,
           (double)c / (iters * 16.0));
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 8);
    run<0, 0, 4>(out, cyc);
    run<2, 0, 4>(out, cyc);
    run<4, 0, 4>(out, cyc);
    run<8, 0, 4>(out, cyc);
    run<16, 0, 4>(out, cyc);
    run<32, 0, 4>(out, cyc);
    run<0, 1, 4>(out, cyc);
    run<0, 2, 4>(out, cyc);
    run<0, 4, 4>(out, cyc);
    run<4, 2, 4>(out, cyc);
    run<0, 0, 8>(out, cyc);
    run<8, 0, 8>(out, cyc);
    run<16, 0, 8>(out, cyc);
    run<0, 2, 8>(out, cyc);
    run<4, 2, 8>(out, cyc);
    return 0;
}
