// Microbenchmark: does a ds_read_b128 stream (one read per NG MFMA gaps, results feeding later MFMAs or not) cost
// v_mfma_f32_32x32x16_f16 pipe time on gfx950?  8 waves per CU (2 per SIMD), like k_gmm_f16.  Development tool.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NG, int USE, int DISTINCT>
__global__ __launch_bounds__(512) void k(float *out, unsigned long long *cyc, int iters) {
    __shared__ __attribute__((aligned(16))) _Float16 lds[40960];
    for (int i = threadIdx.x; i < 40960; i += blockDim.x) lds[i] = (_Float16)(i * 1e-4f);
    __syncthreads();
    f32x16 A[2] = {{0}, {0}};
    f16x8 x[10], y[20];
    for (int s = 0; s < 10; s++) for (int i = 0; i < 8; i++) x[s][i] = (_Float16)(threadIdx.x * 1e-3f + s);
    for (int s = 0; s < 20; s++) for (int i = 0; i < 8; i++) y[s][i] = (_Float16)(1.0f + s);
    typedef const __attribute__((address_space(3))) f16x8 *lp;
    unsigned base = (threadIdx.x & 63) * 16;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
        f16x8 nx[10];
#pragma unroll
        for (int i = 0; i < 30; i++) {
            const int s = i % 10;
            A[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x[DISTINCT ? s : 0], y[DISTINCT ? (i % 20) : 0], A[i & 1], 0, 0, 0);
            if (NG > 0 && i % NG == 0 && i / NG < 10) {
                asm volatile("" : "+v"(base));
                nx[i / NG] = *(lp)(size_t)(base + (i / NG) * 1024 + (it & 1) * 10240);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        if (NG > 0 && USE) {
#pragma unroll
            for (int s = 0; s < 10; s++) if (30 / NG > s) x[s] = nx[s];
        } else if (NG > 0) {
#pragma unroll
            for (int s = 0; s < 10; s++) if (30 / NG > s) asm volatile("" ::"v"(nx[s]));
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float sum = 0;
    for (int i = 0; i < 16; i++) sum += A[0][i] + A[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = sum;
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMin(&cyc[0], t0); atomicMax(&cyc[1], t1); }
}

template <int NG, int USE, int DISTINCT>
void run(float *out, unsigned long long *cyc) {
    const int iters = 1000;
    unsigned long long init[2] = {~0ull, 0ull}, r[2];
    hipMemcpy(cyc, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<NG, USE, DISTINCT>), dim3(256), dim3(512), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(r, cyc, 16, hipMemcpyDeviceToHost);
    printf("ds_read_b128 every %d gaps (0 = none), results %s, operands %s: %.2f cycles per SIMD-MFMA\n", NG, USE ? "used as next A" : "discarded",
           DISTINCT ? "distinct registers" : "one register pair", (double)(r[1] - r[0]) / (iters * 30.0 * 2));
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16);
    run<0, 0, 0>(out, cyc);
    run<0, 0, 0>(out, cyc);
    run<0, 0, 1>(out, cyc);
    run<3, 0, 0>(out, cyc);
    run<3, 0, 1>(out, cyc);
    run<3, 1, 1>(out, cyc);
    run<1, 1, 1>(out, cyc);
    return 0;
}
