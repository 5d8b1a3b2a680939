// Microbenchmark: what does a logsumexp epilogue (v_sub / v_exp / v_add per accumulator element) cost beside
// v_mfma_f32_32x32x16_f16 on gfx950, with one or two waves per SIMD?  Per gap: one MFMA (two alternating accumulator
// chains), then NE elements of {sub, exp2, add}.  SRC 0: the elements are plain registers; SRC 1: they are the
// accumulators the previous round of MFMAs wrote (two sets that swap roles, like k_gmm_f16).
// Prints SIMD cycles per MFMA (wall cycles / MFMAs issued on that SIMD).  Development tool (not part of libsrx).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NE, int SRC, int WAVES, int TRANS, int PIPE>
__global__ __launch_bounds__(64 * WAVES) void k(float *out, unsigned long long *cyc, int iters) {
    f32x16 A[2] = {{0}, {0}}, B[2] = {{0}, {0}};
    f16x8 x, y;
    for (int i = 0; i < 8; i++) { x[i] = (_Float16)(threadIdx.x * 1e-3f); y[i] = (_Float16)1.0f; }
    float plain[16];
    for (int i = 0; i < 16; i++) plain[i] = -(float)i;
    float mx = 0.25f, sum0 = 0.0f, sum1 = 0.0f;
    float E[NE + 1] = {0}, Dd[NE + 1] = {0};
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    auto round = [&](f32x16 (&acc)[2], const f32x16 (&accp)[2]) {
#pragma unroll
        for (int i = 0; i < 16; i++) {   // 16 gaps
            acc[i & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, y, acc[i & 1], 0, 0, 0);
            if (PIPE) {
                // software-pipelined: each op consumes what the previous gap produced (no dependent back-to-back issue)
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    if (e & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(E[e])); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(E[e]));
                }
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    if (TRANS) asm volatile("v_exp_f32 %0, %1" : "=v"(E[e]) : "v"(Dd[e])); else asm volatile("v_mul_f32 %0, %1, %1" : "=v"(E[e]) : "v"(Dd[e]));
                }
#pragma unroll
                for (int e = 0; e < NE; e++) {
                    const int idx = (i * NE + e) % 32;
                    const float a = SRC ? accp[idx / 16][idx % 16] : plain[idx % 16];
                    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(Dd[e]) : "v"(a), "v"(mx));
                }
            } else
#pragma unroll
            for (int e = 0; e < NE; e++) {
                const int idx = (i * NE + e) % 32;
                const float a = SRC ? accp[idx / 16][idx % 16] : plain[idx % 16];
                float d;
                asm volatile("v_sub_f32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(mx));
                if (TRANS) asm volatile("v_exp_f32 %0, %0" : "+v"(d)); else asm volatile("v_mul_f32 %0, %0, %0" : "+v"(d));
                if (e & 1) asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum1) : "v"(d)); else asm volatile("v_add_f32 %0, %0, %1" : "+v"(sum0) : "v"(d));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    for (int it = 0; it < iters; it++) {
        round(A, B);
        round(B, A);
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    float s = sum0 + sum1;
    for (int i = 0; i < 16; i++) s += A[0][i] + A[1][i] + B[0][i] + B[1][i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    // block 0, first to start .. last to finish (an older wave can starve its partner, so one wave's own time misleads)
    if ((threadIdx.x & 63) == 0 && blockIdx.x == 0) { atomicMin(&cyc[0], t0); atomicMax(&cyc[1], t1); }
}

template <int NE, int SRC, int WAVES, int TRANS, int PIPE = 0>
void run(float *out, unsigned long long *cyc) {
    const int iters = 1000;
    unsigned long long init[2] = {~0ull, 0ull}, r[2];
    hipMemcpy(cyc, init, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL((k<NE, SRC, WAVES, TRANS, PIPE>), dim3(256), dim3(64 * WAVES), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipMemcpy(r, cyc, 16, hipMemcpyDeviceToHost);
    const unsigned long long c = r[1] - r[0];
    // s_memtime counts at 100 MHz on gfx950?  report raw units per MFMA too
    printf("%s waves/SIMD %d  elems/gap %d  src %s  %s : %.2f memtime units per SIMD-MFMA\n", PIPE ? "pipelined" : "chained  ", WAVES / 4, NE, SRC ? "acc " : "plain",
           TRANS ? "exp" : "mul", (double)c / (iters * 32.0 * (WAVES / 4)));
}

int main() {
    float *out;
    unsigned long long *cyc;
    hipMalloc(&out, 256 * 512 * 4);
    hipMalloc(&cyc, 16);
    run<0, 0, 4, 1>(out, cyc);
    run<0, 0, 4, 1>(out, cyc);
    run<1, 0, 4, 1>(out, cyc);
    run<2, 0, 4, 1>(out, cyc);
    run<1, 1, 4, 1>(out, cyc);
    run<2, 1, 4, 1>(out, cyc);
    run<2, 1, 4, 0>(out, cyc);
    run<0, 0, 8, 1>(out, cyc);
    run<1, 0, 8, 1>(out, cyc);
    run<2, 0, 8, 1>(out, cyc);
    run<1, 1, 8, 1>(out, cyc);
    run<2, 1, 8, 1>(out, cyc);
    run<2, 1, 8, 0>(out, cyc);
    run<3, 1, 8, 1>(out, cyc);
    run<1, 1, 4, 1, 1>(out, cyc);
    run<2, 1, 4, 1, 1>(out, cyc);
    run<3, 1, 4, 1, 1>(out, cyc);
    run<1, 1, 8, 1, 1>(out, cyc);
    run<2, 1, 8, 1, 1>(out, cyc);
    run<3, 1, 8, 1, 1>(out, cyc);
    return 0;
}
