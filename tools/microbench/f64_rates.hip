// Microbenchmark: issue cost (SIMD cycles per wave instruction) of the f64 / conversion instructions the statistics kernels
// are made of, on gfx950: eight independent chains per lane, one wave per SIMD slot (WAVES waves per workgroup of one CU-full
// grid), s_memtime around a long unrolled loop.  Development tool (not part of libsrx).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/f64_rates.hip -o gpurun_tmp/f64_rates && gpurun_tmp/f64_rates
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ __launch_bounds__(1024) void k(double *out, unsigned long long *cyc, int iters) {
    double a[8];
    float f[8];
    for (int i = 0; i < 8; i++) { a[i] = 1.0 + threadIdx.x * 1e-3 + i; f[i] = (float)a[i]; }
    const double c = 1.0000001, d = 1e-9;
    unsigned long long t0, t1;
    __syncthreads();
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 8; r++) {
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (OP == 0) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a[i]) : "v"(c), "v"(d));
                if (OP == 1) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(d));
                if (OP == 2) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 3) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(f[i]) : "v"(a[i]));
                if (OP == 4) asm volatile("v_cvt_f64_f32 %0, %1" : "=v"(a[i]) : "v"(f[i]));
                if (OP == 5) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (OP == 6) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 7) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
                if (OP == 8) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            }
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    double s = 0;
    for (int i = 0; i < 8; i++) s += a[i] + f[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int OP>
static void run(const char *name, double *out, unsigned long long *cyc) {
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {      // waves per SIMD: workgroups of 256 threads (one wave per SIMD of a CU), wps per CU
        const int nblk = 256 * wps;
        hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
        hipEvent_t e0, e1;
        hipEventCreate(&e0);
        hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<OP>, dim3(nblk), dim3(256), 0, 0, out, cyc, iters);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        const double n = (double)iters * 64 * wps;     // wave instructions issued on one SIMD
        printf("%-16s %d waves/SIMD: %.2f ns per wave instruction on a SIMD\n", name, wps, ms * 1e6 / n);
    }
}

int main() {
    double *out;
    unsigned long long *cyc;
    hipMalloc(&out, 8 * 256 * 256 * 8);
    hipMalloc(&cyc, 8 * 256 * 8);
    run<5>("v_fma_f32", out, cyc);
    run<8>("v_mul_f32", out, cyc);
    run<0>("v_fma_f64", out, cyc);
    run<1>("v_add_f64", out, cyc);
    run<2>("v_mul_f64", out, cyc);
    run<3>("v_cvt_f32_f64", out, cyc);
    run<4>("v_cvt_f64_f32", out, cyc);
    run<6>("v_pk_fma_f32", out, cyc);
    run<7>("v_pk_add_f32", out, cyc);
    // (v_add_f64 has no DPP form on gfx950 -- the assembler refuses row_newbcast on it -- so a 64-bit value cannot be
    //  broadcast inside a 16-lane row for free; two v_mov_b32_dpp would cost more than the LDS read they replace)
    return 0;
}
