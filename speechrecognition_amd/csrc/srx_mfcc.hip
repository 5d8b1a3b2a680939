// MFCC extraction on gfx950.
//
// Replaces (reference, paths relative to /root/reference):
//   Clip::rmsAmplitude / normalizeVolume       src/clip.cpp:151-164      -> k_sumsq, k_scale
//   Clip::preemphasis                          src/clip.cpp:199-205      -> fused into the frame load (and k_preemph)
//   hamming_window via Clip::operator[] const  src/utils.cpp:21-23, src/clip.cpp:123-125 -> window table
//   Clip::partial_dft                          src/clip.cpp:174-194      -> GEMM1 (f32 MFMA) + magnitude
//   filterbank::operator()                     src/mfcc.cpp:143-149,161-167 -> GEMM2
//   log(T*,int)                                src/utils.hpp:55-60       -> logf(x+1) epilogue
//   dct                                        src/mfcc.cpp:205-214      -> GEMM3
//   write_features frame loop                  src/mfcc.cpp:261-271      -> one 64-frame tile per workgroup
//
// Layout: a workgroup (4 waves) owns 64 consecutive frames of the ragged batch; wave w owns rows
// 16w..16w+15.  The windowed half-frames (only the first N/2 samples of a window enter the
// reference's transform) are staged once in LDS and are the A operand of v_mfma_f32_16x16x4_f32
// against the [N/2][2*NB] cos|sin table streamed through LDS in 32-row chunks.
#include <cmath>

#include "srx_mfcc_plan.h"

float srx_filter_get(const srx_filter &f, int bin);  // srx_ctx.hip

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ------------------------------------------------------------------------------------------
// K1: per-utterance sum of squares (f32 square, f64 accumulate) and gain target/rms.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void k_sumsq_scale(const float *__restrict__ x, const int64_t *__restrict__ soff,
                                                      float target, float *__restrict__ scale) {
    const int u = blockIdx.x;
    const int64_t b = soff[u], e = soff[u + 1];
    double acc = 0.0;
    // One block streams a whole utterance: 128-bit loads on the 16-byte-aligned middle of it, four in flight per thread (64
    // bytes per lane), the unaligned head and tail (at most three samples each) by the first threads.  (Scalar loads, four
    // in flight, reached 3.1 TB/s.)
    const float *p = x + b;
    const int64_t n = e - b;
    int64_t head = (int64_t)((4 - ((reinterpret_cast<uintptr_t>(p) >> 2) & 3)) & 3);
    head = head < n ? head : n;
    if ((int64_t)threadIdx.x < head) {
        const float s = p[threadIdx.x];
        const float sq = s * s;
        acc += (double)sq;
    }
    const float4 *v = reinterpret_cast<const float4 *>(p + head);
    const int64_t nv = (n - head) >> 2;
    const int64_t st = blockDim.x;
    int64_t i = threadIdx.x;
    auto add4 = [&](const float4 q) {
        const float q0 = q.x * q.x, q1 = q.y * q.y, q2 = q.z * q.z, q3 = q.w * q.w;
        acc += (double)q0;
        acc += (double)q1;
        acc += (double)q2;
        acc += (double)q3;
    };
    for (; i + 3 * st < nv; i += 4 * st) {
        const float4 a0 = v[i], a1 = v[i + st], a2 = v[i + 2 * st], a3 = v[i + 3 * st];
        add4(a0);
        add4(a1);
        add4(a2);
        add4(a3);
    }
    for (; i < nv; i += st) add4(v[i]);
    const int64_t done = head + nv * 4;
    if (done + (int64_t)threadIdx.x < n) {
        const float s = p[done + threadIdx.x];
        const float sq = s * s;
        acc += (double)sq;
    }
    // wave reduce
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[16];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) part[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += part[i];
        float cur = (float)sqrt(t);
        scale[u] = target / cur;
    }
}

// Clip::rmsAmplitude (src/clip.cpp:157-164): sqrt of the f64 sum of f32 squares, one block per utterance.
__global__ __launch_bounds__(1024) void k_rms(const float *__restrict__ x, const int64_t *__restrict__ soff, double *__restrict__ rms) {
    const int u = blockIdx.x;
    const int64_t b = soff[u], e = soff[u + 1];
    double acc = 0.0;
    for (int64_t i = b + threadIdx.x; i < e; i += blockDim.x) {
        float s = x[i];
        float sq = s * s;
        acc += (double)sq;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[16];
    const int w = threadIdx.x >> 6, l = threadIdx.x & 63;
    if (l == 0) part[w] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int i = 0; i < (int)(blockDim.x >> 6); i++) t += part[i];
        rms[u] = sqrt(t);
    }
}

__global__ void k_apply_scale(float *__restrict__ x, const int64_t *__restrict__ soff, const float *__restrict__ scale) {
    const int u = blockIdx.y;
    const int64_t b = soff[u], e = soff[u + 1];
    const float f = scale[u];
    for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x)
        x[i] *= f;
}

// Clip::preemphasis out of place (in -> tmp), then copied back by the caller.
__global__ void k_preemph(const float *__restrict__ x, float *__restrict__ y, const int64_t *__restrict__ soff, float alpha) {
    const int u = blockIdx.y;
    const int64_t b = soff[u], e = soff[u + 1];
    for (int64_t i = b + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < e; i += (int64_t)gridDim.x * blockDim.x) {
        int64_t p = i - b;
        int64_t q = p == 0 ? 1 : p;  // x[0] = x[1] (new value)
        float v = x[b + q];
        float m = alpha * x[b + q - 1];
        y[i] = v - m;
    }
}

// ------------------------------------------------------------------------------------------
// The fused MFCC tile kernel.
// ------------------------------------------------------------------------------------------
struct MfccArgs {
    const float *x;
    const int64_t *soff;
    const int64_t *foff;
    const float *scale;  // may be null
    const float *win, *twid, *fbm, *dctm;
    float *ceps;   // may be null (spectrum only)
    float *spec;   // may be null; [total_frames][bin_max]: the output of Clip::partial_dft for every frame
    int bin_max;
    int64_t total_frames;
    int n_utt, hop, half, KH, NB, NFP, n_ceps, nfilt;
    int preemph;
    float alpha, rootn_inv;
    int ldA, ldM;
};

static constexpr int TILE_F = 64;     // frames per workgroup
static constexpr int KC = 32;         // twiddle rows staged per chunk
static constexpr int GB = 64;         // bins per group (4 MFMA tiles)
static constexpr int LDW = 2 * GB + 16;  // 144: (cos|sin) columns + pad (== 16 mod 32 -> conflict-free B reads)

__device__ __forceinline__ int find_utt(const int64_t *foff, int n_utt, int64_t f) {
    int lo = 0, hi = n_utt;  // foff[lo] <= f < foff[hi]
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (foff[mid] <= f) lo = mid; else hi = mid;
    }
    return lo;
}

__global__ __launch_bounds__(256) void k_mfcc_tile(MfccArgs a) {
    extern __shared__ float lds[];
    float *A = lds;                              // [64][ldA]   windowed half-frames; later MAG [64][ldM] and LF
    float *Wc = lds + TILE_F * (a.ldA > a.ldM ? a.ldA : a.ldM);  // [KC][LDW]
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int64_t F0 = (int64_t)blockIdx.x * TILE_F;

    // ---- stage A: each wave loads its 16 rows ----
    for (int r = 0; r < 16; r++) {
        const int row = 16 * w + r;
        const int64_t f = F0 + row;
        float *Ar = A + row * a.ldA;
        if (f < a.total_frames) {
            const int u = find_utt(a.foff, a.n_utt, f);
            const int64_t ub = a.soff[u];
            const int64_t p0 = (f - a.foff[u]) * (int64_t)a.hop;
            const float sc = a.scale ? a.scale[u] : 1.0f;
            for (int n = lane; n < a.KH; n += 64) {
                float y = 0.0f;
                if (n < a.half) {
                    int64_t p = p0 + n;
                    float v;
                    if (a.preemph) {
                        int64_t q = p == 0 ? 1 : p;
                        float x1 = a.x[ub + q], x0 = a.x[ub + q - 1];
                        if (a.scale) { x1 *= sc; x0 *= sc; }
                        float m = a.alpha * x0;
                        v = x1 - m;
                    } else {
                        v = a.x[ub + p];
                        if (a.scale) v *= sc;
                    }
                    y = v * a.win[n];
                }
                Ar[n] = y;
            }
        } else {
            for (int n = lane; n < a.KH; n += 64) Ar[n] = 0.0f;
        }
    }
    __syncthreads();

    // ---- GEMM1 + magnitude, one group of 64 bins at a time; results parked in registers until
    //      A is no longer needed, so MAG can alias A.  NB/64 groups <= 4 supported (256 bins).
    const int ngroups = a.NB / GB;
    const int arow = 16 * w + (lane & 15);
    const int kq = lane >> 4;
    float mag[4][4][4];  // [group][tile][reg]   (groups beyond ngroups unused)
#pragma unroll
    for (int g = 0; g < 4; g++) {
        if (g < ngroups) {
            f32x4 cre[4], cim[4];
#pragma unroll
            for (int t = 0; t < 4; t++) { cre[t] = (f32x4){0, 0, 0, 0}; cim[t] = (f32x4){0, 0, 0, 0}; }
            for (int k0 = 0; k0 < a.KH; k0 += KC) {
                const int kc = (a.KH - k0) < KC ? (a.KH - k0) : KC;
                __syncthreads();  // previous chunk fully consumed
                // stage twiddle rows k0..k0+kc-1, columns [g*64, g*64+64) of cos and of sin
                for (int idx = tid; idx < kc * 2 * GB; idx += 256) {
                    int kr = idx / (2 * GB), c = idx - kr * 2 * GB;
                    int part = c / GB, cc = c - part * GB;
                    Wc[kr * LDW + c] = a.twid[((int64_t)(k0 + kr) * 2 + part) * a.NB + g * GB + cc];
                }
                __syncthreads();
                for (int ks = 0; ks < kc; ks += 4) {
                    float av = A[arow * a.ldA + k0 + ks + kq];
                    const float *wr = Wc + (ks + kq) * LDW + (lane & 15);
#pragma unroll
                    for (int t = 0; t < 4; t++) {
                        float bc = wr[t * 16];
                        float bs = wr[GB + t * 16];
                        cre[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bc, cre[t], 0, 0, 0);
                        cim[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bs, cim[t], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    float re = cre[t][r], im = cim[t][r];
                    float s2 = re * re;
                    float i2 = im * im;
                    mag[g][t][r] = sqrtf(s2 + i2) * a.rootn_inv;
                }
        }
    }
    __syncthreads();  // everyone done reading A
    float *MAG = lds;  // [64][ldM]
#pragma unroll
    for (int g = 0; g < 4; g++)
        if (g < ngroups)
#pragma unroll
            for (int t = 0; t < 4; t++)
#pragma unroll
                for (int r = 0; r < 4; r++)
                    MAG[(16 * w + kq * 4 + r) * a.ldM + g * GB + t * 16 + (lane & 15)] = mag[g][t][r];
    __syncthreads();
    if (a.spec) {   // Clip::partial_dft's out[0 .. bin_max) of every frame of the tile (src/clip.cpp:174-194)
        for (int idx = tid; idx < TILE_F * a.bin_max; idx += 256) {
            const int row = idx / a.bin_max, b = idx - row * a.bin_max;
            const int64_t f = F0 + row;
            if (f < a.total_frames) a.spec[f * a.bin_max + b] = MAG[row * a.ldM + b];
        }
        if (!a.ceps) return;   // (uniform: no barrier is skipped by part of the workgroup)
    }

    // ---- GEMM2: filterbank  [16 rows][NB] x [NB][NFP], NFP <= 64 (4 tiles) ----
    const int nft = a.NFP / 16;
    f32x4 fbacc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) fbacc[t] = (f32x4){0, 0, 0, 0};
    for (int k = 0; k < a.NB; k += 4) {
        float av = MAG[arow * a.ldM + k + kq];
        const float *br = a.fbm + (int64_t)(k + kq) * a.NFP + (lane & 15);
#pragma unroll
        for (int t = 0; t < 4; t++)
            if (t < nft) fbacc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, br[t * 16], fbacc[t], 0, 0, 0);
    }
    __syncthreads();  // MAG consumed
    // log(x+1) and park as LF [64][ldL], ldL = NFP + 2
    float *LF = lds;
    const int ldL = a.NFP + 2;
#pragma unroll
    for (int t = 0; t < 4; t++)
        if (t < nft)
#pragma unroll
            for (int r = 0; r < 4; r++) LF[(16 * w + kq * 4 + r) * ldL + t * 16 + (lane & 15)] = logf(fbacc[t][r] + 1.0f);
    __syncthreads();

    // ---- GEMM3: DCT  [16 rows][NFP] x [NFP][16] ----
    f32x4 cacc = (f32x4){0, 0, 0, 0};
    for (int k = 0; k < a.NFP; k += 4) {
        float av = LF[arow * ldL + k + kq];
        float bv = a.dctm[(k + kq) * 16 + (lane & 15)];
        cacc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, cacc, 0, 0, 0);
    }
    const int j = lane & 15;
    if (j < a.n_ceps) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int64_t f = F0 + 16 * w + kq * 4 + r;
            if (f < a.total_frames) a.ceps[f * a.n_ceps + j] = cacc[r];   // (a.ceps is non-null here: the spectrum-only call returned)
        }
    }
}

// ------------------------------------------------------------------------------------------
// Plan
// ------------------------------------------------------------------------------------------
extern "C" int srx_mfcc_plan_create(srx_ctx *ctx, const srx_mfcc_cfg *cfg, const srx_filter *fb, int n_filters,
                                    srx_mfcc_plan **out) {
    SRX_ENTER(ctx, "srx_mfcc_plan_create");
    SRX_REQUIRE(ctx, cfg && fb && out, "srx_mfcc_plan_create: null argument");
    SRX_REQUIRE(ctx, cfg->window_len >= 8 && cfg->frame_shift >= 1, "srx_mfcc_plan_create: bad window/shift");
    SRX_REQUIRE(ctx, cfg->n_ceps >= 1 && cfg->n_ceps <= 16, "srx_mfcc_plan_create: n_ceps must be 1..16");
    SRX_REQUIRE(ctx, n_filters >= 1 && n_filters <= 64, "srx_mfcc_plan_create: n_filters must be 1..64");
    const int N = cfg->window_len;
    const int bin_max = (int)(fb[n_filters - 1].center + fb[n_filters - 1].width) + 1;
    SRX_REQUIRE(ctx, bin_max >= 1 && bin_max <= 256, "srx_mfcc_plan_create: bin_max %d outside 1..256", bin_max);
    SRX_REQUIRE(ctx, bin_max <= N, "srx_mfcc_plan_create: bin_max %d exceeds window %d", bin_max, N);
    for (int i = 0; i < n_filters; i++)
        SRX_REQUIRE(ctx, (int)(fb[i].center - fb[i].width) >= 0 && (int)(fb[i].center + fb[i].width) < bin_max,
                    "srx_mfcc_plan_create: filter %d reaches outside [0,bin_max)", i);

    srx_mfcc_plan *p = new srx_mfcc_plan();
    p->ctx = ctx;
    p->cfg = *cfg;
    p->nfilt = n_filters;
    p->bin_max = bin_max;
    p->half = N / 2;
    p->KH = (p->half + 3) / 4 * 4;
    p->NB = (bin_max + GB - 1) / GB * GB;
    p->NFP = (n_filters + 15) / 16 * 16;
    p->rootn_inv = (float)(1.0f / sqrt((double)N));  // src/clip.cpp:176

    // Window (src/utils.cpp:21-23): f64 cosine, rounded to f32 once.
    std::vector<float> win(p->KH, 0.0f);
    for (int n = 0; n < p->half; n++) win[n] = (float)(0.54f + 0.46 * cos(2 * M_PI * n / (N - 1)));
    // Twiddles (src/clip.cpp:182-183): angle expression as written, in f64; mirrored write
    // (src/clip.cpp:186-192): out[j] ends up holding the magnitude of frequency max(j, N-1-j)
    // for j >= N-len when len > N/2.
    std::vector<float> tw((size_t)p->KH * 2 * p->NB, 0.0f);
    const bool mirror = bin_max > N / 2;
    for (int j = 0; j < bin_max; j++) {
        int k = j;
        if (mirror && j >= N - bin_max) k = j > N - 1 - j ? j : N - 1 - j;
        for (int n = 0; n < p->half; n++) {
            double ang = -2 * M_PI * k * n / N;
            tw[((size_t)n * 2 + 0) * p->NB + j] = (float)cos(ang);
            tw[((size_t)n * 2 + 1) * p->NB + j] = (float)sin(ang);
        }
    }
    // Filterbank matrix from filter::get (src/mfcc.cpp:132-140) over each filter's own bin range.
    std::vector<float> fbm((size_t)p->NB * p->NFP, 0.0f);
    for (int i = 0; i < n_filters; i++)
        for (int b = (int)(fb[i].center - fb[i].width); b <= (int)(fb[i].center + fb[i].width); b++)
            fbm[(size_t)b * p->NFP + i] = srx_filter_get(fb[i], b);
    // DCT matrix (src/mfcc.cpp:210): cos(j*(i-0.0f)*M_PI/len), f32 product j*i then f64.
    std::vector<float> dm((size_t)p->NFP * 16, 0.0f);
    for (int jj = 0; jj < cfg->n_ceps; jj++)
        for (int i = 0; i < n_filters; i++) {
            float ji = (float)jj * ((float)i - 0.0f);
            dm[(size_t)i * 16 + jj] = (float)cos((double)ji * M_PI / n_filters);
        }

    auto up = [&](float **d, const std::vector<float> &h) -> hipError_t {
        hipError_t e = hipMalloc((void **)d, h.size() * sizeof(float));
        if (e != hipSuccess) return e;
        return hipMemcpy(*d, h.data(), h.size() * sizeof(float), hipMemcpyHostToDevice);
    };
    hipError_t e;
    if ((e = up(&p->win, win)) != hipSuccess || (e = up(&p->twid, tw)) != hipSuccess ||
        (e = up(&p->fbm, fbm)) != hipSuccess || (e = up(&p->dctm, dm)) != hipSuccess) {
        srx_mfcc_plan_destroy(p);
        return srx_fail(ctx, SRX_ERR_DEVICE, "srx_mfcc_plan_create: table upload failed: %s", hipGetErrorString(e));
    }
    if (srx_mfcc16_supported(p->half, n_filters, cfg->n_ceps, bin_max)) {
        int rc = srx_mfcc16_build(p, win, tw, fbm, dm);
        if (rc != SRX_OK) {
            srx_mfcc_plan_destroy(p);
            return rc;
        }
    }
    if (srx_mfccfft_supported(N, n_filters, cfg->n_ceps)) {
        int rc = srx_mfccfft_build(p, fb);
        if (rc != SRX_OK) {
            srx_mfcc_plan_destroy(p);
            return rc;
        }
    }
    *out = p;
    return SRX_OK;
}

extern "C" int srx_mfcc_plan_set_path(srx_mfcc_plan *p, int path) {
    if (!p) return SRX_ERR_INVALID;
    SRX_REQUIRE(p->ctx, path == SRX_MFCC_AUTO || path == SRX_MFCC_F32 || path == SRX_MFCC_F16X3 || path == SRX_MFCC_FFT,
                "srx_mfcc_plan_set_path: unknown path %d", path);
    SRX_REQUIRE(p->ctx, path != SRX_MFCC_FFT || p->fft_tw,
                "srx_mfcc_plan_set_path: the FFT path covers windows of 256, 320, 400 and 512 samples (half window 8*4*4, 8*5*4, 8*5*5, 8*8*4)");
    SRX_REQUIRE(p->ctx, path != SRX_MFCC_F16X3 || p->ks16 > 0,
                "srx_mfcc_plan_set_path: the split-f16 path covers half windows of 145..160 or 193..208 samples, <= 64 filters, <= 256 bins");
    p->path = path;
    return SRX_OK;
}

extern "C" void srx_mfcc_plan_destroy(srx_mfcc_plan *p) {
    if (!p) return;
    (void)hipFree(p->fft_tw);
    (void)hipFree(p->fft_fbw);
    (void)hipFree(p->fft_fblo);
    (void)hipFree(p->pk16);
    (void)hipFree(p->dct16);
    (void)hipFree(p->win);
    (void)hipFree(p->twid);
    (void)hipFree(p->fbm);
    (void)hipFree(p->dctm);
    delete p;
}

extern "C" int srx_mfcc_plan_bin_max(const srx_mfcc_plan *p) { return p ? p->bin_max : SRX_ERR_INVALID; }

static int round_ld(int n) {  // smallest 2*odd >= n
    int v = n;
    while (!((v % 2 == 0) && ((v / 2) % 2 == 1))) v++;
    return v;
}

static int mfcc_launch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                       const int64_t *frame_off_dev, int64_t total_frames, float *scratch_scale_dev, float *ceps_dev, float *spec_dev) {
    SRX_REQUIRE(ctx, plan && samples_dev && sample_off_dev && frame_off_dev && (ceps_dev || spec_dev), "srx_mfcc_batch_dev: null argument");
    SRX_REQUIRE(ctx, n_utt >= 0 && total_frames >= 0, "srx_mfcc_batch_dev: negative size");
    if (n_utt == 0 || total_frames == 0) return SRX_OK;
    const srx_mfcc_cfg &c = plan->cfg;
    if (c.normalize) {
        SRX_REQUIRE(ctx, scratch_scale_dev, "srx_mfcc_batch_dev: normalize needs scratch_scale_dev");
        {
            SrxProf prof_(ctx, "k_sumsq_scale");
            hipLaunchKernelGGL(k_sumsq_scale, dim3(n_utt), dim3(1024), 0, ctx->stream, samples_dev, sample_off_dev, c.target_rms,
                               scratch_scale_dev);
        }
        SRX_LAUNCH_CHECK(ctx);
    }
    if (!spec_dev && plan->path == SRX_MFCC_FFT)
        return srx_mfccfft_launch(ctx, plan, samples_dev, sample_off_dev, n_utt, frame_off_dev, total_frames,
                                  c.normalize ? scratch_scale_dev : nullptr, ceps_dev);
    if (!spec_dev && plan->ks16 > 0 && plan->path != SRX_MFCC_F32)
        return srx_mfcc16_launch(ctx, plan, samples_dev, sample_off_dev, n_utt, frame_off_dev, total_frames,
                                 c.normalize ? scratch_scale_dev : nullptr, ceps_dev);
    MfccArgs a;
    a.x = samples_dev; a.soff = sample_off_dev; a.foff = frame_off_dev;
    a.scale = c.normalize ? scratch_scale_dev : nullptr;
    a.win = plan->win; a.twid = plan->twid; a.fbm = plan->fbm; a.dctm = plan->dctm;
    a.ceps = ceps_dev; a.spec = spec_dev; a.bin_max = plan->bin_max; a.total_frames = total_frames; a.n_utt = n_utt; a.hop = c.frame_shift;
    a.half = plan->half; a.KH = plan->KH; a.NB = plan->NB; a.NFP = plan->NFP; a.n_ceps = c.n_ceps; a.nfilt = plan->nfilt;
    a.preemph = c.preemphasis; a.alpha = c.alpha; a.rootn_inv = plan->rootn_inv;
    a.ldA = round_ld(plan->KH);
    a.ldM = round_ld(plan->NB);
    const int ldmax = a.ldA > a.ldM ? a.ldA : a.ldM;
    const size_t lds_bytes = ((size_t)TILE_F * ldmax + (size_t)KC * LDW) * sizeof(float);
    SRX_REQUIRE(ctx, lds_bytes <= 160 * 1024, "srx_mfcc_batch_dev: LDS need %zu exceeds 160 KiB", lds_bytes);
    static bool attr_set[SRX_MAX_DEVICES] = {};   // the attribute is per device
    if (!attr_set[ctx->device]) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_mfcc_tile, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device] = true;
    }
    const int64_t nblk = ceil_div64(total_frames, TILE_F);
    SRX_REQUIRE(ctx, nblk < (1ll << 31), "srx_mfcc_batch_dev: too many frames");
    {
        SrxProf prof_(ctx, "k_mfcc_tile");
        hipLaunchKernelGGL(k_mfcc_tile, dim3((unsigned)nblk), dim3(256), lds_bytes, ctx->stream, a);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_mfcc_batch_dev(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev,
                                  const int64_t *sample_off_dev, int n_utt, const int64_t *frame_off_dev,
                                  int64_t total_frames, float *scratch_scale_dev, float *ceps_dev) {
    SRX_ENTER(ctx, "srx_mfcc_batch_dev");
    SRX_REQUIRE(ctx, ceps_dev, "srx_mfcc_batch_dev: null argument");
    return mfcc_launch(ctx, plan, samples_dev, sample_off_dev, n_utt, frame_off_dev, total_frames, scratch_scale_dev, ceps_dev, nullptr);
}

// Clip::partial_dft (src/clip.cpp:174-194) of every frame, stand-alone: the magnitudes the fused kernel keeps in LDS,
// written out (f32 matrix-core path; the plan's normalisation / pre-emphasis settings apply as in srx_mfcc_batch_dev).
extern "C" int srx_partial_dft_batch_dev(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev,
                                         const int64_t *sample_off_dev, int n_utt, const int64_t *frame_off_dev,
                                         int64_t total_frames, float *scratch_scale_dev, float *spec_dev) {
    SRX_ENTER(ctx, "srx_partial_dft_batch_dev");
    SRX_REQUIRE(ctx, spec_dev, "srx_partial_dft_batch_dev: null argument");
    return mfcc_launch(ctx, plan, samples_dev, sample_off_dev, n_utt, frame_off_dev, total_frames, scratch_scale_dev, nullptr, spec_dev);
}

extern "C" int srx_partial_dft_batch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples, const int64_t *sample_off, int n_utt,
                                     float *spec, int64_t *frame_off) {
    SRX_ENTER(ctx, "srx_partial_dft_batch");
    SRX_REQUIRE(ctx, plan && samples && sample_off && spec && frame_off && n_utt >= 0, "srx_partial_dft_batch: bad argument");
    int rc = srx_mfcc_frame_offsets(&plan->cfg, sample_off, n_utt, frame_off);
    if (rc != SRX_OK) return srx_fail(ctx, rc, "%s", srx_last_error(nullptr));
    if (n_utt == 0) return SRX_OK;
    SRX_REQUIRE(ctx, sample_off[0] == 0, "srx_partial_dft_batch: sample_off[0] must be 0");
    const int64_t ns = sample_off[n_utt], nf = frame_off[n_utt];
    DevBuf dx, dso, dfo, dsc, ds;
    SRX_HIP(ctx, dx.alloc(ns * sizeof(float)));
    SRX_HIP(ctx, dso.alloc((n_utt + 1) * sizeof(int64_t)));
    SRX_HIP(ctx, dfo.alloc((n_utt + 1) * sizeof(int64_t)));
    SRX_HIP(ctx, dsc.alloc(n_utt * sizeof(float)));
    SRX_HIP(ctx, ds.alloc(nf * plan->bin_max * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(dx.p, samples, ns * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dso.p, sample_off, (n_utt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dfo.p, frame_off, (n_utt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = srx_partial_dft_batch_dev(ctx, plan, dx.as<float>(), dso.as<int64_t>(), n_utt, dfo.as<int64_t>(), nf, dsc.as<float>(), ds.as<float>());
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(spec, ds.p, nf * plan->bin_max * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

extern "C" int srx_mfcc_batch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples, const int64_t *sample_off,
                              int n_utt, float *ceps, int64_t *frame_off) {
    SRX_ENTER(ctx, "srx_mfcc_batch");
    SRX_REQUIRE(ctx, plan && samples && sample_off && ceps && frame_off && n_utt >= 0, "srx_mfcc_batch: bad argument");
    int rc = srx_mfcc_frame_offsets(&plan->cfg, sample_off, n_utt, frame_off);
    if (rc != SRX_OK) return srx_fail(ctx, rc, "%s", srx_last_error(nullptr));
    if (n_utt == 0) return SRX_OK;
    SRX_REQUIRE(ctx, sample_off[0] == 0, "srx_mfcc_batch: sample_off[0] must be 0");
    const int64_t ns = sample_off[n_utt], nf = frame_off[n_utt];
    DevBuf dx, dso, dfo, dsc, dc;
    SRX_HIP(ctx, dx.alloc(ns * sizeof(float)));
    SRX_HIP(ctx, dso.alloc((n_utt + 1) * sizeof(int64_t)));
    SRX_HIP(ctx, dfo.alloc((n_utt + 1) * sizeof(int64_t)));
    SRX_HIP(ctx, dsc.alloc(n_utt * sizeof(float)));
    SRX_HIP(ctx, dc.alloc(nf * plan->cfg.n_ceps * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(dx.p, samples, ns * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dso.p, sample_off, (n_utt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dfo.p, frame_off, (n_utt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    rc = srx_mfcc_batch_dev(ctx, plan, dx.as<float>(), dso.as<int64_t>(), n_utt, dfo.as<int64_t>(), nf, dsc.as<float>(),
                            dc.as<float>());
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(ceps, dc.p, nf * plan->cfg.n_ceps * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

extern "C" int srx_normalize_volume_dev(srx_ctx *ctx, float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                                        float target_rms, float *scratch_scale_dev) {
    SRX_ENTER(ctx, "srx_normalize_volume_dev");
    SRX_REQUIRE(ctx, samples_dev && sample_off_dev && scratch_scale_dev && n_utt >= 0, "srx_normalize_volume_dev: bad argument");
    if (n_utt == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_sumsq_scale");
        hipLaunchKernelGGL(k_sumsq_scale, dim3(n_utt), dim3(1024), 0, ctx->stream, samples_dev, sample_off_dev, target_rms,
                           scratch_scale_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_apply_scale");
        hipLaunchKernelGGL(k_apply_scale, dim3(64, n_utt), dim3(256), 0, ctx->stream, samples_dev, sample_off_dev, scratch_scale_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_scale_dev(srx_ctx *ctx, float *samples_dev, const int64_t *sample_off_dev, int n_utt, const float *scale_dev) {
    SRX_ENTER(ctx, "srx_scale_dev");
    SRX_REQUIRE(ctx, samples_dev && sample_off_dev && scale_dev && n_utt >= 0, "srx_scale_dev: bad argument");
    if (n_utt == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_apply_scale");
        hipLaunchKernelGGL(k_apply_scale, dim3(64, n_utt), dim3(256), 0, ctx->stream, samples_dev, sample_off_dev, scale_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_rms_amplitude_dev(srx_ctx *ctx, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                                     double *rms_dev) {
    SRX_ENTER(ctx, "srx_rms_amplitude_dev");
    SRX_REQUIRE(ctx, samples_dev && sample_off_dev && rms_dev && n_utt >= 0, "srx_rms_amplitude_dev: bad argument");
    if (n_utt == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_rms");
        hipLaunchKernelGGL(k_rms, dim3(n_utt), dim3(1024), 0, ctx->stream, samples_dev, sample_off_dev, rms_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_preemphasis_dev(srx_ctx *ctx, float *samples_dev, float *scratch_dev, const int64_t *sample_off_dev,
                                   int n_utt, int64_t total_samples, float alpha) {
    SRX_ENTER(ctx, "srx_preemphasis_dev");
    SRX_REQUIRE(ctx, samples_dev && scratch_dev && sample_off_dev && n_utt >= 0 && total_samples >= 0,
                "srx_preemphasis_dev: bad argument");
    if (n_utt == 0 || total_samples == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_preemph");
        hipLaunchKernelGGL(k_preemph, dim3(64, n_utt), dim3(256), 0, ctx->stream, samples_dev, scratch_dev, sample_off_dev, alpha);
    }
    SRX_LAUNCH_CHECK(ctx);
    SRX_HIP(ctx, hipMemcpyAsync(samples_dev, scratch_dev, total_samples * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    return SRX_OK;
}
