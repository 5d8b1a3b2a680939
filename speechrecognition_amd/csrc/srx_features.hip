// Feature post-processing on gfx950: 5-tap (2*WL+1) triangular smoothing with edge renormalisation,
// then clamped central differences for deltas and delta-deltas.
//
// Replaces read_features' smoothing loop (src/mfcc.cpp:296-311) and compute_deltas /
// featurevec::compute_deltas / compute_delta2s (src/mfcc.cpp:73-102).  Every f32 operation is
// performed in the reference's order with no FMA contraction, so the result is bit-identical to
// the scalar restatement.  HBM-bound: 4*n_ceps bytes in, 12*n_ceps bytes out per frame.
#include "srx_common.h"

__device__ __forceinline__ float smooth_at(const float *__restrict__ raw, int64_t base, int64_t n, int64_t i, int q,
                                           int nc, int WL) {
    float s = 0.0f;
    int NF = 0;
    for (int w = -WL; w <= WL; w++) {
        int wt = WL - (w < 0 ? -w : w) + 1;
        int64_t ii = i + w;
        if (ii >= 0 && ii < n) {
            NF += wt;
            float prod = (float)wt * raw[(base + ii) * nc + q];
            s = s + prod;
        }
    }
    return s / (float)NF;
}

__device__ __forceinline__ float delta_at(const float *__restrict__ raw, int64_t base, int64_t n, int64_t i, int q, int nc,
                                          int WL) {
    int64_t j = i - 1, k = i + 1;
    if (j < 0) j = 0;
    if (k == n) k--;
    float a = smooth_at(raw, base, n, k, q, nc, WL);
    float b = smooth_at(raw, base, n, j, q, nc, WL);
    float d = a - b;
    return d / (float)(k - j);
}

__global__ __launch_bounds__(256) void k_features(const float *__restrict__ raw, const int64_t *__restrict__ foff, int n_utt,
                                                  int64_t total_frames, int nc, int WL, float *__restrict__ feats) {
    const int64_t gi = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (gi >= total_frames * nc) return;
    const int64_t f = gi / nc;
    const int q = (int)(gi - f * nc);
    int lo = 0, hi = n_utt;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (foff[mid] <= f) lo = mid; else hi = mid;
    }
    const int64_t base = foff[lo];
    const int64_t n = foff[lo + 1] - base;
    const int64_t i = f - base;
    float c = smooth_at(raw, base, n, i, q, nc, WL);
    float d = delta_at(raw, base, n, i, q, nc, WL);
    int64_t j = i - 1, k = i + 1;
    if (j < 0) j = 0;
    if (k == n) k--;
    float dk = delta_at(raw, base, n, k, q, nc, WL);
    float dj = delta_at(raw, base, n, j, q, nc, WL);
    float dd = dk - dj;
    dd = dd / (float)(k - j);
    float *o = feats + f * (3 * nc);
    o[q] = c;
    o[nc + q] = d;
    o[2 * nc + q] = dd;
}

extern "C" int srx_features_batch_dev(srx_ctx *ctx, int WL, int n_ceps, const float *ceps_dev, const int64_t *frame_off_dev,
                                      int n_utt, int64_t total_frames, float *feats_dev) {
    SRX_REQUIRE(nullptr, ctx, "srx_features_batch_dev: null ctx");
    SRX_REQUIRE(ctx, ceps_dev && frame_off_dev && feats_dev, "srx_features_batch_dev: null argument");
    SRX_REQUIRE(ctx, WL >= 0 && WL <= 64 && n_ceps >= 1 && n_utt >= 0 && total_frames >= 0, "srx_features_batch_dev: bad size");
    if (n_utt == 0 || total_frames == 0) return SRX_OK;
    const int64_t nthr = total_frames * n_ceps;
    const int64_t nblk = ceil_div64(nthr, 256);
    SRX_REQUIRE(ctx, nblk < (1ll << 31), "srx_features_batch_dev: too many frames");
    {
        SrxProf prof_(ctx, "k_features");
        hipLaunchKernelGGL(k_features, dim3((unsigned)nblk), dim3(256), 0, ctx->stream, ceps_dev, frame_off_dev, n_utt, total_frames,
                           n_ceps, WL, feats_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_features_batch(srx_ctx *ctx, int WL, int n_ceps, const float *ceps, const int64_t *frame_off, int n_utt,
                                  float *feats) {
    SRX_REQUIRE(nullptr, ctx, "srx_features_batch: null ctx");
    SRX_REQUIRE(ctx, ceps && frame_off && feats && n_utt >= 0 && n_ceps >= 1, "srx_features_batch: bad argument");
    if (n_utt == 0) return SRX_OK;
    const int64_t nf = frame_off[n_utt];
    DevBuf dc, dfo, df;
    SRX_HIP(ctx, dc.alloc(nf * n_ceps * sizeof(float)));
    SRX_HIP(ctx, dfo.alloc((n_utt + 1) * sizeof(int64_t)));
    SRX_HIP(ctx, df.alloc(nf * 3 * n_ceps * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(dc.p, ceps, nf * n_ceps * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dfo.p, frame_off, (n_utt + 1) * sizeof(int64_t), hipMemcpyHostToDevice, ctx->stream));
    int rc = srx_features_batch_dev(ctx, WL, n_ceps, dc.as<float>(), dfo.as<int64_t>(), n_utt, nf, df.as<float>());
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(feats, df.p, nf * 3 * n_ceps * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

// mean(ut) / variance(ut, mu)  src/utterance.cpp:32-65: sums over all frames of the first D dims.
// Two launches: sum_dev[0..D) = sum x ; then sum_dev[D..2D) = sum (x - mu)^2 with mu = f32(sum/T),
// f = f32(x - mu), f*f in f32 as the reference, accumulated in f64 (the reference keeps f32 running
// sums; f64 here is the more accurate of the two -- documented divergence).
__global__ __launch_bounds__(256) void k_colsum(const float *__restrict__ feats, int ld, int64_t T, int D,
                                                const double *__restrict__ mu_sum, double *__restrict__ out, int pass) {
    const int d = blockIdx.y;
    double acc = 0.0;
    float mu = 0.0f;
    if (pass == 1) mu = (float)mu_sum[d] / (float)(int)T;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < T; t += (int64_t)gridDim.x * blockDim.x) {
        float v = feats[t * ld + d];
        if (pass == 1) {
            float f = v - mu;
            v = f * f;
        }
        acc += (double)v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&out[d], part[0] + part[1] + part[2] + part[3]);
    (void)D;
}

extern "C" int srx_corpus_mean_var_dev(srx_ctx *ctx, const float *feats_dev, int ld, int64_t total_frames, int D,
                                       double *sum_dev) {
    SRX_REQUIRE(nullptr, ctx, "srx_corpus_mean_var_dev: null ctx");
    SRX_REQUIRE(ctx, feats_dev && sum_dev && D >= 1 && ld >= D && total_frames >= 1, "srx_corpus_mean_var_dev: bad argument");
    SRX_HIP(ctx, hipMemsetAsync(sum_dev, 0, 2 * D * sizeof(double), ctx->stream));
    int nb = (int)(ceil_div64(total_frames, 256) < 1024 ? ceil_div64(total_frames, 256) : 1024);
    {
        SrxProf prof_(ctx, "k_colsum");
        hipLaunchKernelGGL(k_colsum, dim3(nb, D), dim3(256), 0, ctx->stream, feats_dev, ld, total_frames, D, sum_dev, sum_dev, 0);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_colsum");
        hipLaunchKernelGGL(k_colsum, dim3(nb, D), dim3(256), 0, ctx->stream, feats_dev, ld, total_frames, D, sum_dev, sum_dev + D, 1);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
