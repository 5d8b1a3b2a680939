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
    if (WL < 0) return raw[(base + i) * nc + q];   // compute_deltas alone (src/mfcc.cpp:95-102): the coefficients as they are
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

// One workgroup = FBK consecutive frames of the batch (they may span utterances).  Three stages through LDS, each value
// computed once: smoothed cepstra for frames F0-2 .. F0+FBK+1, deltas for F0-1 .. F0+FBK, delta-deltas for F0 .. F0+FBK-1.
// A frame's neighbours are clamped to its own utterance (src/mfcc.cpp:79-80,91-92), so every value a frame needs lies
// within two frames of it.  Same operations in the same order as the scalar form (smooth_at; (a - b) / (k - j)).
static constexpr int FBK = 64;

__global__ __launch_bounds__(256) void k_features(const float *__restrict__ raw, const int64_t *__restrict__ foff, int n_utt,
                                                  int64_t total_frames, int nc, int WL, float *__restrict__ feats) {
    extern __shared__ float fsh[];
    float *sm = fsh;                               // [FBK + 4][nc]   frame F0 - 2 + r
    float *dl = fsh + (FBK + 4) * nc;              // [FBK + 2][nc]   frame F0 - 1 + r
    int64_t *ub = reinterpret_cast<int64_t *>(dl + (FBK + 2) * nc);   // [FBK + 4] utterance base of frame F0 - 2 + r
    int64_t *un = ub + (FBK + 4);                                      // [FBK + 4] its length (0: no such frame)
    const int64_t F0 = (int64_t)blockIdx.x * FBK;
    // one binary search per block (a chain of dependent loads; uniform, so scalar loads), on the block's first frame;
    // a row walks forward from that utterance
    int u0;
    {
        const int64_t f0 = F0 - 2 < 0 ? 0 : F0 - 2;
        int lo = 0, hi = n_utt;
        while (hi - lo > 1) {
            int mid = (lo + hi) >> 1;
            if (foff[mid] <= f0) lo = mid; else hi = mid;
        }
        u0 = lo;
    }
    for (int r = threadIdx.x; r < FBK + 4; r += 256) {
        const int64_t f = F0 - 2 + r;
        int64_t base = 0, n = 0;
        if (f >= 0 && f < total_frames) {
            int lo = u0;
            while (lo + 1 < n_utt && foff[lo + 1] <= f) lo++;
            base = foff[lo];
            n = foff[lo + 1] - base;
        }
        ub[r] = base;
        un[r] = n;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (FBK + 4) * nc; i += 256) {
        const int r = i / nc, q = i - r * nc;
        if (un[r] > 0) sm[i] = smooth_at(raw, ub[r], un[r], F0 - 2 + r - ub[r], q, nc, WL);
    }
    __syncthreads();
    // row of frame (base + idx) in sm: its frame number minus (F0 - 2)
    for (int i = threadIdx.x; i < (FBK + 2) * nc; i += 256) {
        const int r = i / nc, q = i - r * nc;       // frame F0 - 1 + r  = sm row r + 1
        const int64_t n = un[r + 1];
        if (n > 0) {
            const int64_t base = ub[r + 1], ii = F0 - 1 + r - base;
            int64_t j = ii - 1, k = ii + 1;
            if (j < 0) j = 0;
            if (k == n) k--;
            const float a = sm[(int)(base + k - (F0 - 2)) * nc + q];
            const float b = sm[(int)(base + j - (F0 - 2)) * nc + q];
            const float d = a - b;
            dl[i] = d / (float)(k - j);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < FBK * nc; i += 256) {
        const int r = i / nc, q = i - r * nc;       // frame F0 + r = sm row r + 2 = dl row r + 1
        const int64_t f = F0 + r;
        if (f >= total_frames) continue;
        const int64_t n = un[r + 2], base = ub[r + 2], ii = f - base;
        int64_t j = ii - 1, k = ii + 1;
        if (j < 0) j = 0;
        if (k == n) k--;
        const float dk = dl[(int)(base + k - (F0 - 1)) * nc + q];
        const float dj = dl[(int)(base + j - (F0 - 1)) * nc + q];
        float dd = dk - dj;
        dd = dd / (float)(k - j);
        float *o = feats + f * (3 * nc);
        o[q] = sm[(r + 2) * nc + q];
        o[nc + q] = dl[(r + 1) * nc + q];
        o[2 * nc + q] = dd;
    }
}

extern "C" int srx_features_batch_dev(srx_ctx *ctx, int WL, int n_ceps, const float *ceps_dev, const int64_t *frame_off_dev,
                                      int n_utt, int64_t total_frames, float *feats_dev) {
    SRX_ENTER(ctx, "srx_features_batch_dev");
    SRX_REQUIRE(ctx, ceps_dev && frame_off_dev && feats_dev, "srx_features_batch_dev: null argument");
    SRX_REQUIRE(ctx, WL >= -1 && WL <= 64 && n_ceps >= 1 && n_utt >= 0 && total_frames >= 0, "srx_features_batch_dev: bad size");
    if (n_utt == 0 || total_frames == 0) return SRX_OK;
    SRX_REQUIRE(ctx, n_ceps <= 64, "srx_features_batch_dev: n_ceps %d > 64", n_ceps);
    const int64_t nblk = ceil_div64(total_frames, FBK);
    SRX_REQUIRE(ctx, nblk < (1ll << 31), "srx_features_batch_dev: too many frames");
    const size_t lds = ((size_t)(FBK + 4) + (FBK + 2)) * n_ceps * sizeof(float) + 2 * (FBK + 4) * sizeof(int64_t);
    {
        SrxProf prof_(ctx, "k_features");
        hipLaunchKernelGGL(k_features, dim3((unsigned)nblk), dim3(256), lds, ctx->stream, ceps_dev, frame_off_dev, n_utt, total_frames,
                           n_ceps, WL, feats_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_features_batch(srx_ctx *ctx, int WL, int n_ceps, const float *ceps, const int64_t *frame_off, int n_utt,
                                  float *feats) {
    SRX_ENTER(ctx, "srx_features_batch");
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
                                                const double *__restrict__ mu_sum, double *__restrict__ out, int pass,
                                                const float *__restrict__ mu_given) {
    const int d = blockIdx.y;
    double acc = 0.0;
    float mu = 0.0f;
    if (pass == 1) mu = mu_given ? mu_given[d] : (float)mu_sum[d] / (float)(int)T;
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
    SRX_ENTER(ctx, "srx_corpus_mean_var_dev");
    SRX_REQUIRE(ctx, feats_dev && sum_dev && D >= 1 && ld >= D && total_frames >= 1, "srx_corpus_mean_var_dev: bad argument");
    SRX_HIP(ctx, hipMemsetAsync(sum_dev, 0, 2 * D * sizeof(double), ctx->stream));
    int nb = (int)(ceil_div64(total_frames, 256) < 1024 ? ceil_div64(total_frames, 256) : 1024);
    {
        SrxProf prof_(ctx, "k_colsum");
        hipLaunchKernelGGL(k_colsum, dim3(nb, D), dim3(256), 0, ctx->stream, feats_dev, ld, total_frames, D, sum_dev, sum_dev, 0, nullptr);
    }
    SRX_LAUNCH_CHECK(ctx);
    {
        SrxProf prof_(ctx, "k_colsum");
        hipLaunchKernelGGL(k_colsum, dim3(nb, D), dim3(256), 0, ctx->stream, feats_dev, ld, total_frames, D, sum_dev, sum_dev + D, 1, nullptr);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_corpus_sums_dev(srx_ctx *ctx, const float *feats_dev, int ld, int64_t total_frames, int D, const float *mu_dev,
                                   double *out_dev) {
    SRX_ENTER(ctx, "srx_corpus_sums_dev");
    SRX_REQUIRE(ctx, out_dev && D >= 1 && ld >= D && total_frames >= 0 && (feats_dev || total_frames == 0), "srx_corpus_sums_dev: bad argument");
    SRX_HIP(ctx, hipMemsetAsync(out_dev, 0, D * sizeof(double), ctx->stream));
    if (total_frames == 0) return SRX_OK;   // an empty shard contributes zeros
    int nb = (int)(ceil_div64(total_frames, 256) < 1024 ? ceil_div64(total_frames, 256) : 1024);
    {
        SrxProf prof_(ctx, "k_colsum");
        hipLaunchKernelGGL(k_colsum, dim3(nb, D), dim3(256), 0, ctx->stream, feats_dev, ld, total_frames, D, (const double *)nullptr, out_dev,
                           mu_dev ? 1 : 0, mu_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
