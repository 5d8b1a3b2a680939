// Diagonal-covariance GMM emission scoring on gfx950.
//
// Replaces (reference): gaussian::operator() src/gmm.cpp:41-49, gmm::operator() src/gmm.cpp:99-109,
// acoustic_model::operator() src/gmm.cpp:184-186, and the model bookkeeping of src/gmm.cpp:13-38,
// 171-175 (construct / set_wt / copy).
//
// Two scoring paths:
//   SRX_SCORE_EXACT  one thread per (frame, tied state), the reference's per-dimension loop in its
//                    own operation order, f64 standing in for the 80-bit long double.
//   SRX_SCORE_MFMA   logp(x) = const + sum_d [ (-1/(2 s2_d)) x'_d^2 + (mu'_d / s2_d) x'_d ],  x' = x - c,
//                    mu' = mu - c (c = mean of all component means: kills the cancellation of the
//                    expanded square), evaluated as a [rows x (2D+2)] x [(2D+2) x frames] contraction on
//                    v_mfma_f32_32x32x2_f32 (exact f32, k-ordered fma chain).  Gaussians are the MFMA
//                    rows, frames the columns, so a lane's 16 accumulator registers hold the
//                    components of whole states and the mixture log-sum-exp needs no cross-lane step.
#include <cmath>

#include "srx_common.h"

// ------------------------------------------------------------------------------------------
// model bookkeeping
// ------------------------------------------------------------------------------------------
extern "C" int srx_model_create(srx_ctx *ctx, int G, int M, int D, srx_model **out) {
    SRX_ENTER(ctx, "srx_model_create");
    SRX_REQUIRE(ctx, out && G >= 1 && M >= 1 && D >= 1 && D <= 128, "srx_model_create: bad shape G=%d M=%d D=%d", G, M, D);
    srx_model *m = new srx_model();
    m->ctx = ctx;
    m->G = G; m->M = M; m->D = D;
    const size_t n = (size_t)G * M;
    hipError_t e;
    if ((e = hipMalloc((void **)&m->mean, n * D * sizeof(double))) != hipSuccess ||
        (e = hipMalloc((void **)&m->var, n * D * sizeof(double))) != hipSuccess ||
        (e = hipMalloc((void **)&m->weight, n * sizeof(float))) != hipSuccess ||
        (e = hipMalloc((void **)&m->logwt, n * sizeof(float))) != hipSuccess ||
        (e = hipMalloc((void **)&m->center, D * sizeof(float))) != hipSuccess) {
        srx_model_destroy(m);
        return srx_fail(ctx, SRX_ERR_NOMEM, "srx_model_create: %s", hipGetErrorString(e));
    }
    *out = m;
    return SRX_OK;
}

extern "C" void srx_model_destroy(srx_model *m) {
    if (!m) return;
    (void)hipFree(m->mean);
    (void)hipFree(m->var);
    (void)hipFree(m->weight);
    (void)hipFree(m->logwt);
    (void)hipFree(m->center);
    (void)hipFree(m->packed);
    (void)hipFree(m->packed16);
    (void)hipFree(m->packed16s);
    (void)hipFree(m->lconst);
    (void)hipFree(m->colstats);
    (void)hipFree(m->flag16);
    delete m;
}

extern "C" int srx_model_set(srx_model *m, const double *mean, const double *var, const float *weight) {
    SRX_REQUIRE(nullptr, m, "srx_model_set: null model");
    srx_ctx *ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_REQUIRE(ctx, mean && var && weight, "srx_model_set: null argument");
    const size_t n = (size_t)m->G * m->M;
    SRX_HIP(ctx, hipMemcpyAsync(m->mean, mean, n * m->D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(m->var, var, n * m->D * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(m->weight, weight, n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    // gaussian::set_wt (src/gmm.cpp:35-38): logwt = log(wt) in f32, on the host like the reference
    std::vector<float> lw(n);
    for (size_t i = 0; i < n; i++) lw[i] = logf(weight[i]);
    SRX_HIP(ctx, hipMemcpyAsync(m->logwt, lw.data(), n * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    m->version++;
    return SRX_OK;
}

extern "C" int srx_model_get(srx_model *m, double *mean, double *var, float *weight) {
    SRX_REQUIRE(nullptr, m, "srx_model_get: null model");
    srx_ctx *ctx = m->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    const size_t n = (size_t)m->G * m->M;
    if (mean) SRX_HIP(ctx, hipMemcpyAsync(mean, m->mean, n * m->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (var) SRX_HIP(ctx, hipMemcpyAsync(var, m->var, n * m->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (weight) SRX_HIP(ctx, hipMemcpyAsync(weight, m->weight, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

extern "C" uint64_t srx_model_version(const srx_model *m) { return m ? m->version : 0; }

extern "C" int srx_model_copy(srx_model *dst, const srx_model *src) {
    SRX_REQUIRE(nullptr, dst && src, "srx_model_copy: null model");
    srx_ctx *ctx = dst->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_REQUIRE(ctx, dst->G == src->G && dst->M == src->M && dst->D == src->D, "srx_model_copy: shape mismatch");
    const size_t n = (size_t)dst->G * dst->M;
    SRX_HIP(ctx, hipMemcpyAsync(dst->mean, src->mean, n * dst->D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dst->var, src->var, n * dst->D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dst->weight, src->weight, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dst->logwt, src->logwt, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    dst->version++;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------
// EXACT path: gaussian::operator() in the reference's order (src/gmm.cpp:41-49).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gmm_exact(const float *__restrict__ feats, int ld, int64_t T, int G, int M, int D,
                                                   const double *__restrict__ mean, const double *__restrict__ var,
                                                   const float *__restrict__ logwt, int mode, float *__restrict__ out,
                                                   int64_t st_t, int64_t st_g) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * G) return;
    const int64_t t = idx / G;
    const int g = (int)(idx - t * G);
    const float *fv = feats + t * ld;
    float r;
    if (mode == SRX_MIX_COMPONENT0) {
        const size_t o = (size_t)g * M;
        r = srx_logprob_exact(fv, mean + o * D, var + o * D, logwt[o], D);
    } else {
        const size_t o = (size_t)g * M;
        r = srx_lse_exact(fv, mean + o * D, var + o * D, logwt + o, M, D);
    }
    out[t * st_t + (int64_t)g * st_g] = r;
}

// MFMA paths: srx_gmm_mfma.hip (f32 inputs), srx_gmm_f16.hip (split-f16 inputs)
int srx_gmm_score_f16x3(srx_ctx *ctx, srx_model *m, int mix_mode, const float *feats, int ld, int64_t T, float *out, int64_t st_t,
                        int64_t st_g);
int srx_gmm_score_mfma(srx_ctx *ctx, srx_model *m, int mix_mode, const float *feats, int ld, int64_t T, float *out, int64_t st_t,
                       int64_t st_g);

extern "C" int srx_gmm_score_dev(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats_dev, int ld,
                                 int64_t T, float *out_dev, int64_t stride_t, int64_t stride_g) {
    SRX_ENTER(ctx, "srx_gmm_score_dev");
    SRX_REQUIRE(ctx, m && feats_dev && out_dev, "srx_gmm_score_dev: null argument");
    SRX_REQUIRE(ctx, ld >= m->D && T >= 0, "srx_gmm_score_dev: ld=%d < D=%d", ld, m->D);
    SRX_REQUIRE(ctx, (stride_g == 1 && stride_t >= m->G) || (stride_t == 1 && stride_g >= T),
                "srx_gmm_score_dev: output strides must be frame-major (stride_g=1, stride_t>=G) or state-major (stride_t=1, stride_g>=T)");
    SRX_REQUIRE(ctx, mix_mode == SRX_MIX_COMPONENT0 || mix_mode == SRX_MIX_LOGSUMEXP, "srx_gmm_score_dev: bad mix_mode");
    if (T == 0) return SRX_OK;
    if (score_path == SRX_SCORE_EXACT) {
        const int64_t n = T * m->G;
        SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_gmm_score_dev: too large");
        {
            SrxProf prof_(ctx, "k_gmm_exact");
            hipLaunchKernelGGL(k_gmm_exact, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, feats_dev, ld, T, m->G, m->M,
                               m->D, m->mean, m->var, m->logwt, mix_mode, out_dev, stride_t, stride_g);
        }
        SRX_LAUNCH_CHECK(ctx);
        return SRX_OK;
    }
    if (score_path == SRX_SCORE_MFMA_F16X3)
        return srx_gmm_score_f16x3(ctx, m, mix_mode, feats_dev, ld, T, out_dev, stride_t, stride_g);
    SRX_REQUIRE(ctx, score_path == SRX_SCORE_MFMA, "srx_gmm_score_dev: bad score_path");
    return srx_gmm_score_mfma(ctx, m, mix_mode, feats_dev, ld, T, out_dev, stride_t, stride_g);
}

extern "C" int srx_gmm_score(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats, int ld, int64_t T,
                             float *out) {
    SRX_ENTER(ctx, "srx_gmm_score");
    SRX_REQUIRE(ctx, m && feats && out && T >= 0, "srx_gmm_score: bad argument");
    if (T == 0) return SRX_OK;
    DevBuf df, dout;
    SRX_HIP(ctx, df.alloc((size_t)T * ld * sizeof(float)));
    SRX_HIP(ctx, dout.alloc((size_t)T * m->G * sizeof(float)));
    SRX_HIP(ctx, hipMemcpyAsync(df.p, feats, (size_t)T * ld * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
    int rc = srx_gmm_score_dev(ctx, m, mix_mode, score_path, df.as<float>(), ld, T, dout.as<float>(), m->G, 1);
    if (rc != SRX_OK) return rc;
    SRX_HIP(ctx, hipMemcpyAsync(out, dout.p, (size_t)T * m->G * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}
