// Embedded-training sufficient statistics on gfx950.
//
// Replaces the statistics half of hmm::train (src/hmm.cpp:274-341): the zeta pass (:274-287),
// sum_zetas/clear/weights (:288-296), the mean pass (:298-317), the variance pass (:319-339) and
// gaussian::blend (src/gmm.cpp:82-88).  The reference walks every utterance sequentially and
// re-evaluates gaussian::prob three times per aligned frame; here the per-frame log-densities are
// evaluated once (in the reference's operation order, f64 for long double), kept, and the three
// accumulations run frame-parallel with f64 atomics -- LDS-privatised per workgroup when the table
// fits, straight to HBM otherwise.  zeta = w * exp(logp - shift[g][m]) with shift[g][m] the largest
// log-density component (g,m) saw: means and variances are ratios of one component's zeta sums, so the
// shift cancels; the mixture weights re-apply exp(shift[g][m] - max_m shift[g][m]) before normalising.
// This keeps exp() inside f64 range where the reference relies on the 80-bit exponent.
#include <cmath>

#include "srx_common.h"

__device__ __forceinline__ void atomic_max_float(float *addr, float v) {
    // total order trick: non-negative floats order like ints, negative floats reversed like uints
    if (v >= 0.0f)
        atomicMax(reinterpret_cast<int *>(addr), __float_as_int(v));
    else
        atomicMin(reinterpret_cast<unsigned int *>(addr), __float_as_uint(v));
}

// one workgroup per utterance: its frames' path states -> tied ids (-1 for an utterance without a path).  (One thread per frame with
// a binary search over the frame offsets was 17 dependent loads per frame at 100,000 utterances: 0.67 ms per 5e7 frames.)
__global__ __launch_bounds__(256) void k_gather_gid(int n_utt, const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                    const int32_t *__restrict__ tied, const int32_t *__restrict__ path,
                                                    const int32_t *__restrict__ status, int64_t total, int32_t *__restrict__ gid) {
    const int u = blockIdx.x;
    if (u >= n_utt) return;
    const int64_t f0 = foff[u], f1 = foff[u + 1] < total ? foff[u + 1] : total;
    const int s0 = soff[u];
    const bool ok = status[u] == 0;
    for (int64_t f = f0 + threadIdx.x; f < f1; f += 256) gid[f] = ok ? tied[s0 + path[f]] : -1;
}

__global__ void k_fill_f32(float *p, float v, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

__global__ __launch_bounds__(256) void k_model_lconst(const double *__restrict__ var, int64_t n, double *__restrict__ lconst) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const float l2pi = (float)-0.4515827052894549;  // (float)log(M_2_PI), src/gmm.cpp:43
    if (i < n) {
        lconst[i] = (double)l2pi + log(var[i]);
        lconst[n + i] = 1.0 / var[i];
    }
}

// one thread per (frame, component): log-density under A, and the per-state maximum
__global__ __launch_bounds__(256) void k_em_logp(const float *__restrict__ feats, int ld, const int32_t *__restrict__ gid, int64_t T,
                                                 int M, int D, const double *__restrict__ mean, const double *__restrict__ rvar,
                                                 const double *__restrict__ lconst, const float *__restrict__ logwt,
                                                 float *__restrict__ logp, float *__restrict__ shift) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= T * M) return;
    const int64_t t = idx / M;
    const int m = (int)(idx - t * M);
    const int g = gid[t];
    if (g < 0) { logp[idx] = 0.0f; return; }
    const size_t o = (size_t)g * M + m;
    const float lp = srx_logprob_exact_lc(feats + t * ld, mean + o * D, rvar + o * D, lconst + o * D, logwt[o], D);
    logp[idx] = lp;
    if (lp == lp) atomic_max_float(&shift[o], lp);
}

// Accumulation.  PASS 1: row o = [zeta, zeta*x_0 .. zeta*x_{D-1}] (stride 1+D).
//                PASS 2: row o = [zeta*diff_d^2] (stride D), diff = f32(x_d - meanB[o][d]) (src/hmm.cpp:331).
// If the whole table fits the dynamic LDS it is privatised per workgroup (LOCAL), else atomics go to HBM.
template <int PASS, bool LOCAL>
__global__ __launch_bounds__(256) void k_em_accum(const float *__restrict__ feats, int ld, const int32_t *__restrict__ gid, int64_t T,
                                                  int M, int D, int nrows, const float *__restrict__ weight,
                                                  const float *__restrict__ logp, const float *__restrict__ shift,
                                                  const double *__restrict__ meanB, double *__restrict__ stats) {
    extern __shared__ double tab[];
    const int stride = PASS == 1 ? 1 + D : D;
    if (LOCAL) {
        for (int i = threadIdx.x; i < nrows * stride; i += 256) tab[i] = 0.0;
        __syncthreads();
    }
    double *dst = LOCAL ? tab : stats;
    for (int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x; idx < T * M; idx += (int64_t)gridDim.x * 256) {
        const int64_t t = idx / M;
        const int m = (int)(idx - t * M);
        const int g = gid[t];
        if (g < 0) continue;
        const size_t o = (size_t)g * M + m;
        // zeta = weight * prob  (src/hmm.cpp:282,307,328), shifted
        const double zeta = (double)weight[o] * exp((double)logp[idx] - (double)shift[o]);
        const float *fv = feats + t * ld;
        double *row = dst + o * stride;
        if (PASS == 1) {
            atomicAdd(&row[0], zeta);
            for (int e = 0; e < D; e++) atomicAdd(&row[1 + e], zeta * (double)fv[e]);
        } else {
            for (int e = 0; e < D; e++) {
                const float diff = (float)((double)fv[e] - meanB[o * D + e]);
                atomicAdd(&row[e], zeta * (double)diff * (double)diff);
            }
        }
    }
    if (LOCAL) {
        __syncthreads();
        for (int i = threadIdx.x; i < nrows * stride; i += 256) {
            const double v = tab[i];
            if (v != 0.0) atomicAdd(&stats[i], v);
        }
    }
}

// weights and means of B from stats1  (src/hmm.cpp:288-296, 317; gaussian::clear sets var = 0)
__global__ __launch_bounds__(256) void k_em_means(int G, int M, int D, const double *__restrict__ stats1,
                                                  const float *__restrict__ shift, double *__restrict__ meanB,
                                                  double *__restrict__ varB, float *__restrict__ wB, float *__restrict__ lwB) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * M * D) return;
    const int e = (int)(idx % D);
    const size_t o = (size_t)(idx / D);
    const int g = (int)(o / M);
    const double zs = stats1[o * (1 + D)];
    meanB[idx] = stats1[o * (1 + D) + 1 + e] / zs;
    varB[idx] = 0.0;
    if (e == 0) {
        // zetasum[o] = stats1[o][0] * exp(shift[o]); normalise within the state (src/hmm.cpp:291-295)
        float smax = -__builtin_huge_valf();
        for (int m = 0; m < M; m++) smax = fmaxf(smax, shift[(size_t)g * M + m]);
        double zg = 0.0, zo = 0.0;
        for (int m = 0; m < M; m++) {
            const size_t om = (size_t)g * M + m;
            const float sh = shift[om];
            const double z = (sh == -__builtin_huge_valf()) ? 0.0 : stats1[om * (1 + D)] * exp((double)sh - (double)smax);
            zg += z;
            if (om == o) zo = z;
        }
        const float w = (float)(zo / zg);
        wB[o] = w;
        lwB[o] = logf(w);
    }
}

// variances of B, then gaussian::blend(A, keep)  (src/hmm.cpp:339-341, src/gmm.cpp:82-88)
__global__ __launch_bounds__(256) void k_em_finish(int G, int M, int D, const double *__restrict__ stats1,
                                                   const double *__restrict__ stats2, const double *__restrict__ meanA,
                                                   const double *__restrict__ varA, const float *__restrict__ wA,
                                                   double *__restrict__ meanB, double *__restrict__ varB, float *__restrict__ wB,
                                                   float *__restrict__ lwB, float keep) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * M * D) return;
    const int e = (int)(idx % D);
    const size_t o = (size_t)(idx / D);
    const double zs = stats1[o * (1 + D)];
    const double vnew = stats2[idx] / zs;
    const float omk = 1 - keep;
    meanB[idx] = meanB[idx] * (double)keep + meanA[idx] * (double)omk;
    varB[idx] = vnew * (double)keep + varA[idx] * (double)omk;
    if (e == 0) {
        const float a = wB[o] * keep;
        const float b = wA[o] * omk;
        const float w = a + b;
        wB[o] = w;
        lwB[o] = logf(w);
    }
}

// state_model::augment over every frame of every final path (src/hmm.cpp:46-51, 350-356):
// probsum[cls] += exp(tr[1]) of the path state (hmm_state::get_adv_prob, src/hmm.cpp:26-28), count[cls] += 1.
// The reference keeps f32 running sums in path order; here the sums are f64 atomics (order-free).
__global__ __launch_bounds__(256) void k_state_model(int n_utt, const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                     const int32_t *__restrict__ cls, const float *__restrict__ trlogp,
                                                     const int32_t *__restrict__ path, const int32_t *__restrict__ status,
                                                     int64_t total, double *__restrict__ probsum, int32_t *__restrict__ count) {
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= total) return;
    int lo = 0, hi = n_utt;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (foff[mid] <= f) lo = mid; else hi = mid;
    }
    if (status[lo] != 0) return;
    const int s = soff[lo] + path[f];
    const int c = cls[s];
    if (c < 0) return;
    atomicAdd(&probsum[c], (double)expf(trlogp[(size_t)s * 3 + 1]));
    atomicAdd(&count[c], 1);
}

extern "C" int srx_state_model_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *cls_dev, const int32_t *path_dev,
                                   const int32_t *status_dev, int n_classes, double *probsum_dev, int32_t *count_dev) {
    SRX_ENTER(ctx, "srx_state_model_dev");
    SRX_REQUIRE(ctx, hmm && cls_dev && path_dev && status_dev && probsum_dev && count_dev && n_classes >= 1,
                "srx_state_model_dev: bad argument");
    SRX_HIP(ctx, hipMemsetAsync(probsum_dev, 0, (size_t)n_classes * sizeof(double), ctx->stream));
    SRX_HIP(ctx, hipMemsetAsync(count_dev, 0, (size_t)n_classes * sizeof(int32_t), ctx->stream));
    if (hmm->n_utt == 0 || hmm->total_frames == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_state_model");
        hipLaunchKernelGGL(k_state_model, dim3((unsigned)ceil_div64(hmm->total_frames, 256)), dim3(256), 0, ctx->stream, hmm->n_utt,
                           hmm->frame_off_dev, hmm->state_off_dev, cls_dev, hmm->trlogp_dev, path_dev, status_dev,
                           hmm->total_frames, probsum_dev, count_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_em_gather_gid_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *path_dev, const int32_t *status_dev,
                                     int64_t total_frames, int32_t *gid_dev) {
    SRX_ENTER(ctx, "srx_em_gather_gid_dev");
    SRX_REQUIRE(ctx, hmm && path_dev && status_dev && gid_dev && total_frames >= 0, "srx_em_gather_gid_dev: bad argument");
    if (total_frames == 0 || hmm->n_utt == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_gather_gid");
        hipLaunchKernelGGL(k_gather_gid, dim3((unsigned)hmm->n_utt), dim3(256), 0, ctx->stream, hmm->n_utt,
                           hmm->frame_off_dev, hmm->state_off_dev, hmm->tied_dev, path_dev, status_dev, total_frames, gid_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

// srx_train_seg.hip: sort-by-state segmented reduction for tables beyond LDS
bool srx_em_seg_applicable(int G, int M, int D);
int srx_em_logp_seg(srx_ctx *ctx, const float *feats, int ld, int64_t T, int G, int M, int D, const double *mean, const double *rvar,
                    const double *lconst, const float *logwt, float *logp, float *shift);
int srx_em_seg_sort(srx_ctx *ctx, const int32_t *gid, int64_t T, int G);
template <int PASS>
int srx_em_accum_seg(srx_ctx *ctx, const float *feats, int ld, const int32_t *gid, int64_t T, int G, int M, int D, const float *weight,
                     const float *logp, const float *shift, const double *meanB, double *stats);

static int ensure_lconst(srx_ctx *ctx, srx_model *A);

extern "C" int srx_em_shift_dev(srx_ctx *ctx, srx_model *A, const float *feats_dev, int ld, const int32_t *gid_dev, int64_t T,
                                float *logp_dev, float *shift_dev) {
    SRX_ENTER(ctx, "srx_em_shift_dev");
    SRX_REQUIRE(ctx, A && feats_dev && gid_dev && logp_dev && shift_dev && T >= 0 && ld >= A->D, "srx_em_shift_dev: bad argument");
    {
        SrxProf prof_(ctx, "k_fill_f32");
        hipLaunchKernelGGL(k_fill_f32, dim3((A->G * A->M + 255) / 256), dim3(256), 0, ctx->stream, shift_dev, -__builtin_huge_valf(),
                           A->G * A->M);
    }
    SRX_LAUNCH_CHECK(ctx);
    // a new alignment (usually under the same pointer): whatever sort the workspace holds belongs to the previous one
    ctx->em_gid = nullptr;
    if (T == 0) return SRX_OK;
    // tables beyond LDS: the passes of this alignment run as a segmented reduction over frames sorted by state; the
    // sort belongs to the alignment, and this is the first step every iteration takes with a new one
    const bool seg = (size_t)A->G * A->M * (1 + A->D) * sizeof(double) > 64 * 1024 && srx_em_seg_applicable(A->G, A->M, A->D);
    if (seg) {
        int rc = srx_em_seg_sort(ctx, gid_dev, T, A->G);
        if (rc != SRX_OK) return rc;
    }
    {
        int rc = ensure_lconst(ctx, A);
        if (rc != SRX_OK) return rc;
    }
    if (seg)
        return srx_em_logp_seg(ctx, feats_dev, ld, T, A->G, A->M, A->D, A->mean, A->lconst + (size_t)A->G * A->M * A->D, A->lconst, A->logwt,
                               logp_dev, shift_dev);
    const int64_t n = T * A->M;
    SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_em_shift_dev: too large");
    {
        SrxProf prof_(ctx, "k_em_logp");
        hipLaunchKernelGGL(k_em_logp, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, feats_dev, ld, gid_dev, T, A->M, A->D,
                           A->mean, A->lconst + (size_t)A->G * A->M * A->D, A->lconst, A->logwt, logp_dev, shift_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

template <int PASS>
static int launch_accum(srx_ctx *ctx, srx_model *A, srx_model *B, const float *feats, int ld, const int32_t *gid, int64_t T,
                        const float *logp, const float *shift, double *stats) {
    const int stride = PASS == 1 ? 1 + A->D : A->D;
    const int nrows = A->G * A->M;
    const size_t bytes = (size_t)nrows * stride * sizeof(double);
    SRX_HIP(ctx, hipMemsetAsync(stats, 0, bytes, ctx->stream));
    if (T == 0) return SRX_OK;
    const int64_t n = T * A->M;
    int64_t nb = ceil_div64(n, 256 * 8);
    if (nb > 8 * ctx->num_cus) nb = 8 * ctx->num_cus;
    if (nb < 1) nb = 1;
    const double *meanB = B ? B->mean : nullptr;
    if (bytes <= 64 * 1024) {
        {
            SrxProf prof_(ctx, PASS == 1 ? "k_em_accum1" : "k_em_accum2");
            hipLaunchKernelGGL((k_em_accum<PASS, true>), dim3((unsigned)nb), dim3(256), bytes, ctx->stream, feats, ld, gid, T, A->M, A->D,
                               nrows, A->weight, logp, shift, meanB, stats);
        }
    } else if (srx_em_seg_applicable(A->G, A->M, A->D)) {
        return srx_em_accum_seg<PASS>(ctx, feats, ld, gid, T, A->G, A->M, A->D, A->weight, logp, shift, meanB, stats);
    } else {
        {
            SrxProf prof_(ctx, PASS == 1 ? "k_em_accum1" : "k_em_accum2");
            hipLaunchKernelGGL((k_em_accum<PASS, false>), dim3((unsigned)nb), dim3(256), 0, ctx->stream, feats, ld, gid, T, A->M, A->D,
                               nrows, A->weight, logp, shift, meanB, stats);
        }
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_em_pass1_dev(srx_ctx *ctx, srx_model *A, const float *feats_dev, int ld, const int32_t *gid_dev, int64_t T,
                                const float *logp_dev, const float *shift_dev, double *stats1_dev) {
    SRX_ENTER(ctx, "srx_em_pass1_dev");
    SRX_REQUIRE(ctx, A && feats_dev && gid_dev && logp_dev && shift_dev && stats1_dev && T >= 0 && ld >= A->D,
                "srx_em_pass1_dev: bad argument");
    return launch_accum<1>(ctx, A, nullptr, feats_dev, ld, gid_dev, T, logp_dev, shift_dev, stats1_dev);
}

extern "C" int srx_em_means_dev(srx_ctx *ctx, srx_model *B, const double *stats1_dev, const float *shift_dev) {
    SRX_ENTER(ctx, "srx_em_means_dev");
    SRX_REQUIRE(ctx, B && stats1_dev && shift_dev, "srx_em_means_dev: null argument");
    const int64_t n = (int64_t)B->G * B->M * B->D;
    {
        SrxProf prof_(ctx, "k_em_means");
        hipLaunchKernelGGL(k_em_means, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, B->G, B->M, B->D, stats1_dev, shift_dev,
                           B->mean, B->var, B->weight, B->logwt);
    }
    SRX_LAUNCH_CHECK(ctx);
    B->version++;
    return SRX_OK;
}

extern "C" int srx_em_pass2_dev(srx_ctx *ctx, srx_model *A, srx_model *B, const float *feats_dev, int ld, const int32_t *gid_dev,
                                int64_t T, const float *logp_dev, const float *shift_dev, double *stats2_dev) {
    SRX_ENTER(ctx, "srx_em_pass2_dev");
    SRX_REQUIRE(ctx, A && B && feats_dev && gid_dev && logp_dev && shift_dev && stats2_dev && T >= 0 && ld >= A->D,
                "srx_em_pass2_dev: bad argument");
    SRX_REQUIRE(ctx, A->G == B->G && A->M == B->M && A->D == B->D, "srx_em_pass2_dev: model shape mismatch");
    return launch_accum<2>(ctx, A, B, feats_dev, ld, gid_dev, T, logp_dev, shift_dev, stats2_dev);
}

// srx_train_seg.hip
bool srx_em_fused_applicable(int G, int M, int D);
int srx_em_fused_row(int D);
int srx_em_fused_launch(srx_ctx *ctx, const float *feats, int ld, const int32_t *gid, int64_t T, int G, int M, int D, const double *meanA,
                        const double *rvarA, const float *weight, const float *logp, const float *shift, double *stats);
int srx_em_fused_finish_launch(srx_ctx *ctx, int G, int M, int D, const double *stats, const float *shift, const double *lconstA,
                               double *peak_scratch, const double *meanA, const double *varA, const float *wA, const float *lwA, double *meanB,
                               double *varB, float *wB, float *lwB, float keep, int32_t *underflow);

static int ensure_lconst(srx_ctx *ctx, srx_model *A) {
    if (A->lconst_version != A->version || !A->lconst) {
        const int64_t nl = (int64_t)A->G * A->M * A->D;
        // [G*M*D] lconst, [G*M*D] 1/var, then [G*M] scratch for the components' peak log-densities (srx_em_fused_finish_dev)
        if (!A->lconst) SRX_HIP(ctx, hipMalloc((void **)&A->lconst, ((size_t)2 * nl + (size_t)A->G * A->M) * sizeof(double)));
        {
            SrxProf prof_(ctx, "k_model_lconst");
            hipLaunchKernelGGL(k_model_lconst, dim3((unsigned)ceil_div64(nl, 256)), dim3(256), 0, ctx->stream, A->var, nl, A->lconst);
        }
        SRX_LAUNCH_CHECK(ctx);
        A->lconst_version = A->version;
    }
    return SRX_OK;
}

extern "C" int64_t srx_em_fused_len(const srx_model *A) {
    if (!A || !srx_em_fused_applicable(A->G, A->M, A->D)) return 0;
    return (int64_t)A->G * A->M * srx_em_fused_row(A->D);
}

extern "C" int srx_em_fused_dev(srx_ctx *ctx, srx_model *A, const float *feats_dev, int ld, const int32_t *gid_dev, int64_t T,
                                const float *logp_dev, const float *shift_dev, double *stats12_dev) {
    SRX_ENTER(ctx, "srx_em_fused_dev");
    SRX_REQUIRE(ctx, A && feats_dev && gid_dev && stats12_dev && T >= 0 && ld >= A->D, "srx_em_fused_dev: bad argument");
    SRX_REQUIRE(ctx, (logp_dev == nullptr) == (shift_dev == nullptr), "srx_em_fused_dev: logp_dev and shift_dev go together");
    SRX_REQUIRE(ctx, srx_em_fused_applicable(A->G, A->M, A->D),
                "srx_em_fused_dev: needs M <= 16, D <= 39, G <= 16383 (have M %d, D %d, G %d); use the srx_em_shift/pass1/pass2 steps", A->M, A->D,
                A->G);
    int rc = ensure_lconst(ctx, A);
    if (rc != SRX_OK) return rc;
    return srx_em_fused_launch(ctx, feats_dev, ld, gid_dev, T, A->G, A->M, A->D, A->mean, A->lconst + (size_t)A->G * A->M * A->D, A->weight,
                               logp_dev, shift_dev, stats12_dev);
}

extern "C" int srx_em_fused_finish_dev(srx_ctx *ctx, srx_model *A, srx_model *B, const double *stats12_dev, const float *shift_dev,
                                       float blend_keep, int32_t *underflow_dev) {
    SRX_ENTER(ctx, "srx_em_fused_finish_dev");
    SRX_REQUIRE(ctx, A && B && stats12_dev, "srx_em_fused_finish_dev: null argument");
    SRX_REQUIRE(ctx, A->G == B->G && A->M == B->M && A->D == B->D, "srx_em_fused_finish_dev: model shape mismatch");
    SRX_REQUIRE(ctx, srx_em_fused_applicable(A->G, A->M, A->D), "srx_em_fused_finish_dev: shape outside the fused path");
    int rc = ensure_lconst(ctx, A);
    if (rc != SRX_OK) return rc;
    const size_t nl = (size_t)A->G * A->M * A->D;
    rc = srx_em_fused_finish_launch(ctx, A->G, A->M, A->D, stats12_dev, shift_dev, A->lconst, A->lconst + 2 * nl, A->mean, A->var, A->weight,
                                    A->logwt, B->mean, B->var, B->weight, B->logwt, blend_keep, underflow_dev);
    if (rc != SRX_OK) return rc;
    B->version++;
    return SRX_OK;
}

extern "C" int srx_em_finish_dev(srx_ctx *ctx, srx_model *A, srx_model *B, const double *stats1_dev, const double *stats2_dev,
                                 float blend_keep) {
    SRX_ENTER(ctx, "srx_em_finish_dev");
    SRX_REQUIRE(ctx, A && B && stats1_dev && stats2_dev, "srx_em_finish_dev: null argument");
    SRX_REQUIRE(ctx, A->G == B->G && A->M == B->M && A->D == B->D, "srx_em_finish_dev: model shape mismatch");
    const int64_t n = (int64_t)B->G * B->M * B->D;
    {
        SrxProf prof_(ctx, "k_em_finish");
        hipLaunchKernelGGL(k_em_finish, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, B->G, B->M, B->D, stats1_dev,
                           stats2_dev, A->mean, A->var, A->weight, B->mean, B->var, B->weight, B->logwt, blend_keep);
    }
    SRX_LAUNCH_CHECK(ctx);
    B->version++;
    return SRX_OK;
}
