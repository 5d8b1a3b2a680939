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

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr float kNegInf = -__builtin_huge_valf();

// ------------------------------------------------------------------------------------------
// model bookkeeping
// ------------------------------------------------------------------------------------------
extern "C" int srx_model_create(srx_ctx *ctx, int G, int M, int D, srx_model **out) {
    SRX_REQUIRE(nullptr, ctx, "srx_model_create: null ctx");
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
    delete m;
}

extern "C" int srx_model_set(srx_model *m, const double *mean, const double *var, const float *weight) {
    SRX_REQUIRE(nullptr, m, "srx_model_set: null model");
    srx_ctx *ctx = m->ctx;
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
    m->dirty = true;
    return SRX_OK;
}

extern "C" int srx_model_get(srx_model *m, double *mean, double *var, float *weight) {
    SRX_REQUIRE(nullptr, m, "srx_model_get: null model");
    srx_ctx *ctx = m->ctx;
    const size_t n = (size_t)m->G * m->M;
    if (mean) SRX_HIP(ctx, hipMemcpyAsync(mean, m->mean, n * m->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (var) SRX_HIP(ctx, hipMemcpyAsync(var, m->var, n * m->D * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (weight) SRX_HIP(ctx, hipMemcpyAsync(weight, m->weight, n * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    return SRX_OK;
}

extern "C" int srx_model_copy(srx_model *dst, const srx_model *src) {
    SRX_REQUIRE(nullptr, dst && src, "srx_model_copy: null model");
    srx_ctx *ctx = dst->ctx;
    SRX_REQUIRE(ctx, dst->G == src->G && dst->M == src->M && dst->D == src->D, "srx_model_copy: shape mismatch");
    const size_t n = (size_t)dst->G * dst->M;
    SRX_HIP(ctx, hipMemcpyAsync(dst->mean, src->mean, n * dst->D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dst->var, src->var, n * dst->D * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dst->weight, src->weight, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    SRX_HIP(ctx, hipMemcpyAsync(dst->logwt, src->logwt, n * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
    dst->dirty = true;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------
// EXACT path: gaussian::operator() in the reference's order (src/gmm.cpp:41-49).
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ float gaussian_logprob_exact(const float *__restrict__ fv, const double *__restrict__ mean,
                                                        const double *__restrict__ var, float logwt, int D) {
    float res = 0.0f;
    const float l2pi = (float)-0.4515827052894549;  // (float)log(M_2_PI), src/gmm.cpp:43
    for (int i = 0; i < D; i++) {
        float t = (float)((double)fv[i] - mean[i]);
        float tt = t * t;
        double term = ((double)l2pi + log(var[i])) + (double)tt / var[i];
        res = (float)((double)res + term);
    }
    float h = -0.5f * res;
    return h + logwt;
}

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
        r = gaussian_logprob_exact(fv, mean + o * D, var + o * D, logwt[o], D);
    } else {
        double s = 0.0;
        for (int m = 0; m < M; m++) {
            const size_t o = (size_t)g * M + m;
            s += exp((double)gaussian_logprob_exact(fv, mean + o * D, var + o * D, logwt[o], D));
        }
        r = (float)log(s);
    }
    out[t * st_t + (int64_t)g * st_g] = r;
}

// ------------------------------------------------------------------------------------------
// MFMA path: packing
// ------------------------------------------------------------------------------------------
// centre[d] = f32( mean over all (g,m) of mean[g][m][d] ), one block per d.
__global__ __launch_bounds__(256) void k_center(const double *__restrict__ mean, int n, int D, float *__restrict__ center) {
    const int d = blockIdx.x;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        double v = mean[(size_t)i * D + d];
        if (v == v && fabs(v) < 1e300) acc += v;  // NaN/inf means (unvisited states after EM) do not poison the shift
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    __shared__ double part[4];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) center[d] = (float)((part[0] + part[1] + part[2] + part[3]) / n);
}

// Row i (0..31) of an MFMA tile lives in lane half hr = (i>>2)&1, accumulator register r = (i&3) + 4*(i>>3).
// Register r of half hr holds component r%MP of tile-local state hr*(16/MP) + r/MP.
// packed layout: [row tile][kstep s = 0..D][h = 0,1][row i = 0..31]
//   s < D: h=0 -> coefficient of x'_d^2 = -1/(2 var), h=1 -> coefficient of x'_d = mu'/var
//   s = D: h=0 -> constant term, h=1 -> 0
__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ mean, const double *__restrict__ var,
                                              const float *__restrict__ logwt, const float *__restrict__ center, int G, int M,
                                              int D, int MP, int Muse, int ntiles, float *__restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int KS = D + 1;
    if (idx >= (int64_t)ntiles * KS * 64) return;
    const int i = (int)(idx & 31);
    const int h = (int)((idx >> 5) & 1);
    const int s = (int)((idx >> 6) % KS);
    const int tile = (int)((idx >> 6) / KS);
    const int hr = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
    const int st = hr * (16 / MP) + r / MP, c = r % MP;
    const int g = tile * (32 / MP) + st;
    float v;
    if (g >= G || c >= Muse) {
        v = (s == D && h == 0) ? kNegInf : 0.0f;
    } else {
        const size_t o = (size_t)g * M + c;
        if (s < D) {
            const double iv = 1.0 / var[o * D + s];
            v = h == 0 ? (float)(-0.5 * iv) : (float)(iv * (mean[o * D + s] - (double)center[s]));
        } else if (h == 1) {
            v = 0.0f;
        } else {
            const double l2pi = (double)(float)-0.4515827052894549;
            double acc = 0.0;
            for (int d = 0; d < D; d++) {
                const double mu = mean[o * D + d] - (double)center[d];
                const double s2 = var[o * D + d];
                acc += l2pi + log(s2) + mu * mu / s2;
            }
            v = (float)((double)logwt[o] - 0.5 * acc);
        }
    }
    packed[idx] = v;
}

static int padded_components(int M) {
    int mp = 1;
    while (mp < M) mp <<= 1;
    return mp;
}

static int ensure_packed(srx_ctx *ctx, srx_model *m, int mix_mode) {
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    const int MP = padded_components(Muse);
    if (MP > 16) return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "MFMA scoring supports at most 16 mixture components (got %d)", Muse);
    if (!m->dirty && m->packed_mode == mix_mode && m->packed) return SRX_OK;
    const int spt = 32 / MP;  // states per row tile
    int ntiles = (m->G + spt - 1) / spt;
    ntiles = (ntiles + 3) / 4 * 4;  // whole groups of 4 tiles so any RT divides
    const size_t bytes = (size_t)ntiles * (m->D + 1) * 64 * sizeof(float);
    if (bytes > m->packed_bytes) {
        (void)hipFree(m->packed);
        m->packed = nullptr;
        m->packed_bytes = 0;
        SRX_HIP(ctx, hipMalloc((void **)&m->packed, bytes));
        m->packed_bytes = bytes;
    }
    hipLaunchKernelGGL(k_center, dim3(m->D), dim3(256), 0, ctx->stream, m->mean, m->G * m->M, m->D, m->center);
    SRX_LAUNCH_CHECK(ctx);
    const int64_t n = (int64_t)ntiles * (m->D + 1) * 64;
    hipLaunchKernelGGL(k_pack, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, m->mean, m->var, m->logwt, m->center,
                       m->G, m->M, m->D, MP, Muse, ntiles, m->packed);
    SRX_LAUNCH_CHECK(ctx);
    m->dirty = false;
    m->packed_mode = mix_mode;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------
// MFMA path: scoring kernel.  Workgroup = 4 waves sharing one parameter tile of RT row tiles
// (32 Gaussians each) in LDS; every wave walks its own sequence of 32-frame tiles (no workgroup
// barrier inside the loop, so the waves of a SIMD drift apart and one's epilogue/stores overlap
// the other's MFMAs).  out element (t, g) lives at out[t*st_t + g*st_g]: st_t == 1 (state-major)
// makes every store 128 contiguous bytes per half-wave.
// ------------------------------------------------------------------------------------------
static constexpr int WF = 32;        // frames per wave tile (MFMA columns)
static constexpr int NXR = 20;       // staging registers per lane: covers WF*D <= 1280 (D <= 40) in one batch

// Copies this wave's WF frames (rows of D floats, row stride ld) into its LDS slab xs[frame][DP]
// through registers.  `load` issues the global loads (so they can fly during the MFMA loop of the
// previous tile), `store` writes them to LDS.  Rows past T read as zero.
template <bool CONTIG>
__device__ __forceinline__ void stage_load(float (&rx)[NXR], const float *__restrict__ feats, int ld, int64_t T, int64_t t0, int D,
                                           int lane, int batch) {
    const int nel = WF * D;
    const int64_t lim = (T - t0) * (int64_t)D;  // elements that exist (contiguous case)
#pragma unroll
    for (int i = 0; i < NXR; i++) {
        const int idx = batch * (NXR * 64) + i * 64 + lane;
        float v = 0.0f;
        if (CONTIG) {
            if (idx < nel && idx < lim) v = feats[t0 * ld + idx];
        } else {
            const int t = idx / D, d = idx - t * D;
            if (idx < nel && t0 + t < T) v = feats[(t0 + t) * ld + d];
        }
        rx[i] = v;
    }
}
template <bool CONTIG>
__device__ __forceinline__ void stage_store(const float (&rx)[NXR], float *xs, int D, int DP, int lane, int batch) {
    const int nel = WF * D;
#pragma unroll
    for (int i = 0; i < NXR; i++) {
        const int idx = batch * (NXR * 64) + i * 64 + lane;
        if (idx < nel) {
            if (CONTIG && DP == D) {
                xs[idx] = rx[i];
            } else {
                const int t = idx / D, d = idx - t * D;
                xs[t * DP + d] = rx[i];
            }
        }
    }
}

template <int MP, int RT, bool CONTIG>
__global__ __launch_bounds__(256, 2) void k_gmm_mfma(const float *__restrict__ feats, int ld, int64_t T,
                                                     const float *__restrict__ center, const float *__restrict__ packed, int D,
                                                     int DP, int G, float *__restrict__ out, int64_t st_t, int64_t st_g) {
    extern __shared__ float lds[];
    const int KS = D + 1;
    float *Pl = lds;                                   // [RT][KS][64]
    float *cl = lds + RT * KS * 64;                    // [D] centring vector
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    float *xs = cl + ((D + 3) & ~3) + w * (WF * DP);   // this wave's [WF][DP] (DP odd: conflict-free column reads)
    const int j = lane & 31, h = lane >> 5;

    {
        const float4 *src = reinterpret_cast<const float4 *>(packed + (size_t)blockIdx.y * RT * KS * 64);
        float4 *dst = reinterpret_cast<float4 *>(Pl);
        const int n4 = RT * KS * 16;
        for (int i = tid; i < n4; i += 256) dst[i] = src[i];
        for (int i = tid; i < D; i += 256) cl[i] = center[i];
    }
    __syncthreads();
    const int64_t nwt = (T + WF - 1) / WF;
    constexpr int NST = 16 / MP;          // states per lane half per row tile
    constexpr int SPT = 32 / MP;          // states per row tile
    const float *pl = Pl + lane;
    const float *xrow = xs + j * DP;
    const int nbatch = (WF * D + NXR * 64 - 1) / (NXR * 64);
    const int64_t wstride = (int64_t)gridDim.x * 4;
    int64_t wt = (int64_t)blockIdx.x * 4 + w;
    float rx[NXR];
    if (wt < nwt) stage_load<CONTIG>(rx, feats, ld, T, wt * WF, D, lane, 0);
    for (; wt < nwt; wt += wstride) {
        const int64_t t0 = wt * WF;
        // ---- finish staging tile wt (batch 0 is already in registers) ----
        stage_store<CONTIG>(rx, xs, D, DP, lane, 0);
        for (int bt = 1; bt < nbatch; bt++) {
            stage_load<CONTIG>(rx, feats, ld, T, t0, D, lane, bt);
            stage_store<CONTIG>(rx, xs, D, DP, lane, bt);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- prefetch the next tile's first batch; it lands while the MFMAs run ----
        if (wt + wstride < nwt) stage_load<CONTIG>(rx, feats, ld, T, (wt + wstride) * WF, D, lane, 0);

        f32x16 acc[RT];
#pragma unroll
        for (int r = 0; r < RT; r++)
#pragma unroll
            for (int e = 0; e < 16; e++) acc[r][e] = 0.0f;
        // software-pipelined k loop: operands of step s+1 are read while step s multiplies
        float an[RT], xn, cn;
#pragma unroll
        for (int r = 0; r < RT; r++) an[r] = pl[(r * KS) * 64];
        xn = xrow[0];
        cn = cl[0];
        for (int s = 0; s < D; s++) {
            float ac[RT];
#pragma unroll
            for (int r = 0; r < RT; r++) ac[r] = an[r];
            const float xv = xn - cn;
            const int sn = s + 1;
#pragma unroll
            for (int r = 0; r < RT; r++) an[r] = pl[(r * KS + sn) * 64];
            if (sn < D) {
                xn = xrow[sn];
                cn = cl[sn];
            }
            const float b = h ? xv : xv * xv;
#pragma unroll
            for (int r = 0; r < RT; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(ac[r], b, acc[r], 0, 0, 0);
        }
        {
            const float b = h ? 0.0f : 1.0f;
#pragma unroll
            for (int r = 0; r < RT; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(an[r], b, acc[r], 0, 0, 0);
        }
        __builtin_amdgcn_wave_barrier();
        // ---- epilogue: per-state log-sum-exp over the lane's registers, then store ----
        const int64_t fr = t0 + j;
        if (fr < T) {
            float *orow = out + fr * st_t;
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int gbase = ((int)blockIdx.y * RT + r) * SPT + h * NST;
#pragma unroll
                for (int st = 0; st < NST; st++) {
                    float v;
                    if (MP == 1) {
                        v = acc[r][st];
                    } else {
                        float mx = acc[r][st * MP];
#pragma unroll
                        for (int c = 1; c < MP; c++) mx = fmaxf(mx, acc[r][st * MP + c]);
                        float sum = 0.0f;
#pragma unroll
                        for (int c = 0; c < MP; c++) sum += __expf(acc[r][st * MP + c] - mx);
                        v = mx + __logf(sum);
                    }
                    const int g = gbase + st;
                    if (g < G) orow[(int64_t)g * st_g] = v;
                }
            }
        }
    }
}

template <int MP, int RT, bool CONTIG>
static int launch_mfma2(srx_ctx *ctx, srx_model *m, const float *feats, int ld, int64_t T, float *out, int64_t st_t, int64_t st_g) {
    const int KS = m->D + 1;
    const int DP = m->D | 1;  // odd row stride
    const size_t lds_bytes = ((size_t)RT * KS * 64 + ((m->D + 3) & ~3) + (size_t)4 * WF * DP) * sizeof(float);
    SRX_REQUIRE(ctx, lds_bytes <= 160 * 1024, "srx_gmm_score: D=%d needs %zu bytes of LDS", m->D, lds_bytes);
    static bool attr_set = false;
    if (!attr_set) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_gmm_mfma<MP, RT, CONTIG>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         160 * 1024));
        attr_set = true;
    }
    const int spt = 32 / MP;
    const int ntiles = (m->G + spt - 1) / spt;
    const int ngroups = (ntiles + RT - 1) / RT;
    const int64_t nbt = (T + 4 * WF - 1) / (4 * WF);   // workgroup-sized frame tiles
    // ~16 workgroups per resident slot (2 per CU) for load balance; each reloads a 40 KB parameter tile
    int64_t gx = (int64_t)(32 * ctx->num_cus + ngroups - 1) / ngroups;
    if (gx > nbt) gx = nbt;
    if (gx < 1) gx = 1;
    hipLaunchKernelGGL((k_gmm_mfma<MP, RT, CONTIG>), dim3((unsigned)gx, (unsigned)ngroups), dim3(256), lds_bytes, ctx->stream, feats,
                       ld, T, m->center, m->packed, m->D, DP, m->G, out, st_t, st_g);
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

template <int MP, int RT>
static int launch_mfma(srx_ctx *ctx, srx_model *m, const float *feats, int ld, int64_t T, float *out, int64_t st_t, int64_t st_g) {
    if (ld == m->D) return launch_mfma2<MP, RT, true>(ctx, m, feats, ld, T, out, st_t, st_g);
    return launch_mfma2<MP, RT, false>(ctx, m, feats, ld, T, out, st_t, st_g);
}

template <int MP>
static int dispatch_rt(srx_ctx *ctx, srx_model *m, const float *feats, int ld, int64_t T, float *out, int64_t st_t, int64_t st_g) {
    const int spt = 32 / MP;
    const int ntiles = (m->G + spt - 1) / spt;
    if (ntiles >= 4) return launch_mfma<MP, 4>(ctx, m, feats, ld, T, out, st_t, st_g);
    if (ntiles >= 2) return launch_mfma<MP, 2>(ctx, m, feats, ld, T, out, st_t, st_g);
    return launch_mfma<MP, 1>(ctx, m, feats, ld, T, out, st_t, st_g);
}

extern "C" int srx_gmm_score_dev(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats_dev, int ld,
                                 int64_t T, float *out_dev, int64_t stride_t, int64_t stride_g) {
    SRX_REQUIRE(nullptr, ctx, "srx_gmm_score_dev: null ctx");
    SRX_REQUIRE(ctx, m && feats_dev && out_dev, "srx_gmm_score_dev: null argument");
    SRX_REQUIRE(ctx, ld >= m->D && T >= 0, "srx_gmm_score_dev: ld=%d < D=%d", ld, m->D);
    SRX_REQUIRE(ctx, (stride_g == 1 && stride_t >= m->G) || (stride_t == 1 && stride_g >= T),
                "srx_gmm_score_dev: output strides must be frame-major (stride_g=1, stride_t>=G) or state-major (stride_t=1, stride_g>=T)");
    SRX_REQUIRE(ctx, mix_mode == SRX_MIX_COMPONENT0 || mix_mode == SRX_MIX_LOGSUMEXP, "srx_gmm_score_dev: bad mix_mode");
    if (T == 0) return SRX_OK;
    if (score_path == SRX_SCORE_EXACT) {
        const int64_t n = T * m->G;
        SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_gmm_score_dev: too large");
        hipLaunchKernelGGL(k_gmm_exact, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, feats_dev, ld, T, m->G, m->M,
                           m->D, m->mean, m->var, m->logwt, mix_mode, out_dev, stride_t, stride_g);
        SRX_LAUNCH_CHECK(ctx);
        return SRX_OK;
    }
    SRX_REQUIRE(ctx, score_path == SRX_SCORE_MFMA, "srx_gmm_score_dev: bad score_path");
    int rc = ensure_packed(ctx, m, mix_mode);
    if (rc != SRX_OK) return rc;
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    switch (padded_components(Muse)) {
        case 1: return dispatch_rt<1>(ctx, m, feats_dev, ld, T, out_dev, stride_t, stride_g);
        case 2: return dispatch_rt<2>(ctx, m, feats_dev, ld, T, out_dev, stride_t, stride_g);
        case 4: return dispatch_rt<4>(ctx, m, feats_dev, ld, T, out_dev, stride_t, stride_g);
        case 8: return dispatch_rt<8>(ctx, m, feats_dev, ld, T, out_dev, stride_t, stride_g);
        case 16: return dispatch_rt<16>(ctx, m, feats_dev, ld, T, out_dev, stride_t, stride_g);
    }
    return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_gmm_score_dev: unsupported component count");
}

extern "C" int srx_gmm_score(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats, int ld, int64_t T,
                             float *out) {
    SRX_REQUIRE(nullptr, ctx, "srx_gmm_score: null ctx");
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
