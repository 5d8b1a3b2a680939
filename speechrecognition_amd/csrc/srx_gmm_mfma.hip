// GMM emission scoring as an f32 MFMA contraction (the SRX_SCORE_MFMA path of srx_gmm_score_dev).
//
// Replaces gaussian::operator() (src/gmm.cpp:41-49) + gmm::operator() (src/gmm.cpp:99-109) +
// acoustic_model::operator() (src/gmm.cpp:184-186) for all frames x all tied states.
//
//   logp_{g,m}(x) = const_{g,m} + sum_d [ (-1/(2 s2_d)) x'_d^2 + (mu'_d / s2_d) x'_d ],   x' = x - c, mu' = mu - c
// (c = mean of all component means: removes the cancellation of the expanded square), i.e.
//   [Gaussians x K] . [K x frames],  K = 2D+2 :  rows (x'^2_0, x'_0, x'^2_1, x'_1, ..., 1, 0).
//
// Three launches:
//   k_center / k_pack  (only when the model changed)  parameters -> packed A operand, MFMA lane order
//   k_gmm_prep         features -> centred operand rows Xc[T][KP] = (x'_0 .. x'_{D-1}, 1, 0...), 164 B/frame at D = 39
//   k_gmm_mfma         the contraction on v_mfma_f32_32x32x2_f32 + in-lane log-sum-exp + store
//
// k_gmm_mfma: workgroup = 8 waves (one per CU, two per SIMD) sharing one parameter tile of RT row
// tiles (32 Gaussians each) in LDS.  Every wave walks its own sequence of 32-frame tiles: the tile's
// operand rows are contiguous in Xc and are copied into one of the wave's two LDS slabs by LDS-DMA
// (global_load_lds_dwordx4: no VGPRs, no VALU) a whole tile ahead, so the copy flies under the MFMAs;
// the wave waits with a counted vmcnt that leaves its own output stores in flight.  x'^2 is formed in
// registers (lanes 0-31 feed the quadratic row of a k-step, lanes 32-63 the linear one).  Gaussians are the
// MFMA rows and frames the columns, so one lane's 16 accumulator registers hold the components of whole
// states for one frame: the mixture log-sum-exp needs no cross-lane traffic.  No workgroup barrier
// inside the loop: the two waves of a SIMD drift apart and overlap MFMA with VALU/stores.
#include <cmath>

#include "srx_common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

static constexpr float kNegInf = -__builtin_huge_valf();
static constexpr int WF = 32;      // frames per wave tile (MFMA columns)
static constexpr int NW = 8;       // waves per workgroup
static constexpr float kLn2 = 0.69314718055994530942f;

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
// packed layout: [row tile][kstep s = 0..KSP)[h = 0,1][row i = 0..31]
//   s < D : h=0 -> coefficient of x'_d^2 = -1/(2 var), h=1 -> coefficient of x'_d = mu'/var
//   s = D : h=0 -> constant term, h=1 -> 0;   s > D : 0 (K padded to a multiple of 8 steps)
// With more than one component everything is pre-multiplied by log2(e): the accumulators are then
// log2-densities and the log-sum-exp runs on v_exp_f32 / v_log_f32 without per-element scaling.
__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ mean, const double *__restrict__ var,
                                              const float *__restrict__ logwt, const float *__restrict__ center, int G, int M,
                                              int D, int KSP, int MP, int Muse, int ntiles, float *__restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)ntiles * KSP * 64) return;
    const int i = (int)(idx & 31);
    const int h = (int)((idx >> 5) & 1);
    const int s = (int)((idx >> 6) % KSP);
    const int tile = (int)((idx >> 6) / KSP);
    const int hr = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
    const int st = hr * (16 / MP) + r / MP, c = r % MP;
    const int g = tile * (32 / MP) + st;
    const double scale = MP > 1 ? 1.4426950408889634 : 1.0;
    float v = 0.0f;
    if (s <= D) {
        if (g >= G || c >= Muse) {
            v = (s == D && h == 0) ? kNegInf : 0.0f;
        } else {
            const size_t o = (size_t)g * M + c;
            if (s < D) {
                const double iv = 1.0 / var[o * D + s];
                v = h == 0 ? (float)(scale * -0.5 * iv) : (float)(scale * iv * (mean[o * D + s] - (double)center[s]));
            } else if (h == 0) {
                const double l2pi = (double)(float)-0.4515827052894549;  // (float)log(M_2_PI), src/gmm.cpp:43
                double acc = 0.0;
                for (int d = 0; d < D; d++) {
                    const double mu = mean[o * D + d] - (double)center[d];
                    const double s2 = var[o * D + d];
                    acc += l2pi + log(s2) + mu * mu / s2;
                }
                v = (float)(scale * ((double)logwt[o] - 0.5 * acc));
            }
        }
    }
    packed[idx] = v;
}

static int padded_components(int M) {
    int mp = 1;
    while (mp < M) mp <<= 1;
    return mp;
}

static int ensure_packed(srx_ctx *ctx, srx_model *m, int mix_mode, int KSP) {
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    const int MP = padded_components(Muse);
    if (MP > 16) return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "MFMA scoring supports at most 16 mixture components (got %d)", Muse);
    if (!m->dirty && m->packed_mode == mix_mode && m->packed) return SRX_OK;
    const int spt = 32 / MP;  // states per row tile
    int ntiles = (m->G + spt - 1) / spt;
    ntiles = (ntiles + 3) / 4 * 4;  // whole groups of 4 tiles so any RT divides
    const size_t bytes = ((size_t)ntiles * KSP * 64 + 1024) * sizeof(float);
    if (bytes > m->packed_bytes) {
        (void)hipFree(m->packed);
        m->packed = nullptr;
        m->packed_bytes = 0;
        SRX_HIP(ctx, hipMalloc((void **)&m->packed, bytes));
        m->packed_bytes = bytes;
    }
    {
        SrxProf prof_(ctx, "k_center");
        hipLaunchKernelGGL(k_center, dim3(m->D), dim3(256), 0, ctx->stream, m->mean, m->G * m->M, m->D, m->center);
    }
    SRX_LAUNCH_CHECK(ctx);
    const int64_t n = (int64_t)ntiles * KSP * 64;
    {
        SrxProf prof_(ctx, "k_pack");
        hipLaunchKernelGGL(k_pack, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, m->mean, m->var, m->logwt, m->center,
                           m->G, m->M, m->D, KSP, MP, Muse, ntiles, m->packed);
    }
    SRX_LAUNCH_CHECK(ctx);
    m->dirty = false;
    m->packed_mode = mix_mode;
    return SRX_OK;
}

// Operand rows: Xc[t][k], k < KP:  k < D -> x'_k = x_k - c_k, k = D -> 1 (the constant's k-step), else 0.
// Rows T .. Tpad-1 (padding to whole wave tiles) are all zero.
__global__ __launch_bounds__(256) void k_gmm_prep(const float *__restrict__ feats, int ld, int64_t T, int64_t Tpad, int D, int KP,
                                                  const float *__restrict__ center, float *__restrict__ Xc) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= Tpad * KP) return;
    const int64_t t = idx / KP;
    const int k = (int)(idx - t * KP);
    float v = 0.0f;
    if (t < T) {
        if (k < D)
            v = feats[t * ld + k] - center[k];
        else if (k == D)
            v = 1.0f;
    }
    Xc[idx] = v;
}

template <int MP, int RT>
__global__ __launch_bounds__(512, 2) void k_gmm_mfma(const float *__restrict__ Xc, int64_t T, int64_t nwt, int ngroups,
                                                     const float *__restrict__ packed, int KSP, int KP, int slab, int ndma, int G,
                                                     float *__restrict__ out, int64_t st_t, int64_t st_g, float *__restrict__ dump) {
    extern __shared__ float lds[];
    float *Pl = lds;                                   // [RT][KSP][64]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *xs = lds + RT * KSP * 64 + w * (2 * slab);  // this wave's two [WF][KP] slabs (KP odd: conflict-free)
    const int j = lane & 31, h = lane >> 5;
    constexpr int NST = 16 / MP;          // states per lane half per row tile
    constexpr int SPT = 32 / MP;          // states per row tile
    constexpr int NSTORE = RT * NST > 63 ? 63 : RT * NST;  // stores per tile per lane (vmcnt is 6 bits)
    const float *pl = Pl + lane;
    const int nblk = KSP / 4;             // even
    const size_t tile_floats = (size_t)WF * KP;

    auto dma = [&](int64_t wt, int buf) {
        // one wave tile = WF*KP contiguous floats of Xc -> one of this wave's slabs, 1 KiB per instruction
        const char *g = reinterpret_cast<const char *>(Xc + wt * tile_floats) + lane * 16;
        float *dst = xs + buf * slab;
        for (int i = 0; i < ndma; i++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + i * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + i * 256), 16, 0, 0);
    };

    // Static, even partition of the (parameter group, wave tile) space over the workgroups: the grid
    // is one workgroup per CU, and workgroup b owns the contiguous range [lo, hi) of group-major tile
    // indices, so it reloads the parameter tile only when its range crosses into the next group.
    const int64_t ntot = (int64_t)ngroups * nwt;
    const int64_t lo = ntot * blockIdx.x / gridDim.x, hi = ntot * (blockIdx.x + 1) / gridDim.x;
    for (int64_t base = lo; base < hi;) {
        const int grp = (int)(base / nwt);
        const int64_t t_lo = base - (int64_t)grp * nwt;
        const int64_t t_hi = (hi - (int64_t)grp * nwt) < nwt ? (hi - (int64_t)grp * nwt) : nwt;
        base = (int64_t)grp * nwt + t_hi;

        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();  // every wave is done with the previous parameter tile
        {
            const float4 *src = reinterpret_cast<const float4 *>(packed + (size_t)grp * RT * KSP * 64);
            float4 *dst = reinterpret_cast<float4 *>(Pl);
            const int n4 = RT * KSP * 16;
            for (int i = tid; i < n4; i += 64 * NW) dst[i] = src[i];
        }
        __syncthreads();

        int64_t wt = t_lo + w;
        int buf = 0;
        if (wt < t_hi) dma(wt, 0);
        bool first = true;
        for (; wt < t_hi; wt += NW, buf ^= 1) {
            // in flight, oldest first: [this tile's DMA][previous tile's NSTORE stores]: wait for the DMA only
            // (a counted wait is only safe when at least NSTORE younger operations exist: not on the first pass)
            if (first)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            first = false;
            __builtin_amdgcn_wave_barrier();
            // next tile -> the other slab (its last reader was the k-loop two tiles ago); clamped so the
            // number of outstanding operations is the same on the last pass
            dma(wt + NW < t_hi ? wt + NW : wt, buf ^ 1);

            const float *xrow = xs + buf * slab + j * KP;
            f32x16 acc[RT];
            float a0[4][RT], x0[4], a1[4][RT], x1[4];
#define SRX_LDBLK(A, X, BLK)                                                            \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                     \
        X[u] = xrow[(BLK) * 4 + u];                                                     \
        _Pragma("unroll") for (int r = 0; r < RT; r++) A[u][r] = pl[(r * KSP + (BLK) * 4 + u) * 64]; \
    }
#define SRX_B(X) (h ? (X) : (X) * (X))
#define SRX_MMBLK(A, X)                                                                 \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                     \
        const float b = SRX_B(X[u]);                                                    \
        _Pragma("unroll") for (int r = 0; r < RT; r++)                                  \
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(A[u][r], b, acc[r], 0, 0, 0); \
    }
            SRX_LDBLK(a0, x0, 0)
            SRX_LDBLK(a1, x1, 1)
            {
                const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                const float b0 = SRX_B(x0[0]);
#pragma unroll
                for (int r = 0; r < RT; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[0][r], b0, z, 0, 0, 0);
#pragma unroll
                for (int u = 1; u < 4; u++) {
                    const float b = SRX_B(x0[u]);
#pragma unroll
                    for (int r = 0; r < RT; r++) acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[u][r], b, acc[r], 0, 0, 0);
                }
            }
            for (int blk = 2; blk < nblk; blk += 2) {
                SRX_LDBLK(a0, x0, blk)
                SRX_MMBLK(a1, x1)
                SRX_LDBLK(a1, x1, blk + 1)
                SRX_MMBLK(a0, x0)
            }
            SRX_MMBLK(a1, x1)
#undef SRX_LDBLK
#undef SRX_MMBLK
#undef SRX_B
            // ---- epilogue: per-state log-sum-exp over the lane's registers, then exactly NSTORE stores ----
            const int64_t fr = wt * WF + j;
            float *orow = out + fr * st_t;
            int nst = 0;
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int gbase = (grp * RT + r) * SPT + h * NST;
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
                        for (int c = 0; c < MP; c++) sum += __builtin_amdgcn_exp2f(acc[r][st * MP + c] - mx);
                        v = (mx + __builtin_amdgcn_logf(sum)) * kLn2;
                    }
                    const int g = gbase + st;
                    // out-of-range lanes store to a dump slot instead of branching, so the count stays exact
                    float *dst = (fr < T && g < G) ? orow + (int64_t)g * st_g : dump + lane;
                    if (nst < NSTORE || (fr < T && g < G)) *dst = v;
                    nst++;
                }
            }
        }
    }
}

template <int MP, int RT>
static int launch_mfma(srx_ctx *ctx, srx_model *m, const float *Xc, int64_t T, int64_t nwt, int KSP, int KP, float *out,
                       int64_t st_t, int64_t st_g, float *dump) {
    const int ndma = (WF * KP * 4 + 1023) / 1024;
    const int slab = ndma * 256;  // floats; the last DMA piece may overrun the tile, never the slab
    const size_t lds_bytes = ((size_t)RT * KSP * 64 + (size_t)NW * 2 * slab + 64) * sizeof(float);
    SRX_REQUIRE(ctx, lds_bytes <= 160 * 1024, "srx_gmm_score: D=%d needs %zu bytes of LDS (max 160 KiB)", m->D, lds_bytes);
    static bool attr_set = false;
    if (!attr_set) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_gmm_mfma<MP, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set = true;
    }
    const int spt = 32 / MP;
    const int ntiles = (m->G + spt - 1) / spt;
    const int ngroups = (ntiles + RT - 1) / RT;
    const int64_t ntot = (int64_t)ngroups * nwt;
    int64_t nb = ctx->num_cus;  // one 8-wave workgroup per CU, each with an equal share of the tile space
    if (nb > (ntot + NW - 1) / NW) nb = (ntot + NW - 1) / NW;
    if (nb < 1) nb = 1;
    {
        SrxProf prof_(ctx, "k_gmm_mfma");
        hipLaunchKernelGGL((k_gmm_mfma<MP, RT>), dim3((unsigned)nb), dim3(64 * NW), lds_bytes, ctx->stream, Xc, T, nwt, ngroups,
                           m->packed, KSP, KP, slab, ndma, m->G, out, st_t, st_g, dump);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

template <int MP>
static int dispatch_rt(srx_ctx *ctx, srx_model *m, const float *Xc, int64_t T, int64_t nwt, int KSP, int KP, float *out,
                       int64_t st_t, int64_t st_g, float *dump) {
    const int spt = 32 / MP;
    const int ntiles = (m->G + spt - 1) / spt;
    if (ntiles >= 4) return launch_mfma<MP, 4>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
    if (ntiles >= 2) return launch_mfma<MP, 2>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
    return launch_mfma<MP, 1>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
}

// Called by srx_gmm_score_dev (srx_gmm.hip) after argument validation.
int srx_gmm_score_mfma(srx_ctx *ctx, srx_model *m, int mix_mode, const float *feats, int ld, int64_t T, float *out, int64_t st_t,
                       int64_t st_g) {
    const int KSP = (m->D + 1 + 7) / 8 * 8;  // k-steps, padded to two register blocks of four
    const int KP = KSP + 1;                  // operand row length (odd): x'_0..x'_{D-1}, 1, 0...
    int rc = ensure_packed(ctx, m, mix_mode, KSP);
    if (rc != SRX_OK) return rc;
    const int64_t nwt = (T + WF - 1) / WF;
    const int64_t Tpad = nwt * WF;
    void *ws = nullptr;
    // +4 KiB: the last DMA piece of a tile reads past it; +256 B: dump slots for out-of-range lanes' stores
    rc = srx_workspace(ctx, 0, ((size_t)Tpad * KP + 1024 + 64) * sizeof(float), &ws);
    if (rc != SRX_OK) return rc;
    float *Xc = reinterpret_cast<float *>(ws);
    float *dump = Xc + (size_t)Tpad * KP + 1024;
    const int64_t n = Tpad * KP;
    SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_gmm_score: too many frames");
    {
        SrxProf prof_(ctx, "k_gmm_prep");
        hipLaunchKernelGGL(k_gmm_prep, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, feats, ld, T, Tpad, m->D, KP,
                           m->center, Xc);
    }
    SRX_LAUNCH_CHECK(ctx);
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    switch (padded_components(Muse)) {
        case 1: return dispatch_rt<1>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
        case 2: return dispatch_rt<2>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
        case 4: return dispatch_rt<4>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
        case 8: return dispatch_rt<8>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
        case 16: return dispatch_rt<16>(ctx, m, Xc, T, nwt, KSP, KP, out, st_t, st_g, dump);
    }
    return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_gmm_score: unsupported component count");
}
