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
#ifndef SRX_V_NW
#define SRX_V_NW 8
#endif
static constexpr int NW = SRX_V_NW;       // waves per workgroup
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
// packed layout: [row tile][k block = s/4][h = 0,1][row i = 0..31][u = s%4]: the four k-steps of a register block are
// one 16-byte LDS read per lane (ds_read_b128)
//   s < D : h=0 -> coefficient of x'_d^2 = -1/(2 var), h=1 -> coefficient of x'_d = mu'/var
//   s = D : h=0 -> constant term, h=1 -> 0;   s > D : 0 (K padded to a multiple of 8 steps)
// With more than one component everything is pre-multiplied by log2(e): the accumulators are then
// log2-densities and the log-sum-exp runs on v_exp_f32 / v_log_f32 without per-element scaling.
__global__ __launch_bounds__(256) void k_pack(const double *__restrict__ mean, const double *__restrict__ var,
                                              const float *__restrict__ logwt, const float *__restrict__ center, int G, int M,
                                              int D, int KSP, int MP, int Muse, int ntiles, float *__restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)ntiles * KSP * 64) return;
    const int u = (int)(idx & 3);
    const int i = (int)((idx >> 2) & 31);
    const int h = (int)((idx >> 7) & 1);
    const int kblk = (int)((idx >> 8) % (KSP / 4));
    const int tile = (int)((idx >> 8) / (KSP / 4));
    const int s = kblk * 4 + u;
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
    if (m->packed_version == m->version && m->packed_mode == mix_mode && m->packed) return SRX_OK;
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
    m->packed_version = m->version;
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

// -DSRX_V_STAMP builds a diagnostic variant that prints, for one workgroup, the shader cycles each wave spent
// waiting for its operand DMA, in the k loop and in the epilogue (s_memtime stamps; never enabled in libsrx.so).
#ifdef SRX_V_STAMP
#define SRX_STAMP(var) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory"); __builtin_amdgcn_sched_barrier(0); }
#else
#define SRX_STAMP(var)
#endif
template <int MP, int RT>
__global__ __launch_bounds__(64 * NW, NW / 4) void k_gmm_mfma(const float *__restrict__ Xc, int64_t T, int64_t nwt, int ngroups,
                                                     const float *__restrict__ packed, int KSP, int KP, int slab, int ndma, int G,
                                                     float *__restrict__ out, int64_t st_t, int64_t st_g, float *__restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float *Pl = lds;                                   // [RT][KSP/4][64][4]
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    float *xs = lds + RT * KSP * 64 + w * (2 * slab);  // this wave's two [WF][KP] slabs (KP odd: conflict-free)
    const int j = lane & 31, h = lane >> 5;
    constexpr int NST = 16 / MP;          // states per lane half per row tile
    constexpr int SPT = 32 / MP;          // states per row tile
    constexpr int NSTORE = RT * NST > 63 ? 63 : RT * NST;  // stores per tile per lane (vmcnt is 6 bits)
    int *ctr = reinterpret_cast<int *>(lds + RT * KSP * 64 + NW * 2 * slab);  // tile hand-out counter
    const float4 *pl4 = reinterpret_cast<const float4 *>(Pl) + lane;
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
    // Inside a workgroup the tiles are handed to the waves dynamically (see below).
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
            if (tid == 0) *ctr = 0;
        }
        __syncthreads();

        // Tiles of this segment are numbered 0 .. nseg-1.  The first NW are handed out statically (wave w
        // takes tile w); after that every wave draws its next tile from an LDS counter one tile ahead of
        // need.  The older wave of a SIMD wins the MFMA arbitration and runs faster than the younger one:
        // with a static split it would finish early and leave the younger wave running alone.
        const int nseg = (int)(t_hi - t_lo);
        int cur = w;
        int buf = 0;
        if (cur < nseg) dma(t_lo + cur, 0);
#ifdef SRX_V_STAMP
        unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0, sum_wait = 0, sum_k = 0, sum_epi = 0, ntile = 0, tbeg = 0, tend = 0;
        SRX_STAMP(tbeg)
#endif
        bool first = true;
        for (; cur < nseg; buf ^= 1) {
            SRX_STAMP(ts0)
            int drawn = 0;
            if (lane == 0) drawn = atomicAdd(ctr, 1);
            // in flight, oldest first: [this tile's DMA][previous tile's NSTORE stores]: wait for the DMA only
            // (a counted wait is only safe when at least NSTORE younger operations exist: not on the first pass)
            if (first)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NSTORE) : "memory");
            first = false;
            __builtin_amdgcn_wave_barrier();
            const int nxt = NW + __builtin_amdgcn_readfirstlane(drawn);
            const int64_t wt = t_lo + cur;
            // next tile -> the other slab (its last reader was the k-loop two tiles ago); on the last pass the
            // current tile is fetched again so that the number of outstanding operations stays the same
            dma(nxt < nseg ? t_lo + nxt : wt, buf ^ 1);

            SRX_STAMP(ts1)
            const float *xrow = xs + buf * slab + j * KP;
            f32x16 acc[RT];
            float a0[4][RT], x0[4], a1[4][RT], x1[4];
            const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#define SRX_SB __builtin_amdgcn_sched_barrier(0);
#define SRX_B(X) (h ? (X) : (X) * (X))
            // One k block = 4 k-steps x RT row tiles = 4*RT MFMAs on operands already in registers, with the
            // NEXT block's LDS reads (RT x ds_read_b128 of parameters, 2 x ds_read2_b32 of x) spread one or two
            // per group of RT MFMAs and the order pinned.  Left to itself the scheduler either sinks each read
            // next to its use (exposed LDS latency) or clumps them between blocks, and six back-to-back LDS
            // reads stall the wave's own MFMA stream.  (Measured with tools/microbench/mfma_f32_issue.hip: a wave's
            // VALU instructions do NOT hide under its own f32-input MFMAs on gfx950 -- about 5.5 cycles each are
            // added -- only another wave of the SIMD overlaps them, hence two waves per SIMD and as few VALU
            // instructions per k-step as possible.)
#define SRX_BLOCK(AC, XC, AN, XN, BN, LOADNEXT, ZERO)                                                   \
    _Pragma("unroll") for (int u = 0; u < 4; u++) {                                                     \
        if (LOADNEXT) {                                                                                 \
            _Pragma("unroll") for (int r = u; r < RT; r += 4) {                                         \
                const float4 q = pl4[(r * nblk + (BN)) * 64];                                           \
                AN[0][r] = q.x; AN[1][r] = q.y; AN[2][r] = q.z; AN[3][r] = q.w;                         \
            }                                                                                           \
            if (u < 2) {                                                                                \
                XN[2 * u] = xrow[(BN) * 4 + 2 * u];                                                     \
                XN[2 * u + 1] = xrow[(BN) * 4 + 2 * u + 1];                                             \
            }                                                                                           \
        }                                                                                               \
        const float b = SRX_B(XC[u]);                                                                   \
        _Pragma("unroll") for (int r = 0; r < RT; r++)                                                  \
            acc[r] = __builtin_amdgcn_mfma_f32_32x32x2f32(AC[u][r], b, ((ZERO) && u == 0) ? z : acc[r], 0, 0, 0); \
        SRX_SB                                                                                          \
    }
#pragma unroll
            for (int u = 0; u < 4; u++) x0[u] = xrow[u];
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const float4 q = pl4[(r * nblk) * 64];
                a0[0][r] = q.x; a0[1][r] = q.y; a0[2][r] = q.z; a0[3][r] = q.w;
            }
            SRX_SB
            SRX_BLOCK(a0, x0, a1, x1, 1, true, true)
            for (int blk = 1; blk < nblk - 1; blk += 2) {
                SRX_BLOCK(a1, x1, a0, x0, blk + 1, true, false)
                SRX_BLOCK(a0, x0, a1, x1, blk + 2, true, false)
            }
            SRX_BLOCK(a1, x1, a0, x0, 0, false, false)
#undef SRX_SB
#undef SRX_BLOCK
#undef SRX_B
            SRX_STAMP(ts2)
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
#ifdef SRX_V_STAMP
            SRX_STAMP(ts3)
            sum_wait += ts1 - ts0; sum_k += ts2 - ts1; sum_epi += ts3 - ts2; ntile++;
#endif
            cur = nxt;
        }
#ifdef SRX_V_STAMP
        SRX_STAMP(tend)
        if (blockIdx.x == 7 && lane == 0)
            printf("wave %d tiles %llu total %llu wait %llu kloop %llu epi %llu (per tile: total %llu wait %llu k %llu epi %llu)\n", w, ntile, tend - tbeg,
                   sum_wait, sum_k, sum_epi, (tend - tbeg) / (ntile ? ntile : 1), sum_wait / (ntile ? ntile : 1), sum_k / (ntile ? ntile : 1), sum_epi / (ntile ? ntile : 1));
#endif
    }
}

template <int MP, int RT>
static int launch_mfma(srx_ctx *ctx, srx_model *m, const float *Xc, int64_t T, int64_t nwt, int KSP, int KP, float *out,
                       int64_t st_t, int64_t st_g, float *dump) {
    const int ndma = (WF * KP * 4 + 1023) / 1024;
    const int slab = ndma * 256;  // floats; the last DMA piece may overrun the tile, never the slab
    const size_t lds_bytes = ((size_t)RT * KSP * 64 + (size_t)NW * 2 * slab + 64) * sizeof(float);
    SRX_REQUIRE(ctx, lds_bytes <= 160 * 1024, "srx_gmm_score: D=%d needs %zu bytes of LDS (max 160 KiB)", m->D, lds_bytes);
    static bool attr_set[SRX_MAX_DEVICES] = {};   // the attribute is per device
    if (!attr_set[ctx->device]) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_gmm_mfma<MP, RT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device] = true;
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
