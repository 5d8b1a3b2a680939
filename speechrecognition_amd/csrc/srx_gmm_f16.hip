// GMM emission scoring on the f16 matrix cores with split operands (SRX_SCORE_MFMA_F16X3).
//
// Same contraction as srx_gmm_mfma.hip -- replaces gaussian::operator() (src/gmm.cpp:41-49), gmm::operator()
// (src/gmm.cpp:99-109) and acoustic_model::operator() (src/gmm.cpp:184-186) for all frames x all tied states:
//   logp_{g,m}(x) = const_{g,m} + sum_d [ (-s_d^2/(2 var_d)) (x'_d/s_d)^2 + (s_d mu'_d/var_d) (x'_d/s_d) ],  x' = x - c, mu' = mu - c
// i.e. [Gaussians x K] . [K x frames] with K = 2D+1 <= 80 -- but on v_mfma_f32_32x32x16_f16, 16x the rate of the
// f32-input MFMA.  f16 alone (11 significant bits) cannot hold the 1e-4 bar, so both operands are split
//   a = a_hi + a_lo,  b = b_hi + b_lo   (a_hi = f16(a), a_lo = f16(a - a_hi); f16 subnormals are honoured by the MFMA)
// and three products are accumulated in f32:  a_lo b_hi + a_hi b_lo + a_hi b_hi.  The dropped a_lo b_lo term and the
// rounding of the lo parts are ~2^-22 relative per product (measured against f64: <= 2e-6 of max(|logp|, 1)).
// c centres the data (mean of all component means); s_d is a power of two per dimension (span of the model / 8), so
// that the frame operand is O(1..30) where the model lives: exact to apply, keeps the lo parts out of the subnormal
// range and squares far from 65504.
//
// What does not fit is handled explicitly, never silently:
//   * a model coefficient beyond the f16 range (variance below ~1e-5 of the span^2): k_pack16 raises a flag, the
//     host reads it when the model changes, and scoring goes to the f32-input path (srx_gmm_mfma.hip);
//   * a frame further than ~30 spans from the centre, or NaN: k_prep16 flags the frame, k_gmm_fixup16 re-evaluates its
//     row with the exact per-dimension formula.
//
// Data movement.  Both operands are stored in MFMA fragment order, one 16-byte piece per lane:
//   A' [row tile][k-step 0..2KS)[lane][8 halves]   hi segments (k-steps 0..KS) then lo segments; 1 KiB per (tile, k-step)
//   B' [frame tile][k-step 0..2KS)[lane][8 halves]
// A workgroup (8 waves, 2 per SIMD) owns 512 frames and one eighth of the rows.  Each wave keeps the fragments of its
// 64 frames in registers for the whole kernel (B stationary: 2 tiles x 2KS x 4 VGPRs) and streams row tiles from LDS,
// where the workgroup double-buffers chunks of NT row tiles with LDS-DMA.  The eighth of the rows a workgroup works on
// is blockIdx.x % 8, i.e. its XCD under round-robin placement, so each XCD's L2 holds "its" 1/8 of A' (1.25 MB at
// 2000 x 16 x 39) and every chunk after the first comes from L2.  Gaussians are the MFMA rows, frames the columns:
// a lane's 16 accumulators are the components of whole states for one frame, the mixture log-sum-exp is in-lane.
#include <atomic>
#include <cmath>

#include "srx_common.h"
#include <utility>

#include "srx_gmm_f16.h"

int padded_components16(int M) {
    int mp = 1;
    while (mp < M) mp <<= 1;
    return mp;
}

// ------------------------------------------------------------------------------------------
// model side
// ------------------------------------------------------------------------------------------
// colstats[0][d] = centre c_d, [1][d] = s_d, [2][d] = 1/s_d.  One block per dimension.
__global__ __launch_bounds__(256) void k_colstats16(const double *__restrict__ mean, const double *__restrict__ var, int n, int D,
                                                    float *__restrict__ cs) {
    const int d = blockIdx.x;
    __shared__ double red[4];
    __shared__ float cshared;
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        double v = mean[(size_t)i * D + d];
        if (v == v && fabs(v) < 1e300) acc += v;
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) cshared = (float)((red[0] + red[1] + red[2] + red[3]) / n);
    __syncthreads();
    const double c = (double)cshared;
    double mx = 0.0, vx = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) {
        double m = mean[(size_t)i * D + d], v = var[(size_t)i * D + d];
        if (m == m && fabs(m) < 1e300) mx = fmax(mx, fabs(m - c));
        if (v == v && v > 0 && v < 1e300) vx = fmax(vx, v);
    }
    for (int o = 32; o > 0; o >>= 1) {
        mx = fmax(mx, __shfl_down(mx, o, 64));
        vx = fmax(vx, __shfl_down(vx, o, 64));
    }
    __syncthreads();
    __shared__ double red2[4];
    if ((threadIdx.x & 63) == 0) {
        red[threadIdx.x >> 6] = mx;
        red2[threadIdx.x >> 6] = vx;
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        mx = fmax(fmax(red[0], red[1]), fmax(red[2], red[3]));
        vx = fmax(fmax(red2[0], red2[1]), fmax(red2[2], red2[3]));
        double span = mx + 8.0 * sqrt(vx);
        int e = 0;
        if (span > 0 && span < 1e300) e = (int)ceil(log2(span / 8.0));
        e = e < -40 ? -40 : (e > 40 ? 40 : e);
        cs[d] = cshared;
        cs[D + d] = (float)ldexp(1.0, e);
        cs[2 * D + d] = (float)ldexp(1.0, -e);
    }
}

// One thread per (row tile, k-step, lane): 8 halves.  Row i of a tile <-> (lane half hr = (i>>2)&1, accumulator register
// r = (i&3) + 4*(i>>3)); register r of half hr is component r % MP of tile-local state hr*(16/MP) + r/MP.
__global__ __launch_bounds__(256) void k_pack16(const double *__restrict__ mean, const double *__restrict__ var,
                                                const float *__restrict__ logwt, const float *__restrict__ cs, int G, int M, int D,
                                                int KS, int MP, int Muse, int ntiles_alloc, f16x8 *__restrict__ packed,
                                                int *__restrict__ flag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)ntiles_alloc * 2 * KS * 64) return;
    const int lane = (int)(idx & 63);
    const int s = (int)((idx >> 6) % (2 * KS));
    const int tile = (int)((idx >> 6) / (2 * KS));
    const int i = lane & 31, hh = lane >> 5;
    const int hr = (i >> 2) & 1, r = (i & 3) + 4 * (i >> 3);
    const int st = hr * (16 / MP) + r / MP, c = r % MP;
    const int g = tile * (32 / MP) + st;
    const double scale = MP > 1 ? 1.4426950408889634 : 1.0;
    const bool lo = s >= KS;
    const int k0 = (lo ? s - KS : s) * 16 + hh * 8;
    f16x8 out;
    bool bad = false;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = coef16(mean, var, logwt, cs, G, M, D, Muse, g, c, k0 + e, scale);
        if (fabsf(v) > kF16Max && v == v) bad = true;   // NaN parameters pass through as NaN, like the f32 path
        const _Float16 hi = (_Float16)v;
        out[e] = lo ? (_Float16)(v - (float)hi) : hi;
    }
    packed[idx] = out;
    if (bad) atomicOr(flag, 1);
}

int ensure_packed16(srx_ctx *ctx, srx_model *m, int mix_mode, int KS) {
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    const int MP = padded_components16(Muse);
    if (m->packed16_version == m->version && m->packed16_mode == mix_mode && m->packed16) return SRX_OK;
    const int spt = 32 / MP;
    const int ntiles = (m->G + spt - 1) / spt;
    const int tps = (ntiles + NSLICE - 1) / NSLICE + 8;   // slices are rounded up to whole output groups (<= 8 tiles)
    const int ntiles_alloc = tps * NSLICE + NT16;  // whole slices, plus one chunk the last DMA may run into
    const size_t bytes = (size_t)ntiles_alloc * 2 * KS * 1024;
    if (bytes > m->packed16_bytes) {
        (void)hipFree(m->packed16);
        m->packed16 = nullptr;
        m->packed16_bytes = 0;
        SRX_HIP(ctx, hipMalloc(&m->packed16, bytes));
        m->packed16_bytes = bytes;
    }
    const int D = m->D;
    const bool first = !m->colstats;
    if (first) SRX_HIP(ctx, hipMalloc((void **)&m->colstats, 6 * (size_t)D * sizeof(float)));
    if (!m->flag16) SRX_HIP(ctx, hipMalloc((void **)&m->flag16, sizeof(int)));
    // Centre and scale of the model as it is now (candidate).  The values IN USE are kept while they still suit the model:
    // the frame-side operands depend only on them, so a caller that scores the same frames under a model that changes a
    // little from call to call (the iterations of hmm::train) can keep those operands (srx_gmm_f16_align.hip).  They are
    // replaced when a scale moved by more than 2x up / 4x down, a centre by more than two scaled units, or when the
    // coefficients no longer fit f16 with them.
    {
        SrxProf prof_(ctx, "k_colstats16");
        hipLaunchKernelGGL(k_colstats16, dim3(D), dim3(256), 0, ctx->stream, m->mean, m->var, m->G * m->M, D, m->colstats + 3 * D);
    }
    SRX_LAUNCH_CHECK(ctx);
    bool keep = false;
    if (!first && m->colstats_epoch > 0) {
        std::vector<float> h(6 * (size_t)D);
        SRX_HIP(ctx, hipMemcpyAsync(h.data(), m->colstats, h.size() * sizeof(float), hipMemcpyDeviceToHost, ctx->stream));
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        keep = true;
        for (int d = 0; d < D && keep; d++) {
            const float c0 = h[d], s0 = h[D + d], c1 = h[3 * D + d], s1 = h[4 * D + d];
            keep = s1 <= 2.0f * s0 && s1 >= 0.25f * s0 && fabsf(c1 - c0) <= 2.0f * s0;   // false for NaN
        }
    }
    auto adopt = [&]() -> int {
        SRX_HIP(ctx, hipMemcpyAsync(m->colstats, m->colstats + 3 * D, 3 * (size_t)D * sizeof(float), hipMemcpyDeviceToDevice, ctx->stream));
        // epochs are unique across models (a plan's frame-operand cache is keyed on the epoch: a model freed and another
        // allocated at its address must not look like the same centre / scale vector)
        static std::atomic<uint64_t> next_epoch{1};
        m->colstats_epoch = next_epoch.fetch_add(1);
        return SRX_OK;
    };
    if (!keep) {
        int rc = adopt();
        if (rc != SRX_OK) return rc;
    }
    const int64_t n = (int64_t)ntiles_alloc * 2 * KS * 64;
    int flag = 0;
    for (int attempt = 0; attempt < 2; attempt++) {
        SRX_HIP(ctx, hipMemsetAsync(m->flag16, 0, sizeof(int), ctx->stream));
        {
            SrxProf prof_(ctx, "k_pack16");
            hipLaunchKernelGGL(k_pack16, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, m->mean, m->var, m->logwt, m->colstats,
                               m->G, m->M, m->D, KS, MP, Muse, ntiles_alloc, (f16x8 *)m->packed16, m->flag16);
        }
        SRX_LAUNCH_CHECK(ctx);
        // the model changed: one 4-byte read-back decides whether its coefficients fit f16
        SRX_HIP(ctx, hipMemcpyAsync(&flag, m->flag16, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        if (flag == 0 || !keep) break;
        keep = false;            // the kept values no longer fit: take the model's own and pack once more
        int rc = adopt();
        if (rc != SRX_OK) return rc;
    }
    m->f16_ok = flag == 0;
    m->packed16_version = m->version;
    m->packed16_mode = mix_mode;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------
// frame side: B' fragments + per-frame "re-evaluate" flags
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_prep16(const float *__restrict__ feats, int ld, int64_t T, int64_t nft, int D, int KS,
                                                const float *__restrict__ cs, f16x8 *__restrict__ Bp, unsigned char *__restrict__ fflag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= nft * 2 * KS * 64) return;
    const int lane = (int)(idx & 63);
    const int s = (int)((idx >> 6) % (2 * KS));
    const int64_t ft = (idx >> 6) / (2 * KS);
    const int j = lane & 31, hh = lane >> 5;
    const int64_t t = ft * 32 + j;
    const bool lo = s >= KS;
    const int k0 = (lo ? s - KS : s) * 16 + hh * 8;
    bool bad = false;
    const f16x8 out = prep16_frag(feats, ld, t, t < T, D, k0, lo, cs, bad);
    Bp[idx] = out;
    if (bad && !lo) fflag[t] = 1;
}

// Frames flagged by k_prep16: the whole row of scores, exact evaluation (same code as SRX_SCORE_EXACT).
__global__ __launch_bounds__(256) void k_gmm_fixup16(const unsigned char *__restrict__ fflag, const float *__restrict__ feats, int ld,
                                                     int64_t T, int G, int M, int D, const double *__restrict__ mean,
                                                     const double *__restrict__ var, const float *__restrict__ logwt, int mode,
                                                     float *__restrict__ out, int64_t st_t, int64_t st_g) {
    // one wave per 64 frames; a flagged frame is then evaluated by the whole wave, lanes over states
    const int lane = threadIdx.x & 63;
    const int64_t t0 = ((int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) * 64;
    if (t0 >= T) return;
    const int64_t tl = t0 + lane;
    const unsigned long long mask = __ballot(tl < T && fflag[tl] != 0);
    if (mask == 0) return;
    for (int b = 0; b < 64; b++) {
        if (!((mask >> b) & 1ull)) continue;
        const int64_t t = t0 + b;
        const float *fv = feats + t * ld;
        for (int g = lane; g < G; g += 64) {
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
    }
}

// ------------------------------------------------------------------------------------------
// the contraction
// ------------------------------------------------------------------------------------------
__device__ __forceinline__ void wait_vmcnt_at_most(int n) {
    // s_waitcnt takes an immediate; n is wave-uniform
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(3)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(5)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
    }
}

// WIDE: state-major output (stride_t == 1).  The per-state results of a group of row tiles are staged in a wave-private
// LDS tile [state][32 frames] and leave as global_store_dwordx4 (four consecutive frames per lane, 1 KiB per
// instruction).  A CU retires roughly one store instruction per ~100 cycles whatever its width: with one dword per
// lane the 4 GB emission matrix cost as much as all the MFMAs and did not overlap them.
template <int MP, int KS, bool WIDE>
__global__ __launch_bounds__(64 * NW16, SRX_F16_WAVES_PER_SIMD) void k_gmm_f16(const f16x8 *__restrict__ Bp, int64_t T, const f16x8 *__restrict__ Ap,
                                                          int ntiles, int tps, int G, float *__restrict__ out, int64_t st_t,
                                                          int64_t st_g, float *__restrict__ dump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds16[];
    typedef F16Cfg<MP> Cfg;
    constexpr int KS2 = 2 * KS;
    constexpr int NT = Cfg::NT;
    constexpr int TILE_BYTES = KS2 * 1024;
    constexpr int CHUNK_BYTES = NT * TILE_BYTES;
    constexpr int NPIECE = NT * KS2;             // 1 KiB DMA pieces per chunk
    constexpr int NST = Cfg::NST, SPT = Cfg::SPT, GT = Cfg::GT, RG = Cfg::RG, U = Cfg::U;
    static_assert(NT % U == 0 && U % GT == 0 && U % 2 == 0, "chunks hold whole unrolled steps");
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int slice = blockIdx.x % NSLICE;
    const int64_t fb = blockIdx.x / NSLICE;
    // A slice holds whole unrolled steps (U tiles); the last step of the model may run up to U-1 tiles past ntiles: those
    // are padding in A' (states >= G, never stored).  The steps are dealt out evenly, and the slices that carry one step
    // more rotate with the frame block: a fixed split (7 x 128 + 104 tiles at G = 2,000) left one XCD idle for the last
    // fifth of every round, while this one shifts a slice's range by at most `extra` steps, so it stays in its XCD's L2.
    (void)tps;
    const int ntiles_r = (ntiles + U - 1) / U * U;
    const int units = ntiles_r / U, base = units / NSLICE, extra = units % NSLICE;
    int lo_u = slice * base, sz_u = base;
    const int rot = (int)(fb % NSLICE);
#pragma unroll
    for (int s2 = 0; s2 < NSLICE; s2++) {
        const int ex = ((s2 + rot) % NSLICE) < extra ? 1 : 0;
        if (s2 < slice) lo_u += ex;
        if (s2 == slice) sz_u += ex;
    }
    const int tile_lo = lo_u * U;
    const int tile_hi = tile_lo + sz_u * U;
    if (tile_lo >= tile_hi) return;
    const int nchunks = (tile_hi - tile_lo + NT - 1) / NT;
    // LDS addressing by integer byte offset through an address-space-3 pointer.  (Reads through a pointer the compiler
    // can trace to the LDS allocation get an s_waitcnt vmcnt(0) in front of them while an LDS-DMA is outstanding -- that
    // also drains every global store of the wave.  The DMA is ordered by hand below: counted vmcnt + barrier.)
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const lds_ptr lds3 = (lds_ptr)lds16;
    constexpr unsigned STAGE_BASE = 2 * CHUNK_BYTES;             // staging area, per wave [FT][RG][32] floats
    const unsigned stage_w = STAGE_BASE + w * Cfg::STAGE_FLOATS * 4;
    const unsigned stage_wr = stage_w + ((h * NST) * 32 + j) * 4;               // this lane's column, row h*NST
    const unsigned stage_rd = stage_w + ((lane >> 3) * 32 + (lane & 7) * 4) * 4;  // row lane/8, four frames from 4*(lane%8)

    auto dma = [&](int chunk, int buf) __attribute__((always_inline)) {
        // NPIECE contiguous KiB of A' -> LDS buffer `buf`; piece p is issued by wave p % NW16
        const char *g = reinterpret_cast<const char *>(Ap) + ((size_t)(tile_lo + chunk * NT)) * TILE_BYTES + lane * 16;
        unsigned char *dst = lds16 + buf * CHUNK_BYTES;
        for (int p = w; p < NPIECE; p += NW16)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(g + (size_t)p * 1024),
                                             (__attribute__((address_space(3))) void *)(dst + p * 1024), 16, 0, 0);
    };
    dma(0, 0);

    // this wave's frames: fragments stay in registers for the whole kernel
    const int64_t ft0 = (fb * NW16 + w) * FT_PER_WAVE;
    f16x8 Bf[FT_PER_WAVE][KS2];
#pragma unroll
    for (int f = 0; f < FT_PER_WAVE; f++)
#pragma unroll
        for (int s = 0; s < KS2; s++) Bf[f][s] = Bp[((ft0 + f) * KS2 + s) * 64 + lane];
    // wave-uniform: every frame of this wave exists, so a lane's four consecutive frames can leave as one dwordx4
    const bool full_frames = (ft0 + FT_PER_WAVE) * 32 <= T;

    // Software pipeline: the epilogue of row tile n-1 (VALU: max / exp2 / sum / log per state) is issued between the
    // MFMAs of row tile n, which write a second set of accumulators.  Per 32x32 tile the matrix core is busy 15 x 32
    // cycles while the epilogue needs ~320 cycles of vector issue per lane: left as two phases they serialise (the two
    // waves of a SIMD run in lockstep between workgroup barriers), interleaved they overlap.  The interleave is written
    // out by hand -- slice i of the epilogue after MFMA i, order pinned with sched_barrier -- because the scheduler
    // otherwise clusters the MFMAs.
    constexpr int NMF = 3 * KS * FT_PER_WAVE;          // MFMAs per row tile
    // two accumulator sets that swap roles from tile to tile (no register copies): one is written by the MFMAs of the
    // current tile while the epilogue consumes the other
    f32x16 accA[FT_PER_WAVE], accB[FT_PER_WAVE];
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int f = 0; f < FT_PER_WAVE; f++) accA[f] = accB[f] = z;
    LseEpilogue16<MP, KS> epi;
    int pend_stores = 0;        // store instructions issued since the last DMA issue (wave-uniform)

    // Slice `i` of the epilogue (LseEpilogue16) of the tile whose accumulators are in accp; `pr` = that tile's position in
    // its output group; tile_p = its index (non-WIDE only; < 0: none).  i and pr are compile-time.
    auto epi_slice = [&](auto ic, auto prc, const f32x16 (&accp)[FT_PER_WAVE], int tile_p) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value, pr = decltype(prc)::value;
        epi.template slice<i>(accp, [&](int f, int st, float v) __attribute__((always_inline)) {
            if (WIDE) {
                // staging row = state within the group; the offset is an immediate of the ds_write
                *reinterpret_cast<__attribute__((address_space(3))) float *>(lds3 + stage_wr + ((f * RG + pr * SPT + st) * 32) * 4) = v;
            } else {
                const int64_t fr = (ft0 + f) * 32 + j;
                const int g = tile_p * SPT + h * NST + st;
                // branch-free (a conditional store would split the basic block and undo the interleave):
                // lanes with nothing to write store to a dump slot
                float *dst = (tile_p >= 0 && fr < T && g < G) ? out + fr * st_t + (int64_t)g * st_g : dump + lane;
                *dst = v;
            }
        });
    };

    // WIDE: write the staged output group whose first state is gfirst: per instruction 8 states x 32 frames, a lane stores
    // four consecutive frames of one state (global_store_dwordx4; 4-byte alignment is enough for it, which the compiler
    // does not assume -- it splits a float4 store into x3 + x1).
    float *const out_lane = out + (int64_t)(lane >> 3) * st_g + ft0 * 32 + (lane & 7) * 4;
    auto flush_group = [&](int gfirst) __attribute__((always_inline)) {
#pragma unroll
        for (int f = 0; f < FT_PER_WAVE; f++)
#pragma unroll
            for (int b = 0; b < RG / 8; b++) {
                const f32x4 v = *reinterpret_cast<const __attribute__((address_space(3))) f32x4 *>(lds3 + stage_rd + ((f * RG + b * 8) * 32) * 4);
                const int g = gfirst + b * 8 + (lane >> 3);
                float *dst = out_lane + ((int64_t)(gfirst + b * 8) * st_g + f * 32);
                if (g < G) {
                    if (full_frames) {
                        asm volatile("global_store_dwordx4 %0, %1, off" ::"v"(dst), "v"(v) : "memory");
                    } else {
                        const int64_t fr = (ft0 + f) * 32 + (lane & 7) * 4;
                        if (fr < T) dst[0] = v[0];
                        if (fr + 1 < T) dst[1] = v[1];
                        if (fr + 2 < T) dst[2] = v[2];
                        if (fr + 3 < T) dst[3] = v[3];
                    }
                }
            }
        pend_stores += FT_PER_WAVE * (RG / 8);
    };

    // fragment n (in order of use: lo 0..KS-1, then hi 0..KS-1; storage order is hi first) of the tile at LDS offset `off`
    auto frag = [&](unsigned off, int n) __attribute__((always_inline)) {
        const int s = n < KS ? KS + n : n - KS;
        return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(lds3 + off + s * 1024);
    };

    f16x8 AfA[KS2], AfB[KS2];
    for (int c = 0; c < nchunks; c++) {
        // This wave's DMA pieces of chunk c must have landed.  In flight, oldest first: [pieces of chunk c][the stores
        // issued since those pieces were issued].  vmcnt retires in order, so waiting until at most that many operations
        // are outstanding waits for the pieces and leaves the stores in flight.  (Waves that may take the scalar-store
        // tail path, and the dword-store layout, simply drain.)
        if (c == 0 || !WIDE || !full_frames)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            wait_vmcnt_at_most(pend_stores);
        __syncthreads();                                  // ... everybody's; and everybody is done with the other buffer
        if (c + 1 < nchunks) dma(c + 1, (c + 1) & 1);
        pend_stores = 0;
        const unsigned cb = (c & 1) * CHUNK_BYTES + lane * 16;   // LDS byte offset of this lane's 16 bytes of the chunk's first fragment
        const int nt = (tile_hi - tile_lo - c * NT) < NT ? (tile_hi - tile_lo - c * NT) : NT;   // a multiple of U
        // One row tile (position u in its unrolled step): its MFMAs (fragments Afc) write `acc` while the epilogue of the
        // previous tile reads `accp` and the fragments of the next tile are fetched into Afn, one ds_read_b128 per MFMA gap
        // (their address passes through a pin, which keeps each read in its gap).  After the last tile of a chunk the
        // fetch reads past the chunk -- still inside this workgroup's LDS (the other chunk buffer or the staging area) --
        // and is not used: the first tile of a chunk is read again after the barrier.
        auto do_tile = [&](auto uc, int tt, f32x16 (&acc)[FT_PER_WAVE], const f32x16 (&accp)[FT_PER_WAVE], const f16x8 (&Afc)[KS2],
                           f16x8 (&Afn)[KS2]) __attribute__((always_inline)) {
            constexpr int u = decltype(uc)::value;
            const int tl = c * NT + tt;                   // slice-local index of this tile
            unsigned nxt = cb + (tt + 1) * TILE_BYTES;
            // small terms first: a_lo b_hi, then a_hi b_lo, then a_hi b_hi
            static_for16(std::make_integer_sequence<int, NMF>{}, [&](auto ic) __attribute__((always_inline)) {
                constexpr int i = decltype(ic)::value;
                constexpr int p = i / (KS * FT_PER_WAVE), s_ = (i / FT_PER_WAVE) % KS, f = i % FT_PER_WAVE;
                const f16x8 a_ = p == 0 ? Afc[KS + s_] : Afc[s_];
                const f16x8 b_ = p == 1 ? Bf[f][KS + s_] : Bf[f][s_];
                acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, i < FT_PER_WAVE ? z : acc[f], 0, 0, 0);
                // (from gap 1 on: a read placed ahead of the tile's first MFMA would be caught by the wait on Afc there)
                if constexpr (i >= 1 && i <= KS2) {
                    asm volatile("" : "+v"(nxt));
                    Afn[i - 1 < KS ? KS + i - 1 : i - 1 - KS] = frag(nxt, i - 1);
                }
                epi_slice(ic, std::integral_constant<int, (u + U - 1) % GT>{}, accp, tile_lo + tl - 1 < tile_lo ? -1 : tile_lo + tl - 1);
                __builtin_amdgcn_sched_barrier(0);
            });
            // the previous tile closed an output group
            if constexpr (WIDE && u % GT == 0)
                if (tl > 0) flush_group((tile_lo + tl - GT) * SPT);
        };
#pragma unroll
        for (int n = 0; n < KS2; n++) AfA[n < KS ? KS + n : n - KS] = frag(cb, n);
        __builtin_amdgcn_sched_barrier(0);
        // U row tiles per step: the two accumulator sets (and fragment sets) swap roles without register copies, and the
        // position of a tile in its output group is a constant
        for (int tt = 0; tt < nt; tt += U)
            static_for16(std::make_integer_sequence<int, U>{}, [&](auto uc) __attribute__((always_inline)) {
                constexpr int u = decltype(uc)::value;
                if constexpr (u % 2 == 0)
                    do_tile(uc, tt + u, accA, accB, AfA, AfB);
                else
                    do_tile(uc, tt + u, accB, accA, AfB, AfA);
            });
    }
    // drain: the epilogue of the last tile (odd position: accB), which closes the last group
    static_for16(std::make_integer_sequence<int, NMF>{}, [&](auto ic) __attribute__((always_inline)) {
        epi_slice(ic, std::integral_constant<int, GT - 1>{}, accB, tile_hi - 1);
    });
    if (WIDE) flush_group((tile_hi - GT) * SPT);
}

template <int MP, int KS>
static int launch_f16(srx_ctx *ctx, srx_model *m, const f16x8 *Bp, int64_t T, int64_t nfb, float *out, int64_t st_t, int64_t st_g,
                      float *dump) {
    typedef F16Cfg<MP> Cfg;
    const size_t lds_bytes = (size_t)2 * Cfg::NT * 2 * KS * 1024 + (size_t)NW16 * Cfg::STAGE_FLOATS * sizeof(float);
    SRX_REQUIRE(ctx, lds_bytes <= 160 * 1024, "srx_gmm_score: LDS budget exceeded (%zu bytes)", lds_bytes);
    static bool attr_set[SRX_MAX_DEVICES] = {};   // the attribute is per device
    if (!attr_set[ctx->device]) {
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_gmm_f16<MP, KS, true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        SRX_HIP(ctx, hipFuncSetAttribute((const void *)k_gmm_f16<MP, KS, false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        attr_set[ctx->device] = true;
    }
    const int spt = 32 / MP;
    const int ntiles = (m->G + spt - 1) / spt;
    constexpr int TQ = Cfg::U;   // a slice holds whole unrolled steps (output groups, accumulator pairs)
    const int tps = ((ntiles + NSLICE - 1) / NSLICE + TQ - 1) / TQ * TQ;
    SRX_REQUIRE(ctx, nfb * NSLICE < (1ll << 31), "srx_gmm_score: too many frames for one launch");
    {
        SrxProf prof_(ctx, "k_gmm_f16");
        if (st_t == 1)
            hipLaunchKernelGGL((k_gmm_f16<MP, KS, true>), dim3((unsigned)(nfb * NSLICE)), dim3(64 * NW16), lds_bytes, ctx->stream, Bp, T,
                               (const f16x8 *)m->packed16, ntiles, tps, m->G, out, st_t, st_g, dump);
        else
            hipLaunchKernelGGL((k_gmm_f16<MP, KS, false>), dim3((unsigned)(nfb * NSLICE)), dim3(64 * NW16), lds_bytes, ctx->stream, Bp, T,
                               (const f16x8 *)m->packed16, ntiles, tps, m->G, out, st_t, st_g, dump);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

template <int KS>
static int dispatch_mp16(srx_ctx *ctx, srx_model *m, int MP, const f16x8 *Bp, int64_t T, int64_t nfb, float *out, int64_t st_t,
                         int64_t st_g, float *dump) {
    switch (MP) {
        case 1: return launch_f16<1, KS>(ctx, m, Bp, T, nfb, out, st_t, st_g, dump);
        case 2: return launch_f16<2, KS>(ctx, m, Bp, T, nfb, out, st_t, st_g, dump);
        case 4: return launch_f16<4, KS>(ctx, m, Bp, T, nfb, out, st_t, st_g, dump);
        case 8: return launch_f16<8, KS>(ctx, m, Bp, T, nfb, out, st_t, st_g, dump);
        case 16: return launch_f16<16, KS>(ctx, m, Bp, T, nfb, out, st_t, st_g, dump);
    }
    return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_gmm_score: unsupported component count");
}

int srx_gmm_score_mfma(srx_ctx *ctx, srx_model *m, int mix_mode, const float *feats, int ld, int64_t T, float *out, int64_t st_t,
                       int64_t st_g);

// Called by srx_gmm_score_dev (srx_gmm.hip) after argument validation.
int srx_gmm_score_f16x3(srx_ctx *ctx, srx_model *m, int mix_mode, const float *feats, int ld, int64_t T, float *out, int64_t st_t,
                        int64_t st_g) {
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    if (2 * m->D + 1 > 80 || Muse > 16)  // K beyond five k-steps / more components than a lane's accumulators
        return srx_gmm_score_mfma(ctx, m, mix_mode, feats, ld, T, out, st_t, st_g);
    const int KS = 2 * m->D + 1 <= 32 ? 2 : 5;
    int rc = ensure_packed16(ctx, m, mix_mode, KS);
    if (rc != SRX_OK) return rc;
    if (!m->f16_ok) return srx_gmm_score_mfma(ctx, m, mix_mode, feats, ld, T, out, st_t, st_g);
    const int MP = padded_components16(Muse);
    const int64_t nfb = (T + FRAMES_PER_WG - 1) / FRAMES_PER_WG;
    const int64_t nft = nfb * NW16 * FT_PER_WAVE;
    const size_t b_bytes = (size_t)nft * 2 * KS * 1024;
    void *ws = nullptr;
    const size_t flag_bytes = ((size_t)T + 255) / 256 * 256;
    rc = srx_workspace(ctx, 0, b_bytes + flag_bytes + 256, &ws);
    if (rc != SRX_OK) return rc;
    f16x8 *Bp = reinterpret_cast<f16x8 *>(ws);
    unsigned char *fflag = reinterpret_cast<unsigned char *>(ws) + b_bytes;
    float *dump = reinterpret_cast<float *>(fflag + flag_bytes);   // 64 floats for the stores of lanes with nothing to write
    SRX_HIP(ctx, hipMemsetAsync(fflag, 0, (size_t)T, ctx->stream));
    const int64_t n = nft * 2 * KS * 64;
    SRX_REQUIRE(ctx, ceil_div64(n, 256) < (1ll << 31), "srx_gmm_score: too many frames");
    {
        SrxProf prof_(ctx, "k_prep16");
        hipLaunchKernelGGL(k_prep16, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, feats, ld, T, nft, m->D, KS, m->colstats,
                           Bp, fflag);
    }
    SRX_LAUNCH_CHECK(ctx);
    rc = KS == 2 ? dispatch_mp16<2>(ctx, m, MP, Bp, T, nfb, out, st_t, st_g, dump)
                 : dispatch_mp16<5>(ctx, m, MP, Bp, T, nfb, out, st_t, st_g, dump);
    if (rc != SRX_OK) return rc;
    {
        SrxProf prof_(ctx, "k_gmm_fixup16");
        hipLaunchKernelGGL(k_gmm_fixup16, dim3((unsigned)ceil_div64(T, 256)), dim3(256), 0, ctx->stream, fflag, feats, ld, T, m->G, m->M,
                           m->D, m->mean, m->var, m->logwt, mix_mode, out, st_t, st_g);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
