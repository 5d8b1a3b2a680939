// Alignment-mode emission scoring: each utterance's frames are scored against that utterance's own HMM states only
// (what the reference does lazily inside hmm::viterbi, src/hmm.cpp:166: (*acm)(*fvs[i], target->ctx) for the <= 3 targets of
// every active state) instead of against all G tied states.  SURVEY.md 8(d): 2(2D+1)*S*M flop per frame, S ~ 50, against
// 2(2D+1)*G*M for the dense matrix -- 40x less work and a compact emission buffer (sum_u S_u*T_u floats) at G = 2000.
//
// Same arithmetic as srx_gmm_f16.hip (split-f16 operands, identical coefficient routine and k order), so the scores are
// bit-identical to the dense kernel's.  What differs is the data movement: a wave owns 64 frames of one utterance and
// walks that utterance's row tiles; the A' fragments of a tile are gathered per lane from a per-state table
// [G][k-step][component][k-half] straight into registers one tile ahead (no LDS, no barriers: waves are independent,
// the table of an utterance's ~50 states is a few hundred KB of L2 hits), and the wave builds its own B' fragments from
// the feature rows (nobody else needs them).  Output: per utterance a frame-major block [T_u][S_u] at emis_off[u] (one or two cache lines per frame for the
// Viterbi wave that reads it), staged per wave through LDS in slabs of 64 frames x 16 states.
#include "srx_gmm_f16.h"

struct srx_align_plan {
    srx_ctx *ctx = nullptr;
    int n_utt = 0;
    int64_t n_wb = 0;             // 64-frame wave blocks
    int64_t total_emis = 0;       // floats of the compact emission buffer
    int32_t *wb_utt = nullptr;    // [n_wb] utterance of a wave block
    int32_t *wb_f0 = nullptr;     // [n_wb] its first frame within the utterance
    int64_t *emis_off = nullptr;  // [n_utt+1] device
    std::vector<int64_t> emis_off_host;
};

extern "C" int srx_align_plan_create(srx_ctx *ctx, const int64_t *frame_off, const int32_t *state_off, int n_utt, srx_align_plan **out) {
    SRX_ENTER(ctx, "srx_align_plan_create");
    SRX_REQUIRE(ctx, frame_off && state_off && out && n_utt >= 0, "srx_align_plan_create: bad argument");
    srx_align_plan *p = new srx_align_plan();
    p->ctx = ctx;
    p->n_utt = n_utt;
    p->emis_off_host.assign((size_t)n_utt + 1, 0);
    std::vector<int32_t> wu, wf;
    for (int u = 0; u < n_utt; u++) {
        const int64_t T = frame_off[u + 1] - frame_off[u];
        const int64_t S = state_off[u + 1] - state_off[u];
        if (T < 0 || S < 1 || T >= (1ll << 31) || S * T >= (1ll << 31)) {   // k_viterbi indexes a block with 32 bits
            delete p;
            return srx_fail(ctx, SRX_ERR_INVALID, "srx_align_plan_create: utterance %d has %lld frames, %lld states", u, (long long)T,
                            (long long)S);
        }
        p->emis_off_host[u + 1] = p->emis_off_host[u] + S * T;
        for (int64_t f = 0; f < T; f += 64) {
            wu.push_back(u);
            wf.push_back((int32_t)f);
        }
    }
    p->n_wb = (int64_t)wu.size();
    p->total_emis = p->emis_off_host[n_utt];
    hipError_t e = hipSuccess;
    if ((e = hipMalloc((void **)&p->wb_utt, (wu.size() + 1) * 4)) != hipSuccess || (e = hipMalloc((void **)&p->wb_f0, (wf.size() + 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&p->emis_off, ((size_t)n_utt + 1) * 8)) != hipSuccess ||
        (e = hipMemcpy(p->wb_utt, wu.data(), wu.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->wb_f0, wf.data(), wf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->emis_off, p->emis_off_host.data(), ((size_t)n_utt + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipFree(p->wb_utt); (void)hipFree(p->wb_f0); (void)hipFree(p->emis_off);
        delete p;
        return srx_fail(ctx, SRX_ERR_DEVICE, "srx_align_plan_create: %s", hipGetErrorString(e));
    }
    *out = p;
    return SRX_OK;
}

extern "C" void srx_align_plan_destroy(srx_align_plan *p) {
    if (!p) return;
    (void)hipFree(p->wb_utt);
    (void)hipFree(p->wb_f0);
    (void)hipFree(p->emis_off);
    delete p;
}

extern "C" int64_t srx_align_plan_emission_count(const srx_align_plan *p) { return p ? p->total_emis : SRX_ERR_INVALID; }

extern "C" int srx_align_plan_emission_offsets(const srx_align_plan *p, int64_t *off) {
    if (!p || !off) return SRX_ERR_INVALID;
    std::memcpy(off, p->emis_off_host.data(), p->emis_off_host.size() * 8);
    return SRX_OK;
}

int srx_align_plan_n_utt(const srx_align_plan *p) { return p ? p->n_utt : -1; }

extern "C" const int64_t *srx_align_plan_emission_offsets_dev(const srx_align_plan *p) { return p ? p->emis_off : nullptr; }

// ------------------------------------------------------------------------------------------
// per-state table: [g][k-step 0..2KS)[component 0..MP)[k-half][8 halves]; hi k-steps then lo k-steps (as the dense table)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_pack16_state(const double *__restrict__ mean, const double *__restrict__ var,
                                                      const float *__restrict__ logwt, const float *__restrict__ cs, int G, int M, int D,
                                                      int KS, int MP, int Muse, f16x8 *__restrict__ packed) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (int64_t)G * 2 * KS * MP * 2) return;
    const int kh = (int)(idx & 1);
    const int c = (int)((idx >> 1) % MP);
    const int s = (int)(((idx >> 1) / MP) % (2 * KS));
    const int g = (int)(((idx >> 1) / MP) / (2 * KS));
    const double scale = MP > 1 ? 1.4426950408889634 : 1.0;
    const bool lo = s >= KS;
    const int k0 = (lo ? s - KS : s) * 16 + kh * 8;
    f16x8 out;
#pragma unroll
    for (int e = 0; e < 8; e++) {
        const float v = coef16(mean, var, logwt, cs, G, M, D, Muse, g, c, k0 + e, scale);
        const _Float16 hi = (_Float16)v;
        out[e] = lo ? (_Float16)(v - (float)hi) : hi;
    }
    packed[idx] = out;
}

static int ensure_packed16s(srx_ctx *ctx, srx_model *m, int mix_mode, int KS) {
    int rc = ensure_packed16(ctx, m, mix_mode, KS);   // column statistics and the "fits f16" verdict
    if (rc != SRX_OK) return rc;
    if (m->packed16s_version == m->version && m->packed16s_mode == mix_mode && m->packed16s) return SRX_OK;
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    const int MP = padded_components16(Muse);
    const int64_t n = (int64_t)m->G * 2 * KS * MP * 2;
    const size_t bytes = (size_t)n * 16;
    if (bytes > m->packed16s_bytes) {
        (void)hipFree(m->packed16s);
        m->packed16s = nullptr;
        m->packed16s_bytes = 0;
        SRX_HIP(ctx, hipMalloc(&m->packed16s, bytes));
        m->packed16s_bytes = bytes;
    }
    {
        SrxProf prof_(ctx, "k_pack16_state");
        hipLaunchKernelGGL(k_pack16_state, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, m->mean, m->var, m->logwt,
                           m->colstats, m->G, m->M, m->D, KS, MP, Muse, (f16x8 *)m->packed16s);
    }
    SRX_LAUNCH_CHECK(ctx);
    m->packed16s_version = m->version;
    m->packed16s_mode = mix_mode;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------
// exact evaluation of (utterance state, frame) pairs: all frames (flags == nullptr) or the flagged ones
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gmm_exact_align(const unsigned char *__restrict__ fflag, const float *__restrict__ feats, int ld,
                                                         const int32_t *__restrict__ wb_utt, const int32_t *__restrict__ wb_f0, int64_t n_wb,
                                                         const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                         const int32_t *__restrict__ tied, const int64_t *__restrict__ eoff, int M, int D,
                                                         const double *__restrict__ mean, const double *__restrict__ var,
                                                         const float *__restrict__ logwt, int mode, float *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t wb = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (wb >= n_wb) return;
    const int u = wb_utt[wb], lf0 = wb_f0[wb];
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u], S = soff[u + 1] - s0;
    const int nf = T - lf0 < 64 ? T - lf0 : 64;
    unsigned long long mask = nf == 64 ? ~0ull : ((1ull << nf) - 1);
    if (fflag) mask = __ballot(lane < nf && fflag[f0 + lf0 + lane] != 0);
    if (mask == 0) return;
    float *ob = out + eoff[u];
    for (int b = 0; b < nf; b++) {
        if (!((mask >> b) & 1ull)) continue;
        const int lf = lf0 + b;
        const float *fv = feats + (f0 + lf) * ld;
        for (int s = lane; s < S; s += 64) {
            const int g = tied[s0 + s];
            float r;
            if (mode == SRX_MIX_COMPONENT0) {
                const size_t o = (size_t)g * M;
                r = srx_logprob_exact(fv, mean + o * D, var + o * D, logwt[o], D);
            } else {
                const size_t o = (size_t)g * M;
                r = srx_lse_exact(fv, mean + o * D, var + o * D, logwt + o, M, D);
            }
            ob[(int64_t)lf * S + s] = r;
        }
    }
}

// ------------------------------------------------------------------------------------------
// the contraction
// ------------------------------------------------------------------------------------------
static constexpr int NWA = 4;   // waves per workgroup (independent of each other)

template <int MP, int KS>
__global__ __launch_bounds__(64 * NWA, 2) void k_gmm_f16_align(const float *__restrict__ feats, int ld, int D, const float *__restrict__ cs,
                                                               const f16x8 *__restrict__ Aps, const int32_t *__restrict__ wb_utt,
                                                               const int32_t *__restrict__ wb_f0, int64_t n_wb,
                                                               const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                               const int32_t *__restrict__ tied, const int64_t *__restrict__ eoff,
                                                               float *__restrict__ out, unsigned char *__restrict__ fflag,
                                                               float *__restrict__ dump) {
    typedef F16Cfg<MP> Cfg;
    constexpr int KS2 = 2 * KS;
    constexpr int NST = Cfg::NST, SPT = Cfg::SPT;
    constexpr int SB = SPT > 16 ? SPT : 16;     // states per output slab
    constexpr int TPS = SB / SPT;               // row tiles per slab
    constexpr int LDR = SB + 1;                 // staging row stride (odd: the transposing writes spread over the banks)
    __shared__ float stage_all[NWA][64 * LDR];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float *const stage = stage_all[w];          // [64 frames][SB states] of this wave
    const int64_t wb = (int64_t)blockIdx.x * NWA + w;
    if (wb >= n_wb) return;
    const int u = wb_utt[wb], lf0 = wb_f0[wb];
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u], S = soff[u + 1] - s0;
    float *const ob = out + eoff[u];
    const int j = lane & 31, hh = lane >> 5;

    // ---- this wave's frames: B' fragments built in place (k_prep16's routine) ----
    f16x8 Bf[FT_PER_WAVE][KS2];
#pragma unroll
    for (int f = 0; f < FT_PER_WAVE; f++) {
        const int lf = lf0 + f * 32 + j;
        const bool valid = lf < T;
        const int64_t t = f0 + (valid ? lf : 0);
        bool bad = false;
#pragma unroll
        for (int s = 0; s < KS; s++) prep16_frag2(feats, ld, t, valid, D, s * 16 + hh * 8, cs, bad, Bf[f][s], Bf[f][KS + s]);
        if (bad) fflag[t] = 1;
    }

    // ---- A' gather: fragment row i = lane % 32 is component r % MP of tile-local state hr*NST + r/MP (k_pack16's map) ----
    const int i_ = lane & 31, kh = lane >> 5;
    const int hr = (i_ >> 2) & 1, r_ = (i_ & 3) + 4 * (i_ >> 3);
    const int sl_lane = hr * NST + r_ / MP;
    const f16x8 *const Alane = Aps + ((r_ % MP) * 2 + kh);
    constexpr int64_t STATE_STRIDE = (int64_t)KS2 * MP * 2;   // f16x8 units per state
    const int ntile = (S + SPT - 1) / SPT;
    const int ntile_r = (ntile + 1) & ~1;                      // tiles in pairs; a padding tile repeats the last state, not stored
    auto state_of = [&](int n) __attribute__((always_inline)) {
        int sl = n * SPT + sl_lane;
        sl = sl < S ? sl : S - 1;
        return tied[s0 + sl];
    };

    constexpr int NMF = 3 * KS * FT_PER_WAVE;
    f32x16 accA[FT_PER_WAVE], accB[FT_PER_WAVE];
    const f32x16 z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
    for (int f = 0; f < FT_PER_WAVE; f++) accA[f] = accB[f] = z;
    LseEpilogue16<MP, KS> epi;
    // slice i of the log-sum-exp epilogue of row tile `tp` (< 0: none), see LseEpilogue16
    auto epi_slice = [&](auto ic, const f32x16 (&accp)[FT_PER_WAVE], int tp) __attribute__((always_inline)) {
        constexpr int i = decltype(ic)::value;
        epi.template slice<i>(accp, [&](int f, int st, float v) __attribute__((always_inline)) {
            // staging row = frame, column = state within the slab (a tile before the first writes a column that the
            // first real tile overwrites)
            const int tq = tp < 0 ? 0 : tp % TPS;
            stage[(f * 32 + j) * LDR + tq * SPT + hh * NST + st] = v;
        });
    };
    // writes slab `sb` (states sb*SB ..): lane f owns frame row f of the [64][SB] block (row stride LDR is odd, so the
    // 64 lanes read 64 different banks) and stores its SB states as 128-bit pieces -- 4-byte alignment is enough for
    // global_store_dwordx4 -- into the frame-major output; the slab that holds the row's end goes element by element
    auto flush_slab = [&](int sb) __attribute__((always_inline)) {
        typedef float f4u_ __attribute__((ext_vector_type(4), aligned(4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[SB];
#pragma unroll
        for (int c = 0; c < SB; c++) v[c] = stage[lane * LDR + c];
        const int lf = lf0 + lane, c0 = sb * SB;
        if (lf < T) {
            float *dst = ob + (int64_t)lf * S + c0;
            if (c0 + SB <= S) {
#pragma unroll
                for (int c = 0; c < SB; c += 4) *reinterpret_cast<f4u_ *>(dst + c) = (f4u_){v[c], v[c + 1], v[c + 2], v[c + 3]};
            } else {
#pragma unroll
                for (int c = 0; c < SB; c++)
                    if (c0 + c < S) dst[c] = v[c];
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // One row tile n: MFMAs on fragments Afc into `acc`; the epilogue of tile n-1 reads `accp`; the fragments of tile n+1
    // (state g1) are fetched into Afn, one global load per MFMA gap; the state of tile n+2 is looked up for the tile after.
    int g1 = state_of(1), g2 = 0;
    auto do_tile = [&](int n, f32x16 (&acc)[FT_PER_WAVE], const f32x16 (&accp)[FT_PER_WAVE], const f16x8 (&Afc)[KS2],
                       f16x8 (&Afn)[KS2]) __attribute__((always_inline)) {
        const f16x8 *pn = Alane + (int64_t)g1 * STATE_STRIDE;
        static_for16(std::make_integer_sequence<int, NMF>{}, [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int p = i / (KS * FT_PER_WAVE), s_ = (i / FT_PER_WAVE) % KS, f = i % FT_PER_WAVE;
            const f16x8 a_ = p == 0 ? Afc[KS + s_] : Afc[s_];
            const f16x8 b_ = p == 1 ? Bf[f][KS + s_] : Bf[f][s_];
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, i < FT_PER_WAVE ? z : acc[f], 0, 0, 0);
            if constexpr (i >= 1 && i <= KS2) {
                asm volatile("" : "+v"(pn));
                constexpr int s2 = i - 1 < KS ? KS + i - 1 : i - 1 - KS;   // lo fragments first, as they are used first
                // the pin hides where pn points; without the address space the loads become FLAT ones, which count
                // on both memory counters and made every tile wait for its own prefetch
                Afn[s2] = ((const __attribute__((address_space(1))) f16x8 *)pn)[s2 * MP * 2];
            }
            if constexpr (i == KS2 + 1) g2 = state_of(n + 2);
            epi_slice(ic, accp, n - 1);
            __builtin_amdgcn_sched_barrier(0);
        });
        g1 = g2;
        if (n >= 1 && (n - 1) % TPS == TPS - 1) flush_slab((n - 1) / TPS);   // tile n-1 closed a slab
    };
    f16x8 AfA[KS2], AfB[KS2];
    {
        const f16x8 *p0 = Alane + (int64_t)state_of(0) * STATE_STRIDE;
#pragma unroll
        for (int s2 = 0; s2 < KS2; s2++) AfA[s2] = p0[s2 * MP * 2];
    }
    for (int n = 0; n < ntile_r; n += 2) {
        do_tile(n, accA, accB, AfA, AfB);
        do_tile(n + 1, accB, accA, AfB, AfA);
    }
    static_for16(std::make_integer_sequence<int, NMF>{}, [&](auto ic) __attribute__((always_inline)) { epi_slice(ic, accB, ntile_r - 1); });
    flush_slab((ntile_r - 1) / TPS);   // the slab the last tile belongs to
}

template <int MP, int KS>
static int launch_align(srx_ctx *ctx, srx_model *m, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *feats, int ld,
                        float *out, unsigned char *fflag, float *dump) {
    SrxProf prof_(ctx, "k_gmm_f16_align");
    hipLaunchKernelGGL((k_gmm_f16_align<MP, KS>), dim3((unsigned)ceil_div64(plan->n_wb, NWA)), dim3(64 * NWA), 0, ctx->stream, feats, ld, m->D,
                       m->colstats, (const f16x8 *)m->packed16s, plan->wb_utt, plan->wb_f0, plan->n_wb, hmm->frame_off_dev,
                       hmm->state_off_dev, hmm->tied_dev, plan->emis_off, out, fflag, dump);
    return SRX_OK;
}

template <int KS>
static int dispatch_align(srx_ctx *ctx, srx_model *m, int MP, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *feats, int ld,
                          float *out, unsigned char *fflag, float *dump) {
    switch (MP) {
        case 1: return launch_align<1, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump);
        case 2: return launch_align<2, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump);
        case 4: return launch_align<4, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump);
        case 8: return launch_align<8, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump);
        case 16: return launch_align<16, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump);
    }
    return srx_fail(ctx, SRX_ERR_UNSUPPORTED, "srx_gmm_score_align_dev: unsupported component count");
}

static int launch_exact_align(srx_ctx *ctx, srx_model *m, int mix_mode, const srx_hmm_batch *hmm, const srx_align_plan *plan,
                              const float *feats, int ld, const unsigned char *fflag, float *out) {
    SrxProf prof_(ctx, fflag ? "k_gmm_fixup16_align" : "k_gmm_exact_align");
    hipLaunchKernelGGL(k_gmm_exact_align, dim3((unsigned)ceil_div64(plan->n_wb, 4)), dim3(256), 0, ctx->stream, fflag, feats, ld, plan->wb_utt,
                       plan->wb_f0, plan->n_wb, hmm->frame_off_dev, hmm->state_off_dev, hmm->tied_dev, plan->emis_off, m->M, m->D, m->mean,
                       m->var, m->logwt, mix_mode, out);
    return SRX_OK;
}

extern "C" int srx_gmm_score_align_dev(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats_dev, int ld,
                                       const srx_hmm_batch *hmm, const srx_align_plan *plan, float *emis_dev) {
    SRX_ENTER(ctx, "srx_gmm_score_align_dev");
    SRX_REQUIRE(ctx, m && feats_dev && hmm && plan && emis_dev, "srx_gmm_score_align_dev: null argument");
    SRX_REQUIRE(ctx, mix_mode == SRX_MIX_COMPONENT0 || mix_mode == SRX_MIX_LOGSUMEXP, "srx_gmm_score_align_dev: unknown mix mode %d", mix_mode);
    SRX_REQUIRE(ctx, score_path == SRX_SCORE_MFMA_F16X3 || score_path == SRX_SCORE_EXACT,
                "srx_gmm_score_align_dev: score path must be SRX_SCORE_MFMA_F16X3 or SRX_SCORE_EXACT");
    SRX_REQUIRE(ctx, ld >= m->D, "srx_gmm_score_align_dev: ld %d < D %d", ld, m->D);
    SRX_REQUIRE(ctx, plan->n_utt == hmm->n_utt, "srx_gmm_score_align_dev: plan is for %d utterances, batch has %d", plan->n_utt, hmm->n_utt);
    SRX_REQUIRE(ctx, ceil_div64(plan->n_wb, NWA) < (1ll << 31), "srx_gmm_score_align_dev: too many frames for one launch");
    if (plan->n_wb == 0) return SRX_OK;
    const int Muse = mix_mode == SRX_MIX_COMPONENT0 ? 1 : m->M;
    bool f16 = score_path == SRX_SCORE_MFMA_F16X3 && 2 * m->D + 1 <= 80 && Muse <= 16;
    const int KS = 2 * m->D + 1 <= 32 ? 2 : 5;
    if (f16) {
        int rc = ensure_packed16s(ctx, m, mix_mode, KS);
        if (rc != SRX_OK) return rc;
        f16 = m->f16_ok;
    }
    if (!f16) {
        launch_exact_align(ctx, m, mix_mode, hmm, plan, feats_dev, ld, nullptr, emis_dev);
        SRX_LAUNCH_CHECK(ctx);
        return SRX_OK;
    }
    void *ws = nullptr;
    const size_t flag_bytes = ((size_t)hmm->total_frames + 255) / 256 * 256;
    int rc = srx_workspace(ctx, 0, flag_bytes + 256, &ws);
    if (rc != SRX_OK) return rc;
    unsigned char *fflag = reinterpret_cast<unsigned char *>(ws);
    float *dump = reinterpret_cast<float *>(fflag + flag_bytes);
    SRX_HIP(ctx, hipMemsetAsync(fflag, 0, (size_t)hmm->total_frames, ctx->stream));
    const int MP = padded_components16(Muse);
    rc = KS == 2 ? dispatch_align<2>(ctx, m, MP, hmm, plan, feats_dev, ld, emis_dev, fflag, dump)
                 : dispatch_align<5>(ctx, m, MP, hmm, plan, feats_dev, ld, emis_dev, fflag, dump);
    if (rc != SRX_OK) return rc;
    SRX_LAUNCH_CHECK(ctx);
    launch_exact_align(ctx, m, mix_mode, hmm, plan, feats_dev, ld, fflag, emis_dev);
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
