// Alignment-mode emission scoring: each utterance's frames are scored against that utterance's own HMM states only
// (what the reference does lazily inside hmm::viterbi, src/hmm.cpp:166: (*acm)(*fvs[i], target->ctx) for the <= 3 targets of
// every active state) instead of against all G tied states.  SURVEY.md 8(d): 2(2D+1)*S*M flop per frame, S ~ 50, against
// 2(2D+1)*G*M for the dense matrix -- 40x less work and a compact emission buffer (sum_u S_u*T_u floats) at G = 2000.
//
// Same arithmetic as srx_gmm_f16.hip (split-f16 operands, identical coefficient routine and k order), so the scores are
// bit-identical to the dense kernel's.  What differs is the data movement.  A workgroup is NWA waves x 64 consecutive
// frames of ONE utterance, so all its waves want the same row tiles (the utterance's states): the A' fragments of a chunk
// of tiles are gathered ONCE per workgroup from the per-state table [G][k-step][component][k-half] into LDS by LDS-DMA
// with per-lane source addresses (a lane of the DMA fetches the 16 bytes that the same lane of the MFMA will read, so
// LDS holds the dense kernel's fragment order and reads are one conflict-free ds_read_b128 per fragment), double
// buffered, one barrier per chunk -- the structure of k_gmm_f16, where round 1's kernel had every wave gather every
// tile for itself from L2 (3.3 KB per frame, ~8 TB/s of L2 traffic, and ~100 extra vector instructions per tile that
// did not overlap the MFMAs).  Each wave builds its own B' fragments from the feature rows (nobody else needs them)
// and keeps them in registers.  Output: per utterance a frame-major block [T_u][S_u] at emis_off[u] (one or two cache
// lines per frame for the Viterbi wave that reads it), staged per wave through LDS in slabs of 64 frames x 16 states.
#include "srx_gmm_f16.h"

#ifndef SRX_ALIGN_NWA
#define SRX_ALIGN_NWA 4
#endif
static constexpr int WBF = 32 * FT_PER_WAVE;            // frames of a wave block (what one wave keeps in registers)
#ifndef SRX_ALIGN_WPE
#define SRX_ALIGN_WPE 2                                 // waves per SIMD k_gmm_f16_align is compiled for
#endif
static constexpr int SRX_ALIGN_WAVES = SRX_ALIGN_NWA;   // NWA: waves per workgroup of k_gmm_f16_align (64 frames each, one utterance)

struct srx_align_plan {
    srx_ctx *ctx = nullptr;
    int n_utt = 0;
    int64_t n_wb = 0;             // 64-frame wave blocks
    int64_t total_emis = 0;       // floats of the compact emission buffer
    int32_t *wb_utt = nullptr;    // [n_wb] utterance of a wave block
    int32_t *wb_f0 = nullptr;     // [n_wb] its first frame within the utterance
    int64_t n_wg = 0;             // workgroup items of the matrix-core kernel: NWA * 64 consecutive frames of one utterance
    int32_t *wg_utt = nullptr;    // [n_wg]
    int32_t *wg_f0 = nullptr;     // [n_wg]
    int32_t *wg_wb0 = nullptr;    // [n_wg] index of the item's first wave block (wave w of the item owns wave block wg_wb0 + w)
    // Frame-operand cache: the split-f16 B' fragments of every wave block, [n_wb][FT_PER_WAVE][2*KS][lane] f16x8 (320 bytes
    // per frame at KS = 5), and the "re-evaluate exactly" flag of every frame.  They depend on the features and on the
    // centre / scale vector of the model only (srx_model::colstats, which ensure_packed16 keeps from call to call while it
    // suits the model), so the iterations of hmm::train -- same frames, a model that moves a little -- build them once.
    // Building them inside the scoring kernel cost 1,700 vector instructions per 64-frame wave block, a quarter of the
    // kernel's vector issue, in a phase where the wave runs no MFMA.
    bool cache_frames = false;   // srx_align_plan_cache_frames: the caller promises that the feature buffer does not change
    void *bp = nullptr;
    size_t bp_bytes = 0;
    unsigned char *bp_flag = nullptr;   // [total frames]
    size_t bp_flag_bytes = 0;
    const float *bp_feats = nullptr;    // what the cache was built from
    int bp_ld = 0, bp_KS = 0;
    const srx_model *bp_model = nullptr;
    uint64_t bp_epoch = 0;
    int64_t bp_frames = -1;
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
    std::vector<int32_t> wu, wf, gu, gf, gw;
    for (int u = 0; u < n_utt; u++) {
        const int64_t T = frame_off[u + 1] - frame_off[u];
        const int64_t S = state_off[u + 1] - state_off[u];
        if (T < 0 || S < 1 || T >= (1ll << 31) || S * T >= (1ll << 31)) {   // k_viterbi indexes a block with 32 bits
            delete p;
            return srx_fail(ctx, SRX_ERR_INVALID, "srx_align_plan_create: utterance %d has %lld frames, %lld states", u, (long long)T,
                            (long long)S);
        }
        p->emis_off_host[u + 1] = p->emis_off_host[u] + S * T;
        for (int64_t f = 0; f < T; f += WBF) {
            wu.push_back(u);
            wf.push_back((int32_t)f);
        }
        for (int64_t f = 0; f < T; f += WBF * SRX_ALIGN_WAVES) {
            gu.push_back(u);
            gf.push_back((int32_t)f);
            gw.push_back((int32_t)(wu.size() - (size_t)((T + WBF - 1) / WBF) + (size_t)(f / WBF)));
        }
    }
    p->n_wb = (int64_t)wu.size();
    p->n_wg = (int64_t)gu.size();
    p->total_emis = p->emis_off_host[n_utt];
    hipError_t e = hipSuccess;
    if ((e = hipMalloc((void **)&p->wb_utt, (wu.size() + 1) * 4)) != hipSuccess || (e = hipMalloc((void **)&p->wb_f0, (wf.size() + 1) * 4)) != hipSuccess ||
        (e = hipMalloc((void **)&p->wg_utt, (gu.size() + 1) * 4)) != hipSuccess || (e = hipMalloc((void **)&p->wg_f0, (gf.size() + 1) * 4)) != hipSuccess ||
        (e = hipMemcpy(p->wg_utt, gu.data(), gu.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->wg_f0, gf.data(), gf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc((void **)&p->wg_wb0, (gw.size() + 1) * 4)) != hipSuccess ||
        (e = hipMemcpy(p->wg_wb0, gw.data(), gw.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMalloc((void **)&p->emis_off, ((size_t)n_utt + 1) * 8)) != hipSuccess ||
        (e = hipMemcpy(p->wb_utt, wu.data(), wu.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->wb_f0, wf.data(), wf.size() * 4, hipMemcpyHostToDevice)) != hipSuccess ||
        (e = hipMemcpy(p->emis_off, p->emis_off_host.data(), ((size_t)n_utt + 1) * 8, hipMemcpyHostToDevice)) != hipSuccess) {
        (void)hipFree(p->wb_utt); (void)hipFree(p->wb_f0); (void)hipFree(p->emis_off); (void)hipFree(p->wg_utt); (void)hipFree(p->wg_f0); (void)hipFree(p->wg_wb0);
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
    (void)hipFree(p->wg_utt);
    (void)hipFree(p->wg_f0);
    (void)hipFree(p->wg_wb0);
    (void)hipFree(p->bp);
    (void)hipFree(p->bp_flag);
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

extern "C" int srx_align_plan_cache_frames(srx_align_plan *p, int on) {
    if (!p) return SRX_ERR_INVALID;
    p->cache_frames = on != 0;
    if (!on) {
        (void)hipSetDevice(p->ctx->device);
        (void)hipStreamSynchronize(p->ctx->stream);
        (void)hipFree(p->bp);
        (void)hipFree(p->bp_flag);
        p->bp = nullptr;
        p->bp_flag = nullptr;
        p->bp_bytes = p->bp_flag_bytes = 0;
        p->bp_frames = -1;
    }
    return SRX_OK;
}

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
    // the table holds coefficients scaled with the centre / scale vector in force when it was built: a replaced vector (a dense
    // score in the other mix mode can adopt a new one without any change of the model's version) makes it stale
    if (m->packed16s_version == m->version && m->packed16s_mode == mix_mode && m->packed16s_epoch == m->colstats_epoch && m->packed16s)
        return SRX_OK;
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
    m->packed16s_epoch = m->colstats_epoch;
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
    const int nf = T - lf0 < WBF ? T - lf0 : WBF;
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
#ifndef SRX_ALIGN_SLAB
#define SRX_ALIGN_SLAB 8    // states per output slab: with 8 the workgroup's LDS is 50 KiB and a third workgroup fits a CU once
#endif                      // the waves without frames of the resident ones have left (16: 59 KiB, two workgroups)
static constexpr int NWA = SRX_ALIGN_WAVES;   // waves per workgroup
#ifdef SRX_ALIGN_EXP_STAMP
// diagnostic build: cycles (s_memtime) of wave 0 of every workgroup, summed per phase; printed by the launcher
static __device__ unsigned long long g_stamp[8];
#define SRX_STAMP(k) do { if (tid == 0) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); atomicAdd(&g_stamp[k], now_ - t_prev_); t_prev_ = now_; } } while (0)
#else
#define SRX_STAMP(k) do { } while (0)
#endif
static constexpr int NTA = 2;                 // row tiles per LDS chunk (two accumulator sets swap roles from tile to tile)
#ifndef SRX_ALIGN_RING
#define SRX_ALIGN_RING 1    // 1: three single-tile LDS buffers, DMA two tiles ahead, one barrier per tile; 0: two 2-tile chunk buffers
#endif
static constexpr bool RINGA = SRX_ALIGN_RING != 0;
static constexpr int NBUFA = 3;

__device__ __forceinline__ void align_wait_vmcnt_at_most(int n) {   // s_waitcnt takes an immediate; n is wave-uniform
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

// The frame operands of every wave block (the plan's cache): thread = (wave block, frame tile f, k-step s, lane) builds
// the high and the low fragment with k_prep16's routine; frames beyond f16's range or NaN get their flag.
__global__ __launch_bounds__(256) void k_prep16_align(const float *__restrict__ feats, int ld, int D, int KS, const float *__restrict__ cs,
                                                      const int32_t *__restrict__ wb_utt, const int32_t *__restrict__ wb_f0, int64_t n_wb,
                                                      const int64_t *__restrict__ foff, f16x8 *__restrict__ Bp,
                                                      unsigned char *__restrict__ fflag) {
    const int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n_wb * FT_PER_WAVE * KS * 64) return;
    const int lane = (int)(idx & 63);
    const int s = (int)((idx >> 6) % KS);
    const int f = (int)(((idx >> 6) / KS) % FT_PER_WAVE);
    const int64_t wb = (idx >> 6) / KS / FT_PER_WAVE;
    const int u = wb_utt[wb];
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int lf = wb_f0[wb] + f * 32 + (lane & 31);
    const bool valid = lf < T;
    const int64_t t = f0 + (valid ? lf : 0);
    bool bad = false;
    f16x8 hi, lo;
    prep16_frag2(feats, ld, t, valid, D, s * 16 + (lane >> 5) * 8, cs, bad, hi, lo);
    f16x8 *dst = Bp + ((wb * FT_PER_WAVE + f) * 2 * KS) * 64 + lane;
    dst[(int64_t)s * 64] = hi;
    dst[(int64_t)(KS + s) * 64] = lo;
    if (bad) fflag[t] = 1;
}

// CACHED: the wave's B' fragments come from the plan's cache (Bp); otherwise the wave builds them itself.
template <int MP, int KS, bool CACHED>
__global__ __launch_bounds__(64 * NWA, SRX_ALIGN_WPE) void k_gmm_f16_align(const float *__restrict__ feats, int ld, int D, const float *__restrict__ cs,
                                                               const f16x8 *__restrict__ Aps, const f16x8 *__restrict__ Bp,
                                                               const int32_t *__restrict__ wg_utt, const int32_t *__restrict__ wg_wb0,
                                                               const int32_t *__restrict__ wg_f0, int64_t n_wg, int64_t n_wb,
                                                               const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                               const int32_t *__restrict__ tied, const int64_t *__restrict__ eoff,
                                                               float *__restrict__ out, unsigned char *__restrict__ fflag,
                                                               float *__restrict__ dump) {
    typedef F16Cfg<MP> Cfg;
    constexpr int KS2 = 2 * KS;
    constexpr int NST = Cfg::NST, SPT = Cfg::SPT;
    constexpr int SB = SPT > SRX_ALIGN_SLAB ? SPT : SRX_ALIGN_SLAB;     // states per output slab
    constexpr int TPS = SB / SPT;               // row tiles per slab
    constexpr int LDR = SB + 1;                 // staging row stride (odd: the transposing writes spread over the banks)
    constexpr int TILE_BYTES = KS2 * 1024;
    constexpr int CHUNK_BYTES = NTA * TILE_BYTES;
    constexpr int NPIECE = NTA * KS2;           // 1 KiB DMA pieces per chunk
    // LDS: [chunk buffer 0][chunk buffer 1][per-wave staging 64 x LDR floats][tied ids of the utterance's states]
    constexpr int STAGE_BYTES = WBF * LDR * 4;
    constexpr int STAGE_BASE = RINGA ? NBUFA * TILE_BYTES : 2 * CHUNK_BYTES;
    constexpr int TIED_BASE = STAGE_BASE + NWA * STAGE_BYTES;
    static_assert(RINGA || STAGE_BYTES * NWA + 256 * 4 >= TILE_BYTES, "a fragment fetch past the second chunk buffer must stay inside the allocation");
    __shared__ __attribute__((aligned(16))) unsigned char lds_all[TIED_BASE + 256 * 4];
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr;
    const lds_ptr lds3 = (lds_ptr)lds_all;
    (void)dump;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t item = blockIdx.x;
    if (item >= n_wg) return;
#ifdef SRX_ALIGN_EXP_STAMP
    unsigned long long t_prev_ = __builtin_amdgcn_s_memtime();
    if (tid == 0) atomicAdd(&g_stamp[7], 1ull);
#endif
    const int u = wg_utt[item], gf0 = wg_f0[item];
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u], S = soff[u + 1] - s0;
    if (S > 256) return;   // the launcher rejects such a batch (max_states); never index past the table
    int32_t *const s_tied = reinterpret_cast<int32_t *>(lds_all + TIED_BASE);
    // Start-up order: the tied id of "this thread's" state first (S <= 256 = one load per thread), then -- CACHED -- the
    // wave's 20 operand loads, so that the wait for the tied ids (LDS table, barrier, first DMA) leaves the operand loads
    // in flight instead of queueing behind them.
    const int tied_mine = tid < S ? tied[s0 + tid] : 0;
    f16x8 Bf[FT_PER_WAVE][KS2];
    if (CACHED) {
        int64_t wbi = (int64_t)wg_wb0[item] + w;
        wbi = wbi < n_wb ? wbi : n_wb - 1;      // a wave without frames (it returns below) still forms a valid address
        const f16x8 *bsrc = Bp + (wbi * FT_PER_WAVE * KS2) * 64 + lane;
#pragma unroll
        for (int f = 0; f < FT_PER_WAVE; f++)
#pragma unroll
            for (int s = 0; s < KS2; s++) Bf[f][s] = bsrc[(f * KS2 + s) * 64];
    }
    s_tied[tid] = tied_mine;
    for (int s_ = tid + 64 * NWA; s_ < S; s_ += 64 * NWA) s_tied[s_] = tied[s0 + s_];   // (workgroups narrower than 256 threads)
    __syncthreads();
    // waves of this workgroup that have frames (a finished wave no longer counts at the barriers below)
    SRX_STAMP(0);   // item lookup, tied ids into LDS, barrier
    const int rem = T - gf0;
    const int nwa = rem >= WBF * NWA ? NWA : (rem + WBF - 1) / WBF;
    if (w >= nwa) return;
    const int lf0 = gf0 + w * WBF;
    float *const stage = reinterpret_cast<float *>(lds_all + STAGE_BASE + w * STAGE_BYTES);   // [64 frames][SB states] of this wave
    float *const ob = out + eoff[u];
    const int j = lane & 31, hh = lane >> 5;

    // ---- A' gather by LDS-DMA: fragment row i = lane % 32 is component r % MP of tile-local state hr*NST + r/MP (k_pack16's
    // map); the lane's 16 bytes of k-step s2 of state g sit at Aps[(g*KS2 + s2)*MP*2 + (r % MP)*2 + kh] ----
    const int i_ = lane & 31, kh = lane >> 5;
    const int hr = (i_ >> 2) & 1, r_ = (i_ & 3) + 4 * (i_ >> 3);
    const int sl_lane = hr * NST + r_ / MP;
    const f16x8 *const Alane = Aps + ((r_ % MP) * 2 + kh);
    constexpr int64_t STATE_STRIDE = (int64_t)KS2 * MP * 2;   // f16x8 units per state
    const int ntile = (S + SPT - 1) / SPT;
    const int ntile_r = (ntile + NTA - 1) / NTA * NTA;         // whole chunks; a padding tile repeats the last state, not stored
    const int nchunks = ntile_r / NTA;
    auto dma = [&](int chunk, int buf) __attribute__((always_inline)) {
        // piece p = (tile tt of the chunk, k-step s2) -> 1 KiB at buf*CHUNK_BYTES + p*1024; issued by wave p % nwa
        for (int p = w; p < NPIECE; p += nwa) {
            const int tt = p / KS2, s2 = p - tt * KS2;
            int sl = (chunk * NTA + tt) * SPT + sl_lane;
            sl = sl < S ? sl : S - 1;
#ifdef SRX_ALIGN_EXP_HOT
            sl = sl_lane;             // diagnostic build: every tile gathers the first tile's states (L2-hot table rows)
#endif
            const int g = s_tied[sl];
            const f16x8 *src = Alane + (int64_t)g * STATE_STRIDE + s2 * MP * 2;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(lds_all + buf * CHUNK_BYTES + p * 1024), 16, 0, 0);
        }
    };
    // ring form: one tile per buffer
    auto dma_tile = [&](int n, int buf) __attribute__((always_inline)) {
        int sl = n * SPT + sl_lane;
        sl = sl < S ? sl : S - 1;
        const f16x8 *src = Alane + (int64_t)s_tied[sl] * STATE_STRIDE;
        for (int p = w; p < KS2; p += nwa)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + p * MP * 2),
                                             (__attribute__((address_space(3))) void *)(lds_all + buf * TILE_BYTES + p * 1024), 16, 0, 0);
    };
    if (RINGA) {
        dma_tile(0, 0);
        dma_tile(1, 1);      // ntile_r >= 2
    } else {
        dma(0, 0);
    }

    // ---- this wave's frames: B' fragments from the plan's cache (loads issued above), or built in place (k_prep16's
    // routine); kept in registers for the whole kernel ----
    if (!CACHED) {
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
    }

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
    // global_store_dwordx4 -- into the frame-major output; the slab that holds the row's end goes element by element.
    // pend_stores: store instructions issued since the last DMA issue (wave-uniform; < 0: unknown, drain at the next wait)
    int pend_stores = 0;
    auto flush_slab = [&](int sb) __attribute__((always_inline)) {
        typedef float f4u_ __attribute__((ext_vector_type(4), aligned(4)));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        float v[SB];
        const int row = lane < WBF ? lane : 0;       // (one frame tile per wave: the upper half of the lanes has no row)
#pragma unroll
        for (int c = 0; c < SB; c++) v[c] = stage[row * LDR + c];
        const int lf = lane < WBF ? lf0 + lane : T, c0 = sb * SB;
#ifdef SRX_ALIGN_EXP_NOSTORE
        if (lf0 >= 0) return;                    // diagnostic build: no output stores
#endif
        if (c0 + SB <= S) {                      // wave-uniform
            if (lf < T) {
                float *dst = ob + (int64_t)lf * S + c0;
#pragma unroll
                for (int c = 0; c < SB; c += 4) *reinterpret_cast<f4u_ *>(dst + c) = (f4u_){v[c], v[c + 1], v[c + 2], v[c + 3]};
            }
            if (pend_stores >= 0) pend_stores += SB / 4;
        } else {
            if (lf < T) {
                float *dst = ob + (int64_t)lf * S + c0;
#pragma unroll
                for (int c = 0; c < SB; c++)
                    if (c0 + c < S) dst[c] = v[c];
            }
            pend_stores = -1;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    };

    // fragment n (in order of use: lo 0..KS-1, then hi 0..KS-1; storage order is hi first) of the tile at LDS offset `off`
    auto frag = [&](unsigned off, int n) __attribute__((always_inline)) {
        const int s = n < KS ? KS + n : n - KS;
        return *reinterpret_cast<const __attribute__((address_space(3))) f16x8 *>(lds3 + off + s * 1024);
    };

    f16x8 AfA[KS2], AfB[KS2];
    // One row tile n (fragments Afc, LDS offset of the NEXT tile's first fragment in nxt0): its MFMAs write `acc` while the
    // epilogue of tile n-1 reads `accp` and the fragments of the next tile are fetched into Afn, one ds_read_b128 per MFMA gap.
    auto do_tile = [&](int n, unsigned nxt0, f32x16 (&acc)[FT_PER_WAVE], const f32x16 (&accp)[FT_PER_WAVE], const f16x8 (&Afc)[KS2],
                       f16x8 (&Afn)[KS2]) __attribute__((always_inline)) {
        unsigned nxt = nxt0;
        static_for16(std::make_integer_sequence<int, NMF>{}, [&](auto ic) __attribute__((always_inline)) {
            constexpr int i = decltype(ic)::value;
            constexpr int p = i / (KS * FT_PER_WAVE), s_ = (i / FT_PER_WAVE) % KS, f = i % FT_PER_WAVE;
            const f16x8 a_ = p == 0 ? Afc[KS + s_] : Afc[s_];
            const f16x8 b_ = p == 1 ? Bf[f][KS + s_] : Bf[f][s_];
            acc[f] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a_, b_, i < FT_PER_WAVE ? z : acc[f], 0, 0, 0);
            if constexpr (i >= 1 && i <= KS2) {
                asm volatile("" : "+v"(nxt));
                Afn[i - 1 < KS ? KS + i - 1 : i - 1 - KS] = frag(nxt, i - 1);
            }
#ifdef SRX_ALIGN_EXP_NOEPI
            if constexpr (i == NMF - 1) stage[lane * LDR] = accp[0][0] + accp[FT_PER_WAVE - 1][15];   // diagnostic build: MFMAs only
#else
            epi_slice(ic, accp, n - 1);
#endif
            __builtin_amdgcn_sched_barrier(0);
        });
        if (n >= 1 && (n - 1) % TPS == TPS - 1) flush_slab((n - 1) / TPS);   // tile n-1 closed a slab
    };
    if (RINGA) {
        // Tile n sits in buffer n % 3.  At the barrier in front of tile n the pieces of tiles n and n+1 have landed (a wave
        // waits for its own pieces of n+1, issued one tile ago: in flight, oldest first, are [pieces of n+1][stores since], so
        // waiting until at most that many operations are outstanding leaves the stores in flight), so the fragments of tile
        // n+1 can be fetched under tile n's MFMAs; the buffer of tile n-1 is free for the DMA of tile n+2.
        auto ring_step = [&](int n, f32x16 (&acc)[FT_PER_WAVE], const f32x16 (&accp)[FT_PER_WAVE], const f16x8 (&Afc)[KS2],
                             f16x8 (&Afn)[KS2]) __attribute__((always_inline)) {
            if (n == 0 || pend_stores < 0)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            else
                align_wait_vmcnt_at_most(pend_stores);
            __syncthreads();
            if (n == 0) SRX_STAMP(1);
            if (n + 2 < ntile_r) dma_tile(n + 2, (n + 2) % NBUFA);
            pend_stores = 0;
            do_tile(n, ((n + 1) % NBUFA) * TILE_BYTES + lane * 16, acc, accp, Afc, Afn);
        };
#ifdef SRX_ALIGN_EXP_TWICE
        // diagnostic build: the tile loop runs twice per workgroup (same tiles, same output) -- kernel time minus the normal
        // build's is the marginal cost of one pass of the loop
        for (int rep_ = 0; rep_ < 2; rep_++) {
        if (rep_ == 1) {
            __syncthreads();
            dma_tile(0, 0);
            dma_tile(1, 1);
            pend_stores = -1;
        }
#endif
        for (int n = 0; n < ntile_r; n += 2) {
            if (n == 0) {
                // (the first tile's own fragments: after its barrier; everything later arrives under the previous tile)
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
#pragma unroll
                for (int q = 0; q < KS2; q++) AfA[q < KS ? KS + q : q - KS] = frag(lane * 16, q);
                __builtin_amdgcn_sched_barrier(0);
            }
            ring_step(n, accA, accB, AfA, AfB);
            ring_step(n + 1, accB, accA, AfB, AfA);
        }
#ifdef SRX_ALIGN_EXP_TWICE
        }
#endif
    } else {
    for (int c = 0; c < nchunks; c++) {
        // This wave's DMA pieces of chunk c must have landed.  In flight, oldest first: [pieces of chunk c][stores issued
        // since].  vmcnt retires in order, so waiting until at most that many operations are outstanding waits for the
        // pieces and leaves the stores in flight (the first chunk also waits for the feature loads of the prologue).
        if (c == 0 || pend_stores < 0)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        else
            align_wait_vmcnt_at_most(pend_stores);
        __syncthreads();                                  // ... everybody's; and everybody is done with the other buffer
        if (c == 0) SRX_STAMP(1);                         // first DMA issue, operand loads, their wait, first barrier
        if (c + 1 < nchunks) dma(c + 1, (c + 1) & 1);
        pend_stores = 0;
        const unsigned cb = (c & 1) * CHUNK_BYTES + lane * 16;   // this lane's 16 bytes of the chunk's first fragment
        // After the last tile of a chunk the fragment fetch reads past the chunk -- still inside this workgroup's LDS -- and is
        // not used: the first tile of a chunk is read again after the barrier.
#pragma unroll
        for (int n = 0; n < KS2; n++) AfA[n < KS ? KS + n : n - KS] = frag(cb, n);
        __builtin_amdgcn_sched_barrier(0);
        static_assert(NTA == 2, "two tiles per chunk: accumulator and fragment sets alternate A, B");
        do_tile(c * NTA, cb + TILE_BYTES, accA, accB, AfA, AfB);
        do_tile(c * NTA + 1, cb + 2 * TILE_BYTES, accB, accA, AfB, AfA);
    }
    }
    SRX_STAMP(2);   // the tile loop
    static_for16(std::make_integer_sequence<int, NMF>{}, [&](auto ic) __attribute__((always_inline)) { epi_slice(ic, accB, ntile_r - 1); });
    flush_slab((ntile_r - 1) / TPS);   // the slab the last tile belongs to
#ifdef SRX_ALIGN_EXP_STAMP
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    SRX_STAMP(3);   // last epilogue, last flush, stores drained
    if (tid == 0) atomicAdd(&g_stamp[6], (unsigned long long)ntile_r);
#endif
}

template <int MP, int KS>
static int launch_align(srx_ctx *ctx, srx_model *m, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *feats, int ld,
                        float *out, unsigned char *fflag, float *dump, const f16x8 *Bp) {
    SrxProf prof_(ctx, "k_gmm_f16_align");
    if (Bp)
        hipLaunchKernelGGL((k_gmm_f16_align<MP, KS, true>), dim3((unsigned)plan->n_wg), dim3(64 * NWA), 0, ctx->stream, feats, ld, m->D,
                           m->colstats, (const f16x8 *)m->packed16s, Bp, plan->wg_utt, plan->wg_wb0, plan->wg_f0, plan->n_wg, plan->n_wb,
                           hmm->frame_off_dev, hmm->state_off_dev, hmm->tied_dev, plan->emis_off, out, fflag, dump);
    else
        hipLaunchKernelGGL((k_gmm_f16_align<MP, KS, false>), dim3((unsigned)plan->n_wg), dim3(64 * NWA), 0, ctx->stream, feats, ld, m->D,
                           m->colstats, (const f16x8 *)m->packed16s, Bp, plan->wg_utt, plan->wg_wb0, plan->wg_f0, plan->n_wg, plan->n_wb,
                           hmm->frame_off_dev, hmm->state_off_dev, hmm->tied_dev, plan->emis_off, out, fflag, dump);
#ifdef SRX_ALIGN_EXP_STAMP
    {
        unsigned long long h[8] = {0};
        (void)hipStreamSynchronize(ctx->stream);
        (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamp), sizeof h);
        const double n = h[7] ? (double)h[7] : 1.0;
        fprintf(stderr, "[align stamps] WGs %llu  setup %.0f  first-chunk %.0f  tile-loop %.0f (%.1f tiles)  tail %.0f  cycles per WG (wave 0)\n", h[7],
                h[0] / n, h[1] / n, h[2] / n, h[6] / n, h[3] / n);
        unsigned long long z8[8] = {0};
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_stamp), z8, sizeof z8);
    }
#endif
    return SRX_OK;
}

template <int KS>
static int dispatch_align(srx_ctx *ctx, srx_model *m, int MP, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *feats, int ld,
                          float *out, unsigned char *fflag, float *dump, const f16x8 *Bp) {
    switch (MP) {
        case 1: return launch_align<1, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump, Bp);
        case 2: return launch_align<2, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump, Bp);
        case 4: return launch_align<4, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump, Bp);
        case 8: return launch_align<8, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump, Bp);
        case 16: return launch_align<16, KS>(ctx, m, hmm, plan, feats, ld, out, fflag, dump, Bp);
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

// The plan's frame-operand cache for (feats, ld, model's centre/scale epoch): (re)built when any of them changed.  Returns
// nullptr in *Bp when the cache is off or device memory is short (the kernel then builds the operands itself).
static int ensure_frame_cache(srx_ctx *ctx, srx_model *m, srx_align_plan *plan, const srx_hmm_batch *hmm, const float *feats, int ld, int KS,
                              const f16x8 **Bp, unsigned char **fflag) {
    *Bp = nullptr;
    *fflag = nullptr;
    if (!plan->cache_frames) return SRX_OK;
    const size_t bytes = (size_t)plan->n_wb * FT_PER_WAVE * 2 * KS * 1024;
    const size_t fbytes = ((size_t)hmm->total_frames + 255) / 256 * 256 + 256;
    if (bytes > plan->bp_bytes || fbytes > plan->bp_flag_bytes) {
        size_t free_b = 0, total_b = 0;
        (void)hipMemGetInfo(&free_b, &total_b);
        if (free_b + plan->bp_bytes < bytes + fbytes + ((size_t)2 << 30)) return SRX_OK;   // keep 2 GiB of headroom: no cache
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        (void)hipFree(plan->bp);
        (void)hipFree(plan->bp_flag);
        plan->bp = nullptr;
        plan->bp_flag = nullptr;
        plan->bp_bytes = plan->bp_flag_bytes = 0;
        plan->bp_frames = -1;
        if (hipMalloc(&plan->bp, bytes) != hipSuccess || hipMalloc((void **)&plan->bp_flag, fbytes) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(plan->bp);
            plan->bp = nullptr;
            return SRX_OK;
        }
        plan->bp_bytes = bytes;
        plan->bp_flag_bytes = fbytes;
    }
    const bool fresh = plan->bp_frames == hmm->total_frames && plan->bp_feats == feats && plan->bp_ld == ld && plan->bp_KS == KS &&
                       plan->bp_model == m && plan->bp_epoch == m->colstats_epoch;
    if (!fresh) {
        SRX_HIP(ctx, hipMemsetAsync(plan->bp_flag, 0, (size_t)hmm->total_frames, ctx->stream));
        const int64_t n = plan->n_wb * FT_PER_WAVE * KS * 64;
        {
            SrxProf prof_(ctx, "k_prep16_align");
            hipLaunchKernelGGL(k_prep16_align, dim3((unsigned)ceil_div64(n, 256)), dim3(256), 0, ctx->stream, feats, ld, m->D, KS, m->colstats,
                               plan->wb_utt, plan->wb_f0, plan->n_wb, hmm->frame_off_dev, (f16x8 *)plan->bp, plan->bp_flag);
        }
        SRX_LAUNCH_CHECK(ctx);
        plan->bp_frames = hmm->total_frames;
        plan->bp_feats = feats;
        plan->bp_ld = ld;
        plan->bp_KS = KS;
        plan->bp_model = m;
        plan->bp_epoch = m->colstats_epoch;
    }
    *Bp = (const f16x8 *)plan->bp;
    *fflag = plan->bp_flag;
    return SRX_OK;
}

extern "C" int srx_gmm_score_align_dev(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats_dev, int ld,
                                       const srx_hmm_batch *hmm, const srx_align_plan *plan_c, float *emis_dev) {
    srx_align_plan *plan = const_cast<srx_align_plan *>(plan_c);   // the frame-operand cache lives in the plan
    SRX_ENTER(ctx, "srx_gmm_score_align_dev");
    SRX_REQUIRE(ctx, m && feats_dev && hmm && plan && emis_dev, "srx_gmm_score_align_dev: null argument");
    SRX_REQUIRE(ctx, mix_mode == SRX_MIX_COMPONENT0 || mix_mode == SRX_MIX_LOGSUMEXP, "srx_gmm_score_align_dev: unknown mix mode %d", mix_mode);
    SRX_REQUIRE(ctx, score_path == SRX_SCORE_MFMA_F16X3 || score_path == SRX_SCORE_EXACT,
                "srx_gmm_score_align_dev: score path must be SRX_SCORE_MFMA_F16X3 or SRX_SCORE_EXACT");
    SRX_REQUIRE(ctx, ld >= m->D, "srx_gmm_score_align_dev: ld %d < D %d", ld, m->D);
    SRX_REQUIRE(ctx, plan->n_utt == hmm->n_utt, "srx_gmm_score_align_dev: plan is for %d utterances, batch has %d", plan->n_utt, hmm->n_utt);
    SRX_REQUIRE(ctx, plan->n_wg < (1ll << 31) && ceil_div64(plan->n_wb, 4) < (1ll << 31), "srx_gmm_score_align_dev: too many frames for one launch");
    SRX_REQUIRE(ctx, hmm->max_states >= 1 && hmm->max_states <= 256, "srx_gmm_score_align_dev: max_states %d outside 1..256", hmm->max_states);
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
    const f16x8 *Bp = nullptr;
    unsigned char *fflag = nullptr;
    int rc = ensure_frame_cache(ctx, m, plan, hmm, feats_dev, ld, KS, &Bp, &fflag);
    if (rc != SRX_OK) return rc;
    float *dump = nullptr;
    if (!Bp) {   // no cache: per-call flags in the context's workspace, operands built inside the kernel
        void *ws = nullptr;
        const size_t flag_bytes = ((size_t)hmm->total_frames + 255) / 256 * 256;
        rc = srx_workspace(ctx, 0, flag_bytes + 256, &ws);
        if (rc != SRX_OK) return rc;
        fflag = reinterpret_cast<unsigned char *>(ws);
        dump = reinterpret_cast<float *>(fflag + flag_bytes);
        SRX_HIP(ctx, hipMemsetAsync(fflag, 0, (size_t)hmm->total_frames, ctx->stream));
    }
    const int MP = padded_components16(Muse);
    rc = KS == 2 ? dispatch_align<2>(ctx, m, MP, hmm, plan, feats_dev, ld, emis_dev, fflag, dump, Bp)
                 : dispatch_align<5>(ctx, m, MP, hmm, plan, feats_dev, ld, emis_dev, fflag, dump, Bp);
    if (rc != SRX_OK) return rc;
    SRX_LAUNCH_CHECK(ctx);
    launch_exact_align(ctx, m, mix_mode, hmm, plan, feats_dev, ld, fflag, emis_dev);
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
