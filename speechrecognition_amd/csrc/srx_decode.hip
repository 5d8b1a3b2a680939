// hmm::viterbi (src/hmm.cpp:150-224) on an ARBITRARY transition list -- the lexicon HMM of
// hmm::hmm(pronlex&, unigram_model&, acoustic_model*, state_model*) (src/hmm.cpp:92-129): a start state with one
// transition per word (LMSF * unigram log-probability), per word a chain of phone states and a terminating silence, and
// back-edges to the start state -- with a finite beam (src/hmm.cpp:165) and either end condition: the last state, as
// compiled (src/hmm.cpp:210-215), or the best active state, the recognition variant the reference has commented out
// (src/hmm.cpp:196-209).  Also the host-side constructor of that topology.
//
// The reference's semantics are kept, including the one that makes it unusual: `logp` is carried across a source
// state's transitions (src/hmm.cpp:164-169), so the candidate of a source's k-th transition contains the transition and
// emission scores of its transitions 0..k-1.  For the start state that is a chain over every word of the lexicon, in
// f32, in order -- a serial dependence, evaluated as such (one lane per fan-out source) so that the score is bit-identical.
//
// One workgroup per utterance, the frame loop inside the kernel.  Per frame:
//   A  every source with more than three transitions (the start state) walks its transition list once and leaves the
//      running values in `pre`;
//   B  every state, as a TARGET, takes the best of its incoming candidates (the few transitions of an ordinary source are
//      re-accumulated on the fly; candidates of a fan-out source come from `pre`), visiting its sources in descending
//      index with a strict > -- the tie rule of the alignment kernel (exact ties go to the higher source index; the
//      reference's own order is unordered_map iteration order).  Targets with many incoming edges (the start state: two
//      per word) are reduced by the whole workgroup;
//   C  the frame's best value sets the next frame's pruning threshold.
// delta and the activity flags (the reference's `layer` maps: a state is in the map or not) live in global memory, so
// the state count is bounded by memory, not by LDS; back-pointers are one int32 per (frame, state).  Every state is
// visited every frame (activity is a flag, not a list): the sweep is dense, which a lexicon of a few thousand states
// does not notice.  The full 125 k-word lexicon (10^6 states) goes through the frontier sweep of srx_decode_frontier.hip
// (srx_decode_plan_set_sweep; chosen automatically), which gives the same results bit for bit.
// NaN emission scores (tied states whose parameters are NaN because training never visited them) are outside the
// contract: the reference's result then depends on its map order; here a NaN candidate that arrives first is kept, as
// in the reference, for ordinary targets, while the workgroup reduction of a fan-in target may pick differently.
#include <algorithm>
#include <vector>

#include "srx_common.h"
#include "srx_decode_plan.h"

// ------------------------------------------------------------------------------------------ host: topology
extern "C" int srx_hmm_build_lexicon(int n_words, const int32_t *phone_off, const int32_t *phones, const float *uni_logp, float lmsf,
                                     int32_t *tr_off, int32_t *tr_tgt, float *tr_logp, int32_t *ctx, int32_t *word_of,
                                     int64_t *n_states_out, int64_t *n_tr_out) {
    if (n_words < 0 || !phone_off || (n_words > 0 && (!phones || !uni_logp)))
        return srx_fail(nullptr, SRX_ERR_INVALID, "srx_hmm_build_lexicon: bad arguments");
    const float LOG_HALF = -0.6931471805f, LOG_THIRD = -1.09861228866f;  // src/hmm.hpp:25-26
    int64_t n_states = 1, n_tr = n_words;
    for (int w = 0; w < n_words; w++) {
        const int len = phone_off[w + 1] - phone_off[w];
        if (len < 1) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_hmm_build_lexicon: word %d has no phones", w);
        n_states += len + 1;
        n_tr += 2 * (int64_t)len + 1 + 2;
    }
    if (n_states_out) *n_states_out = n_states;
    if (n_tr_out) *n_tr_out = n_tr;
    if (!tr_off) return SRX_OK;   // counting call
    if (n_tr > 0 && (!tr_tgt || !tr_logp)) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_hmm_build_lexicon: null output");
    if (n_states >= (1ll << 31) || n_tr >= (1ll << 31)) return srx_fail(nullptr, SRX_ERR_INVALID, "srx_hmm_build_lexicon: too large");
    // the start state: one transition per word, in word order (src/hmm.cpp:108)
    int64_t q = 0, s = 1;
    tr_off[0] = 0;
    for (int w = 0; w < n_words; w++) {
        tr_tgt[q] = (int32_t)s;
        tr_logp[q] = lmsf * uni_logp[w];
        q++;
        s += phone_off[w + 1] - phone_off[w] + 1;
    }
    if (ctx) ctx[0] = ctx[1] = ctx[2] = 0;   // hmm_state(phspec(SIL, BEGIN)): context (SIL, SIL, SIL), src/hmm.cpp:13-15
    if (word_of) word_of[0] = -1;
    s = 1;
    for (int w = 0; w < n_words; w++) {
        const int32_t *p = phones + phone_off[w];
        const int len = phone_off[w + 1] - phone_off[w];
        const int64_t s0 = s, e = s0 + len;   // e: the terminating silence (src/hmm.cpp:107)
        for (int64_t i = s0; i < e; i++) {
            const bool last = i == e - 1;
            tr_off[i] = (int32_t)q;
            tr_tgt[q] = (int32_t)i;       tr_logp[q] = last ? LOG_THIRD : LOG_HALF; q++;   // src/hmm.cpp:111
            tr_tgt[q] = (int32_t)(i + 1); tr_logp[q] = last ? LOG_THIRD : LOG_HALF; q++;   // :112
            if (last) { tr_tgt[q] = 0; tr_logp[q] = LOG_THIRD; q++; }                       // :113
            if (ctx) {
                ctx[i * 3] = i == s0 ? 0 : p[i - s0 - 1];
                ctx[i * 3 + 1] = p[i - s0];
                ctx[i * 3 + 2] = i + 1 == e ? 0 : p[i - s0 + 1];
            }
            if (word_of) word_of[i] = w;
        }
        tr_off[e] = (int32_t)q;
        tr_tgt[q] = (int32_t)e; tr_logp[q] = LOG_HALF; q++;   // self loop, src/hmm.cpp:121
        tr_tgt[q] = 0;          tr_logp[q] = LOG_HALF; q++;   // back to the start state, :122
        if (ctx) {
            ctx[e * 3] = p[len - 1];
            ctx[e * 3 + 1] = 0;
            ctx[e * 3 + 2] = 0;
        }
        if (word_of) word_of[e] = w;
        s = e + 1;
    }
    tr_off[n_states] = (int32_t)q;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ plan: the graph on the device
template <class T>
static hipError_t to_device(T **dst, const std::vector<T> &h) {
    hipError_t e = hipMalloc((void **)dst, (h.size() ? h.size() : 1) * sizeof(T));
    if (e != hipSuccess) return e;
    return h.empty() ? hipSuccess : hipMemcpy(*dst, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
}

// Suffix maxima / magnitude sums of the wide sources' transition scores at tile boundaries (they move with the scores).
static int wide_suffix_tables(srx_decode_plan *p, const float *tr_logp) {
    srx_ctx *ctx = p->ctx;
    std::vector<int32_t> off(1, 0);
    std::vector<float> lpmax;
    std::vector<double> lpabs;
    for (size_t w = 0; w < p->h_wide_src.size(); w++) {
        const int s = p->h_wide_src[w];
        const int t0 = p->h_tr_off[s], K = p->h_tr_off[s + 1] - t0;
        const int nt = (K + FR_TILE - 1) / FR_TILE;
        const size_t base = lpmax.size();
        lpmax.resize(base + nt);
        lpabs.resize(base + nt);
        float mx = -__builtin_huge_valf();
        double sum = 0.0;
        for (int k = K - 1; k >= 0; k--) {
            const float v = tr_logp[t0 + k];
            if (v != v) mx = __builtin_huge_valf();   // a NaN score: +inf from here down, the kernel's test then never passes
            else if (v > mx) mx = v;
            sum += fabs((double)v);
            if (k % FR_TILE == 0) {
                lpmax[base + k / FR_TILE] = mx;
                lpabs[base + k / FR_TILE] = sum;
            }
        }
        off.push_back((int32_t)lpmax.size());
    }
    if (!p->wtile_off) {
        hipError_t e;
        if ((e = to_device(&p->wtile_off, off)) != hipSuccess || (e = to_device(&p->wtile_lpmax, lpmax)) != hipSuccess ||
            (e = to_device(&p->wtile_lpabs, lpabs)) != hipSuccess)
            return srx_fail(ctx, SRX_ERR_NOMEM, "srx_decode_plan: %s", hipGetErrorString(e));
        return SRX_OK;
    }
    if (!lpmax.empty()) {
        SRX_HIP(ctx, hipMemcpy(p->wtile_lpmax, lpmax.data(), lpmax.size() * sizeof(float), hipMemcpyHostToDevice));
        SRX_HIP(ctx, hipMemcpy(p->wtile_lpabs, lpabs.data(), lpabs.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    return SRX_OK;
}

extern "C" void srx_decode_plan_destroy(srx_decode_plan *p) {
    if (!p) return;
    (void)hipFree(p->tr_off); (void)hipFree(p->tr_tgt); (void)hipFree(p->tied); (void)hipFree(p->tr_logp);
    (void)hipFree(p->in_off); (void)hipFree(p->in_src); (void)hipFree(p->in_k); (void)hipFree(p->in_pre);
    (void)hipFree(p->fan_src); (void)hipFree(p->fan_pre); (void)hipFree(p->big_tgt);
    (void)hipFree(p->wide_of); (void)hipFree(p->wide_src); (void)hipFree(p->wide_pre); (void)hipFree(p->wpre_fan); (void)hipFree(p->wpre_tied);
    (void)hipFree(p->wtile_off); (void)hipFree(p->wtile_lpmax); (void)hipFree(p->wtile_lpabs);
    delete p;
}

extern "C" int srx_decode_plan_create(srx_ctx *ctx, int n_states, const int32_t *tr_off, const int32_t *tr_tgt, const float *tr_logp,
                                      const int32_t *tied, srx_decode_plan **out) {
    SRX_ENTER(ctx, "srx_decode_plan_create");
    SRX_REQUIRE(ctx, out && n_states >= 1 && tr_off && tr_tgt && tr_logp && tied, "srx_decode_plan_create: bad argument");
    const int n_tr = tr_off[n_states];
    SRX_REQUIRE(ctx, tr_off[0] == 0 && n_tr >= 0, "srx_decode_plan_create: malformed transition offsets");
    std::vector<int32_t> indeg((size_t)n_states + 1, 0);
    for (int s = 0; s < n_states; s++) {
        SRX_REQUIRE(ctx, tr_off[s + 1] >= tr_off[s], "srx_decode_plan_create: malformed transition offsets at state %d", s);
        for (int ti = tr_off[s]; ti < tr_off[s + 1]; ti++) {
            SRX_REQUIRE(ctx, tr_tgt[ti] >= 0 && tr_tgt[ti] < n_states, "srx_decode_plan_create: transition %d targets state %d of %d", ti,
                        tr_tgt[ti], n_states);
            indeg[(size_t)tr_tgt[ti] + 1]++;
        }
    }
    std::vector<int32_t> in_off((size_t)n_states + 1, 0);
    for (int j = 0; j < n_states; j++) in_off[(size_t)j + 1] = in_off[j] + indeg[(size_t)j + 1];
    std::vector<int32_t> in_src((size_t)n_tr), in_k((size_t)n_tr), in_pre((size_t)n_tr, -1), fill(in_off.begin(), in_off.end() - 1);
    std::vector<int32_t> fan_src, fan_pre, big_tgt;
    std::vector<int32_t> pre_of_state((size_t)n_states, -1);
    int n_pre = 0;
    for (int s = 0; s < n_states; s++)
        if (tr_off[s + 1] - tr_off[s] >= FANOUT_MIN) {
            pre_of_state[s] = n_pre;
            fan_src.push_back(s);
            fan_pre.push_back(n_pre);
            n_pre += tr_off[s + 1] - tr_off[s];
        }
    // sources descending, so that each target's list comes out in the visiting order (ascending k within a source)
    for (int s = n_states - 1; s >= 0; s--)
        for (int ti = tr_off[s]; ti < tr_off[s + 1]; ti++) {
            const int j = tr_tgt[ti];
            const int q = fill[j]++;
            in_src[q] = s;
            in_k[q] = ti - tr_off[s];
            in_pre[q] = pre_of_state[s] >= 0 ? pre_of_state[s] + (ti - tr_off[s]) : -1;
        }
    for (int j = 0; j < n_states; j++)
        if (in_off[(size_t)j + 1] - in_off[j] >= FANIN_MIN) big_tgt.push_back(j);
    std::vector<int32_t> wide_of((size_t)n_states, -1), wide_src, wide_pre, wpre_fan, wpre_tied;
    for (int s = 0; s < n_states; s++)
        if (tr_off[s + 1] - tr_off[s] >= FR_FAN_MIN) {
            wide_of[s] = (int32_t)wide_src.size();
            wide_src.push_back(s);
            wide_pre.push_back((int32_t)wpre_fan.size());
            wpre_fan.insert(wpre_fan.end(), (size_t)(tr_off[s + 1] - tr_off[s]), wide_of[s]);
            for (int ti = tr_off[s]; ti < tr_off[s + 1]; ti++) wpre_tied.push_back(tied[tr_tgt[ti]]);
        }
    srx_decode_plan *p = new srx_decode_plan();
    p->ctx = ctx;
    p->n_wide = (int)wide_src.size();
    p->n_wpre = (int)wpre_fan.size();
    p->n_states = n_states;
    p->n_tr = n_tr;
    p->n_fan = (int)fan_src.size();
    p->n_big = (int)big_tgt.size();
    p->n_pre = n_pre;
    std::vector<int32_t> h_off(tr_off, tr_off + n_states + 1), h_tgt(tr_tgt, tr_tgt + n_tr), h_tied(tied, tied + n_states);
    std::vector<float> h_lp(tr_logp, tr_logp + n_tr);
    hipError_t e;
    if ((e = to_device(&p->tr_off, h_off)) != hipSuccess || (e = to_device(&p->tr_tgt, h_tgt)) != hipSuccess ||
        (e = to_device(&p->tied, h_tied)) != hipSuccess || (e = to_device(&p->tr_logp, h_lp)) != hipSuccess ||
        (e = to_device(&p->in_off, in_off)) != hipSuccess || (e = to_device(&p->in_src, in_src)) != hipSuccess ||
        (e = to_device(&p->in_k, in_k)) != hipSuccess || (e = to_device(&p->in_pre, in_pre)) != hipSuccess ||
        (e = to_device(&p->fan_src, fan_src)) != hipSuccess || (e = to_device(&p->fan_pre, fan_pre)) != hipSuccess ||
        (e = to_device(&p->big_tgt, big_tgt)) != hipSuccess || (e = to_device(&p->wide_of, wide_of)) != hipSuccess ||
        (e = to_device(&p->wide_src, wide_src)) != hipSuccess || (e = to_device(&p->wide_pre, wide_pre)) != hipSuccess ||
        (e = to_device(&p->wpre_fan, wpre_fan)) != hipSuccess || (e = to_device(&p->wpre_tied, wpre_tied)) != hipSuccess) {
        srx_decode_plan_destroy(p);
        return srx_fail(ctx, e == hipErrorOutOfMemory ? SRX_ERR_NOMEM : SRX_ERR_DEVICE, "srx_decode_plan_create: %s", hipGetErrorString(e));
    }
    p->h_tr_off.assign(tr_off, tr_off + n_states + 1);
    p->h_wide_src = wide_src;
    for (int s_ = 0; s_ < n_states; s_++) p->n_cols = std::max(p->n_cols, tied[s_] + 1);
    {
        const int rc = wide_suffix_tables(p, tr_logp);
        if (rc != SRX_OK) {
            srx_decode_plan_destroy(p);
            return rc;
        }
    }
    *out = p;
    return SRX_OK;
}

// hmm::update_states rewrote the transition log-probabilities (src/hmm.cpp:361-366): same topology, new values.
extern "C" int srx_decode_plan_set_transitions(srx_decode_plan *p, const float *tr_logp) {
    if (!p || !tr_logp) return SRX_ERR_INVALID;
    srx_ctx *ctx = p->ctx;
    SRX_HIP(ctx, hipSetDevice(ctx->device));
    SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
    SRX_HIP(ctx, hipMemcpy(p->tr_logp, tr_logp, (size_t)p->n_tr * sizeof(float), hipMemcpyHostToDevice));
    return wide_suffix_tables(p, tr_logp);
}

extern "C" int srx_decode_plan_set_sweep(srx_decode_plan *p, int sweep, size_t max_workspace_bytes) {
    if (!p) return SRX_ERR_INVALID;
    srx_ctx *ctx = p->ctx;
    SRX_REQUIRE(ctx, sweep == SRX_DECODE_AUTO || sweep == SRX_DECODE_DENSE || sweep == SRX_DECODE_FRONTIER, "srx_decode_plan_set_sweep: sweep %d", sweep);
    p->sweep = sweep;
    if (max_workspace_bytes) p->max_workspace = max_workspace_bytes;
    return SRX_OK;
}

// ------------------------------------------------------------------------------------------ the kernel
static constexpr int DEC_THREADS = 256;

struct DecBest {   // a candidate: value and source (or state) index; "none" = idx < 0
    float v;
    int idx;
};
// the tie rule of a descending-index visit with strict >: larger value wins, equal values go to the larger index
__device__ __forceinline__ DecBest dec_better(DecBest a, DecBest b) {
    if (a.idx < 0) return b;
    if (b.idx < 0) return a;
    if (b.v > a.v || (b.v == a.v && b.idx > a.idx)) return b;
    return a;
}
__device__ __forceinline__ DecBest dec_block_reduce(DecBest x, DecBest *red) {
    for (int o = 32; o > 0; o >>= 1) {
        DecBest y;
        y.v = __shfl_xor(x.v, o, 64);
        y.idx = __shfl_xor(x.idx, o, 64);
        x = dec_better(x, y);
    }
    __syncthreads();   // red[] may still be read from the previous reduction
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = x;
    __syncthreads();
    DecBest r = red[0];
    for (int w = 1; w < DEC_THREADS / 64; w++) r = dec_better(r, red[w]);
    return r;
}

__global__ __launch_bounds__(DEC_THREADS) void k_viterbi_graph(int n_states, const int32_t *__restrict__ tr_off, const int32_t *__restrict__ tr_tgt,
                                                               const float *__restrict__ tr_logp, const int32_t *__restrict__ tied,
                                                               const int32_t *__restrict__ in_off, const int32_t *__restrict__ in_src,
                                                               const int32_t *__restrict__ in_k, const int32_t *__restrict__ in_pre,
                                                               int n_fan, const int32_t *__restrict__ fan_src, const int32_t *__restrict__ fan_pre,
                                                               int n_big, const int32_t *__restrict__ big_tgt, int n_pre,
                                                               const float *__restrict__ emis, int64_t st_t, int64_t st_g,
                                                               const int64_t *__restrict__ foff, const int64_t *__restrict__ bp_off, float beam,
                                                               int final_mode, float *__restrict__ dws, unsigned char *__restrict__ aws,
                                                               float *__restrict__ pws, int32_t *__restrict__ bpws, int32_t *__restrict__ path,
                                                               float *__restrict__ final_logp, int32_t *__restrict__ final_state,
                                                               int32_t *__restrict__ status) {
    __shared__ DecBest red[DEC_THREADS / 64];
    const int u = blockIdx.x, tid = threadIdx.x;
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    float *d_cur = dws + (size_t)u * 2 * n_states, *d_nxt = d_cur + n_states;
    unsigned char *a_cur = aws + (size_t)u * 2 * n_states, *a_nxt = a_cur + n_states;
    float *pre = pws + (size_t)u * (n_pre > 0 ? n_pre : 1);
    int32_t *bp = bpws + bp_off[u];
    for (int j = tid; j < n_states; j += DEC_THREADS) {   // curr = {0: 0.0f}, src/hmm.cpp:157
        d_cur[j] = 0.0f;
        a_cur[j] = j == 0;
    }
    float cur_best = -1e10f;                              // src/hmm.cpp:156
    __syncthreads();
    for (int t = 0; t < T; t++) {
        const float *e_t = emis + (f0 + t) * st_t;
        const float thresh = cur_best - beam;            // src/hmm.cpp:165
        // ---- A: running values of the fan-out sources, transition by transition (src/hmm.cpp:166-169)
        for (int fi = tid; fi < n_fan; fi += DEC_THREADS) {
            const int s = fan_src[fi];
            if (a_cur[s] && !(d_cur[s] < thresh)) {
                float x = d_cur[s];
                float *dst = pre + fan_pre[fi];
                for (int ti = tr_off[s]; ti < tr_off[s + 1]; ti++) {
                    x = x + tr_logp[ti];
                    x = x + e_t[(int64_t)tied[tr_tgt[ti]] * st_g];
                    dst[ti - tr_off[s]] = x;
                }
            }
        }
        if (n_fan > 0) __syncthreads();
        // the candidate that source s offers through its k-th transition
        auto candidate = [&](int q) __attribute__((always_inline)) -> float {
            const int ip = in_pre[q];
            if (ip >= 0) return pre[ip];
            const int s = in_src[q], k = in_k[q];
            float x = d_cur[s];
            for (int ti = tr_off[s]; ti <= tr_off[s] + k; ti++) {
                x = x + tr_logp[ti];
                x = x + e_t[(int64_t)tied[tr_tgt[ti]] * st_g];
            }
            return x;
        };
        // ---- B: every state as a target (src/hmm.cpp:171-183)
        float fbv = -1e10f;        // next_best_logp: fmaxf over the accepted values, from -1e10f (src/hmm.cpp:161, 176, 181)
        for (int j = tid; j < n_states; j += DEC_THREADS) {
            const int q0 = in_off[j], q1 = in_off[j + 1];
            if (q1 - q0 >= FANIN_MIN) continue;
            DecBest b = {0.0f, -1};
            for (int q = q0; q < q1; q++) {
                const int s = in_src[q];
                if (!a_cur[s] || d_cur[s] < thresh) continue;
                const float c = candidate(q);
                if (b.idx < 0 || c > b.v) { b.v = c; b.idx = s; }
            }
            a_nxt[j] = b.idx >= 0;
            if (b.idx >= 0) {
                d_nxt[j] = b.v;
                bp[(size_t)t * n_states + j] = b.idx;
                fbv = fmaxf(fbv, b.v);
            }
        }
        for (int bi = 0; bi < n_big; bi++) {   // targets with many incoming edges: all threads, then one reduction
            const int j = big_tgt[bi];
            DecBest b = {0.0f, -1};
            for (int q = in_off[j] + tid; q < in_off[j + 1]; q += DEC_THREADS) {
                const int s = in_src[q];
                if (!a_cur[s] || d_cur[s] < thresh) continue;
                DecBest c = {candidate(q), s};
                b = dec_better(b, c);
            }
            b = dec_block_reduce(b, red);
            if (tid == 0) {
                a_nxt[j] = b.idx >= 0;
                if (b.idx >= 0) {
                    d_nxt[j] = b.v;
                    bp[(size_t)t * n_states + j] = b.idx;
                }
            }
            if (b.idx >= 0) fbv = fmaxf(fbv, b.v);
        }
        // ---- C: current_best_logp = next_best_logp (src/hmm.cpp:187)
        for (int o = 32; o > 0; o >>= 1) fbv = fmaxf(fbv, __shfl_xor(fbv, o, 64));
        __syncthreads();
        if ((tid & 63) == 0) red[tid >> 6].v = fbv;
        __syncthreads();
        cur_best = red[0].v;
        for (int w = 1; w < DEC_THREADS / 64; w++) cur_best = fmaxf(cur_best, red[w].v);
        float *tf = d_cur; d_cur = d_nxt; d_nxt = tf;
        unsigned char *ta = a_cur; a_cur = a_nxt; a_nxt = ta;
        __syncthreads();
    }
    // ---- end condition
    DecBest fin = {0.0f, -1};
    if (T >= 1) {
        if (final_mode == 0) {            // the last state, src/hmm.cpp:210-215
            if (a_cur[n_states - 1]) { fin.v = d_cur[n_states - 1]; fin.idx = n_states - 1; }
        } else {                          // the best active state above -1e10f, src/hmm.cpp:196-209 (commented out there)
            for (int j = tid; j < n_states; j += DEC_THREADS)
                if (a_cur[j] && d_cur[j] > -1e10f) {
                    DecBest c = {d_cur[j], j};
                    fin = dec_better(fin, c);
                }
            fin = dec_block_reduce(fin, red);
        }
    }
    if (tid == 0) {
        if (fin.idx < 0) {
            status[u] = 1;                // "NO PATH" / throw 1729
        } else {
            status[u] = 0;
            final_logp[u] = fin.v;
            if (final_state) final_state[u] = fin.idx;
            int b = fin.idx;              // src/hmm.cpp:219-222
            for (int t = T - 1; t >= 0; t--) {
                path[f0 + t] = b;
                b = bp[(size_t)t * n_states + b];
            }
        }
    }
}

extern "C" int srx_viterbi_graph_dev(srx_ctx *ctx, const srx_decode_plan *plan, const float *emis_dev, int64_t stride_t, int64_t stride_g,
                                     const int64_t *frame_off, int n_utt, float beam_width, int final_mode, int32_t *path_dev,
                                     float *final_logp_dev, int32_t *final_state_dev, int32_t *status_dev) {
    SRX_ENTER(ctx, "srx_viterbi_graph_dev");
    SRX_REQUIRE(ctx, plan && emis_dev && frame_off && path_dev && final_logp_dev && status_dev && n_utt >= 0, "srx_viterbi_graph_dev: null argument");
    SRX_REQUIRE(ctx, final_mode == 0 || final_mode == 1, "srx_viterbi_graph_dev: final_mode %d", final_mode);
    if (n_utt == 0) return SRX_OK;
    const int n = plan->n_states;
    int64_t tmax = 0;
    for (int u = 0; u < n_utt; u++) {
        SRX_REQUIRE(ctx, frame_off[u + 1] >= frame_off[u] && frame_off[u + 1] - frame_off[u] < (1ll << 31), "srx_viterbi_graph_dev: bad frame offsets");
        tmax = std::max<int64_t>(tmax, frame_off[u + 1] - frame_off[u]);
    }
    // The dense sweep keeps one int32 back-pointer per (frame, state) and visits every state every frame with one workgroup per
    // utterance: right for lexica of some thousands of states and many utterances at once (they go in groups whose back-pointers
    // fit the plan's workspace limit).  Beyond that -- or when one utterance alone would not fit -- the frontier sweep
    // (srx_decode_frontier.hip).
    const size_t limit = plan->max_workspace;
    const bool dense_fits = n <= 32768 && (size_t)tmax * n * 4 + 4 <= limit / 2;
    if (plan->sweep == SRX_DECODE_FRONTIER || (plan->sweep == SRX_DECODE_AUTO && !dense_fits))
        return srx_viterbi_frontier(ctx, plan, emis_dev, stride_t, stride_g, frame_off, n_utt, beam_width, final_mode, path_dev, final_logp_dev,
                                    final_state_dev, status_dev);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t npre = (size_t)(plan->n_pre > 0 ? plan->n_pre : 1);
    int u0 = 0;
    while (u0 < n_utt) {
        // a group of consecutive utterances: back-pointers within half the limit (SRX_DECODE_DENSE forced on a single utterance
        // larger than that still runs, alone)
        int U = 0;
        std::vector<int64_t> bp_off(1, 0);
        while (u0 + U < n_utt) {
            const int64_t add = (frame_off[u0 + U + 1] - frame_off[u0 + U]) * n;
            if (U > 0 && (size_t)(bp_off.back() + add) * 4 > limit / 2) break;
            bp_off.push_back(bp_off.back() + add);
            U++;
        }
        // workspace (slot 1, shared with the alignment kernel's back-pointers): frame and back-pointer offsets, delta, flags,
        // running values, back-pointers
        const size_t b_foff = up(((size_t)U + 1) * 8), b_bpo = b_foff, b_d = up((size_t)U * 2 * n * 4), b_a = up((size_t)U * 2 * n),
                     b_p = up((size_t)U * npre * 4), b_bp = up((size_t)bp_off[U] * 4 + 4);
        void *ws = nullptr;
        int rc = srx_workspace(ctx, 1, b_foff + b_bpo + b_d + b_a + b_p + b_bp, &ws);
        if (rc != SRX_OK) return rc;
        char *w = reinterpret_cast<char *>(ws);
        int64_t *foff_d = reinterpret_cast<int64_t *>(w);
        int64_t *bpo_d = reinterpret_cast<int64_t *>(w + b_foff);
        float *dws = reinterpret_cast<float *>(w + b_foff + b_bpo);
        unsigned char *aws = reinterpret_cast<unsigned char *>(w + b_foff + b_bpo + b_d);
        float *pws = reinterpret_cast<float *>(w + b_foff + b_bpo + b_d + b_a);
        int32_t *bpws = reinterpret_cast<int32_t *>(w + b_foff + b_bpo + b_d + b_a + b_p);
        SRX_HIP(ctx, hipMemcpyAsync(foff_d, frame_off + u0, ((size_t)U + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        SRX_HIP(ctx, hipMemcpyAsync(bpo_d, bp_off.data(), ((size_t)U + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the two host arrays above are about to go out of scope
        {
            SrxProf prof_(ctx, "k_viterbi_graph");
            hipLaunchKernelGGL(k_viterbi_graph, dim3(U), dim3(DEC_THREADS), 0, ctx->stream, n, plan->tr_off, plan->tr_tgt, plan->tr_logp,
                               plan->tied, plan->in_off, plan->in_src, plan->in_k, plan->in_pre, plan->n_fan, plan->fan_src, plan->fan_pre,
                               plan->n_big, plan->big_tgt, plan->n_pre, emis_dev, stride_t, stride_g, foff_d, bpo_d, beam_width, final_mode, dws, aws,
                               pws, bpws, path_dev, final_logp_dev + u0, final_state_dev ? final_state_dev + u0 : nullptr, status_dev + u0);
        }
        SRX_LAUNCH_CHECK(ctx);
        u0 += U;
        if (u0 < n_utt) SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));   // the next group reuses the workspace
    }
    return SRX_OK;
}
