// hmm::viterbi (src/hmm.cpp:150-224) on the lexicon HMM of src/hmm.cpp:92-129 at the size the reference builds it
// (src/main.cpp:58: every word of res/lexicon/cmu.txt, about 10^6 states): the FRONTIER sweep of srx_viterbi_graph_dev.
//
// The reference keeps its layers in maps: a state is a source at frame t only if it is in the map and its value passes
// `logp < current_best_logp - beam_width` (src/hmm.cpp:165).  This sweep keeps exactly those states, as a list F_t per
// frame, and lets the whole GPU work on one frame (a few small launches per frame) instead of one workgroup sweeping every
// state:
//   push     every member of F_t walks its transitions with `logp` carried across them (src/hmm.cpp:164-169) and offers each
//            candidate to its target with a 64-bit atomic max on (value as ordered bits, source index): the larger value
//            wins, exact ties go to the higher source index -- the rule of the dense sweep (a descending visit with a strict
//            `>`).  The first candidate to reach a target also appends it to the frame's `touched` list.
//   chain    a member with many transitions (the start state: one per word) has its running values computed by ONE lane,
//            serially, in f32 (the chain is a floating-point prefix: any other order changes the bits), from tiles that
//            the rest of the wave gathers and stages in LDS ahead of it;
//   wpush    ... and offered in parallel afterwards.  By then the frame's best value is final, so a chain value that is
//            already below next frame's threshold is not materialised at all: its target could only ever be skipped at
//            src/hmm.cpp:165 (another candidate that beats it is offered anyway).  The last frame keeps everything, because
//            the end condition looks at membership of the map (src/hmm.cpp:196-215).
//   compact  touched targets whose value passes next frame's threshold become F_{t+1}; each records its position-in-F_t
//            back-pointer.  That is the whole back-pointer store: 8 bytes per frontier member instead of 4 bytes per
//            (frame, state).
// The traceback follows positions, one step per frame.  Everything per-state (candidates, positions, values) is O(n_states)
// per utterance in flight; the store is bounded by srx_decode_plan_set_sweep's workspace limit (status 3 when the beam is too
// wide for it).  NaN scores are outside the contract, as in the dense sweep.
#include <algorithm>
#include <vector>

#include "srx_common.h"
#include "srx_decode_plan.h"

namespace {

constexpr int FR_THREADS = 256;

struct FrCtl {
    int n_touched;
    int fcount[2];         // members of F_t (index t & 1) and of F_{t+1}
    unsigned best_bits;    // next_best_logp being built (src/hmm.cpp:161), as ordered bits
    int dead;              // the store is exhausted
    int pad[3];
};

struct FrArgs {
    int n;
    const int32_t *tr_off, *tr_tgt, *tied;
    const float *tr_logp;
    int n_wide, n_wpre;
    const int32_t *wide_of, *wide_src, *wide_pre, *wpre_fan, *wpre_tied, *wtile_off;
    const float *wtile_lpmax;
    const double *wtile_lpabs;
    int n_cols;
    const float *emis;
    int64_t st_t, st_g;
    const int64_t *foff;            // [U + 1] frame offsets of the utterances in flight
    float beam;
    int final_mode;
    unsigned long long *cand;       // [U][n]  best candidate of every target of the layer being built; 0 = none
    int32_t *touched;               // [U][n]  targets that have a candidate
    int32_t *pos;                   // [U][2][n] position of a state in F_t
    float *fval;                    // [U][2][n] value of F_t's p-th member
    float *pre;                     // [U][n_wpre] running values of the wide sources
    int32_t *wact;                  // [U][2][n_wide] 1 + position in F_t of a wide source that is a member, else 0
    int32_t *wdone;                 // [U][n_wide] transitions of a wide member whose running values exist this frame
    FrCtl *ctl;                     // [U]
    int64_t *fs_off;                // [U][tmax + 2] start of F_t in the store
    int32_t *st_state, *st_back;    // [U][cap]
    int64_t cap;
    int tmax;
    int32_t *path;
    float *final_logp;
    int32_t *final_state, *status;  // of the utterances in flight
};

// f32 as an unsigned key with the same order; -0 is stored as +0 (they compare equal)
__device__ __forceinline__ unsigned f2o(float x) {
    unsigned u = __float_as_uint(x);
    if (u == 0x80000000u) u = 0;
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float o2f(unsigned o) { return __uint_as_float((o & 0x80000000u) ? (o & 0x7fffffffu) : ~o); }

// list append for the lanes of a wave whose `take` is set: one atomic per wave
__device__ __forceinline__ int fr_wave_slot(int *counter, bool take) {
    const unsigned long long m = __ballot(take);
    if (!m) return -1;
    const int lane = __lane_id();
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    return take ? base + __popcll(m & ((1ull << lane) - 1ull)) : -1;
}

__device__ __forceinline__ void fr_offer(const FrArgs &a, int u, int j, float x, int s) {
    const unsigned long long key = ((unsigned long long)f2o(x) << 32) | (unsigned)s;
    unsigned long long *c = a.cand + (size_t)u * a.n + j;
    // values only grow within a frame: a (possibly stale) larger value already there settles it
    bool first = false;
    if (__atomic_load_n(c, __ATOMIC_RELAXED) < key) first = atomicMax(c, key) == 0;
    const int slot = fr_wave_slot(&a.ctl[u].n_touched, first);
    if (first) a.touched[(size_t)u * a.n + slot] = j;
}

__device__ __forceinline__ void fr_best(const FrArgs &a, int u, float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    if (__lane_id() == 0 && v > -1e10f) atomicMax(&a.ctl[u].best_bits, f2o(v));
}

__global__ __launch_bounds__(64) void k_fr_init(FrArgs a) {
    const int u = blockIdx.x;
    if (threadIdx.x != 0) return;
    const int T = (int)(a.foff[u + 1] - a.foff[u]);
    // curr = {0: 0.0f}, current_best_logp = -1e10f (src/hmm.cpp:156-157); the start state is a source unless the beam is negative
    const int in0 = T >= 1 && !(0.0f < -1e10f - a.beam);
    FrCtl c = {};
    c.fcount[0] = in0;
    c.best_bits = f2o(-1e10f);
    a.ctl[u] = c;
    int64_t *fo = a.fs_off + (size_t)u * (a.tmax + 2);
    fo[0] = 0;
    fo[1] = in0;
    if (in0) {
        a.st_state[(size_t)u * a.cap] = 0;
        a.st_back[(size_t)u * a.cap] = -1;
        a.fval[(size_t)u * 2 * a.n] = 0.0f;
        a.pos[(size_t)u * 2 * a.n] = 0;
        if (a.wide_of[0] >= 0) a.wact[(size_t)u * 2 * a.n_wide + a.wide_of[0]] = 1;
    }
}

// ordinary members of F_t: transitions walked in the thread
__global__ __launch_bounds__(FR_THREADS) void k_fr_push(FrArgs a, int t) {
    const int u = blockIdx.y;
    const int64_t f0 = a.foff[u];
    const int T = (int)(a.foff[u + 1] - f0);
    if (t >= T || a.ctl[u].dead) return;
    const int cnt = a.ctl[u].fcount[t & 1];
    const float *e_t = a.emis + (f0 + t) * a.st_t;
    const int64_t off = a.fs_off[(size_t)u * (a.tmax + 2) + t];
    const int32_t *fst = a.st_state + (size_t)u * a.cap + off;
    const float *fv = a.fval + ((size_t)u * 2 + (t & 1)) * a.n;
    float lbest = -1e10f;
    for (int p = blockIdx.x * FR_THREADS + threadIdx.x; p < cnt; p += gridDim.x * FR_THREADS) {
        const int s = fst[p];
        const int t0 = a.tr_off[s], t1 = a.tr_off[s + 1];
        if (t1 - t0 >= FR_FAN_MIN) continue;     // k_fr_chain / k_fr_wpush
        float x = fv[p];
        for (int ti = t0; ti < t1; ti++) {
            const int j = a.tr_tgt[ti];
            x = x + a.tr_logp[ti];
            x = x + e_t[(int64_t)a.tied[j] * a.st_g];
            fr_offer(a, u, j, x, s);
            lbest = fmaxf(lbest, x);
        }
    }
    fr_best(a, u, lbest);
}

// the running values of a wide member: one lane, in order; the wave stages the operands
__global__ __launch_bounds__(64) void k_fr_chain(FrArgs a, int t) {
    __shared__ __align__(16) float s_lp[FR_TILE], s_e[FR_TILE], s_x[FR_TILE];
    const int u = blockIdx.y, fi = blockIdx.x, lane = threadIdx.x;
    const int64_t f0 = a.foff[u];
    const int T = (int)(a.foff[u + 1] - f0);
    if (t >= T || a.ctl[u].dead) return;
    const int wp = a.wact[((size_t)u * 2 + (t & 1)) * a.n_wide + fi];
    if (!wp) return;
    const float *e_t = a.emis + (f0 + t) * a.st_t;
    const int s = a.wide_src[fi];
    const int t0 = a.tr_off[s], K = a.tr_off[s + 1] - t0;
    float *pre = a.pre + (size_t)u * a.n_wpre + a.wide_pre[fi];
    float x = a.fval[((size_t)u * 2 + (t & 1)) * a.n + (wp - 1)];
    float best = -1e10f;
    constexpr int PER = FR_TILE / 64;
    // Operands travel two tiles ahead of the chain, without a branch (indices clamped to the last transition), so that no
    // load is waited for where it is issued: emission columns (contiguous, made at plan creation) for tile k+2, the gathered
    // emission scores and the transition scores for tile k+1, while lane 0 adds up tile k.
    const int32_t *wt = a.wpre_tied + a.wide_pre[fi];
    int r_ix[PER];
    float r_lp[PER], r_e[PER];
    auto fetch_ix = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; i++) r_ix[i] = wt[min(k0 + i * 64 + lane, K - 1)];
    };
    auto fetch = [&](int k0) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            r_lp[i] = a.tr_logp[t0 + min(k0 + i * 64 + lane, K - 1)];
            r_e[i] = e_t[(int64_t)r_ix[i] * a.st_g];
        }
    };
    fetch_ix(0);
    fetch(0);
    fetch_ix(FR_TILE);
    // Stopping early.  Once the running value is below (a lower bound of) next frame's threshold AND can never come back up,
    // the rest of the list only produces values that k_fr_wpush would drop (not on an utterance's last frame, where it drops
    // nothing).  "Never come back up" is decided without touching the rest: every remaining step adds a transition score
    // <= lpmax (suffix maximum, from the plan) and an emission score <= gpos (this frame's largest, 0 if all are negative);
    // with d = -(lpmax + gpos) > 0 a step computes fl(fl(x + lp) + g) <= x - d + ulp(M), M bounding every magnitude on the way
    // (|x| + the remaining |lp| + the remaining steps * the frame's largest |emission|), so d >= ulp(M) makes the values
    // non-increasing from here on.  ulp(M) <= 2^-22 M; 2.5e-7 is used.  Any NaN emission switches the test off.
    const bool last_frame = t == T - 1;
    float gpos = 0.0f, gabs = 0.0f;
    int nan_seen = 0;
    for (int c = lane; c < a.n_cols; c += 64) {
        const float v = e_t[(int64_t)c * a.st_g];
        nan_seen |= !(v == v);
        gpos = fmaxf(gpos, v);
        gabs = fmaxf(gabs, fabsf(v));
    }
    for (int o = 32; o > 0; o >>= 1) {
        gpos = fmaxf(gpos, __shfl_xor(gpos, o, 64));
        gabs = fmaxf(gabs, __shfl_xor(gabs, o, 64));
        nan_seen |= __shfl_xor(nan_seen, o, 64);
    }
    const float best_before = o2f(a.ctl[u].best_bits);      // the ordinary members have all offered (k_fr_push ran before)
    const int tile0 = a.wtile_off[fi];
    int ndone = K;
    for (int k0 = 0; k0 < K; k0 += FR_TILE) {
#pragma unroll
        for (int i = 0; i < PER; i++) {
            s_lp[i * 64 + lane] = r_lp[i];
            s_e[i * 64 + lane] = r_e[i];
        }
        __syncthreads();
        fetch(k0 + FR_TILE);                           // in flight while the chain runs
        fetch_ix(k0 + 2 * FR_TILE);
        const int m = min(FR_TILE, K - k0);
        if (lane == 0) {
            // the serial part: two dependent f32 additions per transition (src/hmm.cpp:167-168).  Operands are read two
            // quads ahead of the additions that use them; results go to an array of their own, so that nothing orders the
            // reads behind the writes.
            const float4 *lp4 = reinterpret_cast<const float4 *>(s_lp), *e4 = reinterpret_cast<const float4 *>(s_e);
            float4 *x4 = reinterpret_cast<float4 *>(s_x);
            const int nq = m >> 2;
            float4 l0 = lp4[0], e0 = e4[0], l1 = lp4[1], e1 = e4[1];     // the arrays hold FR_TILE >= 8 floats whatever m is
#pragma unroll 4
            for (int q = 0; q < nq; q++) {
                const int qn = min(q + 2, FR_TILE / 4 - 1);
                const float4 l2 = lp4[qn], e2 = e4[qn];
                float4 o;
                x = x + l0.x; x = x + e0.x; o.x = x;
                x = x + l0.y; x = x + e0.y; o.y = x;
                x = x + l0.z; x = x + e0.z; o.z = x;
                x = x + l0.w; x = x + e0.w; o.w = x;
                x4[q] = o;
                l0 = l1; e0 = e1; l1 = l2; e1 = e2;
            }
            for (int i = nq * 4; i < m; i++) {
                x = x + s_lp[i];
                x = x + s_e[i];
                s_x[i] = x;
            }
        }
        __syncthreads();
        for (int i = lane; i < m; i += 64) {
            const float v = s_x[i];
            pre[k0 + i] = v;
            best = fmaxf(best, v);
        }
        __syncthreads();
        if (!last_frame && !nan_seen && k0 + FR_TILE < K) {
            for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
            const float x_end = __shfl(x, 0, 64);
            const float thr_lb = fmaxf(best_before, best) - a.beam;       // <= next frame's threshold: the frame's best only grows
            const int ti = tile0 + k0 / FR_TILE + 1;
            const double d = -((double)a.wtile_lpmax[ti] + (double)gpos);
            const double M = fabs((double)x_end) + a.wtile_lpabs[ti] + (double)(K - (k0 + FR_TILE)) * (double)gabs;
            if (x_end < thr_lb && d > 0.0 && d >= 2.5e-7 * M) {
                ndone = k0 + FR_TILE;
                break;
            }
        }
    }
    for (int o = 32; o > 0; o >>= 1) best = fmaxf(best, __shfl_xor(best, o, 64));
    if (lane == 0) {
        a.wdone[(size_t)u * a.n_wide + fi] = ndone;
        if (best > -1e10f) atomicMax(&a.ctl[u].best_bits, f2o(best));
    }
}

// the candidates of the wide members, in parallel; the frame's best value is final by now
__global__ __launch_bounds__(FR_THREADS) void k_fr_wpush(FrArgs a, int t) {
    const int u = blockIdx.y;
    const int T = (int)(a.foff[u + 1] - a.foff[u]);
    if (t >= T || a.ctl[u].dead) return;
    const bool last = t == T - 1;
    const float thr_next = o2f(a.ctl[u].best_bits) - a.beam;      // src/hmm.cpp:165 as the next frame will evaluate it
    const int32_t *wact = a.wact + ((size_t)u * 2 + (t & 1)) * a.n_wide;
    const float *pre = a.pre + (size_t)u * a.n_wpre;
    for (int e = blockIdx.x * FR_THREADS + threadIdx.x; e < a.n_wpre; e += gridDim.x * FR_THREADS) {
        const int fi = a.wpre_fan[e];
        if (!wact[fi] || e - a.wide_pre[fi] >= a.wdone[(size_t)u * a.n_wide + fi]) continue;   // not a member / the chain stopped before
        const float x = pre[e];
        if (!last && x < thr_next) continue;
        const int s = a.wide_src[fi];
        fr_offer(a, u, a.tr_tgt[a.tr_off[s] + (e - a.wide_pre[fi])], x, s);
    }
}

// F_{t+1}: the touched targets that next frame's threshold lets through; candidates cleared for the next frame
__global__ __launch_bounds__(FR_THREADS) void k_fr_compact(FrArgs a, int t) {
    const int u = blockIdx.y;
    const int T = (int)(a.foff[u + 1] - a.foff[u]);
    if (t >= T - 1 || a.ctl[u].dead) return;      // the last layer stays as it is for the end condition
    FrCtl *ctl = a.ctl + u;
    const float thr_next = o2f(ctl->best_bits) - a.beam;
    const int nt = ctl->n_touched;
    const int64_t off_next = a.fs_off[(size_t)u * (a.tmax + 2) + t + 1];
    unsigned long long *cand = a.cand + (size_t)u * a.n;
    const int32_t *touched = a.touched + (size_t)u * a.n;
    const int32_t *pos_t = a.pos + ((size_t)u * 2 + (t & 1)) * a.n;
    int32_t *pos_n = a.pos + ((size_t)u * 2 + ((t + 1) & 1)) * a.n;
    float *fv_n = a.fval + ((size_t)u * 2 + ((t + 1) & 1)) * a.n;
    int32_t *wact_n = a.wact + ((size_t)u * 2 + ((t + 1) & 1)) * a.n_wide;
    const int rounds = (nt + FR_THREADS * (int)gridDim.x - 1) / (FR_THREADS * (int)gridDim.x);
    for (int r = 0; r < rounds; r++) {            // whole waves stay in the loop: fr_wave_slot is a wave operation
        const int i = (r * (int)gridDim.x + (int)blockIdx.x) * FR_THREADS + (int)threadIdx.x;
        bool keep = false;
        int j = 0, s = 0;
        float v = 0.0f;
        if (i < nt) {
            j = touched[i];
            const unsigned long long key = cand[j];
            cand[j] = 0;
            v = o2f((unsigned)(key >> 32));
            s = (int)(unsigned)key;
            keep = !(v < thr_next);
        }
        const int p = fr_wave_slot(&ctl->fcount[(t + 1) & 1], keep);
        if (keep) {
            if (off_next + p < a.cap) {
                a.st_state[(size_t)u * a.cap + off_next + p] = j;
                a.st_back[(size_t)u * a.cap + off_next + p] = pos_t[s];
                fv_n[p] = v;
                pos_n[j] = p;
                if (a.wide_of[j] >= 0) wact_n[a.wide_of[j]] = p + 1;
            } else {
                ctl->dead = 1;
            }
        }
    }
}

__global__ __launch_bounds__(64) void k_fr_advance(FrArgs a, int t) {
    const int u = blockIdx.x;
    const int T = (int)(a.foff[u + 1] - a.foff[u]);
    if (t >= T - 1 || a.ctl[u].dead) return;
    FrCtl *ctl = a.ctl + u;
    int32_t *wact = a.wact + ((size_t)u * 2 + (t & 1)) * a.n_wide;
    for (int i = threadIdx.x; i < a.n_wide; i += 64) wact[i] = 0;
    if (threadIdx.x == 0) {
        int64_t *fo = a.fs_off + (size_t)u * (a.tmax + 2);
        fo[t + 2] = fo[t + 1] + ctl->fcount[(t + 1) & 1];
        ctl->n_touched = 0;
        ctl->fcount[t & 1] = 0;
        ctl->best_bits = f2o(-1e10f);
    }
}

struct FrBest {
    float v;
    int idx;
};
__device__ __forceinline__ FrBest fr_better(FrBest x, FrBest y) {
    if (x.idx < 0) return y;
    if (y.idx < 0) return x;
    if (y.v > x.v || (y.v == x.v && y.idx > x.idx)) return y;
    return x;
}

// the end condition (src/hmm.cpp:196-215) and the path (src/hmm.cpp:219-222)
__global__ __launch_bounds__(FR_THREADS) void k_fr_final(FrArgs a) {
    __shared__ FrBest red[FR_THREADS / 64];
    const int u = blockIdx.x, tid = threadIdx.x;
    const int64_t f0 = a.foff[u];
    const int T = (int)(a.foff[u + 1] - f0);
    const FrCtl *ctl = a.ctl + u;
    const unsigned long long *cand = a.cand + (size_t)u * a.n;
    FrBest fin = {0.0f, -1};
    const bool dead = ctl->dead != 0;
    if (T >= 1 && !dead) {
        if (a.final_mode == 0) {
            const unsigned long long key = cand[a.n - 1];
            if (key) { fin.v = o2f((unsigned)(key >> 32)); fin.idx = a.n - 1; }
        } else {
            const int32_t *touched = a.touched + (size_t)u * a.n;
            for (int i = tid; i < ctl->n_touched; i += FR_THREADS) {
                const int j = touched[i];
                const float v = o2f((unsigned)(cand[j] >> 32));
                if (v > -1e10f) {
                    const FrBest c = {v, j};
                    fin = fr_better(fin, c);
                }
            }
            for (int o = 32; o > 0; o >>= 1) {
                FrBest y;
                y.v = __shfl_xor(fin.v, o, 64);
                y.idx = __shfl_xor(fin.idx, o, 64);
                fin = fr_better(fin, y);
            }
            if ((tid & 63) == 0) red[tid >> 6] = fin;
            __syncthreads();
            fin = red[0];
            for (int w = 1; w < FR_THREADS / 64; w++) fin = fr_better(fin, red[w]);
        }
    }
    if (tid != 0) return;
    if (dead) {
        a.status[u] = 3;
    } else if (fin.idx < 0) {
        a.status[u] = 1;
    } else {
        a.status[u] = 0;
        a.final_logp[u] = fin.v;
        if (a.final_state) a.final_state[u] = fin.idx;
        const int64_t *fo = a.fs_off + (size_t)u * (a.tmax + 2);
        const int s = (int)(unsigned)cand[fin.idx];
        int p = a.pos[((size_t)u * 2 + ((T - 1) & 1)) * a.n + s];
        a.path[f0 + T - 1] = fin.idx;
        for (int t = T - 2; t >= 0; t--) {
            const size_t e = (size_t)u * a.cap + fo[t + 1] + p;
            a.path[f0 + t] = a.st_state[e];
            p = a.st_back[e];
        }
    }
}

}  // namespace

int srx_viterbi_frontier(srx_ctx *ctx, const srx_decode_plan *plan, const float *emis_dev, int64_t stride_t, int64_t stride_g,
                         const int64_t *frame_off, int n_utt, float beam_width, int final_mode, int32_t *path_dev, float *final_logp_dev,
                         int32_t *final_state_dev, int32_t *status_dev) {
    const size_t n = (size_t)plan->n_states, nw = (size_t)std::max(plan->n_wide, 1), nwp = (size_t)std::max(plan->n_wpre, 1);
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    int u0 = 0;
    while (u0 < n_utt) {
        // utterances in flight: as many as the workspace limit carries with a store of sixteen full layers each (or all the
        // layers there can be, if that is less)
        const size_t fixed = up(n * 8) + up(n * 4) + 2 * up(2 * n * 4) + up(nwp * 4) + up(2 * nw * 4) + up(nw * 4) + 256;
        int tmax_rem = 0;
        for (int u = u0; u < n_utt; u++) tmax_rem = std::max<int>(tmax_rem, (int)(frame_off[u + 1] - frame_off[u]));
        const size_t store_goal = 8 * std::min(((size_t)tmax_rem + 1) * n + 1, 16 * n + 1);
        const int U = (int)std::min<size_t>({(size_t)(n_utt - u0), (size_t)32,
                                             std::max<size_t>(plan->max_workspace / (fixed + up(((size_t)tmax_rem + 2) * 8) + store_goal), 1)});
        int tmax = 0;
        for (int u = u0; u < u0 + U; u++) tmax = std::max<int>(tmax, (int)(frame_off[u + 1] - frame_off[u]));
        const size_t per_off = up(((size_t)tmax + 2) * 8);
        const size_t head = up(((size_t)U + 1) * 8);
        const size_t room = plan->max_workspace > head + U * (fixed + per_off) ? plan->max_workspace - head - U * (fixed + per_off) : 0;
        // a layer never has more than n members: (tmax + 1) * n entries always suffice
        const size_t want = ((size_t)tmax + 1) * n + 1;
        const size_t cap = std::min(want, room / (8 * (size_t)U));
        SRX_REQUIRE(ctx, cap >= std::min(want, 2 * n + 1), "srx_viterbi_graph_dev: the workspace limit of %zu bytes is too small for %zu states",
                    plan->max_workspace, n);
        const size_t b_wdone = up(nw * 4) * U;
        const size_t b_cand = up(n * 8) * U, b_touch = up(n * 4) * U, b_pos = up(2 * n * 4) * U, b_fval = b_pos, b_pre = up(nwp * 4) * U,
                     b_wact = up(2 * nw * 4) * U, b_ctl = up(sizeof(FrCtl) * U), b_fso = per_off * U, b_st = up(cap * 4 * U);
        void *ws = nullptr;
        int rc = srx_workspace(ctx, 1, head + b_cand + b_touch + b_pos + b_fval + b_pre + b_wact + b_wdone + b_ctl + b_fso + 2 * b_st, &ws);
        if (rc != SRX_OK) return rc;
        char *w = reinterpret_cast<char *>(ws);
        FrArgs a;
        a.n = (int)n;
        a.tr_off = plan->tr_off; a.tr_tgt = plan->tr_tgt; a.tied = plan->tied; a.tr_logp = plan->tr_logp;
        a.n_wide = plan->n_wide; a.n_wpre = plan->n_wpre;
        a.wide_of = plan->wide_of; a.wide_src = plan->wide_src; a.wide_pre = plan->wide_pre; a.wpre_fan = plan->wpre_fan; a.wpre_tied = plan->wpre_tied;
        a.wtile_off = plan->wtile_off; a.wtile_lpmax = plan->wtile_lpmax; a.wtile_lpabs = plan->wtile_lpabs; a.n_cols = plan->n_cols;
        a.emis = emis_dev; a.st_t = stride_t; a.st_g = stride_g;
        a.beam = beam_width; a.final_mode = final_mode;
        a.foff = reinterpret_cast<const int64_t *>(w); w += head;
        // the per-utterance arrays are indexed [u][...] with the UNPADDED element counts: lay them out that way
        a.cand = reinterpret_cast<unsigned long long *>(w); w += b_cand;
        a.touched = reinterpret_cast<int32_t *>(w); w += b_touch;
        a.pos = reinterpret_cast<int32_t *>(w); w += b_pos;
        a.fval = reinterpret_cast<float *>(w); w += b_fval;
        a.pre = reinterpret_cast<float *>(w); w += b_pre;
        a.wact = reinterpret_cast<int32_t *>(w); w += b_wact;
        a.wdone = reinterpret_cast<int32_t *>(w); w += b_wdone;
        a.ctl = reinterpret_cast<FrCtl *>(w); w += b_ctl;
        a.fs_off = reinterpret_cast<int64_t *>(w); w += b_fso;
        a.st_state = reinterpret_cast<int32_t *>(w); w += b_st;
        a.st_back = reinterpret_cast<int32_t *>(w);
        a.cap = (int64_t)cap; a.tmax = tmax;
        a.path = path_dev; a.final_logp = final_logp_dev + u0; a.final_state = final_state_dev ? final_state_dev + u0 : nullptr;
        a.status = status_dev + u0;
        SRX_HIP(ctx, hipMemcpyAsync((void *)a.foff, frame_off + u0, ((size_t)U + 1) * 8, hipMemcpyHostToDevice, ctx->stream));
        SRX_HIP(ctx, hipMemsetAsync(a.cand, 0, (size_t)U * n * 8, ctx->stream));
        SRX_HIP(ctx, hipMemsetAsync(a.wact, 0, (size_t)U * 2 * nw * 4, ctx->stream));
        SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));     // frame_off is the caller's host array
        const unsigned B = (unsigned)std::min<size_t>(128, std::max<size_t>(1, n / 2048));
        const unsigned Bw = (unsigned)std::min<size_t>(256, std::max<size_t>(1, nwp / 1024));
        {
            SrxProf prof_(ctx, "k_viterbi_frontier");
            hipLaunchKernelGGL(k_fr_init, dim3(U), dim3(64), 0, ctx->stream, a);
            for (int t = 0; t < tmax; t++) {
                hipLaunchKernelGGL(k_fr_push, dim3(B, U), dim3(FR_THREADS), 0, ctx->stream, a, t);
                if (plan->n_wide > 0) {
                    hipLaunchKernelGGL(k_fr_chain, dim3(plan->n_wide, U), dim3(64), 0, ctx->stream, a, t);
                    hipLaunchKernelGGL(k_fr_wpush, dim3(Bw, U), dim3(FR_THREADS), 0, ctx->stream, a, t);
                }
                hipLaunchKernelGGL(k_fr_compact, dim3(B, U), dim3(FR_THREADS), 0, ctx->stream, a, t);
                hipLaunchKernelGGL(k_fr_advance, dim3(U), dim3(64), 0, ctx->stream, a, t);
            }
            hipLaunchKernelGGL(k_fr_final, dim3(U), dim3(FR_THREADS), 0, ctx->stream, a);
        }
        SRX_LAUNCH_CHECK(ctx);
        // the next group of utterances reuses the workspace
        if (u0 + U < n_utt) SRX_HIP(ctx, hipStreamSynchronize(ctx->stream));
        u0 += U;
    }
    return SRX_OK;
}
