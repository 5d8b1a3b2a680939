// Batched Viterbi forced alignment on gfx950.
//
// Replaces hmm::viterbi (src/hmm.cpp:150-224) on the left-to-right training HMMs built by
// hmm::hmm(vector<phone>, acoustic_model*) (src/hmm.cpp:56-88), and
// hmm::train_transition_probabilities (src/hmm.cpp:229-243) with hmm_state::set_adv_prob /
// get_adv_prob (src/hmm.cpp:26-35).
//
// The reference is not textbook Viterbi: `logp` is declared outside the transition loop and
// accumulated inside it (src/hmm.cpp:164-169), so the candidate for a source's k-th transition
// carries the transition log-probs and emission scores of transitions 0..k-1 too.  With
// transitions stored as (self, s+1, s+2) the three candidates for target s are
//     c0[s] = (delta[s]   + a[s][0])   + e[s]
//     c1[s] = (c0[s-1]    + a[s-1][1]) + e[s]
//     c2[s] = (c1[s-1]    + a[s-2][2]) + e[s]        (only if state s-2 has 3 transitions)
// -- two dependent neighbour exchanges per frame.  All adds are plain f32 in that order
// (this file is compiled with -ffp-contract=off), so delta is bit-identical to the reference's.
//
// One wavefront per utterance, SPL consecutive states per lane, delta and the transition constants
// in registers, neighbour exchange by DPP wave_shr:1, emission scores prefetched PF frames ahead,
// 2-bit back-pointers packed 16 frames to a word, device-side backtrace with the current state in
// an SGPR.
#include "srx_common.h"

static constexpr int PF = 8;  // emission prefetch depth (frames)

__device__ __forceinline__ float shr1(float v) {
    // lane i receives lane i-1's value; lane 0 receives 0 (never used: its thresholds are never met)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ int shr1i(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x138, 0xf, 0xf, true); }

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

static constexpr int NEVER = 0x7fffffff;

// VEC: state-major dense emissions with 16-byte aligned rows (stride_t == 1, stride_g % 4 == 0): a lane fetches four
// consecutive frames of its state's row per load -- a quarter of the load instructions and of the cache-line lookups
// (64 lanes read 64 different rows).  The frame loop then starts at the aligned frame at or before the utterance's first.
template <int SPL, bool BEAM, bool VEC>
__global__ __launch_bounds__(64) void k_viterbi(int n_utt, const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                const int32_t *__restrict__ tied, const int32_t *__restrict__ ntr,
                                                const float *__restrict__ trlogp, const float *__restrict__ emis, int64_t st_t, int64_t st_g,
                                                const int64_t *__restrict__ ebase, float beam, uint32_t *__restrict__ bpws, int32_t *__restrict__ path,
                                                float *__restrict__ final_logp, int32_t *__restrict__ status) {
    const int u = blockIdx.x;
    if (u >= n_utt) return;
    const int lane = threadIdx.x;
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u];
    const int S = soff[u + 1] - s0;
    constexpr int SL = 64 * SPL;
    uint32_t *bpw_base = bpws + ((size_t)(f0 >> 4) + (size_t)u) * SL;

    // minimum number of steps from state 0 (static reachability of the left-to-right topology)
    __shared__ int dmin[SL + 2];
    if (lane == 0) {
        dmin[0] = NEVER;  // s = -2
        dmin[1] = NEVER;  // s = -1
        for (int s = 0; s < S; s++) {
            int d = s == 0 ? 0 : NEVER;
            if (s >= 1 && dmin[s + 1] != NEVER) d = dmin[s + 1] + 1;
            if (s >= 2 && ntr[s0 + s - 2] == 3 && dmin[s] != NEVER && dmin[s] + 1 < d) d = dmin[s] + 1;
            dmin[s + 2] = d;
        }
        for (int s = S; s < SL; s++) dmin[s + 2] = NEVER;
    }
    __syncthreads();

    float dl[SPL], a0[SPL], a1[SPL], a2[SPL];
    int tv0[SPL], tv1[SPL], tv2[SPL], col[SPL];
    bool act[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) {
        const int s = lane * SPL + q;
        dl[q] = 0.0f;
        a0[q] = a1[q] = a2[q] = 0.0f;
        tv0[q] = tv1[q] = tv2[q] = NEVER;
        col[q] = 0;
        act[q] = (s == 0);
        if (s < S) {
            col[q] = ebase ? s : tied[s0 + s];   // compact per-utterance block: the row is the state itself
            a0[q] = trlogp[(size_t)(s0 + s) * 3];
            tv0[q] = dmin[s + 2];
            if (s >= 1 && ntr[s0 + s - 1] >= 2) {
                a1[q] = trlogp[(size_t)(s0 + s - 1) * 3 + 1];
                tv1[q] = dmin[s + 1];
            }
            if (s >= 2 && ntr[s0 + s - 2] == 3) {
                a2[q] = trlogp[(size_t)(s0 + s - 2) * 3 + 2];
                tv2[q] = dmin[s];
            }
        }
    }
    // `has2[q]`: a skip transition s-2 -> s exists (BEAM mode uses flags instead of thresholds)
    bool has1[SPL], has2[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) {
        const int s = lane * SPL + q;
        has1[q] = (s < S && s >= 1) ? ntr[s0 + s - 1] >= 2 : false;
        has2[q] = (s < S && s >= 2) ? ntr[s0 + s - 2] == 3 : false;
    }

    // dense matrix: frame t of the batch, column tied[s]; compact (alignment-mode scoring): utterance block [T][S] at
    // ebase[u] -- a frame's emissions are one contiguous row, one or two cache lines per wave load
    if (ebase) {
        st_t = S;
        st_g = 1;
    }
    const float *erow = ebase ? emis + ebase[u] : emis + f0 * st_t;
    int64_t eoff[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) eoff[q] = (int64_t)col[q] * st_g;
    float er[PF][SPL];
    const int a = VEC ? (int)(f0 & 3) : 0;                 // frames between the aligned start and the utterance's first
    const int tlast4 = VEC ? (int)(((f0 + T - 1) & ~(int64_t)3) - f0) : 0;   // aligned group holding the last frame
    if (VEC) {
#pragma unroll
        for (int p = 0; p < PF; p += 4) {
            int tt = -a + p;
            if (tt > tlast4) tt = tlast4;
#pragma unroll
            for (int q = 0; q < SPL; q++) {
                const float4 v = T > 0 ? *reinterpret_cast<const float4 *>(erow + tt + eoff[q]) : make_float4(0, 0, 0, 0);
                er[p][q] = v.x; er[p + 1][q] = v.y; er[p + 2][q] = v.z; er[p + 3][q] = v.w;
            }
        }
    } else {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int tt = p < T ? p : (T > 0 ? T - 1 : 0);
#pragma unroll
            for (int q = 0; q < SPL; q++) er[p][q] = T > 0 ? erow[(int64_t)tt * st_t + eoff[q]] : 0.0f;
        }
    }

    uint32_t bw[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) bw[q] = 0;
    float cur_best = -1e10f;

    for (int tb = -a; tb < T; tb += PF) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int t = tb + p;
            float e[SPL];
#pragma unroll
            for (int q = 0; q < SPL; q++) e[q] = er[p][q];
            // prefetch into the slot(s) just consumed: frame t + PF, or with VEC the four frames t - 3 + PF ..
            if (VEC) {
                if (p % 4 == 3) {
                    int tn = t - 3 + PF;
                    if (tn > tlast4) tn = tlast4;
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        const float4 v = *reinterpret_cast<const float4 *>(erow + tn + eoff[q]);
                        er[p - 3][q] = v.x; er[p - 2][q] = v.y; er[p - 1][q] = v.z; er[p][q] = v.w;
                    }
                }
            } else if (t < T) {
                int tn = t + PF;
                if (tn >= T) tn = T - 1;
#pragma unroll
                for (int q = 0; q < SPL; q++) er[p][q] = erow[(int64_t)tn * st_t + eoff[q]];
            }
            if (t >= 0 && t < T) {
                bool ok[SPL];
                float c0[SPL], c1[SPL], c2[SPL];
                if (BEAM) {
                    const float thresh = cur_best - beam;
#pragma unroll
                    for (int q = 0; q < SPL; q++) ok[q] = act[q] && !(dl[q] < thresh);
                } else {
#pragma unroll
                    for (int q = 0; q < SPL; q++) ok[q] = t >= tv0[q];
                }
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    float x = dl[q] + a0[q];
                    c0[q] = x + e[q];
                }
                const float c0p = shr1(c0[SPL - 1]);
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    float x = (q ? c0[q - 1] : c0p) + a1[q];
                    c1[q] = x + e[q];
                }
                const float c1p = shr1(c1[SPL - 1]);
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    float x = (q ? c1[q - 1] : c1p) + a2[q];
                    c2[q] = x + e[q];
                }
                bool v1[SPL], v2[SPL];
                if (BEAM) {
                    int okl = ok[SPL - 1] ? 1 : 0;
                    int okl2 = SPL >= 2 ? (ok[SPL >= 2 ? SPL - 2 : 0] ? 1 : 0) : 0;
                    const int okp1 = shr1i(okl);                         // ok of state (lane*SPL - 1)
                    const int okp2 = SPL >= 2 ? shr1i(okl2) : shr1i(okp1);  // ok of state (lane*SPL - 2)
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        bool o1 = q >= 1 ? ok[q - 1] : (okp1 != 0);
                        bool o2 = q >= 2 ? ok[q - 2] : (q == 1 ? (okp1 != 0) : (okp2 != 0));
                        v1[q] = has1[q] && o1;
                        v2[q] = has2[q] && o2;
                    }
                } else {
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        v1[q] = t >= tv1[q];
                        v2[q] = t >= tv2[q];
                    }
                }
                float nb = -__builtin_huge_valf();
#pragma unroll
                for (int q = 0; q < SPL; q++) {
                    bool have = ok[q];
                    float best = c0[q];
                    uint32_t bp = 0;
                    if (v1[q] && (!have || c1[q] > best)) { best = c1[q]; bp = 1; have = true; }
                    if (v2[q] && (!have || c2[q] > best)) { best = c2[q]; bp = 2; have = true; }
                    if (have) dl[q] = best;
                    if (BEAM) {
                        act[q] = have;
                        if (have) nb = fmaxf(nb, best);
                    }
                    bw[q] |= bp << (2 * (t & 15));
                }
                if (BEAM) cur_best = fmaxf(-1e10f, wave_max(nb));
                if ((t & 15) == 15 || t == T - 1) {
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        bpw_base[(size_t)(t >> 4) * SL + lane * SPL + q] = bw[q];
                        bw[q] = 0;
                    }
                }
            }
        }
    }

    // ---- end condition: the path must end in the last state (src/hmm.cpp:210-215) ----
    const int last = S - 1;
    const int owner = last / SPL, oq = last % SPL;
    bool reach;
    float fin = 0.0f;
    {
        float dsel = dl[0];
        bool asel = BEAM ? act[0] : (T >= 1 && (T - 1) >= tv0[0] - 1 && tv0[0] != NEVER);
#pragma unroll
        for (int q = 1; q < SPL; q++)
            if (oq == q) {
                dsel = dl[q];
                asel = BEAM ? act[q] : (T >= 1 && (T - 1) >= tv0[q] - 1 && tv0[q] != NEVER);
            }
        fin = __shfl(dsel, owner, 64);
        reach = __shfl(asel ? 1 : 0, owner, 64) != 0;
    }
    if (T < 1) reach = false;
    if (!reach) {
        if (lane == 0) status[u] = 1;
        return;
    }
    if (lane == 0) {
        status[u] = 0;
        final_logp[u] = fin;
    }
    // make this wave's back-pointer words visible to its own loads
    __threadfence_block();
    __syncthreads();

    // ---- backtrace (src/hmm.cpp:219-222): current state b is wave-uniform ----
    int b = last;
    int pv = 0;
    for (int kb = (T - 1) >> 4; kb >= 0; kb--) {
        uint32_t w[SPL];
#pragma unroll
        for (int q = 0; q < SPL; q++) w[q] = bpw_base[(size_t)kb * SL + lane * SPL + q];
        const int thi = (kb << 4) + 15 < T - 1 ? (kb << 4) + 15 : T - 1;
        for (int t = thi; t >= (kb << 4); t--) {
            if ((t & 63) == lane) pv = b;
            const int ow = b / SPL, q = b % SPL;
            uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)w[0], ow);
#pragma unroll
            for (int qq = 1; qq < SPL; qq++) {
                uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)w[qq], ow);
                if (q == qq) word = cand;
            }
            b -= (int)((word >> (2 * (t & 15))) & 3u);
            b = __builtin_amdgcn_readfirstlane(b);
            if ((t & 63) == 0) {
                // frames t .. t+63 (clipped) are complete
                if (t + lane < T) path[f0 + t + lane] = pv;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// hmm::train_transition_probabilities (src/hmm.cpp:229-243).  A training path is monotone, so each
// state's frames form one run; every run but the last re-estimates its state's advance probability.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_train_transitions(int n_utt, const int64_t *__restrict__ foff,
                                                           const int32_t *__restrict__ soff, const int32_t *__restrict__ ntr,
                                                           float *__restrict__ trlogp, const int32_t *__restrict__ path,
                                                           const int32_t *__restrict__ status) {
    const int u = blockIdx.x;
    if (u >= n_utt || status[u] != 0) return;
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u];
    const int S = soff[u + 1] - s0;
    __shared__ int cnt[256];
    for (int s = threadIdx.x; s < 256; s += blockDim.x) cnt[s] = 0;
    __syncthreads();
    for (int t = threadIdx.x; t < T; t += blockDim.x) atomicAdd(&cnt[path[f0 + t]], 1);
    __syncthreads();
    const int lastst = T > 0 ? path[f0 + T - 1] : -1;
    for (int s = threadIdx.x; s < S; s += blockDim.x) {
        const int c = cnt[s];
        if (c > 0 && s != lastst) {
            float *tr = trlogp + (size_t)(s0 + s) * 3;
            const float inv = 1.0f / (float)c;
            const float m = inv < 0.55f ? inv : 0.55f;       // min(MAX_ADV_PROB, 1.0f/count)
            const float adv = expf(tr[1]);                   // get_adv_prob
            const float x = 0.2f * m;                        // HMM_BLEND_FAC * min(...)
            const float y = (1 - 0.2f) * adv;
            const float p = x + y;
            const float avail = ntr[s0 + s] == 3 ? (1 - 0.3f) : 1.0f;  // 1 - SIL_SKIP_P
            const float q0 = (1 - p) * avail;
            const float q1 = p * avail;
            tr[0] = logf(q0);
            tr[1] = logf(q1);
        }
    }
}

static int viterbi_launch(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t, int64_t stride_g,
                          const int64_t *ebase_dev, float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev) {
    SRX_REQUIRE(nullptr, ctx, "srx_viterbi_batch_dev: null ctx");
    SRX_REQUIRE(ctx, hmm && emis_dev && path_dev && final_logp_dev && status_dev, "srx_viterbi_batch_dev: null argument");
    SRX_REQUIRE(ctx, hmm->frame_off_dev && hmm->state_off_dev && hmm->tied_dev && hmm->ntr_dev && hmm->trlogp_dev,
                "srx_viterbi_batch_dev: null hmm field");
    SRX_REQUIRE(ctx, hmm->n_utt >= 0 && hmm->max_states >= 1 && hmm->max_states <= 256,
                "srx_viterbi_batch_dev: max_states %d outside 1..256", hmm->max_states);
    if (hmm->n_utt == 0) return SRX_OK;
    const int64_t total_frames = hmm->total_frames;
    SRX_REQUIRE(ctx, total_frames >= 0, "srx_viterbi_batch_dev: negative total_frames");
    const int spl = hmm->max_states <= 64 ? 1 : (hmm->max_states <= 128 ? 2 : 4);
    const size_t words = ((size_t)(total_frames >> 4) + (size_t)hmm->n_utt + 1) * 64 * spl;
    void *ws = nullptr;
    int rc = srx_workspace(ctx, 1, words * sizeof(uint32_t), &ws);
    if (rc != SRX_OK) return rc;
    const bool beam = beam_width < 1e10f;
    dim3 grid(hmm->n_utt), block(64);
    const bool vec = !ebase_dev && stride_t == 1 && stride_g % 4 == 0 && stride_g >= total_frames &&
                     (reinterpret_cast<uintptr_t>(emis_dev) & 15) == 0;
#define SRX_VIT(SPLV, BEAMV)                                                                                              \
    do { if (vec) SRX_VIT2(SPLV, BEAMV, true); else SRX_VIT2(SPLV, BEAMV, false); } while (0)
#define SRX_VIT2(SPLV, BEAMV, VECV)                                                                                       \
    hipLaunchKernelGGL((k_viterbi<SPLV, BEAMV, VECV>), grid, block, 0, ctx->stream, hmm->n_utt, hmm->frame_off_dev,      \
                       hmm->state_off_dev, hmm->tied_dev, hmm->ntr_dev, hmm->trlogp_dev, emis_dev, stride_t, stride_g, ebase_dev, beam_width, \
                       (uint32_t *)ws, path_dev, final_logp_dev, status_dev)
    {
        SrxProf prof_(ctx, "k_viterbi");
        if (spl == 1) { if (beam) SRX_VIT(1, true); else SRX_VIT(1, false); }
        else if (spl == 2) { if (beam) SRX_VIT(2, true); else SRX_VIT(2, false); }
        else { if (beam) SRX_VIT(4, true); else SRX_VIT(4, false); }
    }
#undef SRX_VIT
#undef SRX_VIT2
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_viterbi_batch_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t,
                                     int64_t stride_g, float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev) {
    return viterbi_launch(ctx, hmm, emis_dev, stride_t, stride_g, nullptr, beam_width, path_dev, final_logp_dev, status_dev);
}

extern "C" int srx_viterbi_align_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *emis_dev,
                                     float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev) {
    SRX_REQUIRE(nullptr, ctx, "srx_viterbi_align_dev: null ctx");
    SRX_REQUIRE(ctx, plan, "srx_viterbi_align_dev: null plan");
    return viterbi_launch(ctx, hmm, emis_dev, 1, 0, srx_align_plan_emission_offsets_dev(plan), beam_width, path_dev, final_logp_dev,
                          status_dev);
}

extern "C" int srx_train_transitions_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *path_dev,
                                         const int32_t *status_dev) {
    SRX_REQUIRE(nullptr, ctx, "srx_train_transitions_dev: null ctx");
    SRX_REQUIRE(ctx, hmm && path_dev && status_dev, "srx_train_transitions_dev: null argument");
    SRX_REQUIRE(ctx, hmm->max_states >= 1 && hmm->max_states <= 256, "srx_train_transitions_dev: max_states outside 1..256");
    if (hmm->n_utt == 0) return SRX_OK;
    {
        SrxProf prof_(ctx, "k_train_transitions");
        hipLaunchKernelGGL(k_train_transitions, dim3(hmm->n_utt), dim3(256), 0, ctx->stream, hmm->n_utt, hmm->frame_off_dev,
                           hmm->state_off_dev, hmm->ntr_dev, hmm->trlogp_dev, path_dev, status_dev);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
