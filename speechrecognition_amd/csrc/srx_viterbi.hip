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
#include <cstdlib>
#include <mutex>

#include "srx_common.h"

#include <type_traits>

// Emission prefetch depth in frames.  The frame loop is a dependent chain of about 0.06 us per step and an emission
// load takes 1-2 us to come back (64 lanes, 64 different rows), so the loads in flight have to cover 20-30 steps.
#ifndef SRX_VIT_PF_VEC
#define SRX_VIT_PF_VEC 32
#endif
#ifndef SRX_VIT_PF
#define SRX_VIT_PF 16
#endif

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
// 16-frame back-pointer groups staged in LDS per backtrace chunk.  OCC (waves per SIMD the kernel is compiled for): with
// one wave per utterance and fewer utterances than wave slots the kernel is one dependent chain per SIMD and wants every
// register and a deep staging buffer (OCC = 1: ~200 VGPRs, 16 KiB of LDS -- two waves per SIMD at most); with tens of
// thousands of utterances it is a throughput kernel and four resident waves per SIMD hide each other's latencies (OCC = 4:
// 128 VGPRs and a few spilled dwords, 4 KiB of LDS): 4.22 -> 2.95 ms for 100,000 utterances / 49.9 M frames.
template <int OCC>
struct VitCfg {
    static constexpr int BT_CH = OCC > 1 ? 8 : 32;
};

// One utterance with SPL states per lane.  The LDS arrays belong to the kernel (sized for the batch's widest utterance);
// sl_alloc = 64 * that width is the row stride the back-pointer workspace was laid out with.
// Returns true (no-beam bodies only) when the utterance has to be run again through the BEAM body: see `mx` below.
template <int SPL, bool BEAM, int LAY, int PF, int BT_CH>
__device__ __forceinline__ bool viterbi_utt(const int u, const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                            const int32_t *__restrict__ tied, const int32_t *__restrict__ ntr,
                                            const float *__restrict__ trlogp, const float *__restrict__ emis, int64_t st_t, int64_t st_g,
                                            const int64_t *__restrict__ ebase, float beam, uint32_t *__restrict__ bpws, int32_t *__restrict__ path,
                                            float *__restrict__ final_logp, int32_t *__restrict__ status, int *dmin, int *s_ntr,
                                            uint32_t *s_bp, const int sl_alloc) {
    // LAY: 0 = strided emission matrix, 1 = VEC (see above), 2 = compact per-utterance blocks [T][S] at ebase[u] (a
    // frame's emissions are one contiguous row; the block has fewer than 2^31 entries, so offsets are 32-bit)
    constexpr bool VEC = LAY == 1, COMPACT = LAY == 2;
    const int lane = threadIdx.x;
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u];
    const int S = soff[u + 1] - s0;
    constexpr int SL = 64 * SPL;
    uint32_t *bpw_base = bpws + ((size_t)(f0 >> 4) + (size_t)u) * sl_alloc;   // rows of this utterance's own SL inside its share

    // minimum number of steps from state 0 (static reachability of the left-to-right topology).  The recurrence is
    // serial in s, so it must not wait on memory: the transition counts come into LDS with one coalesced load first,
    // and the two running values stay in registers (the loop's only dependent chain is integer arithmetic).
#pragma unroll
    for (int q = 0; q < SPL; q++) {
        const int s = q * 64 + lane;
        s_ntr[s] = s < S ? ntr[s0 + s] : 0;
    }
    __syncthreads();
    if (lane == 0) {
        dmin[0] = NEVER;  // s = -2
        dmin[1] = NEVER;  // s = -1
        int d1 = NEVER, d2 = NEVER;   // dmin of states s-1 and s-2
        for (int s = 0; s < S; s++) {
            int d = s == 0 ? 0 : NEVER;
            if (s >= 1 && d1 != NEVER) d = d1 + 1;
            if (s >= 2 && s_ntr[s - 2] == 3 && d2 != NEVER && d2 + 1 < d) d = d2 + 1;
            dmin[s + 2] = d;
            d2 = d1;
            d1 = d;
        }
        for (int s = S; s < SL; s++) dmin[s + 2] = NEVER;
    }
    __syncthreads();

#ifdef SRX_VIT_TIMING
    const long long tk0 = wall_clock64();
#endif
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
            col[q] = COMPACT ? s : tied[s0 + s];   // compact per-utterance block: the row is the state itself
            a0[q] = trlogp[(size_t)(s0 + s) * 3];
            tv0[q] = dmin[s + 2];
            if (s >= 1 && s_ntr[s - 1] >= 2) {
                a1[q] = trlogp[(size_t)(s0 + s - 1) * 3 + 1];
                tv1[q] = dmin[s + 1];
            }
            if (s >= 2 && s_ntr[s - 2] == 3) {
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
        has1[q] = (s < S && s >= 1) ? s_ntr[s - 1] >= 2 : false;
        has2[q] = (s < S && s >= 2) ? s_ntr[s - 2] == 3 : false;
    }

    // dense matrix: frame t of the batch, column tied[s]; compact: the utterance's own block
    const float *erow = COMPACT ? emis + ebase[u] : emis + f0 * st_t;
    int64_t eoff[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) eoff[q] = COMPACT ? (int64_t)col[q] : (int64_t)col[q] * st_g;
    uint32_t colb[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) colb[q] = (uint32_t)col[q] * 4u;
    auto emission = [&](const int t, const int q) __attribute__((always_inline)) -> float {
        // compact block: the row's address is wave-uniform (scalar arithmetic) and the lane adds a 32-bit byte offset, which
        // is the load's scalar-base + vector-offset form -- no vector instruction per frame for the address
        // compact block: the row's offset is wave-uniform (scalar arithmetic) and is added to a per-lane base that never changes:
        // one 64-bit vector add per load instead of an index add and an address computation
        if (COMPACT) return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(erow + (size_t)((uint32_t)t * (uint32_t)S)) + colb[q]);
        return erow[(int64_t)t * st_t + eoff[q]];
    };
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
            for (int q = 0; q < SPL; q++) er[p][q] = T > 0 ? emission(tt, q) : 0.0f;
        }
    }

    uint32_t bw[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) bw[q] = 0;
    float cur_best = -1e10f;
    // The reference applies `logp < current_best_logp - beam_width` even at its default width of 1e10
    // (src/hmm.cpp:165): a source at -inf (zero-weight component, underflowed mixture) or more than 1e10 below the
    // frame's best is skipped, and an utterance whose every candidate dies ends in NO PATH.  The no-beam bodies leave
    // the test out and record the largest |delta| any state ever held instead: below 2.5e9 no two values of a frame
    // are 1e10 apart (and none is below the -1e10 floor of current_best_logp), so the test could not have fired and
    // the result is the reference's.  Otherwise the utterance is run again through the BEAM body with the width
    // given, which carries the test and the "is the source in the map" flags exactly.
    float mx[SPL], dprev[SPL];
#pragma unroll
    for (int q = 0; q < SPL; q++) mx[q] = dprev[q] = 0.0f;

    // Back-pointer words are indexed from the aligned start: frame t sits in word (t + a) >> 4 at bit pair (t + a) & 15.
    // Blocks start at multiples of PF from there and PF is a multiple of 16, so inside the unrolled block both the
    // shift and the "word is full" test are compile-time.  A block whose PF frames all exist runs without per-step
    // range tests (GUARD = false); the first block of an unaligned utterance and the last one take the guarded copy.
    static_assert(PF % 16 == 0, "PF must cover whole back-pointer words");
    // STATIC (alignment without a beam only): from frame t_static on every threshold test `t >= tv` has its final
    // value -- each state below S can be continued (ok), and a candidate exists exactly where the transition does --
    // so the step keeps two compares and four selects per state.  Lanes at or above S compute garbage that nothing
    // reads (values only move towards higher lanes).
    auto run_block = [&](auto mode_tag, const int tb) __attribute__((always_inline)) {
        constexpr int MODE = decltype(mode_tag)::value;   // 0 guarded, 1 unguarded, 2 unguarded + static flags
        constexpr bool GUARD = MODE == 0;
        constexpr bool STATIC = MODE == 2;
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
            } else if (!GUARD || t < T) {
                int tn = t + PF;
                if (tn >= T) tn = T - 1;
#pragma unroll
                for (int q = 0; q < SPL; q++) er[p][q] = emission(tn, q);
            }
            if (!GUARD || (t >= 0 && t < T)) {
                bool ok[SPL];
                float c0[SPL], c1[SPL], c2[SPL];
                if (STATIC) {
                } else if (BEAM) {
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
                if (STATIC) {
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        const bool t1 = has1[q] && (c1[q] > c0[q]);
                        const float b1 = t1 ? c1[q] : c0[q];
                        const bool t2 = has2[q] && (c2[q] > b1);
                        dl[q] = t2 ? c2[q] : b1;
                        const uint32_t bp = t2 ? 2u : (t1 ? 1u : 0u);
                        bw[q] |= bp << (2 * (p & 15));
                        // one v_max3 per two frames
                        if (p & 1) mx[q] = fmaxf(fmaxf(mx[q], fabsf(dprev[q])), fabsf(dl[q]));
                        else dprev[q] = dl[q];
                    }
                } else {
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
                    // selects only, no control flow: a candidate replaces the running best if it exists and either
                    // nothing was there yet or it is strictly greater (src/hmm.cpp:172-184)
                    const bool t1 = v1[q] & (!ok[q] | (c1[q] > c0[q]));
                    const float b1 = t1 ? c1[q] : c0[q];
                    const bool h1 = ok[q] | v1[q];
                    const bool t2 = v2[q] & (!h1 | (c2[q] > b1));
                    const float b2 = t2 ? c2[q] : b1;
                    const bool have = h1 | v2[q];
                    dl[q] = have ? b2 : dl[q];
                    const uint32_t bp = t2 ? 2u : (t1 ? 1u : 0u);
                    if (BEAM) {
                        act[q] = have;
                        nb = have ? fmaxf(nb, b2) : nb;
                    } else {
                        mx[q] = fmaxf(mx[q], fabsf(dl[q]));
                    }
                    bw[q] |= bp << (2 * (p & 15));
                }
                if (BEAM) cur_best = fmaxf(-1e10f, wave_max(nb));
                }
                if ((p & 15) == 15 || (GUARD && t == T - 1)) {
#pragma unroll
                    for (int q = 0; q < SPL; q++) {
                        bpw_base[(size_t)((t + a) >> 4) * SL + lane * SPL + q] = bw[q];
                        bw[q] = 0;
                    }
                }
            }
        }
    };
#ifdef SRX_VIT_TIMING
    const long long tk1 = wall_clock64();
#endif
    // frame from which all threshold tests have settled (uniform over the wave)
    int t_static = 0;
    if (!BEAM) {
#pragma unroll
        for (int q = 0; q < SPL; q++) {
            if (tv0[q] != NEVER && tv0[q] > t_static) t_static = tv0[q];
            if (tv1[q] != NEVER && tv1[q] > t_static) t_static = tv1[q];
            if (tv2[q] != NEVER && tv2[q] > t_static) t_static = tv2[q];
        }
        for (int o = 32; o > 0; o >>= 1) {
            const int other = __shfl_xor(t_static, o, 64);
            t_static = other > t_static ? other : t_static;
        }
        t_static = __builtin_amdgcn_readfirstlane(t_static);
    }
    for (int tb = -a; tb < T; tb += PF) {
        const bool whole = tb >= 0 && tb + PF <= T;
        if (BEAM) {
            if (whole) run_block(std::integral_constant<int, 1>{}, tb);
            else run_block(std::integral_constant<int, 0>{}, tb);
        } else {
            if (whole && tb >= t_static) run_block(std::integral_constant<int, 2>{}, tb);
            else run_block(std::integral_constant<int, 0>{}, tb);
        }
    }

#ifdef SRX_VIT_TIMING
    const long long tk2 = wall_clock64();
#endif
    if (!BEAM) {
        bool bad = false;
#pragma unroll
        // fmaxf drops a NaN operand, so a NaN never reaches mx; but a NaN delta is sticky (every comparison against it is false
        // and the state keeps its own NaN candidate), so the final deltas show whether one occurred
        for (int q = 0; q < SPL; q++) bad |= (lane * SPL + q < S) && (!(mx[q] < 2.5e9f) || dl[q] != dl[q]);
        if (__any(bad)) return true;
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
        return false;
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
    // The walk is one dependent chain per frame, so it must not wait on memory either: the back-pointer words come
    // into LDS a chunk of BT_CH groups (16 frames each) at a time with all loads in flight together, and the walk
    // reads the next group's registers from LDS while it steps through the current one.  Lane (t + a) & 63 of `pv`
    // collects the state of frame t; 64 frames go out in one store.  A whole group is walked with compile-time shifts.
    const int ng = ((T - 1 + a) >> 4) + 1;
    // One step: the 2-bit field of frame j in the current state's word is the decrement; it is also OR-ed, in place,
    // into `dec`, from which the lanes rebuild the 16 states of the group afterwards (all scalar work per frame).
    uint32_t dec = 0;
    auto bt_step = [&](const uint32_t (&w)[SPL], const int sh) __attribute__((always_inline)) {
        const int ow = (int)((uint32_t)b / SPL), q = (int)((uint32_t)b % SPL);   // b >= 0 on every valid path
        uint32_t word = (uint32_t)__builtin_amdgcn_readlane((int)w[0], ow);
#pragma unroll
        for (int qq = 1; qq < SPL; qq++) {
            uint32_t cand = (uint32_t)__builtin_amdgcn_readlane((int)w[qq], ow);
            if (q == qq) word = cand;
        }
        const uint32_t field = word & (3u << sh);
        dec |= field;
        b -= (int)(field >> sh);
    };
    auto bt_flush = [&](const int ta0) __attribute__((always_inline)) {   // frames with t + a in ta0 .. ta0 + 63
        const int t = ta0 - a + lane;
        if (t >= 0 && t < T) path[f0 + t] = pv;
    };
    for (int c_hi = ng; c_hi > 0; c_hi -= BT_CH) {
        const int c_lo = c_hi > BT_CH ? c_hi - BT_CH : 0;
#pragma unroll 8
        for (int g = c_lo; g < c_hi; g++) {
#pragma unroll
            for (int q = 0; q < SPL; q++) s_bp[(g - c_lo) * SL + q * 64 + lane] = bpw_base[(size_t)g * SL + q * 64 + lane];
        }
        __syncthreads();
        uint32_t wn[SPL];
#pragma unroll
        for (int q = 0; q < SPL; q++) wn[q] = s_bp[(c_hi - 1 - c_lo) * SL + lane * SPL + q];
        for (int kb = c_hi - 1; kb >= c_lo; kb--) {
            uint32_t w[SPL];
#pragma unroll
            for (int q = 0; q < SPL; q++) w[q] = wn[q];
            if (kb > c_lo) {
#pragma unroll
                for (int q = 0; q < SPL; q++) wn[q] = s_bp[(kb - 1 - c_lo) * SL + lane * SPL + q];
            }
            const int hi = (kb << 4) + 15 < T - 1 + a ? (kb << 4) + 15 : T - 1 + a;   // in units of t + a
            const int lo = (kb << 4) > a ? (kb << 4) : a;
            const int b_in = b;   // state of the group's last frame
            dec = 0;
            if (hi - lo == 15) {
#pragma unroll
                for (int j = 15; j >= 0; j--) bt_step(w, 2 * j);
            } else {
                for (int ta = hi; ta >= lo; ta--) bt_step(w, 2 * (ta & 15));
            }
            // state of frame j of the group = b_in - (sum of the decrements of frames j+1 .. 15); frames outside
            // lo..hi have a zero field and are never stored
            {
                const int j = lane & 15;
                const uint32_t above = (dec >> (2 * j)) >> 2;
                const int sum = __builtin_popcount(above & 0x55555555u) + 2 * __builtin_popcount(above & 0xAAAAAAAAu);
                if ((lane >> 4) == (kb & 3)) pv = b_in - sum;
            }
            if ((kb & 3) == 0) bt_flush(kb << 4);
        }
        __syncthreads();
    }
#ifdef SRX_VIT_TIMING
    if (lane == 0 && T >= 4) {
        const long long tk3 = wall_clock64();
        path[f0] = (int)(tk1 - tk0); path[f0 + 1] = (int)(tk2 - tk1); path[f0 + 2] = (int)(tk3 - tk2);
    }
#endif
    return false;
}

// One wavefront per utterance.  SPLMAX covers the batch's widest utterance; an utterance that fits fewer states per
// lane takes the narrower body (half or a quarter of the vector work per frame).
template <int SPLMAX, bool BEAM, int LAY, int PF, int OCC>
__global__ __launch_bounds__(64, OCC) void k_viterbi(int n_utt, const int64_t *__restrict__ foff, const int32_t *__restrict__ soff,
                                                const int32_t *__restrict__ tied, const int32_t *__restrict__ ntr,
                                                const float *__restrict__ trlogp, const float *__restrict__ emis, int64_t st_t, int64_t st_g,
                                                const int64_t *__restrict__ ebase, float beam, uint32_t *__restrict__ bpws, int32_t *__restrict__ path,
                                                float *__restrict__ final_logp, int32_t *__restrict__ status) {
    constexpr int SLMAX = 64 * SPLMAX;
    constexpr int BT_CH = SPLMAX >= 8 ? 16 : VitCfg<OCC>::BT_CH;   // 64 KB of LDS at most
    __shared__ int dmin[SLMAX + 2];
    __shared__ int s_ntr[SLMAX];
    __shared__ uint32_t s_bp[BT_CH * SLMAX];
    const int u = blockIdx.x;
    if (u >= n_utt) return;
    const int S = soff[u + 1] - soff[u];
    constexpr int PF1 = LAY == 1 ? SRX_VIT_PF_VEC : PF;   // the narrow bodies can afford the deep prefetch
    // S beyond what the launch was sized for (hmm->max_states smaller than the batch's widest utterance) would overrun
    // the LDS arrays: refuse that utterance instead
    if (S > SLMAX || S < 1) {
        if (threadIdx.x == 0) status[u] = 2;
        return;
    }
    bool redo;
    if (SPLMAX >= 2 && S <= 64)
        redo = viterbi_utt<1, BEAM, LAY, PF1, BT_CH>(u, foff, soff, tied, ntr, trlogp, emis, st_t, st_g, ebase, beam, bpws, path, final_logp, status,
                                              dmin, s_ntr, s_bp, SLMAX);
    else if (SPLMAX >= 4 && S <= 128)
        redo = viterbi_utt<2, BEAM, LAY, PF1, BT_CH>(u, foff, soff, tied, ntr, trlogp, emis, st_t, st_g, ebase, beam, bpws, path, final_logp, status,
                                              dmin, s_ntr, s_bp, SLMAX);
    else
        redo = viterbi_utt<SPLMAX, BEAM, LAY, PF, BT_CH>(u, foff, soff, tied, ntr, trlogp, emis, st_t, st_g, ebase, beam, bpws, path, final_logp,
                                                  status, dmin, s_ntr, s_bp, SLMAX);
    if (!BEAM && redo) {   // wave-uniform; rare (non-finite or absurdly large scores): the reference's pruning test decides
        __syncthreads();
        viterbi_utt<SPLMAX, true, LAY, PF, BT_CH>(u, foff, soff, tied, ntr, trlogp, emis, st_t, st_g, ebase, 1e10f, bpws, path, final_logp, status,
                                           dmin, s_ntr, s_bp, SLMAX);
    }
}

// ------------------------------------------------------------------------------------------
// hmm::train_transition_probabilities (src/hmm.cpp:229-243).  A training path is monotone, so each
// state's frames form one run; every run but the last re-estimates its state's advance probability.
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_train_transitions(int n_utt, const int64_t *__restrict__ foff,
                                                           const int32_t *__restrict__ soff, const int32_t *__restrict__ ntr,
                                                           float *__restrict__ trlogp, const int32_t *__restrict__ path,
                                                           const int32_t *__restrict__ status, int max_states) {
    const int u = blockIdx.x;
    if (u >= n_utt || status[u] != 0) return;
    const int64_t f0 = foff[u];
    const int T = (int)(foff[u + 1] - f0);
    const int s0 = soff[u];
    const int S = soff[u + 1] - s0;
    if (S > max_states) return;       // wider than the batch said it was: never index past the counters
    extern __shared__ int cnt[];      // [max_states]
    for (int s = threadIdx.x; s < S; s += blockDim.x) cnt[s] = 0;
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

static int64_t vit_occ_min() {
    static const int64_t v = std::getenv("SRX_VIT_OCC_MIN_UTT") ? std::atoll(std::getenv("SRX_VIT_OCC_MIN_UTT")) : 8192;
    return v;
}

static size_t viterbi_words(const srx_hmm_batch *hmm, int spl) {
    return ((size_t)(hmm->total_frames >> 4) + (size_t)hmm->n_utt + 1) * 64 * spl;
}

// `bp_words` != nullptr: the caller brings the back-pointer words (at least `bp_capacity` of them) and the call allocates
// nothing -- no workspace growth and no profiling events (srx_viterbi_batch_noalloc_dev below).
static int viterbi_launch(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t, int64_t stride_g,
                          const int64_t *ebase_dev, float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev,
                          uint32_t *bp_words = nullptr, size_t bp_capacity = 0) {
    SRX_REQUIRE(ctx, hmm && hmm->n_utt >= 0, "srx_viterbi_batch_dev: null or negative batch");
    if (hmm->n_utt == 0) return SRX_OK;   // an empty corpus: the reference's `for i < ut.size()` does nothing (src/hmm.cpp:258)
    SRX_REQUIRE(ctx, emis_dev && path_dev && final_logp_dev && status_dev, "srx_viterbi_batch_dev: null argument");
    SRX_REQUIRE(ctx, hmm->frame_off_dev && hmm->state_off_dev && hmm->tied_dev && hmm->ntr_dev && hmm->trlogp_dev,
                "srx_viterbi_batch_dev: null hmm field");
    SRX_REQUIRE(ctx, hmm->max_states >= 1, "srx_viterbi_batch_dev: max_states %d", hmm->max_states);
    const int64_t total_frames = hmm->total_frames;
    SRX_REQUIRE(ctx, total_frames >= 0, "srx_viterbi_batch_dev: negative total_frames");
    // 1, 2, 4 or 8 states per lane: up to SRX_VITERBI_MAX_STATES = 512 states per utterance.  A batch may hold wider
    // utterances (max_states says so): each of those gets status 2 and nothing else, and the caller takes it through
    // srx_viterbi_graph_dev, which has no limit (hmm::viterbi of the host layer does)
    const int spl = hmm->max_states <= 64 ? 1 : (hmm->max_states <= 128 ? 2 : (hmm->max_states <= 256 ? 4 : 8));
    const size_t words = viterbi_words(hmm, spl);
    void *ws = bp_words;
    if (bp_words) {
        SRX_REQUIRE(ctx, words <= bp_capacity,
                    "srx_viterbi_batch_noalloc_dev: the batch needs %zu back-pointer words (total_frames / 16 + n_utt + 1 rows of %d), "
                    "the fixed scratch holds %zu; use srx_viterbi_batch_dev", words, 64 * spl, bp_capacity);
    } else {
        int rc = srx_workspace(ctx, 1, words * sizeof(uint32_t), &ws);
        if (rc != SRX_OK) return rc;
    }
    const bool beam = beam_width < 1e10f;
    dim3 grid(hmm->n_utt), block(64);
    // several times more utterances than the latency build could keep resident (two waves per SIMD): the throughput build
    // (VitCfg).  Measured cross-over: equal at 3,000-6,000 utterances, -6 % at 10,000, -30 % at 100,000.
    const int64_t occ_min = vit_occ_min();
    const bool many = hmm->n_utt >= occ_min;
    const bool vec = !ebase_dev && stride_t == 1 && stride_g % 4 == 0 && stride_g >= total_frames &&
                     (reinterpret_cast<uintptr_t>(emis_dev) & 15) == 0;
#define SRX_VIT(SPLV, BEAMV)                                                                                              \
    do { if (vec) SRX_VIT2(SPLV, BEAMV, 1); else if (ebase_dev) SRX_VIT2(SPLV, BEAMV, 2); else SRX_VIT2(SPLV, BEAMV, 0); } while (0)
#define SRX_VIT2(SPLV, BEAMV, LAYV)                                                                                       \
    do { if (many) SRX_VIT3(SPLV, BEAMV, LAYV, 4); else SRX_VIT3(SPLV, BEAMV, LAYV, 1); } while (0)
#define SRX_VIT3(SPLV, BEAMV, LAYV, OCCV)                                                                                 \
    hipLaunchKernelGGL((k_viterbi<SPLV, BEAMV, LAYV, (SPLV >= 8 ? 16 : LAYV == 1 ? (SPLV <= 2 ? SRX_VIT_PF_VEC : 16) : SRX_VIT_PF), OCCV>), grid, block, 0, ctx->stream, hmm->n_utt, hmm->frame_off_dev,      \
                       hmm->state_off_dev, hmm->tied_dev, hmm->ntr_dev, hmm->trlogp_dev, emis_dev, stride_t, stride_g, ebase_dev, beam_width, \
                       (uint32_t *)ws, path_dev, final_logp_dev, status_dev)
    auto launch = [&]() {
        if (spl == 1) { if (beam) SRX_VIT(1, true); else SRX_VIT(1, false); }
        else if (spl == 2) { if (beam) SRX_VIT(2, true); else SRX_VIT(2, false); }
        else if (spl == 4) { if (beam) SRX_VIT(4, true); else SRX_VIT(4, false); }
        else {
            // eight states per lane: the rare wide utterance.  Only the pruning body (it is the general one: a beam of 1e10
            // prunes nothing) and only the one-wave-per-SIMD build are compiled for it -- the body needs the whole register
            // file and still spills on the strided layouts
            if (vec) SRX_VIT3(8, true, 1, 1); else if (ebase_dev) SRX_VIT3(8, true, 2, 1); else SRX_VIT3(8, true, 0, 1);
        }
    };
    if (bp_words) {
        launch();   // no event pair: creating one allocates
    } else {
        SrxProf prof_(ctx, "k_viterbi");
        launch();
    }
#undef SRX_VIT
#undef SRX_VIT2
#undef SRX_VIT3
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}

extern "C" int srx_viterbi_batch_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t,
                                     int64_t stride_g, float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev) {
    SRX_ENTER(ctx, "srx_viterbi_batch_dev");
    return viterbi_launch(ctx, hmm, emis_dev, stride_t, stride_g, nullptr, beam_width, path_dev, final_logp_dev, status_dev);
}

// ------------------------------------------------------------------------------------------
// srx_viterbi_batch_noalloc_dev: srx_viterbi_batch_dev for callers that cannot allocate (HIP graph capture, serving
// loops, harnesses that forbid hipMalloc).  The back-pointer words live in a fixed array of this code object instead of
// the context's grow-only workspace.  Every GPU the library runs on holds its own copy of the array, at its own address,
// until the process ends; `scratch` below is only the host-side handle of that symbol, never an address a kernel can
// use: the per-device address comes from hipGetSymbolAddress (srx_viterbi_noalloc_setup, called by srx_ctx_create with
// the context's device current) and that pointer is what the kernel is given.  The kernel writes every word it later
// reads, so nothing has to be cleared between calls; calls that overlap on one GPU share the array and race.
// ------------------------------------------------------------------------------------------
static constexpr size_t SRX_VIT_SCRATCH_WORDS = (size_t)SRX_VITERBI_NOALLOC_MAX_WORDS;
static __device__ __align__(256) uint32_t scratch[SRX_VIT_SCRATCH_WORDS];
static uint32_t *g_scratch_ptr[SRX_MAX_DEVICES] = {nullptr};
static hipError_t g_scratch_err[SRX_MAX_DEVICES] = {hipSuccess};
static std::mutex g_scratch_mu;

// Once per device, with that device current (srx_ctx_create).  A failure is kept and reported by the entry point; the
// rest of the library does not depend on the array.
void srx_viterbi_noalloc_setup(int device) {
    if (device < 0 || device >= SRX_MAX_DEVICES) return;
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[device]) return;
    void *p = nullptr;
    const hipError_t e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    g_scratch_err[device] = e;
    if (e == hipSuccess) g_scratch_ptr[device] = reinterpret_cast<uint32_t *>(p);
    (void)vit_occ_min();   // reads the environment once, here, not inside the entry point
}

extern "C" int srx_viterbi_batch_noalloc_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t,
                                             int64_t stride_g, float beam_width, int32_t *path_dev, float *final_logp_dev,
                                             int32_t *status_dev) {
    SRX_ENTER(ctx, "srx_viterbi_batch_noalloc_dev");
    uint32_t *ptr = g_scratch_ptr[ctx->device];
    if (!ptr)
        return srx_fail(ctx, SRX_ERR_DEVICE, "srx_viterbi_batch_noalloc_dev: the scratch array has no address on device %d (%s)",
                        ctx->device, hipGetErrorString(g_scratch_err[ctx->device]));
    return viterbi_launch(ctx, hmm, emis_dev, stride_t, stride_g, nullptr, beam_width, path_dev, final_logp_dev, status_dev, ptr,
                          SRX_VIT_SCRATCH_WORDS);
}

int srx_align_plan_n_utt(const srx_align_plan *plan);   // srx_gmm_f16_align.hip

extern "C" int srx_viterbi_align_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *emis_dev,
                                     float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev) {
    SRX_ENTER(ctx, "srx_viterbi_align_dev");
    SRX_REQUIRE(ctx, plan && hmm, "srx_viterbi_align_dev: null plan or batch");
    SRX_REQUIRE(ctx, srx_align_plan_n_utt(plan) == hmm->n_utt, "srx_viterbi_align_dev: plan is for %d utterances, batch has %d",
                srx_align_plan_n_utt(plan), hmm->n_utt);
    return viterbi_launch(ctx, hmm, emis_dev, 1, 0, srx_align_plan_emission_offsets_dev(plan), beam_width, path_dev, final_logp_dev,
                          status_dev);
}

extern "C" int srx_train_transitions_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *path_dev,
                                         const int32_t *status_dev) {
    SRX_ENTER(ctx, "srx_train_transitions_dev");
    SRX_REQUIRE(ctx, hmm && hmm->n_utt >= 0, "srx_train_transitions_dev: null or negative batch");
    if (hmm->n_utt == 0) return SRX_OK;
    SRX_REQUIRE(ctx, path_dev && status_dev, "srx_train_transitions_dev: null argument");
    // one LDS counter per state of the widest utterance (64 KB without opting in to more: 16,384 states)
    SRX_REQUIRE(ctx, hmm->max_states >= 1 && hmm->max_states <= 16384, "srx_train_transitions_dev: max_states %d outside 1..16384", hmm->max_states);
    {
        SrxProf prof_(ctx, "k_train_transitions");
        hipLaunchKernelGGL(k_train_transitions, dim3(hmm->n_utt), dim3(256), (size_t)hmm->max_states * sizeof(int), ctx->stream, hmm->n_utt, hmm->frame_off_dev,
                           hmm->state_off_dev, hmm->ntr_dev, hmm->trlogp_dev, path_dev, status_dev, hmm->max_states);
    }
    SRX_LAUNCH_CHECK(ctx);
    return SRX_OK;
}
