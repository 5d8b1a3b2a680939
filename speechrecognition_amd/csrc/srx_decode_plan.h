// The lexicon / arbitrary-transition-list HMM on the device, shared by the two sweeps of hmm::viterbi on it: the dense one
// (srx_decode.hip: every state every frame, one workgroup per utterance) and the frontier one (srx_decode_frontier.hip).
#pragma once
#include <vector>

#include "srx_common.h"

static constexpr int FANOUT_MIN = 4;    // dense sweep: sources with at least this many transitions get a prefix pass
static constexpr int FANIN_MIN = 17;    // dense sweep: targets with at least this many incoming edges are reduced by the whole workgroup
static constexpr int FR_FAN_MIN = 33;   // frontier sweep: sources with at least this many transitions run their chain in a kernel of its own
static constexpr int FR_TILE = 1024;    // ... in tiles of this many transitions

struct srx_decode_plan {
    srx_ctx *ctx = nullptr;
    int n_states = 0, n_tr = 0, n_fan = 0, n_big = 0, n_pre = 0;
    int32_t *tr_off = nullptr, *tr_tgt = nullptr, *tied = nullptr;
    float *tr_logp = nullptr;
    int32_t *in_off = nullptr, *in_src = nullptr, *in_k = nullptr, *in_pre = nullptr;   // incoming edges per target, sources descending
    int32_t *fan_src = nullptr, *fan_pre = nullptr;                                   // fan-out sources and their offset in `pre`
    int32_t *big_tgt = nullptr;                                                       // fan-in targets
    // frontier sweep: the wide sources (>= FR_FAN_MIN transitions)
    int n_wide = 0, n_wpre = 0;
    int32_t *wide_of = nullptr;       // [n_states] index among the wide sources or -1
    int32_t *wide_src = nullptr;      // [n_wide] state
    int32_t *wide_pre = nullptr;      // [n_wide] offset of the source's transitions in the flattened list
    int32_t *wpre_fan = nullptr;      // [n_wpre] wide source of every flattened transition
    int32_t *wpre_tied = nullptr;     // [n_wpre] emission column of its target
    // what lets the chain of a wide source stop early (srx_decode_frontier.hip): per tile of FR_TILE transitions, over the
    // transitions FROM the tile's first one to the end of the list
    int n_cols = 0;                   // emission columns in use: max tied + 1
    int32_t *wtile_off = nullptr;     // [n_wide + 1] first tile entry of every wide source
    float *wtile_lpmax = nullptr;     // [tiles] largest transition score
    double *wtile_lpabs = nullptr;    // [tiles] sum of their magnitudes
    std::vector<int32_t> h_tr_off, h_wide_src;   // host copies, for srx_decode_plan_set_transitions
    int sweep = 0;                    // SRX_DECODE_AUTO / DENSE / FRONTIER
    size_t max_workspace = (size_t)2 << 30;
};

int srx_viterbi_frontier(srx_ctx *ctx, const srx_decode_plan *plan, const float *emis_dev, int64_t stride_t, int64_t stride_g,
                         const int64_t *frame_off, int n_utt, float beam_width, int final_mode, int32_t *path_dev, float *final_logp_dev,
                         int32_t *final_state_dev, int32_t *status_dev);
