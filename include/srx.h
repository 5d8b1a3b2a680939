/*
 * srx.h -- C ABI of libsrx.so: the MI355X (gfx950) implementation of the acoustic hot path
 * of albinoBandicoot/speechRecognition (MFCC extraction, diagonal-GMM emission scoring,
 * Viterbi forced alignment, embedded-training statistics).
 *
 * The reference has no FFI/plugin layer: it is one C++ program.  The boundary replaced is
 * the C++ surface its headless main() calls.  Each entry point below
 * cites the reference function(s) it stands in for, paths relative to the reference tree.
 * The C++ classes in speechrecognition_amd/host/ (same names as the reference's) are thin
 * wrappers over this header; INTEGRATION.md shows the binding a maintainer would add.
 *
 * Conventions
 *   - every call returns SRX_OK (0) or a negative srx_status; srx_last_error() has the text;
 *   - no C++ exception and no HIP/torch type crosses this boundary: plain pointers + sizes;
 *   - names ending in _dev take DEVICE pointers and are asynchronous on the context's
 *     stream; all other calls take HOST pointers and are synchronous on return;
 *   - a context is used by one host thread at a time;
 *   - there is no CPU fallback: without a gfx950 device srx_ctx_create fails.
 */
#ifndef SRX_H
#define SRX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    SRX_OK = 0,
    SRX_ERR_INVALID = -1,   /* bad argument (shape, null pointer, unsupported size) */
    SRX_ERR_DEVICE = -2,    /* HIP runtime error */
    SRX_ERR_NOMEM = -3,     /* device or host allocation failed */
    SRX_ERR_IO = -4,        /* file could not be opened (reference: `throw 1`, src/mfcc.cpp:237,281) */
    SRX_ERR_UNSUPPORTED = -5
} srx_status;

typedef struct srx_ctx srx_ctx;

/* ---------------------------------------------------------------- context ---------- */
/* One per GPU.  `stream` is a hipStream_t to launch on (NULL: the context creates its own). */
int srx_ctx_create(int device, void *stream, srx_ctx **out);
void srx_ctx_destroy(srx_ctx *ctx);
const char *srx_last_error(const srx_ctx *ctx); /* ctx may be NULL: last creation error */
int srx_sync(srx_ctx *ctx);
/* The hipStream_t the context launches on (the one given to srx_ctx_create, or its own).  No counterpart in the reference
 * (which has no streams): lets a caller order its own work -- copies, RCCL collectives, torch ops -- against libsrx's
 * kernels on the device instead of calling srx_sync. */
void *srx_ctx_stream(const srx_ctx *ctx);
const char *srx_version(void);

/* Per-kernel timing with HIP events on the context's stream.  No counterpart in the reference (which has
 * only a commented-out clock() print, src/mfcc.cpp:260,268-270); bench.py's roofline figures come from it.
 * srx_profile_enable(ctx, 1) clears the records and starts bracketing every kernel launch of this
 * context with an event pair; srx_profile_query synchronises the stream and returns the summed
 * duration and launch count of the kernel with that name ("k_gmm_mfma", "k_mfcc_tile", "k_viterbi",
 * "k_features", "k_gmm_prep", "k_sumsq_scale", "k_em_logp", "k_em_accum1", "k_em_accum2", ...). */
int srx_profile_enable(srx_ctx *ctx, int on);
int srx_profile_query(srx_ctx *ctx, const char *kernel, double *total_ms, int64_t *launches);

/* Device-memory plumbing so that C/C++ callers need no HIP headers. */
int srx_malloc(srx_ctx *ctx, size_t bytes, void **dev);
int srx_free(srx_ctx *ctx, void *dev);
int srx_memcpy_h2d(srx_ctx *ctx, void *dev, const void *host, size_t bytes);
int srx_memcpy_d2h(srx_ctx *ctx, void *host, const void *dev, size_t bytes);
int srx_memset(srx_ctx *ctx, void *dev, int value, size_t bytes);

/* ---------------------------------------------------------------- filterbank -------- */
/* filter {center,width,height}: src/mfcc.hpp:50-60 */
typedef struct {
    float center, width, height;
} srx_filter;

/* filter::filter(c, w, dft_bins, sps)  src/mfcc.cpp:117-128: centre/width in Hz -> bins (ofMap), height
 * normalised over the filter's own bin range.  filter::get(bin)  src/mfcc.cpp:132-140.  Host-side. */
int srx_filter_init(float center_hz, float width_hz, int dft_bins, int sps, srx_filter *out);
float srx_filter_value(const srx_filter *f, int bin);

/* mel_filterbank(end, n, dft_bins, sps)  src/mfcc.cpp:190-201 (+ filter ctor :117-128,
 * ofMap of OpenFrameworks).  Host-side constants; writes n filters and bin_max
 * (src/mfcc.cpp:220-221). */
int srx_mel_filterbank(float end, int n, int dft_bins, int sps, srx_filter *out, int *bin_max);

/* ---------------------------------------------------------------- MFCC -------------- */
typedef struct {
    int window_len;     /* 320 in src/main.cpp:83 (400 for 25 ms) */
    int frame_shift;    /* 160 */
    int n_ceps;         /* NPARAMS = 13, src/mfcc.hpp:17 */
    int normalize;      /* 1: apply Clip::normalizeVolume(target_rms) first (src/main.cpp:79) */
    float target_rms;   /* 1.0f */
    int preemphasis;    /* 1: apply Clip::preemphasis(alpha) (src/mfcc.cpp:238) */
    float alpha;        /* 0.96f */
} srx_mfcc_cfg;

typedef struct srx_mfcc_plan srx_mfcc_plan;

/* Builds the device tables (window, twiddles, filterbank matrix, DCT matrix) for a
 * configuration + filterbank.  Replaces the per-frame recomputation in
 * hamming_window (src/utils.cpp:21-23), Clip::partial_dft (src/clip.cpp:174-194),
 * filter::get (src/mfcc.cpp:132-140) and dct (src/mfcc.cpp:205-214). */
int srx_mfcc_plan_create(srx_ctx *ctx, const srx_mfcc_cfg *cfg, const srx_filter *fb, int n_filters,
                         srx_mfcc_plan **out);
void srx_mfcc_plan_destroy(srx_mfcc_plan *plan);
int srx_mfcc_plan_bin_max(const srx_mfcc_plan *plan);
/* Which matrix-core formulation srx_mfcc_batch* runs (same reference code replaced, same 1e-4 bar):
 * SRX_MFCC_F16X3: operands split into two f16 parts, three products on the f16 matrix cores (half windows of 113..128,
 * 145..160, 193..208 or 241..256 samples: the reference's 320- and BASELINE's 400-sample windows, and 256 / 512);
 * SRX_MFCC_F32: f32 matrix cores, any window; SRX_MFCC_AUTO (default): F16X3 where it applies.  No counterpart in the
 * reference (a tuning knob). */
enum { SRX_MFCC_AUTO = 0, SRX_MFCC_F32 = 1, SRX_MFCC_F16X3 = 2,
       SRX_MFCC_FFT = 3 /* the half-window transform as ONE packed real FFT of N/2 points in LDS (N/2 = 8*5*5, 8*5*4,
                           8*4*4, 8*8*4: windows of 400, 320, 256, 512 samples), f32 vector arithmetic, no matrix cores.
                           Measured 1.7-1.9x slower than F16X3 (profiles/r03_mfcc_fft_vs_dense.json): never picked by AUTO */ };
int srx_mfcc_plan_set_path(srx_mfcc_plan *plan, int path);

/* Frame count of the do/while(window.slide()) loop, src/mfcc.cpp:261-271 + src/clip.cpp:108-112:
 * n = 1 + floor((L - wl - 1)/hop) for L > wl.  Fills frame_off[n_utt+1] (prefix sums) from
 * sample_off[n_utt+1].  Returns SRX_ERR_INVALID if an utterance has L <= wl (the reference
 * reads past its buffer there). */
int srx_mfcc_frame_offsets(const srx_mfcc_cfg *cfg, const int64_t *sample_off, int n_utt, int64_t *frame_off);

/* write_features(Clip, wl, hop, fb, fname) minus the file I/O (src/mfcc.cpp:235-275) for a
 * ragged batch: utterance u owns samples[sample_off[u] .. sample_off[u+1]) and produces
 * ceps[frame_off[u] .. frame_off[u+1])[n_ceps].  The samples are NOT modified (the
 * reference mutates them in place; normalisation and pre-emphasis are fused into the frame
 * load here -- use srx_normalize_volume_dev / srx_preemphasis_dev to reproduce the mutation).
 * `scratch_scale_dev`: n_utt floats of device scratch (per-utterance gain). */
int srx_mfcc_batch_dev(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev,
                       const int64_t *sample_off_dev, int n_utt, const int64_t *frame_off_dev,
                       int64_t total_frames, float *scratch_scale_dev, float *ceps_dev);
/* Host-pointer convenience form (allocates, copies, runs, copies back). */
int srx_mfcc_batch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples, const int64_t *sample_off,
                   int n_utt, float *ceps, int64_t *frame_off);

/* Clip::partial_dft  src/clip.cpp:174-194 of every frame of the batch, stand-alone: spec[frame][bin_max] (bin_max =
 * srx_mfcc_plan_bin_max) = the magnitudes sqrt(re^2 + im^2) / sqrt(window) over the first half window, including the
 * mirror branch for bins beyond N/2 (src/clip.cpp:177, 186-192).  The plan's normalisation / pre-emphasis settings apply
 * as in srx_mfcc_batch*.  (Inside srx_mfcc_batch these values never leave the chip.) */
int srx_partial_dft_batch_dev(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples_dev, const int64_t *sample_off_dev,
                              int n_utt, const int64_t *frame_off_dev, int64_t total_frames, float *scratch_scale_dev,
                              float *spec_dev);
int srx_partial_dft_batch(srx_ctx *ctx, const srx_mfcc_plan *plan, const float *samples, const int64_t *sample_off, int n_utt,
                          float *spec, int64_t *frame_off);

/* ---------------------------------------------------------------- the MFCC stages one by one --------
 * Inside srx_mfcc_batch* the stages below are fused and their intermediates never leave the chip.  The reference also has
 * each as a function of its own; these entry points are those functions, evaluated in the reference's operation order
 * (f64 product and add with the running sum rounded to f32 per add; sequential f32 accumulation of filter taps) with the
 * cosines from host-built f64 tables, so the results equal the CPU restatement bit for bit (log(x+1): to the last place
 * of the device's logf).  One thread per output: correctness-first forms for callers that work a frame at a time; the
 * batch entry points are the fast path.  `_dev` forms take device pointers and row strides in elements. */

/* Clip::partial_dft(out, len, stack) / Clip::dft  src/clip.cpp:166-194 for n_frames windows of N samples each:
 * y[n] = x[n] * window[n] for n < N/2 (Clip::operator[] const, src/clip.cpp:123-125; window == NULL: rectangular_window),
 * re/im accumulated over n < N/2 only (src/clip.cpp:181), out[k] = sqrt(re^2 + im^2) / sqrt(N) for k < len, the mirrored
 * write out[N-k-1] when len > N/2 (src/clip.cpp:186-192: later writes win), added to what out holds when `stack`.
 * len <= N.  Host form: frames[n_frames][N], out[n_frames][len]; device form: rows of at least N/2 samples (only the first
 * half of a window is read). */
int srx_partial_dft_frames_dev(srx_ctx *ctx, const float *frames_dev, int64_t ld, const float *window_dev, int N, int len,
                               int64_t n_frames, int stack, float *out_dev, int64_t ldo);
int srx_partial_dft_frames(srx_ctx *ctx, const float *frames, const float *window /*[N/2] or NULL*/, int N, int len,
                           int64_t n_frames, int stack, float *out);

/* filterbank::operator()(spectrum) = filter::operator() of every filter  src/mfcc.cpp:143-149, 161-167:
 * out[r][i] = sum over b = (int)(c_i - w_i) .. (int)(c_i + w_i) of filter::get(b) * spec[r][b], f32 in that order.
 * `fb` is a HOST array (n_filters <= 64).  A filter that reaches outside [0, n_bins) is refused (the reference reads
 * whatever lies beside its array). */
int srx_filterbank_apply_dev(srx_ctx *ctx, const srx_filter *fb, int n_filters, const float *spec_dev, int64_t ld, int n_bins,
                             int64_t n_rows, float *out_dev, int64_t ldo);
int srx_filterbank_apply(srx_ctx *ctx, const srx_filter *fb, int n_filters, const float *spec, int n_bins, int64_t n_rows,
                         float *out);

/* log(T *buf, int len)  src/utils.hpp:55-60 for T = float: buf[i] = logf(buf[i] + 1), in place. */
int srx_log1p_dev(srx_ctx *ctx, float *buf_dev, int64_t n);
int srx_log1p(srx_ctx *ctx, float *buf, int64_t n);

/* dct(filtered, len, nparams)  src/mfcc.cpp:205-214 for n_rows vectors: out[r][j] = sum_i in[r][i] * cos(j*i*pi/len),
 * j < nparams, the product and the add in f64, the running sum rounded to f32 after every add. */
int srx_dct_dev(srx_ctx *ctx, const float *in_dev, int64_t ld, int len, int nparams, int64_t n_rows, float *out_dev, int64_t ldo);
int srx_dct(srx_ctx *ctx, const float *in, int len, int nparams, int64_t n_rows, float *out);

/* featurevec::compute_deltas / compute_delta2s for ONE vector  src/mfcc.cpp:73-92: out[q] = (hi[q] - lo[q]) / (float)span,
 * hi / lo = the coefficients (or deltas) of the neighbours k and j, span = k - j (0, 1 or 2).  Host pointers.  The batch
 * form -- compute_deltas(vector) of src/mfcc.cpp:95-102 -- is srx_features_batch* with WL = -1. */
int srx_delta(srx_ctx *ctx, const float *hi, const float *lo, int span, float *out, int64_t n);

/* Clip::rmsAmplitude + Clip::normalizeVolume  src/clip.cpp:151-164 (in place, per utterance) */
int srx_normalize_volume_dev(srx_ctx *ctx, float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                             float target_rms, float *scratch_scale_dev);
/* Clip::operator*=(float)  src/clip.cpp:144-148 per utterance: samples of utterance u *= scale_dev[u]. */
int srx_scale_dev(srx_ctx *ctx, float *samples_dev, const int64_t *sample_off_dev, int n_utt, const float *scale_dev);
/* Clip::rmsAmplitude  src/clip.cpp:157-164 per utterance: rms_dev[u] = sqrt(sum of squares) (no 1/N). */
int srx_rms_amplitude_dev(srx_ctx *ctx, const float *samples_dev, const int64_t *sample_off_dev, int n_utt,
                          double *rms_dev);
/* Clip::preemphasis  src/clip.cpp:199-205 (in place, per utterance; needs a same-size scratch) */
int srx_preemphasis_dev(srx_ctx *ctx, float *samples_dev, float *scratch_dev, const int64_t *sample_off_dev,
                        int n_utt, int64_t total_samples, float alpha);

/* read_features smoothing + compute_deltas  src/mfcc.cpp:296-312, 73-102:
 * ceps [total_frames][n_ceps] -> feats [total_frames][3*n_ceps] = [coeffs|deltas|delta2s].
 * WL = -1: no smoothing stage at all -- compute_deltas(vector) alone (src/mfcc.cpp:95-102), coefficients copied through. */
int srx_features_batch_dev(srx_ctx *ctx, int WL, int n_ceps, const float *ceps_dev, const int64_t *frame_off_dev,
                           int n_utt, int64_t total_frames, float *feats_dev);
int srx_features_batch(srx_ctx *ctx, int WL, int n_ceps, const float *ceps, const int64_t *frame_off, int n_utt,
                       float *feats);

/* mean(ut) / variance(ut, mu)  src/utterance.cpp:32-65 over the first D dims. */
int srx_corpus_mean_var_dev(srx_ctx *ctx, const float *feats_dev, int ld, int64_t total_frames, int D,
                            double *sum_dev /*[2*D] scratch+result: sums then sum of squares*/);

/* One half of mean(ut) / variance(ut, mu) for a shard of the corpus: mu_dev == NULL -> out_dev[d] = sum_t x[t][d]
 * (src/utterance.cpp:36-41); otherwise out_dev[d] = sum_t (x[t][d] - mu[d])^2 with the f32 difference and square of
 * src/utterance.cpp:55-58.  The sums of all shards are added (srx_allreduce_sum_f64_dev) before the division by the
 * global frame count. */
int srx_corpus_sums_dev(srx_ctx *ctx, const float *feats_dev, int ld, int64_t total_frames, int D, const float *mu_dev,
                        double *out_dev /*[D]*/);

/* ---------------------------------------------------------------- acoustic model ---- */
typedef struct srx_model srx_model;

typedef enum {
    SRX_MIX_COMPONENT0 = 0, /* gmm::operator() as written: component 0 only (src/gmm.cpp:99-100) */
    SRX_MIX_LOGSUMEXP = 1   /* the commented-out intent, log sum_m exp(g_m) (src/gmm.cpp:103-107) */
} srx_mix_mode;

/* acoustic_model: G tied states x M diagonal Gaussians x D dims (src/gmm.hpp:25-94).
 * mean/var are [G][M][D] doubles (the reference keeps 80-bit long double), weight [G][M]. */
int srx_model_create(srx_ctx *ctx, int G, int M, int D, srx_model **out);
void srx_model_destroy(srx_model *m);
int srx_model_set(srx_model *m, const double *mean, const double *var, const float *weight);
int srx_model_get(srx_model *m, double *mean, double *var, float *weight);
int srx_model_copy(srx_model *dst, const srx_model *src); /* acoustic_model(acoustic_model*) src/gmm.cpp:171-175 */
/* A counter that every call which rewrites the model's parameters advances (srx_model_set, srx_model_copy into it,
 * srx_em_means_dev / srx_em_finish_dev on it); 0 for a null model.  No counterpart in the reference.  What it is for: the
 * utterance HMMs of hmm::train score under main()'s acoustic_model OBJECT (src/utterance.cpp:20, src/main.cpp:45), and
 * hmm::train writes into that object only every second iteration (acm / temp_acm swap, src/hmm.cpp:344-346) -- iterations
 * 2k+1 and 2k+2 align under identical emission scores, so a caller that kept the compact emission buffer of the last
 * srx_gmm_score_align_dev call may skip the call while this counter (and its feature buffer) has not changed. */
uint64_t srx_model_version(const srx_model *m);

/* gaussian::operator()(fv) / gaussian::logprob  src/gmm.cpp:41-54 for ONE Gaussian that is not part of a model, over T
 * feature vectors (host pointers; mean/var [D] doubles for the reference's long double): out[t] in the reference's
 * operation order (the arithmetic of SRX_SCORE_EXACT below).  gaussian::prob is exp() of it (src/gmm.cpp:57-59). */
int srx_gaussian_logprob(srx_ctx *ctx, const float *feats, int ld, int64_t T, int D, const double *mean, const double *var,
                         float logwt, float *out);

typedef enum {
    SRX_SCORE_MFMA = 0,       /* f32-input MFMA contraction over [x^2, x, 1] (exact f32 products) */
    SRX_SCORE_EXACT = 1,      /* per-dimension evaluation in the reference's operation order (f64 for long double) */
    SRX_SCORE_MFMA_F16X3 = 2  /* the same contraction on f16 MFMA with both operands split hi + lo (three products,
                                 f32 accumulate: ~22 significant bits per product); D <= 39, M <= 16; frames or
                                 models outside the f16 range are re-evaluated / routed to SRX_SCORE_MFMA */
} srx_score_path;

/* acoustic_model::operator()(fv, ctx) for every frame and every tied state
 * (src/gmm.cpp:41-49, 99-109, 184-186).  Score (t, g) is written to out[t*stride_t + g*stride_g]:
 * frame-major (stride_g = 1, stride_t >= G) or state-major (stride_t = 1, stride_g >= T; the layout
 * the Viterbi kernel streams best). */
int srx_gmm_score_dev(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats_dev, int ld,
                      int64_t T, float *out_dev, int64_t stride_t, int64_t stride_g);
int srx_gmm_score(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats, int ld, int64_t T,
                  float *out);

/* ---------------------------------------------------------------- HMM / Viterbi ------ */
/* hmm::hmm(vector<phone>, acoustic_model*)  src/hmm.cpp:56-88 (USE_SUBPHONES false): host helper.
 * phones[S] (0 = SIL) -> ntr[S], trlogp[S][3] (targets s, s+1, s+2), ctx[S][3] (prev,curr,next). */
int srx_hmm_build_training(const int32_t *phones, int S, int32_t *ntr, float *trlogp, int32_t *ctx);

/* A ragged batch of left-to-right training HMMs (device pointers).  Utterance u owns
 * states [state_off[u], state_off[u+1]) and frames [frame_off[u], frame_off[u+1]). */
typedef struct {
    int n_utt;
    const int64_t *frame_off_dev; /* [n_utt+1] */
    const int32_t *state_off_dev; /* [n_utt+1] */
    const int32_t *tied_dev;      /* [n_states] column of the emission matrix for each state */
    const int32_t *ntr_dev;       /* [n_states] 1..3 */
    float *trlogp_dev;            /* [n_states][3] (mutated by srx_train_transitions_dev) */
    int max_states;               /* max over utterances of S (host-known) */
    int64_t total_frames;         /* frame_off[n_utt] (host-known) */
} srx_hmm_batch;

/* hmm::viterbi  src/hmm.cpp:150-224 for every utterance of the batch, emission scores given:
 * emis[t*stride_t + tied[s]*stride_g] = (*acm)(*fvs[t], states[s].ctx), t counted over the whole
 * batch (the strides of srx_gmm_score_dev).  Outputs: path[total_frames]
 * (state index within the utterance), final_logp[n_utt] (= curr[best]), status[n_utt]
 * (0 ok, 1 "NO PATH": path/final untouched; 2: the utterance has no state at all, more than SRX_VITERBI_MAX_STATES = 512
 * states, or more than the 64 / 128 / 256 / 512 that hmm->max_states sized the launch for (a caller bug): nothing computed.
 * Status 2 is a refusal, not a property of the audio: take a wide utterance through
 * srx_viterbi_graph_dev, which has no limit and gives the same score and path; hmm::viterbi / hmm::train of the host layer
 * do.  The reference has no limit, src/hmm.cpp:56-88).
 * prevp of the reference = final_logp / T.
 * Score is bit-exact w.r.t. the reference's f32 add order (it accumulates logp across a source's
 * transitions, src/hmm.cpp:164-169); exact ties are resolved towards the higher source index
 * (the reference's order is unordered_map iteration order, i.e. unspecified).
 * beam_width >= 1e10 is the reference default (src/hmm.cpp:150).  The reference still applies its pruning test there
 * (src/hmm.cpp:165: a source at -inf, or 1e10 below the frame's best, is skipped; if every candidate of a frame dies the
 * result is NO PATH): the kernel runs without the test and repeats an utterance with it when some score left +-2.5e9.
 * Fastest layout: state-major with 16-byte aligned rows (stride_t == 1, stride_g a multiple of 4, emis_dev 16-byte
 * aligned): four frames per load. */
enum { SRX_VITERBI_MAX_STATES = 512 };
int srx_viterbi_batch_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t,
                          int64_t stride_g, float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev);

/* srx_viterbi_batch_dev without any allocation.  NO COUNTERPART IN THE REFERENCE: it exists for callers that cannot
 * allocate device memory -- code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc.
 * Same arguments, same kernel, same results bit for bit; the back-pointer words (srx_viterbi_batch_dev keeps them in the
 * context's grow-only workspace) live in a fixed array inside the library instead, SRX_VITERBI_NOALLOC_MAX_WORDS 32-bit
 * words (32 MiB) on every GPU the library runs on, held until the process ends.  The call performs no hipMalloc / hipFree
 * and creates no events (it is not recorded by srx_profile_*).  A batch needs
 *   (total_frames / 16 + n_utt + 1) * 64 * spl words,  spl = 1, 2, 4, 8 for max_states <= 64, 128, 256, more;
 * a larger batch is refused with SRX_ERR_INVALID (use srx_viterbi_batch_dev, or split the batch).  All calls on one GPU
 * share that GPU's array: calls that overlap in time on one GPU (two contexts, two streams) race -- serialise them. */
enum { SRX_VITERBI_NOALLOC_MAX_WORDS = 8 * 1024 * 1024 };
int srx_viterbi_batch_noalloc_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const float *emis_dev, int64_t stride_t,
                                  int64_t stride_g, float beam_width, int32_t *path_dev, float *final_logp_dev,
                                  int32_t *status_dev);

/* Alignment-mode emission scoring: every utterance's frames against that utterance's own states only -- what the
 * reference computes lazily inside hmm::viterbi (src/hmm.cpp:166, (*acm)(*fvs[i], target->ctx) for the targets of the
 * active states) -- instead of the dense [T][G] matrix of srx_gmm_score_dev.  The plan (host frame/state offsets of the
 * batch) fixes the compact layout: utterance u owns emis[off[u] .. off[u+1]) as a frame-major block [T_u][S_u]
 * (emis[off[u] + t*S_u + s]), off[u+1] - off[u] = S_u * T_u < 2^31 (plan creation rejects a larger block: the alignment
 * kernel indexes inside a block with 32 bits).  Scores are bit-identical to srx_gmm_score_dev's on the same path.
 * score_path: SRX_SCORE_MFMA_F16X3 (models f16 cannot hold are evaluated exactly instead) or SRX_SCORE_EXACT. */
typedef struct srx_align_plan srx_align_plan;
int srx_align_plan_create(srx_ctx *ctx, const int64_t *frame_off, const int32_t *state_off, int n_utt, srx_align_plan **out);
void srx_align_plan_destroy(srx_align_plan *plan);
int64_t srx_align_plan_emission_count(const srx_align_plan *plan);             /* floats: off[n_utt] */
int srx_align_plan_emission_offsets(const srx_align_plan *plan, int64_t *off); /* host copy, [n_utt+1] */
const int64_t *srx_align_plan_emission_offsets_dev(const srx_align_plan *plan);
/* srx_align_plan_cache_frames(plan, 1): the caller promises that the feature buffer it scores with this plan does not
 * change between calls.  The split-f16 path then keeps the frame-side operands of the batch (320 bytes per frame at
 * D = 39) in the plan and rebuilds them only when the feature pointer or the model's centre / scale vector changed -- the
 * iterations of hmm::train score the same frames under a model that moves a little, so they are built once instead of
 * inside every scoring call (no counterpart in the reference: a memory-for-time trade sized for 288 GB of HBM).  Off by
 * default (the cache is keyed on the pointer, not on the contents); skipped by itself when device memory is short;
 * (plan, 0) turns it off and frees it.  Scores are the same bits either way. */
int srx_align_plan_cache_frames(srx_align_plan *plan, int on);
int srx_gmm_score_align_dev(srx_ctx *ctx, srx_model *m, int mix_mode, int score_path, const float *feats_dev, int ld,
                            const srx_hmm_batch *hmm, const srx_align_plan *plan, float *emis_dev);
/* srx_viterbi_batch_dev on the compact emissions of srx_gmm_score_align_dev (same kernel, same bit-exactness). */
int srx_viterbi_align_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const srx_align_plan *plan, const float *emis_dev,
                          float beam_width, int32_t *path_dev, float *final_logp_dev, int32_t *status_dev);

/* ---------------------------------------------------------------- lexicon HMM / decoding */
/* hmm::hmm(pronlex&, unigram_model&, acoustic_model*, state_model*)  src/hmm.cpp:92-129 (USE_SUBPHONES false): host helper.
 * Words in the iteration order of the reference's map<string, pronlist> (sorted by spelling); word w has the phones
 * phones[phone_off[w] .. phone_off[w+1]) (its first pronunciation) and the unigram log-probability uni_logp[w]
 * (unigram_model::get, src/langmodel.cpp:21-27); its start transition carries lmsf * uni_logp[w] (LMSF = 5,
 * src/hmm.hpp:33).  State 0 is the start state; then per word its phone states and a terminating silence.
 * Outputs (CSR, transitions in the reference's push_back order): tr_off[n_states+1], tr_tgt / tr_logp[n_tr],
 * ctx[n_states][3] = (prev, curr, next) phone, word_of[n_states] (-1 for the start state).  With tr_off == NULL only the
 * two counts are written. */
int srx_hmm_build_lexicon(int n_words, const int32_t *phone_off, const int32_t *phones, const float *uni_logp, float lmsf,
                          int32_t *tr_off, int32_t *tr_tgt, float *tr_logp, int32_t *ctx, int32_t *word_of, int64_t *n_states,
                          int64_t *n_tr);

/* An HMM with an arbitrary transition list on the device (host arrays in; tied[s] = emission column of state s).
 * srx_decode_plan_set_transitions replaces the log-probabilities after hmm::update_states (src/hmm.cpp:361-366). */
typedef struct srx_decode_plan srx_decode_plan;
int srx_decode_plan_create(srx_ctx *ctx, int n_states, const int32_t *tr_off, const int32_t *tr_tgt, const float *tr_logp,
                           const int32_t *tied, srx_decode_plan **out);
int srx_decode_plan_set_transitions(srx_decode_plan *plan, const float *tr_logp);
void srx_decode_plan_destroy(srx_decode_plan *plan);
/* How srx_viterbi_graph_dev walks the HMM (same results, bit for bit).  SRX_DECODE_DENSE: every state every frame, one
 * workgroup per utterance, int32 back-pointers [frames][n_states] -- small lexica, many utterances at once.
 * SRX_DECODE_FRONTIER: per frame only the states inside the beam (the reference's `layer` map restricted to what
 * src/hmm.cpp:165 lets through) push their transitions; back-pointers are kept per frontier member (8 bytes each), so the
 * memory follows the beam, not frames * n_states; the whole GPU works on one frame -- the full lexicon of
 * src/main.cpp:58 (about 10^6 states).  SRX_DECODE_AUTO (default): dense up to 32,768 states while the longest utterance's
 * back-pointers fit half the workspace limit.  max_workspace_bytes (0: keep, default 2 GiB) bounds both: the dense sweep
 * takes the utterances in groups whose back-pointers fit half of it, the frontier sweep keeps its per-state arrays and its
 * store inside it; an utterance whose frontier outgrows its share gets status 3.  No counterpart in the reference (a
 * tuning knob). */
enum { SRX_DECODE_AUTO = 0, SRX_DECODE_DENSE = 1, SRX_DECODE_FRONTIER = 2 };
int srx_decode_plan_set_sweep(srx_decode_plan *plan, int sweep, size_t max_workspace_bytes);

/* hmm::viterbi  src/hmm.cpp:150-224 on that HMM, for every utterance of a batch (frame_off: HOST array [n_utt+1], frames
 * counted over the emission matrix as in srx_viterbi_batch_dev): beam search with `beam_width` (src/hmm.cpp:165), `logp`
 * carried across a source's transitions as the reference has it (src/hmm.cpp:164-169), exact ties to the higher source
 * index.  final_mode 0: the path must end in the last state, as the reference is compiled (src/hmm.cpp:210-215);
 * final_mode 1: it ends in the best active state above -1e10f, the recognition variant the reference has commented out
 * (src/hmm.cpp:196-209; its `throw 1729` is status 1 here).  Outputs: path[frames] (state indices), final_logp[n_utt],
 * final_state[n_utt] (may be NULL), status[n_utt] (0 ok, 1 NO PATH, 3 the frontier sweep ran out of its workspace share:
 * see srx_decode_plan_set_sweep).  Score and path are bit-identical to the CPU restatement (PARITY UNPINNED: the reference
 * never runs Viterbi on its lexicon HMM).  Workspace: see srx_decode_plan_set_sweep. */
int srx_viterbi_graph_dev(srx_ctx *ctx, const srx_decode_plan *plan, const float *emis_dev, int64_t stride_t, int64_t stride_g,
                          const int64_t *frame_off, int n_utt, float beam_width, int final_mode, int32_t *path_dev,
                          float *final_logp_dev, int32_t *final_state_dev, int32_t *status_dev);

/* hmm::train_transition_probabilities  src/hmm.cpp:229-243 (+ set/get_adv_prob :26-35) for every
 * utterance whose status is 0. */
int srx_train_transitions_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *path_dev,
                              const int32_t *status_dev);

/* state_model::augment over every frame of every final path  src/hmm.cpp:46-51, 350-356 (the tail of
 * hmm::train): cls[n_states] = tied class of each state under the state model's ties (-1: skip);
 * probsum[c] += exp(tr[1]) (hmm_state::get_adv_prob), count[c] += 1.  state_model::operator()
 * (src/hmm.cpp:39-44) is then probsum/count, 0.5 where count == 0.  Sums are f64 (reference: f32). */
int srx_state_model_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *cls_dev, const int32_t *path_dev,
                        const int32_t *status_dev, int n_classes, double *probsum_dev, int32_t *count_dev);

/* ---------------------------------------------------------------- embedded training -- */
/* The statistics half of one EM iteration, hmm::train src/hmm.cpp:274-341, split at the two
 * points where ranks must exchange sums.  A = model the statistics are taken
 * under (hmm::acm), B = model rebuilt (hmm::temp_acm).  gid[t] = tied state of aligned frame t
 * (-1: frame skipped, e.g. NO PATH utterance).
 *   1. srx_em_shift_dev      per-component max log-density -> shift[G*M] (all-reduce MAX)
 *   2. srx_em_pass1_dev      zeta sums and zeta*x sums  -> stats1[G*M*(1+D)] f64 (all-reduce SUM)
 *   3. srx_em_means_dev      weights + means of B from stats1
 *   4. srx_em_pass2_dev      zeta*(x-mean_B)^2 sums     -> stats2[G*M*D] f64     (all-reduce SUM)
 *   5. srx_em_finish_dev     variances of B, then B = keep*B + (1-keep)*A (gaussian::blend)
 * Call them in this order for every new alignment: step 1 also sorts the frames by state when the table is too
 * large for LDS (then passes 2 and 4 are atomics-free segmented reductions, deterministic in frame order).
 * zeta is computed as w*exp(logp - shift[g][m]) in f64 (the reference uses unshifted 80-bit values;
 * every quantity it derives from them is a ratio in which the shift cancels or is re-applied).  The log-densities of
 * step 1 follow gaussian::operator() (src/gmm.cpp:41-49) term by term in f64 with the division by the variance as a
 * multiplication by its f64 reciprocal (the reference divides in 80-bit long double). */
int srx_em_gather_gid_dev(srx_ctx *ctx, const srx_hmm_batch *hmm, const int32_t *path_dev,
                          const int32_t *status_dev, int64_t total_frames, int32_t *gid_dev);
int srx_em_shift_dev(srx_ctx *ctx, srx_model *A, const float *feats_dev, int ld, const int32_t *gid_dev, int64_t T,
                     float *logp_dev /*[T][M] out*/, float *shift_dev /*[G*M] out*/);
int srx_em_pass1_dev(srx_ctx *ctx, srx_model *A, const float *feats_dev, int ld, const int32_t *gid_dev, int64_t T,
                     const float *logp_dev, const float *shift_dev, double *stats1_dev);
int srx_em_means_dev(srx_ctx *ctx, srx_model *B, const double *stats1_dev, const float *shift_dev);
int srx_em_pass2_dev(srx_ctx *ctx, srx_model *A, srx_model *B, const float *feats_dev, int ld,
                     const int32_t *gid_dev, int64_t T, const float *logp_dev, const float *shift_dev,
                     double *stats2_dev);
int srx_em_finish_dev(srx_ctx *ctx, srx_model *A, srx_model *B, const double *stats1_dev, const double *stats2_dev,
                      float blend_keep);

/* The same statistics in ONE pass and ONE exchange (round 3; the five steps above stay as the rounding-faithful form).
 * With x' = x - c (c = the state's centre: the mean of its component means under A, identical on every rank)
 *     S0 = sum zeta,  S1 = sum zeta x',  S2 = sum zeta x'^2   ->   mean = c + S1/S0,  var = S2/S0 - (S1/S0)^2
 * which is the reference's sum zeta (x - NEW mean)^2 / sum zeta (src/hmm.cpp:318-339) without its rounding of (x - mean) to f32
 * (src/hmm.cpp:331: a relative difference of parts in 1e8 on a variance) and without a second sweep over the frames.
 * srx_em_fused_dev sorts the frames by state and accumulates [S1 | S0 | S2 | frame count] per (state, component) on the f64
 * matrix cores into stats12[srx_em_fused_len(A)] (all-reduce SUM it across ranks); srx_em_fused_finish_dev rebuilds B from it
 * and blends with A.  Two modes:
 *   logp_dev / shift_dev given (from srx_em_shift_dev, shift all-reduced MAX): zeta from the rounding-faithful log-densities;
 *       only the variance formula differs from the five-step form (means, weights: same sums).
 *   logp_dev == shift_dev == NULL: the log-densities are evaluated on the matrix cores too, in f64 WITHOUT the reference's f32
 *       roundings of gaussian::operator() (src/gmm.cpp:44-47), relative to each component's peak density (no data-dependent shift:
 *       no pre-pass, no MAX exchange).  zeta then differs from the reference's by the reference's own f32 noise (~1e-6 relative per
 *       frame).  *underflow_dev (may be NULL) receives the number of components that saw frames but whose zeta sum fell below
 *       1e-290 (every frame tens of standard deviations away: the reference's 80-bit exponent holds those, f64 relative to the
 *       peak does not): when it is non-zero, repeat the iteration's statistics through the five steps above.
 * Shapes: M <= 16, D <= 39, G <= 16383 (srx_em_fused_len returns 0 otherwise). */
int64_t srx_em_fused_len(const srx_model *A);
int srx_em_fused_dev(srx_ctx *ctx, srx_model *A, const float *feats_dev, int ld, const int32_t *gid_dev, int64_t T,
                     const float *logp_dev, const float *shift_dev, double *stats12_dev);
int srx_em_fused_finish_dev(srx_ctx *ctx, srx_model *A, srx_model *B, const double *stats12_dev, const float *shift_dev,
                            float blend_keep, int32_t *underflow_dev);

/* ---------------------------------------------------------------- multi-GPU exchange --- */
/* One process per GPU, utterances sharded across the ranks; MFCC, scoring and alignment need no communication.  What
 * the ranks exchange are the sums that the reference's sequential loops over all utterances build:
 *   per EM iteration  MAX of shift[G*M] f32 (after srx_em_shift_dev), SUM of stats1[G*M*(1+D)] f64 (after
 *                     srx_em_pass1_dev; src/hmm.cpp:275-287, 298-316), SUM of stats2[G*M*D] f64 (after srx_em_pass2_dev;
 *                     src/hmm.cpp:319-338) -- every rank then rebuilds the identical model;
 *   once at the end   SUM of the state-model tables of srx_state_model_dev (src/hmm.cpp:350-356);
 *   before training   SUM of the corpus sums of srx_corpus_sums_dev (mean / variance, src/utterance.cpp:32-65).
 * All-reduces are in place on device buffers, asynchronous on the context's stream (ordered with the kernels around
 * them), RCCL over xGMI.  world == 1 makes every call a no-op, so a single-GPU run takes the same path.
 *
 * srx_comm_unique_id: rank 0 obtains SRX_COMM_ID_BYTES opaque bytes (ncclGetUniqueId) and hands them to the other
 * ranks out of band (file, environment, the launcher's store); srx_comm_create joins (collective over the ranks; one
 * GPU per rank).  srx_comm_create_host is a REHEARSAL transport for ranks that share one GPU, which RCCL refuses:
 * buffers cross a POSIX shared-memory segment `shm_name` ("/name") and are summed on the host in rank order.  It exists
 * so that multi-rank control flow can be tested on a one-GPU box; it is not a data path.  `shm_name` must be unique to the
 * launch (the host layer appends its launch nonce): rank 0 removes and recreates the name, but a rank that arrives first
 * could otherwise join a segment a crashed launch left behind.  A copy that fails inside an all-reduce fails the call on
 * EVERY rank (no rank is left waiting) and the communicator stays unusable. */
typedef struct srx_comm srx_comm;
enum { SRX_COMM_ID_BYTES = 128 };
int srx_comm_unique_id(void *id_out);
int srx_comm_create(srx_ctx *ctx, int rank, int world, const void *id, srx_comm **out);
int srx_comm_create_host(srx_ctx *ctx, int rank, int world, const char *shm_name, srx_comm **out);
void srx_comm_destroy(srx_comm *comm);
int srx_comm_rank(const srx_comm *comm);
int srx_comm_world(const srx_comm *comm);
int64_t srx_comm_bytes_reduced(const srx_comm *comm); /* payload bytes handed to all-reduces so far */
int srx_allreduce_max_f32_dev(srx_comm *comm, float *buf_dev, int64_t n);
int srx_allreduce_sum_f64_dev(srx_comm *comm, double *buf_dev, int64_t n);
int srx_allreduce_sum_i32_dev(srx_comm *comm, int32_t *buf_dev, int64_t n);

#ifdef __cplusplus
}
#endif
#endif /* SRX_H */
