/*
 * srx_oracle.c -- CPU restatement of the reference's acoustic hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libsrx.so, HIP) never
 * links, loads or calls anything in this directory.
 *
 * PARITY UNPINNED: the reference (albinoBandicoot/speechRecognition) ships no tests, no
 * golden vectors and no fixtures, and its sources cannot be compiled in this image
 * (every translation unit includes OpenFrameworks' ofMain.h, which is absent; see
 * DESIGN.md "Oracle").  This file is therefore a from-scratch restatement of the
 * reference's arithmetic, statement by statement, with the C++ conversion rules applied
 * by hand (float op double -> double, `float += double` rounds once per add, long double
 * is the x87 80-bit type on x86-64).  The only external anchors are the scalar facts the
 * survey recorded from the reference: they are checked in
 * tests/test_oracle_pins.py.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/).  Third-party arithmetic: OpenFrameworks' ofMap (version unpinned by
 * the reference; project layout suggests 0.9.x) is restated from its published formula.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off, no fast-math).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif
#ifndef M_2_PI
#define M_2_PI 0.63661977236758134308 /* 2/pi */
#endif

/* ------------------------------------------------------------------------------------
 * Filterbank  (src/mfcc.cpp:117-140, 180-201)
 * ---------------------------------------------------------------------------------- */
typedef struct {
    float center, width, height;
} orc_filter;

/* OpenFrameworks ofMap(value, inMin, inMax, outMin, outMax), clamp=false; all float.
 * Published formula (ofMath.cpp): ((v-i0)/(i1-i0)*(o1-o0)+o0), with the degenerate-range guard. */
static float of_map(float v, float i0, float i1, float o0, float o1) {
    if (fabsf(i0 - i1) < 1.1920929e-07f) return o0;
    return ((v - i0) / (i1 - i0) * (o1 - o0) + o0);
}

/* filter::get  src/mfcc.cpp:132-140 */
float orc_filter_get(const orc_filter *f, int bin) {
    float diff = f->center - (float)bin;
    if (diff < 0) diff = -diff;
    if (diff > f->width) return 0;
    diff /= f->width;
    diff = 1 - diff;
    diff *= f->height;
    return diff;
}

/* filter::filter  src/mfcc.cpp:117-128 */
static void filter_init(orc_filter *f, float c, float w, int dft_bins, int sps) {
    f->center = of_map(c, 0, (float)sps, 0, (float)dft_bins);
    f->width = of_map(w, 0, (float)sps, 0, (float)dft_bins);
    f->height = 1;
    float h = 0;
    for (int i = (int)(f->center - f->width); i <= (int)(f->center + f->width); i++) h += orc_filter_get(f, i);
    f->height = 1 / h;
}

/* melToHz / hzToMel  src/mfcc.cpp:180-187  (float overloads of exp/log) */
static float mel_to_hz(float mel) { return 700 * (expf(mel / 1127.0f) - 1); }
static float hz_to_mel(float hz) { return 1127 * logf(1 + hz / 700.0f); }

/* mel_filterbank  src/mfcc.cpp:190-201.  Returns bin_max (src/mfcc.cpp:220-221). */
int orc_mel_filterbank(float end, int n, int dft_bins, int sps, orc_filter *out) {
    float mel_end = hz_to_mel(end);
    float mel_step = mel_end / (float)(n + 1);
    float prev = 0;
    for (int i = 0; i < n; i++) {
        float hz = mel_to_hz((float)(i + 1) * mel_step);
        filter_init(&out[i], hz, hz - prev, dft_bins, sps);
        prev = hz;
    }
    return (int)(out[n - 1].center + out[n - 1].width) + 1;
}

/* ------------------------------------------------------------------------------------
 * Clip operations (src/clip.cpp)
 * ---------------------------------------------------------------------------------- */

/* Clip::rmsAmplitude  src/clip.cpp:157-164 (rectangular window => x*1.0f) */
double orc_rms_amplitude(const float *x, long L) {
    double amp = 0;
    for (long i = 0; i < L; i++) {
        float s = x[i] * 1.0f;
        amp += (double)(s * s);
    }
    return sqrt(amp);
}

/* Clip::normalizeVolume + operator*=(float)  src/clip.cpp:151-154, 144-148 */
void orc_normalize_volume(float *x, long L, float target) {
    float cur = (float)orc_rms_amplitude(x, L);
    float f = target / cur;
    for (long i = 0; i < L; i++) x[i] *= f;
}

/* Clip::preemphasis  src/clip.cpp:199-205 */
void orc_preemphasis(float *x, long L, float alpha) {
    for (long i = L - 1; i > 0; i--) x[i] -= alpha * x[i - 1];
    x[0] = x[1];
}

/* hamming_window  src/utils.cpp:21-23 */
float orc_hamming(int x, int len) { return (float)(0.54f + 0.46 * cos(2 * M_PI * x / (len - 1))); }

/* frame count of the do/while(slide) loop: src/mfcc.cpp:261-271, src/clip.cpp:108-112 */
long orc_num_frames(long L, int wl, int hop) {
    long n = 1;
    long end = wl;
    for (;;) {
        end += hop;
        if (!(end < L)) break;
        n++;
    }
    return n;
}

/* Clip::partial_dft (stack=false)  src/clip.cpp:174-194, through the const operator[]
 * (src/clip.cpp:123-125) with window = hamming_window.  x points at the frame start. */
void orc_partial_dft(const float *x, int N, int len, float *out) {
    float rootn_inv = (float)(1.0f / sqrt((double)N));
    int mirror = len > N / 2;
    for (int k = 0; k < len; k++) {
        float re = 0, im = 0;
        for (int n = 0; n < N / 2; n++) {
            float y = x[n] * orc_hamming(n, N);
            re = (float)((double)re + (double)y * cos(-2 * M_PI * k * n / N));
            im = (float)((double)im + (double)y * sin(-2 * M_PI * k * n / N));
        }
        float res = sqrtf(re * re + im * im) * rootn_inv;
        out[k] = res;
        if (mirror && N - k - 1 < len) out[N - k - 1] = res;
    }
}

/* Clip::partial_dft with its `stack` argument and an arbitrary window  src/clip.cpp:174-194, statement by statement:
 * win[n] = window(n, N) for n < N/2 (NULL: rectangular_window, src/utils.cpp:13-15); out has len entries and is read when
 * stack != 0.  Requires len <= N (beyond that the reference writes out[N-k-1] at a negative index). */
void orc_partial_dft_win(const float *x, const float *win, int N, int len, int stack, float *out) {
    float rootn_inv = (float)(1.0f / sqrt((double)N));
    int mirror = len > N / 2;
    for (int k = 0; k < len; k++) {
        float re = 0, im = 0;
        for (int n = 0; n < N / 2; n++) {
            float y = x[n] * (win ? win[n] : 1.0f);
            re = (float)((double)re + (double)y * cos(-2 * M_PI * k * n / N));
            im = (float)((double)im + (double)y * sin(-2 * M_PI * k * n / N));
        }
        float res = sqrtf(re * re + im * im) * rootn_inv;
        if (stack) {
            out[k] += res;
            if (mirror && N - k - 1 < len) out[N - k - 1] += res;
        } else {
            out[k] = res;
            if (mirror && N - k - 1 < len) out[N - k - 1] = res;
        }
    }
}

/* log(T *buf, int len)  src/utils.hpp:55-60 with T = float */
void orc_log1p(float *buf, long n) {
    for (long i = 0; i < n; i++) buf[i] = logf(buf[i] + 1);
}

/* compute_deltas(vector<featurevec*>)  src/mfcc.cpp:95-102 with featurevec::compute_deltas / compute_delta2s  :73-92:
 * feats n x 3*nparams, coefficients in [0, nparams) on entry; fills the deltas and delta-deltas. */
void orc_compute_deltas(float *feats, long n, int nparams) {
    int ld = 3 * nparams;
    for (long i = 0; i < n; i++) {
        long j = i - 1, k = i + 1;
        if (j < 0) j = 0;
        if (k == n) k--;
        for (int q = 0; q < nparams; q++) feats[i * ld + nparams + q] = (feats[k * ld + q] - feats[j * ld + q]) / (float)(k - j);
    }
    for (long i = 0; i < n; i++) {
        long j = i - 1, k = i + 1;
        if (j < 0) j = 0;
        if (k == n) k--;
        for (int q = 0; q < nparams; q++)
            feats[i * ld + 2 * nparams + q] = (feats[k * ld + nparams + q] - feats[j * ld + nparams + q]) / (float)(k - j);
    }
}

/* filterbank::operator() + filter::operator()  src/mfcc.cpp:143-149,161-167 */
void orc_filterbank_apply(const orc_filter *fb, int nfilt, const float *spec, float *out) {
    for (int f = 0; f < nfilt; f++) {
        float res = 0;
        for (int i = (int)(fb[f].center - fb[f].width); i <= (int)(fb[f].center + fb[f].width); i++)
            res += orc_filter_get(&fb[f], i) * spec[i];
        out[f] = res;
    }
}

/* dct  src/mfcc.cpp:205-214 */
void orc_dct(const float *filtered, int len, int nparams, float *res) {
    for (int j = 0; j < nparams; j++) {
        res[j] = 0;
        for (int i = 0; i < len; i++) {
            float ji = (float)j * ((float)i - 0.0f);
            res[j] = (float)((double)res[j] + (double)filtered[i] * cos((double)ji * M_PI / len));
        }
    }
}

/* mfcc()  src/mfcc.cpp:219-232 ; log(T*,int) src/utils.hpp:55-60.
 * spec_out (optional) receives the bin_max magnitudes. */
void orc_mfcc_frame(const float *x, int N, const orc_filter *fb, int nfilt, int nparams, float *out, float *spec_out) {
    int bin_max = (int)(fb[nfilt - 1].center + fb[nfilt - 1].width) + 1;
    float *dft = (float *)malloc(sizeof(float) * (size_t)(bin_max > N ? bin_max : N));
    float *filtered = (float *)malloc(sizeof(float) * (size_t)nfilt);
    orc_partial_dft(x, N, bin_max, dft);
    if (spec_out) memcpy(spec_out, dft, sizeof(float) * (size_t)bin_max);
    orc_filterbank_apply(fb, nfilt, dft, filtered);
    for (int i = 0; i < nfilt; i++) filtered[i] = logf(filtered[i] + 1);
    orc_dct(filtered, nfilt, nparams, out);
    free(dft);
    free(filtered);
}

/* write_features  src/mfcc.cpp:235-275 minus the file I/O and the dead noise estimate
 * (src/mfcc.cpp:246-254: computed, never used).  Mutates x (pre-emphasis) like the reference.
 * ceps receives n x nparams floats; returns n.  Requires L > wl. */
long orc_write_features(float *x, long L, int wl, int hop, const orc_filter *fb, int nfilt, int nparams, float alpha,
                        float *ceps) {
    orc_preemphasis(x, L, alpha);
    long n = 0;
    long start = 0, end = wl;
    do {
        orc_mfcc_frame(x + start, wl, fb, nfilt, nparams, ceps + n * nparams, NULL);
        n++;
        start += hop;
        end += hop;
    } while (end < L);
    return n;
}

/* read_features smoothing + compute_deltas  src/mfcc.cpp:296-312, 73-102.
 * raw: n x nparams; feats: n x 3*nparams laid out [coeffs | deltas | delta2s] like featurevec. */
void orc_read_features(const float *raw, long n, int nparams, int WL, float *feats) {
    int ld = 3 * nparams;
    for (long i = 0; i < n; i++) {
        float *fv = feats + i * ld;
        for (int j = 0; j < ld; j++) fv[j] = 0;
        int NF = 0;
        for (int w = -WL; w <= WL; w++) {
            int wt = WL - abs(w) + 1;
            if (i + w >= 0 && i + w < n) {
                NF += wt;
                for (int j = 0; j < nparams; j++) fv[j] += (float)wt * raw[(i + w) * nparams + j];
            }
        }
        for (int j = 0; j < nparams; j++) fv[j] /= (float)NF;
    }
    for (long i = 0; i < n; i++) {
        long j = i - 1, k = i + 1;
        if (j < 0) j = 0;
        if (k == n) k--;
        for (int q = 0; q < nparams; q++)
            feats[i * ld + nparams + q] = (feats[k * ld + q] - feats[j * ld + q]) / (float)(k - j);
    }
    for (long i = 0; i < n; i++) {
        long j = i - 1, k = i + 1;
        if (j < 0) j = 0;
        if (k == n) k--;
        for (int q = 0; q < nparams; q++)
            feats[i * ld + 2 * nparams + q] =
                (feats[k * ld + nparams + q] - feats[j * ld + nparams + q]) / (float)(k - j);
    }
}

/* mean / variance over a corpus  src/utterance.cpp:32-65 (f32 running sums, first D dims) */
void orc_corpus_mean_var(const float *feats, int ld, long T, int D, float *mu, float *var) {
    for (int j = 0; j < D; j++) mu[j] = 0;
    for (long i = 0; i < T; i++)
        for (int j = 0; j < D; j++) mu[j] += feats[i * ld + j];
    for (int j = 0; j < D; j++) mu[j] /= (float)(int)T;
    for (int j = 0; j < D; j++) var[j] = 0;
    for (long i = 0; i < T; i++)
        for (int j = 0; j < D; j++) {
            float f = feats[i * ld + j] - mu[j];
            var[j] += f * f;
        }
    for (int j = 0; j < D; j++) var[j] /= (float)(int)T;
}

/* ------------------------------------------------------------------------------------
 * GMM  (src/gmm.cpp)
 * ---------------------------------------------------------------------------------- */

/* gaussian::operator()  src/gmm.cpp:41-49 */
float orc_gaussian_logprob(const float *fv, const long double *mean, const long double *var, float logwt, int D) {
    float res = 0;
    float l2pi = (float)log(M_2_PI);
    for (int i = 0; i < D; i++) {
        float t = (float)((long double)fv[i] - mean[i]);
        long double term = ((long double)l2pi + logl(var[i])) + (long double)(t * t) / var[i];
        res = (float)((long double)res + term);
    }
    return -0.5f * res + logwt;
}

/* gaussian::set_wt  src/gmm.cpp:35-38: logwt = log(wt) on a float -> logf */
void orc_set_wt(const float *w, float *lw, long n) {
    for (long i = 0; i < n; i++) lw[i] = logf(w[i]);
}

/* acoustic_model::operator() over all tied states: out[t][g].
 * mode 0: gmm::operator() as written (component 0 only)  src/gmm.cpp:99-100
 * mode 1: the commented-out intent log(sum_m exp(g_m)) in double  src/gmm.cpp:103-107 */
void orc_gmm_score(const float *feats, int ld, long T, int G, int M, int D, const long double *mean,
                   const long double *var, const float *logwt, int mode, float *out) {
    for (long t = 0; t < T; t++) {
        const float *fv = feats + t * ld;
        for (int g = 0; g < G; g++) {
            if (mode == 0) {
                out[t * G + g] = orc_gaussian_logprob(fv, mean + (size_t)g * M * D, var + (size_t)g * M * D,
                                                      logwt[(size_t)g * M], D);
            } else {
                double res = 0;
                for (int m = 0; m < M; m++) {
                    size_t o = (size_t)g * M + m;
                    res += exp((double)orc_gaussian_logprob(fv, mean + o * D, var + o * D, logwt[o], D));
                }
                out[t * G + g] = (float)log(res);
            }
        }
    }
}

/* ------------------------------------------------------------------------------------
 * HMM  (src/hmm.cpp)
 * ---------------------------------------------------------------------------------- */
#define ORC_LOG_HALF -0.6931471805f   /* src/hmm.hpp:25 */
#define ORC_LOG_THIRD -1.09861228866f /* src/hmm.hpp:26 */
#define ORC_SIL_SKIP_P 0.3f           /* src/hmm.hpp:28 */
#define ORC_MAX_ADV_PROB 0.55f        /* src/hmm.hpp:29 */
#define ORC_HMM_BLEND_FAC 0.2f        /* src/hmm.hpp:30 */

/* hmm::hmm(vector<phone>, acoustic_model*)  src/hmm.cpp:56-88, USE_SUBPHONES=false.
 * phones[S] (0 = SIL).  Writes ntr[S], trlogp[S][3] (targets are s, s+1, s+2 in that order)
 * and ctx[S][3] = (prev, curr, next). */
void orc_build_training_hmm(const int *phones, int S, int *ntr, float *trlogp, int *ctx) {
    for (int i = 0; i < S; i++) {
        ntr[i] = 0;
        trlogp[i * 3] = trlogp[i * 3 + 1] = trlogp[i * 3 + 2] = 0;
    }
    for (int i = 0; i < S - 1; i++) {
        if (phones[i + 1] == 0 && i + 2 < S) {
            ntr[i] = 3;
            trlogp[i * 3] = trlogp[i * 3 + 1] = trlogp[i * 3 + 2] = ORC_LOG_THIRD;
        } else {
            ntr[i] = 2;
            trlogp[i * 3] = trlogp[i * 3 + 1] = ORC_LOG_HALF;
        }
    }
    ntr[S - 1] = 1;
    trlogp[(S - 1) * 3] = 0;
    for (int i = 0; i < S; i++) {
        ctx[i * 3 + 0] = (i == 0) ? 0 : phones[i - 1];
        ctx[i * 3 + 1] = phones[i];
        ctx[i * 3 + 2] = (i + 1 == S) ? 0 : phones[i + 1];
    }
}

/* hmm::viterbi  src/hmm.cpp:150-224 on a left-to-right training HMM, dense restatement.
 *  emis[t*ldE + tied[s]] = (*acm)(*fvs[t], states[s].ctx)  (tied==NULL -> column s)
 *  The reference visits active sources in unordered_map iteration order, which is
 *  stdlib-specific; `order` fixes it: 0 = descending source index, 1 = ascending.  The
 *  score never depends on it; the path does only when an exact float tie sits on a
 *  decision.  *nties counts exact ties seen between a candidate and the stored best.
 *  Returns 0 ok, 1 NO PATH (path untouched, final/prevp untouched). */
int orc_viterbi(const float *emis, long ldE, const int *tied, long T, int S, const int *ntr, const float *trlogp,
                float beam_width, int order, int *path, float *final_logp, float *prevp, long *nties) {
    float *cur = (float *)malloc(sizeof(float) * (size_t)S);
    float *nxt = (float *)malloc(sizeof(float) * (size_t)S);
    char *acur = (char *)calloc((size_t)S, 1);
    char *anxt = (char *)calloc((size_t)S, 1);
    int *back = (int *)malloc(sizeof(int) * (size_t)S * (size_t)(T > 0 ? T : 1));
    long ties_seen = 0;
    float current_best_logp = -1e10f;
    cur[0] = 0;
    acur[0] = 1;
    for (long i = 0; i < T; i++) {
        float next_best_logp = -1e10f;
        memset(anxt, 0, (size_t)S);
        for (int q = 0; q < S; q++) {
            int s = order == 0 ? S - 1 - q : q;
            if (!acur[s]) continue;
            float logp = cur[s];
            if (logp < current_best_logp - beam_width) continue;
            for (int ti = 0; ti < ntr[s]; ti++) {
                int target = s + ti;
                logp += trlogp[s * 3 + ti];
                logp += emis[i * ldE + (tied ? tied[target] : target)];
                if (anxt[target]) {
                    if (logp == nxt[target]) ties_seen++;
                    if (logp > nxt[target]) {
                        nxt[target] = logp;
                        next_best_logp = fmaxf(next_best_logp, logp);
                        back[i * S + target] = s;
                    }
                } else {
                    anxt[target] = 1;
                    nxt[target] = logp;
                    next_best_logp = fmaxf(next_best_logp, logp);
                    back[i * S + target] = s;
                }
            }
        }
        current_best_logp = next_best_logp;
        float *tf = cur; cur = nxt; nxt = tf;
        char *tc = acur; acur = anxt; anxt = tc;
    }
    int rc = 0;
    int best = S - 1;
    if (!acur[best]) {
        rc = 1;
    } else {
        float nlp = cur[best] / (float)T;
        if (final_logp) *final_logp = cur[best];
        if (prevp) *prevp = nlp;
        for (long i = T - 1; i >= 0; i--) {
            path[i] = best;
            best = back[i * S + best];
        }
    }
    if (nties) *nties = ties_seen;
    free(cur); free(nxt); free(acur); free(anxt); free(back);
    return rc;
}

/* hmm_state::set_adv_prob / get_adv_prob  src/hmm.cpp:26-35 */
static void set_adv_prob(int ntr_s, float *tr, float p) {
    float avail = ntr_s == 3 ? (1 - ORC_SIL_SKIP_P) : 1;
    tr[0] = logf((1 - p) * avail);
    tr[1] = logf(p * avail);
}
static float get_adv_prob(const float *tr) { return expf(tr[1]); }

/* hmm::train_transition_probabilities  src/hmm.cpp:229-243 */
void orc_train_transitions(const int *path, long T, const int *ntr, float *trlogp) {
    if (T <= 0) return;
    int curr = path[0];
    int count = 1;
    for (long it = 1; it < T; it++) {
        if (path[it] != curr) {
            float inv = 1.0f / (float)count;
            float m = (inv < ORC_MAX_ADV_PROB) ? inv : ORC_MAX_ADV_PROB; /* std::min(a,b) = b<a ? b : a */
            float p = ORC_HMM_BLEND_FAC * m + (1 - ORC_HMM_BLEND_FAC) * get_adv_prob(&trlogp[curr * 3]);
            set_adv_prob(ntr[curr], &trlogp[curr * 3], p);
            count = 0;
            curr = path[it];
        }
        count++;
    }
}

/* The statistics half of one EM iteration  src/hmm.cpp:274-341 (everything between the
 * Viterbi/transition loops and the pointer swap).
 *  A = model the statistics are taken under (hmm::acm), B = model being rebuilt
 *  (hmm::temp_acm); both [G][M][D] long double mean/var + [G][M] float weight/logwt.
 *  gid[Ttot] = tied state of every aligned frame, all utterances concatenated.
 *  States never visited get zetasum 0 => 0/0 = NaN (reference behaviour). */
void orc_em_statistics(const float *feats, int ld, long Ttot, const int *gid, int G, int M, int D,
                       const long double *A_mean, const long double *A_var, const float *A_weight,
                       const float *A_logwt, long double *B_mean, long double *B_var, float *B_weight,
                       float *B_logwt, long double *B_zetasum /*[G][M] out*/, float blend_keep) {
    size_t GM = (size_t)G * M;
    long double *zg = (long double *)calloc((size_t)G, sizeof(long double));
    /* clear_zetas; pass 1  src/hmm.cpp:274-287 */
    for (size_t o = 0; o < GM; o++) B_zetasum[o] = 0;
    for (long t = 0; t < Ttot; t++) {
        int g = gid[t];
        for (int m = 0; m < M; m++) {
            size_t o = (size_t)g * M + m;
            float lp = orc_gaussian_logprob(feats + t * ld, A_mean + o * D, A_var + o * D, A_logwt[o], D);
            B_zetasum[o] += (long double)A_weight[o] * expl((long double)lp);
        }
    }
    /* sum_zetas; clear; weights  src/hmm.cpp:288-296, src/gmm.cpp:61-67,125-130 */
    for (int g = 0; g < G; g++) {
        zg[g] = 0;
        for (int m = 0; m < M; m++) zg[g] += B_zetasum[(size_t)g * M + m];
    }
    for (size_t o = 0; o < GM; o++) {
        for (int e = 0; e < D; e++) B_mean[o * D + e] = B_var[o * D + e] = 0;
        B_weight[o] = (float)(B_zetasum[o] / zg[o / M]);
        B_logwt[o] = logf(B_weight[o]);
    }
    /* pass 2: means  src/hmm.cpp:298-317 */
    for (long t = 0; t < Ttot; t++) {
        int g = gid[t];
        const float *fv = feats + t * ld;
        for (int k = 0; k < M; k++) {
            size_t o = (size_t)g * M + k;
            float lp = orc_gaussian_logprob(fv, A_mean + o * D, A_var + o * D, A_logwt[o], D);
            long double zeta = (long double)A_weight[o] * expl((long double)lp);
            for (int e = 0; e < D; e++) B_mean[o * D + e] += zeta * (long double)fv[e];
        }
    }
    for (size_t o = 0; o < GM; o++)
        for (int e = 0; e < D; e++) B_mean[o * D + e] /= B_zetasum[o];
    /* pass 3: variances (needs the new means)  src/hmm.cpp:319-339 */
    for (long t = 0; t < Ttot; t++) {
        int g = gid[t];
        const float *fv = feats + t * ld;
        for (int k = 0; k < M; k++) {
            size_t o = (size_t)g * M + k;
            float lp = orc_gaussian_logprob(fv, A_mean + o * D, A_var + o * D, A_logwt[o], D);
            long double zeta = (long double)A_weight[o] * expl((long double)lp);
            for (int e = 0; e < D; e++) {
                float diff = (float)((long double)fv[e] - B_mean[o * D + e]);
                B_var[o * D + e] += zeta * (long double)diff * (long double)diff;
            }
        }
    }
    for (size_t o = 0; o < GM; o++)
        for (int e = 0; e < D; e++) B_var[o * D + e] /= B_zetasum[o];
    /* blend  src/hmm.cpp:341, src/gmm.cpp:82-88: this = this*f + old*(1-f) */
    float f = blend_keep;
    for (size_t o = 0; o < GM; o++) {
        for (int e = 0; e < D; e++) {
            B_mean[o * D + e] = B_mean[o * D + e] * f + A_mean[o * D + e] * (1 - f);
            B_var[o * D + e] = B_var[o * D + e] * f + A_var[o * D + e] * (1 - f);
        }
        B_weight[o] = B_weight[o] * f + A_weight[o] * (1 - f);
        B_logwt[o] = logf(B_weight[o]);
    }
    free(zg);
}

/* state_model::augment over final paths + operator()  src/hmm.cpp:39-51, 350-356.
 * cls[t] = tied class of the path state at frame t, adv[t] = exp(tr[1]) of that state. */
void orc_state_model(const int *cls, const float *adv, long Ttot, int C, float *probsum, int *count) {
    for (int c = 0; c < C; c++) { probsum[c] = 0; count[c] = 0; }
    for (long t = 0; t < Ttot; t++) {
        probsum[cls[t]] += adv[t];
        count[cls[t]] += 1;
    }
}

/* ------------------------------------------------------------------------------------
 * Faithful-cost end-to-end for the CPU baseline: MFCC + features + scoring of the
 * utterance's own states + Viterbi, one utterance.  Same arithmetic as above.
 * ---------------------------------------------------------------------------------- */
long orc_pipeline_utterance(float *x, long L, int wl, int hop, const orc_filter *fb, int nfilt, int nparams,
                            float alpha, int WL, int G, int M, int D, const long double *mean,
                            const long double *var, const float *logwt, int mode, const int *tied, int S,
                            const int *ntr, const float *trlogp, int *path, float *final_logp) {
    orc_normalize_volume(x, L, 1.0f);
    long nmax = orc_num_frames(L, wl, hop);
    float *ceps = (float *)malloc(sizeof(float) * (size_t)nmax * nparams);
    long n = orc_write_features(x, L, wl, hop, fb, nfilt, nparams, alpha, ceps);
    int ld = 3 * nparams;
    float *feats = (float *)malloc(sizeof(float) * (size_t)n * ld);
    orc_read_features(ceps, n, nparams, WL, feats);
    /* the reference evaluates emissions lazily inside Viterbi (src/hmm.cpp:169): score the
       S states of this utterance only, as columns 0..S-1 */
    float *emis = (float *)malloc(sizeof(float) * (size_t)n * S);
    for (long t = 0; t < n; t++)
        for (int s = 0; s < S; s++) {
            size_t g = (size_t)tied[s];
            if (mode == 0) {
                emis[t * S + s] = orc_gaussian_logprob(feats + t * ld, mean + g * M * D, var + g * M * D, logwt[g * M], D);
            } else {
                double res = 0;
                for (int m = 0; m < M; m++)
                    res += exp((double)orc_gaussian_logprob(feats + t * ld, mean + (g * M + m) * D,
                                                            var + (g * M + m) * D, logwt[g * M + m], D));
                emis[t * S + s] = (float)log(res);
            }
        }
    float prevp = 0;
    long nt = 0;
    orc_viterbi(emis, S, NULL, n, S, ntr, trlogp, 1e10f, 0, path, final_logp, &prevp, &nt);
    free(ceps); free(feats); free(emis);
    (void)G;
    return n;
}

/* ------------------------------------------------------------------------------------
 * General-topology HMMs: the lexicon HMM and hmm::viterbi on it  (src/hmm.cpp:92-129, 150-224)
 * PARITY UNPINNED: main() never calls viterbi on the lexicon HMM, the selection of the best final
 * state is commented out in the reference (src/hmm.cpp:196-209) and no output of it exists.
 * ---------------------------------------------------------------------------------- */

/* hmm::hmm(pronlex&, unigram_model&, acoustic_model*, state_model*)  src/hmm.cpp:92-129, USE_SUBPHONES=false.
 * Words in the iteration order of the reference's map<string, pronlist> (sorted by spelling); word w has phones
 * phones[phone_off[w] .. phone_off[w+1]) (pronunciations[0]) and unigram log-probability uni_logp[w]
 * (unigram_model::get, src/langmodel.cpp:21-27); the start transition carries lmsf * uni_logp[w] (LMSF = 5, src/hmm.hpp:33).
 * State 0 is the start state (ctx SIL,SIL,SIL from hmm_state::hmm_state, src/hmm.cpp:13-15); then per word its phone
 * states and a terminating silence.  Outputs: tr_off[n_states+1], tr_tgt/tr_logp[n_tr] (CSR, transitions in push_back
 * order), ctx[n_states][3], word_of[n_states] (-1 for state 0).  Returns n_states; counts only when tr_off == NULL. */
long orc_build_lexicon_hmm(int n_words, const int *phone_off, const int *phones, const float *uni_logp, float lmsf,
                           int *tr_off, int *tr_tgt, float *tr_logp, int *ctx, int *word_of, long *n_tr_out) {
    long n_states = 1, n_tr = n_words;
    for (int w = 0; w < n_words; w++) {
        int len = phone_off[w + 1] - phone_off[w];
        n_states += len + 1;
        n_tr += 2 * len + 1 + 2;   /* 2 per phone state, +1 for the last one's return, 2 for the silence */
    }
    if (n_tr_out) *n_tr_out = n_tr;
    if (!tr_off) return n_states;
    /* state 0's transitions first (one per word, in word order), then the states in index order */
    int *first = (int *)malloc(sizeof(int) * (size_t)(n_words > 0 ? n_words : 1));
    long s = 1;
    for (int w = 0; w < n_words; w++) { first[w] = (int)s; s += phone_off[w + 1] - phone_off[w] + 1; }
    long q = 0;
    tr_off[0] = 0;
    for (int w = 0; w < n_words; w++) { tr_tgt[q] = first[w]; tr_logp[q] = lmsf * uni_logp[w]; q++; }
    ctx[0] = ctx[1] = ctx[2] = 0;
    word_of[0] = -1;
    for (int w = 0; w < n_words; w++) {
        const int *p = phones + phone_off[w];
        int len = phone_off[w + 1] - phone_off[w];
        int s0 = first[w], e = s0 + len;           /* e = terminating silence */
        for (int i = s0; i < e; i++) {
            int last = i == e - 1;
            tr_off[i] = (int)q;
            tr_tgt[q] = i;     tr_logp[q] = last ? ORC_LOG_THIRD : ORC_LOG_HALF; q++;
            tr_tgt[q] = i + 1; tr_logp[q] = last ? ORC_LOG_THIRD : ORC_LOG_HALF; q++;
            if (last) { tr_tgt[q] = 0; tr_logp[q] = ORC_LOG_THIRD; q++; }
            ctx[i * 3 + 0] = i == s0 ? 0 : p[i - s0 - 1];
            ctx[i * 3 + 1] = p[i - s0];
            ctx[i * 3 + 2] = i + 1 == e ? 0 : p[i - s0 + 1];
            word_of[i] = w;
        }
        tr_off[e] = (int)q;
        tr_tgt[q] = e; tr_logp[q] = ORC_LOG_HALF; q++;
        tr_tgt[q] = 0; tr_logp[q] = ORC_LOG_HALF; q++;
        ctx[e * 3 + 0] = len > 0 ? p[len - 1] : 0;   /* states[e-1].ph.first (the start state's SIL for an empty word) */
        ctx[e * 3 + 1] = 0;
        ctx[e * 3 + 2] = 0;
        word_of[e] = w;
    }
    tr_off[n_states] = (int)q;
    free(first);
    return n_states;
}

/* hmm::viterbi  src/hmm.cpp:150-224 on an arbitrary transition list (CSR as above).
 *  emis[t*ldE + tied[s]] = (*acm)(*fvs[t], states[s].ctx).
 *  `logp` accumulates across a source's transitions exactly as written (src/hmm.cpp:164-169).
 *  order: 0 = active sources visited in descending index, 1 = ascending (the reference: unordered_map order).
 *  final_mode 0: the path must end in the last state (src/hmm.cpp:210-215, as compiled);
 *  final_mode 1: the commented-out recognition variant (src/hmm.cpp:196-209): best = the active state with the largest
 *                value > -1e10f, visited in `order` with strict >; none -> status 1 (the reference would `throw 1729`).
 *  Returns 0 ok, 1 NO PATH. */
int orc_viterbi_graph(const float *emis, long ldE, const int *tied, long T, int n_states, const int *tr_off, const int *tr_tgt,
                      const float *tr_logp, float beam_width, int order, int final_mode, int *path, float *final_logp,
                      int *final_state, long *nties) {
    float *cur = (float *)malloc(sizeof(float) * (size_t)n_states);
    float *nxt = (float *)malloc(sizeof(float) * (size_t)n_states);
    char *acur = (char *)calloc((size_t)n_states, 1);
    char *anxt = (char *)calloc((size_t)n_states, 1);
    int *back = (int *)malloc(sizeof(int) * (size_t)n_states * (size_t)(T > 0 ? T : 1));
    long ties_seen = 0;
    float current_best_logp = -1e10f;
    cur[0] = 0;
    acur[0] = 1;
    for (long i = 0; i < T; i++) {
        float next_best_logp = -1e10f;
        memset(anxt, 0, (size_t)n_states);
        for (int q = 0; q < n_states; q++) {
            int s = order == 0 ? n_states - 1 - q : q;
            if (!acur[s]) continue;
            float logp = cur[s];
            if (logp < current_best_logp - beam_width) continue;
            for (int ti = tr_off[s]; ti < tr_off[s + 1]; ti++) {
                int target = tr_tgt[ti];
                logp += tr_logp[ti];
                logp += emis[i * ldE + tied[target]];
                if (anxt[target]) {
                    if (logp == nxt[target]) ties_seen++;
                    if (logp > nxt[target]) {
                        nxt[target] = logp;
                        next_best_logp = fmaxf(next_best_logp, logp);
                        back[i * n_states + target] = s;
                    }
                } else {
                    anxt[target] = 1;
                    nxt[target] = logp;
                    next_best_logp = fmaxf(next_best_logp, logp);
                    back[i * n_states + target] = s;
                }
            }
        }
        current_best_logp = next_best_logp;
        float *tf = cur; cur = nxt; nxt = tf;
        char *tc = acur; acur = anxt; anxt = tc;
    }
    int rc = 0, best = -1;
    if (final_mode == 0) {
        best = n_states - 1;
        if (!acur[best]) rc = 1;
    } else {
        float best_logp = -1e10f;
        for (int q = 0; q < n_states; q++) {
            int s = order == 0 ? n_states - 1 - q : q;
            if (acur[s] && cur[s] > best_logp) { best_logp = cur[s]; best = s; }
        }
        if (best == -1) rc = 1;
    }
    if (rc == 0) {
        if (final_logp) *final_logp = cur[best];
        if (final_state) *final_state = best;
        for (long i = T - 1; i >= 0; i--) {
            path[i] = best;
            best = back[i * n_states + best];
        }
    }
    if (nties) *nties = ties_seen;
    free(cur); free(nxt); free(acur); free(anxt); free(back);
    return rc;
}

/* hmm::update_states  src/hmm.cpp:361-366 with state_model::operator()  src/hmm.cpp:39-44 on CSR transitions:
 * every state but the start state takes the pooled advance probability of its tied class (0.5 where nothing was seen).
 * Every lexicon state has at least two transitions (src/hmm.cpp:108-122), so set_adv_prob's tr[1] exists. */
void orc_update_states(int n_states, const int *tr_off, float *tr_logp, const int *cls, const float *probsum, const int *count) {
    for (int i = 1; i < n_states; i++) {
        int c = cls[i];
        float p = (c < 0 || count[c] == 0) ? 0.5f : probsum[c] / count[c];
        int ntr = tr_off[i + 1] - tr_off[i];
        if (ntr < 2) continue;   /* the reference would write tr[1] out of bounds */
        set_adv_prob(ntr, tr_logp + tr_off[i], p);
    }
}
