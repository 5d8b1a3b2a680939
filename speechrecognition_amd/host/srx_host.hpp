// srx_host.hpp -- the reference's C++ class and function surface for the acoustic hot path, as a thin
// host layer over the C ABI of libsrx.so (include/srx.h).  A caller written against the reference's
// headers (the MFCC and TRAIN blocks of src/main.cpp:25-92, src/utterance.cpp:14-28) compiles against this
// header instead; every computation on the path runs in HIP kernels behind srx.h.  There is no CPU
// implementation of the path in this layer: without a gfx950 device the first call throws srx_error.
//
// What is mirrored (reference file:line in the comment of each declaration):
//   mfcc.hpp   featurevec, filter, filterbank, mel_filterbank, mfcc, write_features, read_features
//   clip.hpp   ClipBuffer, ClipArrayBuffer, Clip (view, slide, rmsAmplitude, normalizeVolume, preemphasis, *=)
//   wav.hpp    read_wav, write_wav, wav_io_exception
//   utils.hpp  readlines, hamming_window
//   pronounce.hpp  phone::phone, phone::context, phone::ties, phspec, findPhone, pronlist, pronlex, pronouncer
//   gmm.hpp    gaussian, gmm, acoustic_model
//   hmm.hpp    hmm_state, state_model, hmm (training-HMM constructor, viterbi, train_transition_probabilities,
//              train, update_states)
//   utterance.hpp  utterance, mean, variance
// Not mirrored (outside the hot path): ofApp, cluster, the bigram model, the microphone
// (ofSoundBuffer) ClipBuffer constructors, hanning_window / rand_gaussian / random_fv / min / max / argmax / convolve.
// Every per-frame function of the path -- Clip::partial_dft / dft, filter::operator(), filterbank::operator(), log(buf, len),
// dct, featurevec::compute_deltas / compute_delta2s, gaussian::operator() / logprob / prob, gmm::operator() -- is one call
// into a reference-order kernel of libsrx (csrc/srx_frameops.hip); the fused batch kernels stay the fast path.
// The EM bookkeeping of gaussian / gmm / acoustic_model (clear, clear_zetas, sum_zetas, divide_means, divide_variances,
// blend) works on the host mirror in long double, as the reference does: it is O(parameters), never O(frames), and the
// 80-bit type has no device form; hmm::train does the same steps on the device (srx_em_means_dev / srx_em_finish_dev).
//
// Additions with no counterpart in the reference are marked [batch]: they hand a whole ragged batch to one
// kernel launch, which is where the throughput comes from (the reference loops utterance by utterance).
#ifndef SRX_HOST_HPP
#define SRX_HOST_HPP

#include <cstdint>
#include <exception>
#include <istream>
#include <list>
#include <map>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

#include "../../include/srx.h"

#ifndef FV_LEN
#define FV_LEN 13  // src/mfcc.hpp:16: dimensions the GMMs score (the static cepstra)
#endif
#ifndef NPARAMS
#define NPARAMS 13  // src/mfcc.hpp:17: cepstra per frame
#endif
#define USE_SUBPHONES false          // src/hmm.hpp:23
#define LOG_HALF -0.6931471805f      // src/hmm.hpp:25
#define LOG_THIRD -1.09861228866f    // src/hmm.hpp:26
#define SIL_SKIP_P 0.3f              // src/hmm.hpp:28
#define MAX_ADV_PROB 0.55f           // src/hmm.hpp:29
#define HMM_BLEND_FAC 0.2f           // src/hmm.hpp:30
#define GMM_BLEND_FAC 0.2f           // src/hmm.hpp:31
#define LMSF 5                       // src/hmm.hpp:33: language model scaling factor
#ifndef EM_ITERS
#define EM_ITERS 80                  // src/hmm.cpp:247
#endif

// ------------------------------------------------------------------------------------------ runtime
// Errors of the C ABI surface as this exception (the reference itself throws `int` for I/O failures and
// wav_io_exception objects; those conventions are kept where the reference has them).
class srx_error : public std::exception {
public:
    int code;
    std::string msg;
    srx_error(int c, const std::string &m) : code(c), msg(m) {}
    const char *what() const throw() { return msg.c_str(); }
};

namespace srx {
srx_ctx *context();            // the process-wide context (device SRX_DEVICE or 0), created on first use
// Multi-GPU (no counterpart in the reference, which is one process): one process per GPU, each holding a shard of the
// utterances; hmm::train, mean and variance complete their sums over all ranks through this communicator.
void init_distributed();       // joins the launch described by RANK / WORLD_SIZE / LOCAL_RANK (+ SRX_COMM_FILE); no-op without
srx_comm *communicator();      // world-1 communicator (collectives are no-ops) unless init_distributed joined a launch
// The id rendezvous of init_distributed, as three host-only steps (file I/O; tested without a GPU).  The file holds a
// magic word, a nonce that every rank of ONE launch computes alike, and rank 0's SRX_COMM_ID_BYTES: a file left behind by
// an earlier launch (or by anything else) does not carry this launch's nonce and is ignored by the waiting ranks.
uint64_t launch_nonce();       // from SRX_COMM_NONCE, TORCHELASTIC_RUN_ID, MASTER_ADDR, MASTER_PORT, WORLD_SIZE
void publish_comm_id(const std::string &file, uint64_t nonce, const unsigned char *id);      // rank 0: unlink, write .tmp, rename
bool await_comm_id(const std::string &file, uint64_t nonce, unsigned char *id, int timeout_ms);   // ranks > 0
int rank();
int world();
std::vector<int> shard_utterances(const std::vector<double> &costs, int world, int rank);   // this rank's utterance indices
void set_verbose(bool on);     // the reference prints constantly (e.g. src/mfcc.cpp:279, src/hmm.cpp:217); default off
bool verbose();
void seed_random(unsigned seed);   // generator behind gaussian::initialize (the reference draws from ofRandom there)
double next_random();              // uniform in [0, 1)
}  // namespace srx

// ------------------------------------------------------------------------------------------ utils.hpp
std::vector<std::string> readlines(const char *fname);                                   // src/utils.cpp:40-51
inline std::vector<std::string> readlines(const std::string &f) { return readlines(f.c_str()); }
float hamming_window(int x, int len);                                                    // src/utils.cpp:21-23
float rectangular_window(int x, int len);                                                // src/utils.cpp:13-15
// log(T *buf, int len) for T = float, src/utils.hpp:55-60: buf[i] = log(buf[i] + 1), in place           (GPU: k_log1p)
void log(float *buf, int len);

// ------------------------------------------------------------------------------------------ clip.hpp
class ClipBuffer {                                                                       // src/clip.hpp:22-30
public:
    virtual float &operator[](int) = 0;
    virtual int length() const = 0;
    virtual ~ClipBuffer();
    virtual float *data() = 0;  // contiguous samples (both reference buffers are contiguous)
};

class ClipArrayBuffer : public ClipBuffer {                                              // src/clip.hpp:32-46
    int n;
public:
    float *buf;
    ClipArrayBuffer(int n);
    ~ClipArrayBuffer();
    float &operator[](int i);
    int length() const;
    float *data();
};

class ClipVectorBuffer : public ClipBuffer {                                             // src/clip.hpp:49-64
    std::vector<float> samples;
public:
    ClipVectorBuffer();
    void append(float *buf, int n);           // src/clip.cpp:62-66 (the ofSoundBuffer overload is not mirrored)
    void ensure(int n);
    float &operator[](int i);
    int length() const;
    float *data();
};

class Clip {                                                                             // src/clip.hpp:66-96
public:
    ClipBuffer *buf;
    int start, end;
    float (*window)(int, int);
    Clip();
    Clip(ClipBuffer &b);
    Clip(ClipBuffer &b, int start, int end);
    Clip(const Clip &c, int start, int end);
    bool slide(int n);                        // src/clip.cpp:108-112
    float &operator[](int idx);               // src/clip.cpp:118-120
    float operator[](int idx) const;          // src/clip.cpp:123-125 (applies the window)
    int length() const;
    void operator+=(Clip &c);                 // src/clip.cpp:128-143: sample-wise over min(length, c.length); the right-hand
    void operator-=(Clip &c);                 //   side is read through ITS window.  Not on the hot path (the reference never
    void operator*=(Clip &c);                 //   calls them from main): plain host loops.
    void operator*=(float f);                 // src/clip.cpp:144-148           (GPU)
    void normalizeVolume(float targetRMS);    // src/clip.cpp:151-154           (GPU: k_sumsq_scale + k_apply_scale)
    double rmsAmplitude() const;              // src/clip.cpp:157-164           (GPU: k_rms)
    void preemphasis(float alpha);            // src/clip.cpp:199-205           (GPU: k_preemph)
    void dft(float *out, bool stack = false) const;                  // src/clip.cpp:166-168
    // src/clip.cpp:174-194: the first `len` magnitudes over the first half of the window, through this clip's window
    // function; len > length()/2 also writes the mirrored bins; `stack` adds to `out`.          (GPU: k_partial_dft_exact)
    void partial_dft(float *out, int len, bool stack = false) const;
};

// ------------------------------------------------------------------------------------------ wav.hpp
class wav_io_exception : public std::exception {                                         // src/wav.hpp:18-27
public:
    const char *m;
    wav_io_exception(const char *msg) : m(msg) {}
    virtual const char *what() const throw() { return m; }
};
extern wav_io_exception wavex_not_riff, wavex_not_wave, wavex_not_pcm, wavex_bad, wavex_fopen_fail;
Clip read_wav(std::string file);                                                         // src/wav.cpp:55-92
void write_wav(Clip c, int sps, short bps, std::string fname);                           // src/wav.cpp:94-134

// ------------------------------------------------------------------------------------------ mfcc.hpp
class featurevec {                                                                       // src/mfcc.hpp:26-46
public:
    float coeffs[NPARAMS];
    float deltas[NPARAMS];
    float delta2s[NPARAMS];
    featurevec();
    featurevec(float *c);
    float &operator[](int i);                 // 0..3*NPARAMS-1 across the three arrays, src/mfcc.cpp:30-38
    void operator+=(featurevec &);            // first FV_LEN entries, src/mfcc.cpp:40-44
    void operator*=(float);
    float dist(featurevec &f);                                        // src/mfcc.cpp:52-59   (k-means only: host)
    featurevec normalize(featurevec &mu, featurevec &sigma);          // src/mfcc.cpp:64-70   (k-means only: host)
    void compute_deltas(std::vector<featurevec *> fvs, int i);        // src/mfcc.cpp:73-81   (GPU: k_delta)
    void compute_delta2s(std::vector<featurevec *> fvs, int i);       // src/mfcc.cpp:84-92   (GPU: k_delta)
};
// src/mfcc.cpp:95-102: deltas of every vector, then delta-deltas of every vector                   (GPU: k_features, WL = -1)
void compute_deltas(std::vector<featurevec *> fvs);

class filter {                                                                           // src/mfcc.hpp:50-60
public:
    float center, width, height;
    filter(float c, float w, int dft_bins = 256, int sps = 16000);   // src/mfcc.cpp:117-128
    float get(int bin) const;                                        // src/mfcc.cpp:132-140
    float operator()(float *spectrum);                               // src/mfcc.cpp:143-149   (GPU: k_filterbank_exact)
};

class filterbank {                                                                       // src/mfcc.hpp:62-72
public:
    std::vector<filter> filters;
    filterbank();
    int length() const;
    float *operator()(float *spectrum);       // src/mfcc.cpp:161-167: new float[length()], caller delete[]s   (GPU: k_filterbank_exact)
};

// src/mfcc.cpp:171-177 (start + i*n as the reference has it, not start + i*step)
filterbank *linear_filterbank(float start, float step, int n, float width, int dft_bins = 256, int sps = 16000);
// src/mfcc.cpp:205-214: new float[nparams], caller delete[]s                                       (GPU: k_dct_exact)
float *dct(float *filtered, int len, int nparams);

filterbank *mel_filterbank(float end, int n, int dft_bins = 256, int sps = 16000);       // src/mfcc.cpp:190-201
// One frame: `c` is the window (c.length() = window length).  Returns new float[NPARAMS]; caller delete[]s
// (src/mfcc.cpp:219-232).  `noise` is ignored like in the reference (the subtraction is commented out).
float *mfcc(Clip c, filterbank &fb, float *noise);
// src/mfcc.cpp:235-275: pre-emphasis (mutates the samples behind `c`, like the reference), framing, MFCC of
// every frame, file = int32 n + n x NPARAMS float32.  throw 1 if the file cannot be opened.
void write_features(Clip c, unsigned window_len, unsigned frame_shift, filterbank &fb, std::string fname);
// src/mfcc.cpp:278-314: read the file, smooth (WL), deltas, delta-deltas.  throw 1 if it cannot be opened.
std::vector<featurevec *> read_features(std::string fname, int WL = 2);

// [batch] write_features for many clips in one launch; clips are NOT mutated unless mutate_samples.
void write_features_batch(std::vector<Clip> &clips, unsigned window_len, unsigned frame_shift, filterbank &fb,
                          const std::vector<std::string> &fnames, bool normalize_volume, bool mutate_samples = false);
// [batch] the MFCC of every frame of every clip, concatenated [sum frames][NPARAMS]; frame_off gets n+1 offsets.
std::vector<float> mfcc_batch(std::vector<Clip> &clips, unsigned window_len, unsigned frame_shift, filterbank &fb,
                              bool normalize_volume, std::vector<int64_t> &frame_off);

// ------------------------------------------------------------------------------------------ pronounce.hpp (subset)
namespace phone {
static const int NUM_PH = 40;
enum phone {                                                                             // src/pronounce.hpp:24-28
    SIL = 0, AA, AE, AH, AO, AW, AY, B, CH, D, DH, EH, ER, EY, F, G, HH, IH, IY, JH, K, L, M, N, NG, OW, OY, P, R, S,
    SH, T, TH, UH, UW, V, W, Y, Z, ZH
};
enum subphone { BEGIN = 0, MIDDLE, END };
class context {                                                                          // src/pronounce.hpp:36-48
public:
    phone prev, curr, next;
    context(phone c) : prev(SIL), curr(c), next(SIL) {}
    context(phone p, phone c, phone n) : prev(p), curr(c), next(n) {}
    bool operator<(const context &c) const;                                              // src/pronounce.cpp:58-66
    context operator++();                                                                // src/pronounce.cpp:69-88
};
class ties {                                                                             // src/pronounce.hpp:52-72
public:
    enum mode { IDENT, PHONE_CLASSES, NULL_CONTEXT };
    std::map<phone, phone> c;
    ties(mode m);                                                                        // src/pronounce.cpp:28-51
    context operator()(context ctx);                                                     // src/pronounce.cpp:54-56
};
extern const char *names[];
}  // namespace phone
typedef std::pair<phone::phone, phone::subphone> phspec;
phone::phone findPhone(std::string s);                                                   // src/pronounce.cpp:16-26

// Text front end of the training path (srx_text.cpp): host string work that picks each utterance's states.
class pronlist {                                                                         // src/pronounce.hpp:84-91
public:
    std::vector<std::vector<phone::phone>> pronunciations;
    pronlist();
    void add(std::vector<phone::phone> pron);
};
class pronlex {                                                                          // src/pronounce.hpp:94-106
public:
    std::map<std::string, pronlist> pr;
    pronlex();
    pronlex(std::istream &in);                             // CMU Pronouncing Dictionary format, src/pronounce.cpp:105-138
    void add(std::string w, std::vector<phone::phone> pron);
    pronlist *get(std::string w);                          // NULL when absent, src/pronounce.cpp:141-147
};
class pronouncer {                                                                       // src/pronounce.hpp:110-121
public:
    pronlex main, prefixes, suffixes;
    pronouncer(const char *main_fn, const char *pre_fn, const char *suf_fn);
    std::vector<phone::phone> pronounce_word(std::string w);   // throws 1 when it gives up, src/pronounce.cpp:178-257
    std::vector<phone::phone> pronounce(std::string s);        // src/pronounce.cpp:260-278
};

// ------------------------------------------------------------------------------------------ langmodel.hpp (subset)
// Unigram log-probabilities: what the lexicon HMM's start transitions are built from (src/hmm.cpp:108).
class unigram_model {                                                                    // src/langmodel.hpp:34-52
public:
    std::map<std::string, float> probs;
    long total_count;
    float logtotal;
    unigram_model(long total_count);                          // an empty table: every word gets -logtotal
    unigram_model(std::istream &in, long total_count);        // one "word count" pair per line
    void put(std::string w, long count);                      // src/langmodel.cpp:16-19
    float get(std::string w) const;                           // src/langmodel.cpp:22-28
};
// Bigram log-probabilities with unigram back-off (src/langmodel.hpp:48-65, src/langmodel.cpp:31-44).  The reference builds none
// (its main() never constructs one); here for callers that switch over with code that does.
class bigram_model {
public:
    std::map<std::string, float> probs;                       // key "w1^w2"
    long total_count;
    float logtotal;
    unigram_model *uni;                                       // the model to fall back on for bigrams never seen
    bigram_model(unigram_model *u, long total_count);
    bigram_model(std::istream &in, unigram_model *u, long total_count);   // one "w1 w2 count" triple per line
    void put(std::string w1, std::string w2, long count);
    float get(std::string w1, std::string w2) const;          // log P(w2 | w1)
};

// ------------------------------------------------------------------------------------------ gmm.hpp
typedef long double prob_t;
typedef float logprob_t;

class gaussian {                                                                         // src/gmm.hpp:25-46
public:
    float weight, logwt;
    prob_t mean[FV_LEN];
    prob_t var[FV_LEN];
    prob_t zetasum;
    gaussian();
    void initialize(featurevec &mu, featurevec &sigma2);   // src/gmm.cpp:26-33 (srx::seed_random stands in for ofRandom's generator)
    void set_wt(float wt);                                 // src/gmm.cpp:35-38
    logprob_t operator()(featurevec &fv);                  // src/gmm.cpp:41-49   (GPU: k_gaussian_exact)
    prob_t prob(featurevec &fv);                           // src/gmm.cpp:57-59: exp((prob_t) of the above)
    logprob_t logprob(featurevec &fv);                     // src/gmm.cpp:52-54
    void clear();                                          // src/gmm.cpp:61-67
    void divide_means();                                   // src/gmm.cpp:69-73
    void divide_variances();                               // src/gmm.cpp:75-79
    void blend(gaussian &g, float f);                      // src/gmm.cpp:82-88
};

class gmm {                                                                              // src/gmm.hpp:48-72
public:
    std::vector<gaussian> &gaussians;
    prob_t zetasum;
    gmm(int n);
    gmm(gmm *g);
    logprob_t operator()(featurevec &fv);                  // src/gmm.cpp:99-100: component 0, as the reference is compiled
    void initialize(featurevec &, featurevec &);
    void clear();                                          // src/gmm.cpp:119-154
    void sum_zetas();
    void clear_zetas();
    void divide_means();
    void divide_variances();
    void blend(gmm *g, float f);
};

typedef std::map<phone::context, gmm *>::iterator acm_iter;

class acoustic_model {                                                                   // src/gmm.hpp:75-94
public:
    std::map<phone::context, gmm *> mixtures;   // host mirror of the parameters (authoritative between calls)
    phone::ties &ties;
    acoustic_model(phone::ties &t, int nmix = 1);          // src/gmm.cpp:160-169
    acoustic_model(acoustic_model *acm);                   // src/gmm.cpp:171-175
    ~acoustic_model();
    logprob_t operator()(featurevec &fv, phone::context ph);   // src/gmm.cpp:184-186 (one frame, one state; GPU)
    void initialize(featurevec &, featurevec &);           // src/gmm.cpp:177-181
    void clear();                                          // src/gmm.cpp:188-223: each loops over the mixtures (host mirror)
    void sum_zetas();                                      //   (prints the maximum zeta sum when srx::set_verbose(true))
    void clear_zetas();
    void divide_means();
    void divide_variances();
    void blend(acoustic_model *a, float f);                // src/gmm.cpp:225-229 (host mirror)
    acm_iter begin();
    acm_iter end();

    // --- device side ---
    int num_states() const { return (int)mixtures.size(); }
    int num_mix() const;
    int tied_index(phone::context ph);                     // column of the emission matrix for ties(ph)
    srx_model *device();                                   // uploads the host mirror if it changed
    void download();                                       // device -> host mirror
    void mark_host_dirty() { host_dirty = true; }
    // [batch] every frame x every tied state -> [T][num_states] (component 0 like gmm::operator(), or log-sum-exp)
    std::vector<float> score(const std::vector<featurevec *> &fvs, bool logsumexp = false, bool exact = false);

private:
    srx_model *dev;
    bool host_dirty;
    std::map<phone::context, int> index;
    void build_index();
};

// ------------------------------------------------------------------------------------------ hmm.hpp
class hmm_state;
typedef std::pair<hmm_state *, float> transition;
typedef std::list<hmm_state *> path;

class hmm_state {                                                                        // src/hmm.hpp:40-53
public:
    phspec ph;
    phone::context ctx;
    std::vector<transition> tr;
    const std::string *word;
    hmm_state(phspec p);
    float get_logprob(int i) const;
    void set_prob(int i, float p);
    float get_adv_prob() const;                            // src/hmm.cpp:26-28
    void set_adv_prob(float p);                            // src/hmm.cpp:31-35
};

class state_model {                                                                      // src/hmm.hpp:55-68
public:
    phone::ties *ties;
    std::map<phone::context, float> probsums;
    std::map<phone::context, int> counts;
    state_model(phone::ties *t) : ties(t) {}
    float operator()(hmm_state *state);                    // src/hmm.cpp:39-44
    void augment(hmm_state *state);                        // src/hmm.cpp:46-51 (single state, host bookkeeping)
};

class utterance;

class hmm {                                                                              // src/hmm.hpp:72-95
public:
    std::vector<hmm_state> states;
    std::unordered_map<hmm_state *, int> idx_map;
    acoustic_model *acm, *temp_acm;
    state_model *sm;
    hmm(std::vector<phone::phone> ph, acoustic_model *ac);         // training HMM, src/hmm.cpp:56-88
    // The lexicon HMM, src/hmm.cpp:92-129: start state, per word (first pronunciation) its phone states and a terminating
    // silence, start transitions LMSF * unigram log-probability.  As in src/main.cpp:58 it is also the trainer object.
    hmm(pronlex &pr, unigram_model &uni, acoustic_model *ac, state_model *smo);
    // A trainer object without the lexicon topology (carries acm / temp_acm / sm only): for callers that train and never decode.
    hmm(acoustic_model *ac, state_model *smo);
    ~hmm();
    int indexOf(hmm_state *);
    // src/hmm.cpp:150-224.  On a training HMM: batched alignment kernel.  On the lexicon HMM: the general-topology kernel
    // with the end condition the reference is compiled with (the path must end in the LAST state, src/hmm.cpp:210).
    path viterbi(std::vector<featurevec *> fvs, float &prevp, float beam_width = 1e10);
    // The recognition variant the reference has commented out (src/hmm.cpp:196-209): the path ends in the best active
    // state.  Empty path when every path was pruned (the reference would `throw 1729`).
    path decode(std::vector<featurevec *> fvs, float &prevp, float beam_width = 1e10);
    // The words a path of the lexicon HMM passes through (a word counts when the path enters its first state).
    std::vector<std::string> transcript(const path &p);
    void train_transition_probabilities(path p);                                          // src/hmm.cpp:229-243
    void train(std::vector<utterance> &ut);                                               // src/hmm.cpp:250-358
    void train(std::vector<utterance> &ut, int iterations);    // same with EM_ITERS as a run-time value
    void update_states();                                                                 // src/hmm.cpp:361-366

private:
    bool general_topology = false;       // lexicon HMM: transitions are an arbitrary list
    srx_decode_plan *dplan = nullptr;    // its device image, built on first use, refreshed by update_states
    path run_graph(std::vector<featurevec *> &fvs, float &prevp, float beam_width, int final_mode);
    void ensure_dplan(const float *tr_lp_now);   // build the device image, or refresh its transition scores (CSR order)
};

// ------------------------------------------------------------------------------------------ utterance.hpp
class utterance {                                                                        // src/utterance.hpp:24-34
public:
    std::string text;
    std::vector<phone::phone> pronunciation;
    class hmm *hmm;
    std::vector<featurevec *> features;
    // One line of a Voxforge PROMPTS file, "<mfc path> WORD WORD ...", pronounced by `pr` (src/utterance.cpp:14-28).
    utterance(std::istream &in, std::string path_base, acoustic_model &acm, pronouncer &pr);
    // [batch] the same line with the phone names already spelled out: "<mfc path> PH PH PH ...".
    utterance(std::istream &in, std::string path_base, acoustic_model &acm);
    utterance(const std::string &text, const std::vector<phone::phone> &pron, const std::vector<featurevec *> &fv,
              acoustic_model &acm);
};
featurevec mean(std::vector<utterance> &);                                               // src/utterance.cpp:32-47
featurevec variance(std::vector<utterance> &, featurevec &);                             // src/utterance.cpp:49-65

#endif  // SRX_HOST_HPP
