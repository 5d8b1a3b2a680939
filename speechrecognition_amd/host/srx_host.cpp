// srx_host.cpp -- implementation of the host layer declared in srx_host.hpp: marshalling between the
// reference's host-side objects (Clip, featurevec, acoustic_model, hmm, utterance) and the C ABI of
// libsrx.so.  No arithmetic of the hot path is performed here; file I/O, containers and bookkeeping only.
#include "srx_host.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>
#include <time.h>

using std::string;
using std::vector;

// ============================================================================================ runtime
namespace {

srx_ctx *g_ctx = nullptr;
srx_comm *g_comm = nullptr;
int g_rank = 0, g_world = 1;
bool g_verbose = false;

void check(int rc) {
    if (rc != SRX_OK) {
        const char *m = srx_last_error(g_ctx);
        throw srx_error(rc, (m && *m) ? m : "libsrx call failed");
    }
}

// typed device allocation through the C ABI
template <class T>
struct Dev {
    T *p = nullptr;
    size_t n = 0;
    Dev() {}
    explicit Dev(size_t count) { alloc(count); }
    Dev(const Dev &) = delete;
    Dev &operator=(const Dev &) = delete;
    ~Dev() { release(); }
    void release() {
        if (p) srx_free(srx::context(), p);
        p = nullptr;
        n = 0;
    }
    void alloc(size_t count) {
        release();
        void *q = nullptr;
        check(srx_malloc(srx::context(), (count ? count : 1) * sizeof(T), &q));
        p = static_cast<T *>(q);
        n = count;
    }
    void upload(const T *h, size_t count) {
        if (count > n) alloc(count);
        check(srx_memcpy_h2d(srx::context(), p, h, count * sizeof(T)));
    }
    void upload(const vector<T> &h) { upload(h.data(), h.size()); }
    void download(T *h, size_t count) const { check(srx_memcpy_d2h(srx::context(), h, p, count * sizeof(T))); }
    vector<T> get() const {
        vector<T> h(n);
        if (n) download(h.data(), n);
        return h;
    }
};

// features of a list of utterances -> one contiguous [T][3*NPARAMS] buffer
void pack_features(const vector<featurevec *> &fvs, vector<float> &out) {
    const size_t ld = 3 * NPARAMS;
    const size_t base = out.size();
    out.resize(base + fvs.size() * ld);
    for (size_t t = 0; t < fvs.size(); t++) {
        float *o = &out[base + t * ld];
        std::memcpy(o, fvs[t]->coeffs, NPARAMS * sizeof(float));
        std::memcpy(o + NPARAMS, fvs[t]->deltas, NPARAMS * sizeof(float));
        std::memcpy(o + 2 * NPARAMS, fvs[t]->delta2s, NPARAMS * sizeof(float));
    }
}

// gmm::operator() as written returns component 0 (src/gmm.cpp:99-100); SRX_MIX=logsumexp selects the
// commented-out intent.  SRX_SCORE_PATH=exact selects the per-dimension evaluation in the reference's
// operation order instead of the MFMA contraction; SRX_SCORE_PATH=f16x3 the split-f16 contraction (fastest).
int env_choice(const char *name, const char *alt, int dflt, int other) {
    const char *v = std::getenv(name);
    return (v && !std::strcmp(v, alt)) ? other : dflt;
}
// How hmm::train takes its statistics (include/srx.h, srx_em_fused_dev): SRX_EM_PATH=faithful -- the five steps that reproduce
// every f32 rounding of the reference, three exchanges; fold -- the same log-densities, variance pass folded into the first
// (one MAX + one SUM); default (fused) -- log-densities on the f64 matrix cores as well, one SUM, with the five-step form as
// the fallback for an iteration that loses a component to f64 underflow.
int em_path_choice() {
    const char *v = std::getenv("SRX_EM_PATH");
    if (v && !std::strcmp(v, "faithful")) return 0;
    if (v && !std::strcmp(v, "fold")) return 1;
    return 2;
}
int g_em_path = em_path_choice();
int g_mix_mode = env_choice("SRX_MIX", "logsumexp", SRX_MIX_COMPONENT0, SRX_MIX_LOGSUMEXP);
int g_score_path = env_choice("SRX_SCORE_PATH", "f16x3", env_choice("SRX_SCORE_PATH", "exact", SRX_SCORE_MFMA, SRX_SCORE_EXACT),
                              SRX_SCORE_MFMA_F16X3);

}  // namespace

namespace srx {
srx_ctx *context() {
    if (!g_ctx) {
        // one process per GPU: SRX_DEVICE, else the launcher's LOCAL_RANK, else device 0
        const char *d = std::getenv("SRX_DEVICE");
        if (!d) d = std::getenv("LOCAL_RANK");
        int rc = srx_ctx_create(d ? std::atoi(d) : 0, nullptr, &g_ctx);
        if (rc != SRX_OK) {
            const char *m = srx_last_error(nullptr);
            throw srx_error(rc, m ? m : "srx_ctx_create failed");
        }
    }
    return g_ctx;
}
// The communicator every exchange of this layer goes through.  World 1 (no launcher) is a communicator whose
// collectives are no-ops, so the single-GPU run takes the same path.
srx_comm *communicator() {
    if (!g_comm) {
        int rc = srx_comm_create(context(), 0, 1, nullptr, &g_comm);
        if (rc != SRX_OK) throw srx_error(rc, srx_last_error(g_ctx));
    }
    return g_comm;
}
int rank() { return g_rank; }
int world() { return g_world; }

// ---- id rendezvous through a file ----
namespace {
const char COMM_MAGIC[8] = {'S', 'R', 'X', 'C', 'O', 'M', 'M', '1'};
const size_t COMM_FILE_BYTES = 8 + 8 + SRX_COMM_ID_BYTES;
}  // namespace

// What every rank of one launch knows and another launch does not: the launcher's rendezvous values.  FNV-1a over them.
// A launcher that sets none of them leaves only WORLD_SIZE: set SRX_COMM_NONCE (any string, the same on every rank, new
// for every launch) when a fixed SRX_COMM_FILE path is reused.
uint64_t launch_nonce() {
    uint64_t h = 0xcbf29ce484222325ull;
    for (const char *name : {"SRX_COMM_NONCE", "TORCHELASTIC_RUN_ID", "MASTER_ADDR", "MASTER_PORT", "WORLD_SIZE"}) {
        const char *v = std::getenv(name);
        for (const char *p = v ? v : ""; *p; p++) h = (h ^ (unsigned char)*p) * 0x100000001b3ull;
        h = (h ^ 0xffu) * 0x100000001b3ull;   // field separator
    }
    return h;
}

void publish_comm_id(const string &file, uint64_t nonce, const unsigned char *id) {
    std::remove(file.c_str());   // whatever an earlier launch left: gone before the new one appears (rename is atomic)
    const string tmp = file + ".tmp";
    FILE *f = std::fopen(tmp.c_str(), "wb");
    if (!f) throw srx_error(SRX_ERR_IO, "cannot write " + tmp);
    const bool ok = std::fwrite(COMM_MAGIC, 1, 8, f) == 8 && std::fwrite(&nonce, 1, 8, f) == 8 &&
                    std::fwrite(id, 1, SRX_COMM_ID_BYTES, f) == (size_t)SRX_COMM_ID_BYTES;
    if (std::fclose(f) != 0 || !ok) throw srx_error(SRX_ERR_IO, "cannot write " + tmp);
    if (std::rename(tmp.c_str(), file.c_str()) != 0) throw srx_error(SRX_ERR_IO, "cannot publish " + file);
}

bool await_comm_id(const string &file, uint64_t nonce, unsigned char *id, int timeout_ms) {
    for (int waited = 0;; waited += 10) {
        FILE *f = std::fopen(file.c_str(), "rb");
        if (f) {
            unsigned char buf[COMM_FILE_BYTES + 1];
            const size_t got = std::fread(buf, 1, sizeof buf, f);
            std::fclose(f);
            uint64_t theirs = 0;
            if (got == COMM_FILE_BYTES && !std::memcmp(buf, COMM_MAGIC, 8)) {
                std::memcpy(&theirs, buf + 8, 8);
                if (theirs == nonce) {
                    std::memcpy(id, buf + 16, SRX_COMM_ID_BYTES);
                    return true;
                }
            }
            // anything else -- another launch's nonce, a bare 128-byte id of an older build, a partial file -- is not ours
        }
        if (waited >= timeout_ms) return false;
        struct timespec ts = {0, 10 * 1000 * 1000};
        nanosleep(&ts, nullptr);
    }
}

// Joins the ranks of one launch (RANK / WORLD_SIZE in the environment, as torchrun, mpirun wrappers and SLURM scripts set
// them).  The 128-byte id of rank 0 travels through the file SRX_COMM_FILE (any path all ranks can read) tagged with the
// launch's nonce: rank 0 removes what is there, writes <file>.tmp and renames it; the others wait for a file that carries
// their nonce; rank 0 removes the file again once every rank has joined.  SRX_COMM_SHM=/name selects the host rehearsal
// transport instead (ranks sharing one GPU, see include/srx.h).
void init_distributed() {
    const char *r = std::getenv("RANK"), *w = std::getenv("WORLD_SIZE");
    if (!r || !w || std::atoi(w) <= 1 || g_comm) return;
    g_rank = std::atoi(r);
    g_world = std::atoi(w);
    srx_ctx *ctx = context();
    int rc;
    if (const char *shm = std::getenv("SRX_COMM_SHM")) {
        // the segment's name carries the launch nonce: a segment left behind by a crashed launch has another name and cannot
        // be joined by mistake (the same caveat as for SRX_COMM_FILE: launches with identical rendezvous values need SRX_COMM_NONCE)
        char name[256];
        snprintf(name, sizeof name, "%s_%016llx", shm, (unsigned long long)launch_nonce());
        rc = srx_comm_create_host(ctx, g_rank, g_world, name, &g_comm);
    } else {
        const char *file = std::getenv("SRX_COMM_FILE");
        if (!file) throw srx_error(SRX_ERR_INVALID, "WORLD_SIZE > 1 needs SRX_COMM_FILE (where rank 0 publishes the RCCL id) or SRX_COMM_SHM");
        unsigned char id[SRX_COMM_ID_BYTES];
        const uint64_t nonce = launch_nonce();
        if (g_rank == 0) {
            rc = srx_comm_unique_id(id);
            if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
            publish_comm_id(file, nonce, id);
        } else if (!await_comm_id(file, nonce, id, 60 * 1000)) {
            throw srx_error(SRX_ERR_IO, string("rank 0 never published an id of this launch in ") + file);
        }
        rc = srx_comm_create(ctx, g_rank, g_world, id, &g_comm);
        if (g_rank == 0) std::remove(file);   // collective: every rank has read it by now
    }
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(ctx));
}

// Greedy longest-processing-time partition of the utterances over the ranks (cost = frames x states): sorted by cost
// descending (stable), each to the currently lightest rank; ascending indices within a rank.  Deterministic: every rank
// computes the same partition.  (speechrecognition_amd/parallel.py::shard_utterances is the same rule.)
vector<int> shard_utterances(const vector<double> &costs, int world_, int rank_) {
    vector<int> order(costs.size());
    for (size_t i = 0; i < order.size(); i++) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return costs[a] > costs[b]; });
    vector<double> load((size_t)world_, 0.0);
    vector<int> mine;
    for (int u : order) {
        int best = 0;
        for (int k = 1; k < world_; k++)
            if (load[k] < load[best]) best = k;
        load[best] += costs[u];
        if (best == rank_) mine.push_back(u);
    }
    std::sort(mine.begin(), mine.end());
    return mine;
}

void set_verbose(bool on) { g_verbose = on; }
// A generator of this layer's own: libc's rand() is shared with whatever else the process loads, and draws made
// on the device runtime's threads made random initialisation differ from run to run.
static uint64_t g_rng = 0x9E3779B97F4A7C15ull;
void seed_random(unsigned seed) { g_rng = 0x9E3779B97F4A7C15ull ^ ((uint64_t)seed * 0xD1342543DE82EF95ull); }
double next_random() {   // splitmix64 -> [0, 1)
    uint64_t z = (g_rng += 0x9E3779B97F4A7C15ull);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (double)(z >> 11) * (1.0 / 9007199254740992.0);
}
bool verbose() { return g_verbose; }
}  // namespace srx

// ============================================================================================ utils
vector<string> readlines(const char *fname) {
    // src/utils.cpp:40-51: at least one (possibly empty) line is always returned
    vector<string> lines;
    std::ifstream in(fname);
    string ln;
    std::getline(in, ln);
    do {
        lines.push_back(ln);
        std::getline(in, ln);
    } while (in.good());
    return lines;
}

float hamming_window(int x, int len) { return 0.54f + 0.46 * std::cos(2 * M_PI * x / (len - 1)); }
float rectangular_window(int, int) { return 1; }
void log(float *buf, int len) {
    if (len > 0) check(srx_log1p(srx::context(), buf, len));
}

// ============================================================================================ clip
ClipBuffer::~ClipBuffer() {}
ClipArrayBuffer::ClipArrayBuffer(int count) : n(count), buf(new float[count > 0 ? count : 1]()) {}
ClipArrayBuffer::~ClipArrayBuffer() { delete[] buf; }
float &ClipArrayBuffer::operator[](int i) { return buf[i]; }
int ClipArrayBuffer::length() const { return n; }
float *ClipArrayBuffer::data() { return buf; }

ClipVectorBuffer::ClipVectorBuffer() {}
void ClipVectorBuffer::ensure(int additional) {
    if (samples.capacity() - samples.size() < (size_t)additional) samples.reserve(samples.capacity() + additional);
}
void ClipVectorBuffer::append(float *b, int n) {
    ensure(n);
    for (int i = 0; i < n; i++) samples.push_back(b[i]);
}
float &ClipVectorBuffer::operator[](int i) { return samples[i]; }
int ClipVectorBuffer::length() const { return (int)samples.size(); }
float *ClipVectorBuffer::data() { return samples.data(); }

Clip::Clip() : buf(nullptr), start(0), end(0), window(rectangular_window) {}
Clip::Clip(ClipBuffer &b) : buf(&b), start(0), end(b.length()), window(rectangular_window) {}
Clip::Clip(ClipBuffer &b, int s, int e) : buf(&b), start(s), end(e), window(rectangular_window) {}
Clip::Clip(const Clip &c, int s, int e) : buf(c.buf), start(c.start + s), end(c.start + e), window(rectangular_window) {}
bool Clip::slide(int n) {
    start += n;
    end += n;
    return start >= 0 && end < buf->length();
}
float &Clip::operator[](int idx) { return (*buf)[start + idx]; }
float Clip::operator[](int idx) const { return (*buf)[start + idx] * window(idx, length()); }
int Clip::length() const { return end - start; }

namespace {
// one utterance on the device: samples + {0, L} offsets
struct DevClip {
    Dev<float> x;
    Dev<int64_t> off;
    int64_t L;
    explicit DevClip(const Clip &c) : L(c.length()) {
        x.upload(c.buf->data() + c.start, (size_t)L);
        int64_t o[2] = {0, L};
        off.upload(o, 2);
    }
    void store(Clip &c) { x.download(c.buf->data() + c.start, (size_t)L); }
};
}  // namespace

// src/clip.cpp:128-143.  Not on the hot path: the right-hand clip is read through the const operator[] (its window applies)
void Clip::operator+=(Clip &c) {
    const Clip &r = c;
    for (int i = 0; i < std::min(length(), c.length()); i++) (*this)[i] += r[i];
}
void Clip::operator-=(Clip &c) {
    const Clip &r = c;
    for (int i = 0; i < std::min(length(), c.length()); i++) (*this)[i] -= r[i];
}
void Clip::operator*=(Clip &c) {
    const Clip &r = c;
    for (int i = 0; i < std::min(length(), c.length()); i++) (*this)[i] *= r[i];
}

void Clip::dft(float *out, bool stack) const { partial_dft(out, length(), stack); }

void Clip::partial_dft(float *out, int len, bool stack) const {
    const int N = length();
    if (len <= 0) return;
    if (N < 2) throw srx_error(SRX_ERR_INVALID, "Clip::partial_dft: a window of fewer than two samples");
    // the window is a host function pointer: its N/2 values are evaluated here (src/clip.cpp:124 evaluates them per access),
    // the multiplication and everything after it run on the device
    vector<float> w;
    if (window != rectangular_window) {
        w.resize((size_t)(N / 2));
        for (int n = 0; n < N / 2; n++) w[(size_t)n] = window(n, N);
    }
    check(srx_partial_dft_frames(srx::context(), buf->data() + start, w.empty() ? nullptr : w.data(), N, len, 1, stack ? 1 : 0, out));
}

void Clip::operator*=(float f) {
    if (length() <= 0) return;
    DevClip d(*this);
    Dev<float> sc(1);
    sc.upload(&f, 1);
    check(srx_scale_dev(srx::context(), d.x.p, d.off.p, 1, sc.p));
    d.store(*this);
}

void Clip::normalizeVolume(float targetRMS) {
    if (length() <= 0) return;
    DevClip d(*this);
    Dev<float> sc(1);
    check(srx_normalize_volume_dev(srx::context(), d.x.p, d.off.p, 1, targetRMS, sc.p));
    d.store(*this);
}

double Clip::rmsAmplitude() const {
    if (length() <= 0) return 0.0;
    DevClip d(*this);
    Dev<double> r(1);
    check(srx_rms_amplitude_dev(srx::context(), d.x.p, d.off.p, 1, r.p));
    double v = 0;
    r.download(&v, 1);
    return v;
}

void Clip::preemphasis(float alpha) {
    if (length() <= 1) return;
    DevClip d(*this);
    Dev<float> tmp((size_t)d.L);
    check(srx_preemphasis_dev(srx::context(), d.x.p, tmp.p, d.off.p, 1, d.L, alpha));
    d.store(*this);
}

// ============================================================================================ wav
wav_io_exception wavex_not_riff("Not an RIFF file"), wavex_not_wave("Not a WAV file"),
    wavex_not_pcm("Not a PCM-encoded WAV file"), wavex_bad("Bad WAV file"), wavex_fopen_fail("Could not open WAV file");

Clip read_wav(string fname) {
    FILE *in = std::fopen(fname.c_str(), "rb");
    if (!in) throw wavex_fopen_fail;
    struct Closer {
        FILE *f;
        ~Closer() { std::fclose(f); }
    } closer{in};
    unsigned char h[44];
    if (std::fread(h, 1, 12, in) != 12 || std::memcmp(h, "RIFF", 4)) throw wavex_not_riff;
    if (std::memcmp(h + 8, "WAVE", 4)) throw wavex_not_wave;
    if (std::fread(h, 1, 8, in) != 8 || std::memcmp(h, "fmt ", 4)) throw wavex_bad;
    int32_t sc1size;
    std::memcpy(&sc1size, h + 4, 4);
    unsigned char f[16];
    if (std::fread(f, 1, 16, in) != 16) throw wavex_bad;
    int16_t aformat, nchannels, bps;
    std::memcpy(&aformat, f, 2);
    std::memcpy(&nchannels, f + 2, 2);
    std::memcpy(&bps, f + 14, 2);
    if (aformat != 1) throw wavex_not_pcm;
    if (sc1size > 16) std::fseek(in, sc1size - 16, SEEK_CUR);  // the reference assumes 16 (its check is commented out)
    if (std::fread(h, 1, 8, in) != 8 || std::memcmp(h, "data", 4)) throw wavex_bad;
    int32_t data_len;
    std::memcpy(&data_len, h + 4, 4);
    if (nchannels < 1 || (bps != 8 && bps != 16)) throw wavex_bad;
    // The reference trusts the header (src/wav.cpp:84-88) and reads past what the file holds.  Here a length the file cannot
    // back is cut to what is there (a truncated file gives the frames it really has; 0xFFFFFFFF, the "unknown length" of
    // streamed files, reads to the end), so that a header cannot ask for gigabytes.
    {
        const long here = std::ftell(in);
        long remain = 0;
        if (here >= 0 && std::fseek(in, 0, SEEK_END) == 0) {
            remain = std::ftell(in) - here;
            std::fseek(in, here, SEEK_SET);
        }
        if (remain < 0) remain = 0;
        if (data_len < 0 || (long)data_len > remain) data_len = (int32_t)std::min<long>(remain, 0x7fffffffL);
    }
    const int nframes = data_len / nchannels / (bps / 8);
    if (srx::verbose()) std::cout << "There are " << nframes << " audio frames" << std::endl;
    vector<unsigned char> raw((size_t)data_len);
    const size_t got = std::fread(raw.data(), 1, raw.size(), in);
    ClipArrayBuffer *abuf = new ClipArrayBuffer(nframes);
    const int bytes = bps / 8;
    for (int i = 0; i < nframes; i++) {
        // src/wav.cpp:36-53: 8-bit s/255.0f, 16-bit s/32768.0f; two channels are averaged (only the first two are read)
        float ch[2] = {0, 0};
        const int nread = nchannels == 1 ? 1 : 2;
        for (int c = 0; c < nread; c++) {
            const size_t o = ((size_t)i * nchannels + c) * bytes;
            if (o + bytes > got) break;
            if (bps == 8) {
                ch[c] = raw[o] / 255.0f;
            } else {
                int16_t s;
                std::memcpy(&s, &raw[o], 2);
                ch[c] = s / 32768.0f;
            }
        }
        (*abuf)[i] = nchannels == 1 ? ch[0] : (ch[0] + ch[1]) / 2;
    }
    return Clip(*abuf);
}

void write_wav(Clip c, int sps, short bps, string fname) {
    if (bps != 8 && bps != 16) throw 2;
    FILE *out = std::fopen(fname.c_str(), "wb");
    if (!out) throw 1;
    const int16_t aformat = 1, nchannels = 1;
    const int32_t data_len = bps / 8 * c.length(), sc1size = 16, chsize = data_len + 36;
    const int32_t byte_rate = sps * bps / 8 * nchannels;
    const int16_t block_align = nchannels * bps / 8;
    const int32_t sps32 = sps;
    std::fwrite("RIFF", 1, 4, out);
    std::fwrite(&chsize, 4, 1, out);
    std::fwrite("WAVEfmt ", 1, 8, out);
    std::fwrite(&sc1size, 4, 1, out);
    std::fwrite(&aformat, 2, 1, out);
    std::fwrite(&nchannels, 2, 1, out);
    std::fwrite(&sps32, 4, 1, out);
    std::fwrite(&byte_rate, 4, 1, out);
    std::fwrite(&block_align, 2, 1, out);
    std::fwrite(&bps, 2, 1, out);
    std::fwrite("data", 1, 4, out);
    std::fwrite(&data_len, 4, 1, out);
    for (int i = 0; i < c.length(); i++) {
        float v = c[i];
        v = v < -1 ? -1 : (v > 1 ? 1 : v);
        if (bps == 8) {
            unsigned char s = (unsigned char)((v / 2 + 0.5f) * 255);
            std::fwrite(&s, 1, 1, out);
        } else {
            int16_t s = (int16_t)(v * 32767);
            std::fwrite(&s, 2, 1, out);
        }
    }
    std::fclose(out);
}

// ============================================================================================ featurevec / filterbank
featurevec::featurevec() {
    for (int i = 0; i < NPARAMS; i++) coeffs[i] = deltas[i] = delta2s[i] = 0;
}
featurevec::featurevec(float *c) {
    for (int i = 0; i < NPARAMS; i++) {
        coeffs[i] = c[i];
        deltas[i] = delta2s[i] = 0;
    }
}
float &featurevec::operator[](int i) {
    if (i < NPARAMS) return coeffs[i];
    if (i < 2 * NPARAMS) return deltas[i - NPARAMS];
    return delta2s[i - 2 * NPARAMS];
}
void featurevec::operator+=(featurevec &f) {
    for (int i = 0; i < FV_LEN; i++) (*this)[i] += f[i];
}
void featurevec::operator*=(float f) {
    for (int i = 0; i < FV_LEN; i++) (*this)[i] *= f;
}

float featurevec::dist(featurevec &f) {
    float d = 0;
    for (int i = 0; i < FV_LEN; i++) {
        float diff = (*this)[i] - f[i];
        d += diff * diff;
    }
    return std::sqrt(d);
}
featurevec featurevec::normalize(featurevec &mu, featurevec &var) {
    featurevec res;
    for (int i = 0; i < FV_LEN; i++) res[i] = ((*this)[i] - mu[i]) / std::sqrt(var[i]);
    return res;
}
void featurevec::compute_deltas(vector<featurevec *> fvs, int i) {
    int j = i - 1, k = i + 1;
    if (j < 0) j = 0;
    if (k == (int)fvs.size()) k--;
    check(srx_delta(srx::context(), fvs[(size_t)k]->coeffs, fvs[(size_t)j]->coeffs, k - j, deltas, NPARAMS));
}
void featurevec::compute_delta2s(vector<featurevec *> fvs, int i) {
    int j = i - 1, k = i + 1;
    if (j < 0) j = 0;
    if (k == (int)fvs.size()) k--;
    check(srx_delta(srx::context(), fvs[(size_t)k]->deltas, fvs[(size_t)j]->deltas, k - j, delta2s, NPARAMS));
}
void compute_deltas(vector<featurevec *> fvs) {
    const size_t n = fvs.size();
    if (n == 0) return;
    vector<float> raw(n * NPARAMS), feats(n * 3 * NPARAMS);
    for (size_t i = 0; i < n; i++) std::memcpy(&raw[i * NPARAMS], fvs[i]->coeffs, NPARAMS * sizeof(float));
    int64_t fo[2] = {0, (int64_t)n};
    check(srx_features_batch(srx::context(), -1, NPARAMS, raw.data(), fo, 1, feats.data()));
    for (size_t i = 0; i < n; i++) {
        std::memcpy(fvs[i]->deltas, &feats[i * 3 * NPARAMS + NPARAMS], NPARAMS * sizeof(float));
        std::memcpy(fvs[i]->delta2s, &feats[i * 3 * NPARAMS + 2 * NPARAMS], NPARAMS * sizeof(float));
    }
}

float filter::operator()(float *spectrum) {
    srx_filter f = {center, width, height};
    const int n_bins = (int)(center + width) + 1;
    if (n_bins <= 0) return 0.0f;   // the reference's loop body never runs
    float res = 0;
    check(srx_filterbank_apply(srx::context(), &f, 1, spectrum, n_bins, 1, &res));
    return res;
}
float *filterbank::operator()(float *spectrum) {
    vector<srx_filter> f;
    int n_bins = 0;
    for (const filter &x : filters) {
        f.push_back({x.center, x.width, x.height});
        n_bins = std::max(n_bins, (int)(x.center + x.width) + 1);
    }
    float *res = new float[filters.size() ? filters.size() : 1];
    if (filters.empty()) return res;
    try {
        for (size_t i0 = 0; i0 < f.size(); i0 += 64) {   // the C ABI takes 64 filters per call
            const int m = (int)std::min<size_t>(64, f.size() - i0);
            check(srx_filterbank_apply(srx::context(), &f[i0], m, spectrum, n_bins, 1, res + i0));
        }
    } catch (...) {
        delete[] res;
        throw;
    }
    return res;
}
filterbank *linear_filterbank(float start, float step, int n, float width, int dft_bins, int sps) {
    (void)step;   // src/mfcc.cpp:174 steps by n, not by `step`
    filterbank *fb = new filterbank();
    for (int i = 0; i < n; i++) fb->filters.push_back(filter(start + i * n, width, dft_bins, sps));
    return fb;
}
float *dct(float *filtered, int len, int nparams) {
    float *res = new float[nparams > 0 ? nparams : 1];
    if (nparams <= 0) return res;
    if (len <= 0) {   // the reference's inner loop never runs
        for (int j = 0; j < nparams; j++) res[j] = 0;
        return res;
    }
    try {
        check(srx_dct(srx::context(), filtered, len, nparams, 1, res));
    } catch (...) {
        delete[] res;
        throw;
    }
    return res;
}

filter::filter(float c, float w, int dft_bins, int sps) {
    srx_filter f;
    int rc = srx_filter_init(c, w, dft_bins, sps, &f);
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    center = f.center;
    width = f.width;
    height = f.height;
    if (srx::verbose()) std::cout << "Filter centered on " << center << " of width " << width << std::endl;
}
float filter::get(int bin) const {
    srx_filter f = {center, width, height};
    return srx_filter_value(&f, bin);
}
filterbank::filterbank() {}
int filterbank::length() const { return (int)filters.size(); }

filterbank *mel_filterbank(float end, int n, int dft_bins, int sps) {
    vector<srx_filter> f((size_t)n);
    int bm = 0;
    int rc = srx_mel_filterbank(end, n, dft_bins, sps, f.data(), &bm);
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    filterbank *fb = new filterbank();
    for (int i = 0; i < n; i++) {
        filter x(0, 1, dft_bins, sps);
        x.center = f[i].center;
        x.width = f[i].width;
        x.height = f[i].height;
        fb->filters.push_back(x);
    }
    return fb;
}

namespace {
// MFCC plans are cached per configuration + filterbank constants
struct PlanKey {
    int wl, hop, normalize, preemph;
    vector<float> fb;
    bool operator<(const PlanKey &o) const {
        if (wl != o.wl) return wl < o.wl;
        if (hop != o.hop) return hop < o.hop;
        if (normalize != o.normalize) return normalize < o.normalize;
        if (preemph != o.preemph) return preemph < o.preemph;
        return fb < o.fb;
    }
};
std::map<PlanKey, srx_mfcc_plan *> g_plans;

srx_mfcc_plan *get_plan(unsigned wl, unsigned hop, filterbank &fb, bool normalize, bool preemph, srx_mfcc_cfg *cfg_out) {
    PlanKey k{(int)wl, (int)hop, normalize ? 1 : 0, preemph ? 1 : 0, {}};
    vector<srx_filter> f;
    for (const filter &x : fb.filters) {
        f.push_back({x.center, x.width, x.height});
        k.fb.push_back(x.center);
        k.fb.push_back(x.width);
        k.fb.push_back(x.height);
    }
    srx_mfcc_cfg cfg;
    cfg.window_len = (int)wl;
    cfg.frame_shift = (int)hop;
    cfg.n_ceps = NPARAMS;
    cfg.normalize = normalize ? 1 : 0;
    cfg.target_rms = 1.0f;
    cfg.preemphasis = preemph ? 1 : 0;
    cfg.alpha = 0.96f;  // src/mfcc.cpp:238
    if (cfg_out) *cfg_out = cfg;
    auto it = g_plans.find(k);
    if (it != g_plans.end()) return it->second;
    srx_mfcc_plan *p = nullptr;
    check(srx_mfcc_plan_create(srx::context(), &cfg, f.data(), (int)f.size(), &p));
    g_plans[k] = p;
    return p;
}

void write_mfc(const string &fname, const float *ceps, int n) {
    // src/mfcc.cpp:256-273: int32 frame count, then n x NPARAMS float32, native endian
    FILE *out = std::fopen(fname.c_str(), "wb");
    if (!out) throw 1;
    const int32_t n32 = n;
    std::fwrite(&n32, 4, 1, out);
    std::fwrite(ceps, sizeof(float), (size_t)n * NPARAMS, out);
    std::fclose(out);
}
}  // namespace

vector<float> mfcc_batch(vector<Clip> &clips, unsigned wl, unsigned hop, filterbank &fb, bool normalize_volume,
                         vector<int64_t> &frame_off) {
    srx_mfcc_cfg cfg;
    srx_mfcc_plan *plan = get_plan(wl, hop, fb, normalize_volume, true, &cfg);
    const int n = (int)clips.size();
    vector<int64_t> soff((size_t)n + 1, 0);
    for (int u = 0; u < n; u++) soff[u + 1] = soff[u] + clips[u].length();
    vector<float> x((size_t)soff[n]);
    for (int u = 0; u < n; u++)
        std::memcpy(&x[(size_t)soff[u]], clips[u].buf->data() + clips[u].start, (size_t)clips[u].length() * sizeof(float));
    frame_off.assign((size_t)n + 1, 0);
    int rc = srx_mfcc_frame_offsets(&cfg, soff.data(), n, frame_off.data());
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    vector<float> ceps((size_t)frame_off[n] * NPARAMS);
    check(srx_mfcc_batch(srx::context(), plan, x.data(), soff.data(), n, ceps.data(), frame_off.data()));
    return ceps;
}

void write_features_batch(vector<Clip> &clips, unsigned wl, unsigned hop, filterbank &fb, const vector<string> &fnames,
                          bool normalize_volume, bool mutate_samples) {
    if (fnames.size() != clips.size()) throw srx_error(SRX_ERR_INVALID, "write_features_batch: one file name per clip");
    vector<int64_t> fo;
    vector<float> ceps = mfcc_batch(clips, wl, hop, fb, normalize_volume, fo);
    for (size_t u = 0; u < clips.size(); u++)
        write_mfc(fnames[u], &ceps[(size_t)fo[u] * NPARAMS], (int)(fo[u + 1] - fo[u]));
    if (mutate_samples)
        for (Clip &c : clips) {
            if (normalize_volume) c.normalizeVolume(1.0f);
            c.preemphasis(0.96f);
        }
}

void write_features(Clip c, unsigned window_len, unsigned frame_shift, filterbank &fb, string fname) {
    // the reference opens the file first and throws 1 before touching the samples (src/mfcc.cpp:236-238)
    FILE *probe = std::fopen(fname.c_str(), "wb");
    if (!probe) throw 1;
    std::fclose(probe);
    c.preemphasis(0.96f);  // in place, like the reference
    srx_mfcc_cfg cfg;
    srx_mfcc_plan *plan = get_plan(window_len, frame_shift, fb, false, false, &cfg);
    int64_t soff[2] = {0, c.length()}, fo[2] = {0, 0};
    int rc = srx_mfcc_frame_offsets(&cfg, soff, 1, fo);
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    vector<float> ceps((size_t)fo[1] * NPARAMS);
    check(srx_mfcc_batch(srx::context(), plan, c.buf->data() + c.start, soff, 1, ceps.data(), fo));
    write_mfc(fname, ceps.data(), (int)fo[1]);
}

float *mfcc(Clip c, filterbank &fb, float *) {
    const int wl = c.length();
    srx_mfcc_cfg cfg;
    srx_mfcc_plan *plan = get_plan((unsigned)wl, (unsigned)(wl / 2 > 0 ? wl / 2 : 1), fb, false, false, &cfg);
    vector<float> x((size_t)wl + 1, 0.0f);  // one frame needs L = wl + 1 under the reference's frame-count rule
    std::memcpy(x.data(), c.buf->data() + c.start, (size_t)wl * sizeof(float));
    int64_t soff[2] = {0, wl + 1}, fo[2] = {0, 0};
    float *out = new float[NPARAMS];
    check(srx_mfcc_batch(srx::context(), plan, x.data(), soff, 1, out, fo));
    return out;
}

vector<featurevec *> read_features(string fname, int WL) {
    FILE *in = std::fopen(fname.c_str(), "rb");
    if (!in) throw 1;
    int32_t n = 0;
    if (std::fread(&n, 4, 1, in) != 1) n = 0;
    vector<float> raw((size_t)(n > 0 ? n : 0) * NPARAMS);
    const size_t got = n > 0 ? std::fread(raw.data(), sizeof(float), raw.size(), in) : 0;
    std::fclose(in);
    if (srx::verbose()) std::cout << "\treading features... found " << n << " FVs" << std::endl;
    vector<featurevec *> fvs;
    if (n <= 0 || got != raw.size()) return fvs;
    int64_t fo[2] = {0, n};
    vector<float> feats((size_t)n * 3 * NPARAMS);
    check(srx_features_batch(srx::context(), WL, NPARAMS, raw.data(), fo, 1, feats.data()));
    for (int i = 0; i < n; i++) {
        featurevec *fv = new featurevec();
        const float *f = &feats[(size_t)i * 3 * NPARAMS];
        std::memcpy(fv->coeffs, f, NPARAMS * sizeof(float));
        std::memcpy(fv->deltas, f + NPARAMS, NPARAMS * sizeof(float));
        std::memcpy(fv->delta2s, f + 2 * NPARAMS, NPARAMS * sizeof(float));
        fvs.push_back(fv);
    }
    return fvs;
}

// ============================================================================================ phone
namespace phone {
const char *names[] = {"_",  "AA", "AE", "AH", "AO", "AW", "AY", "B",  "CH", "D",  "DH", "EH", "ER", "EY",
                       "F",  "G",  "HH", "IH", "IY", "JH", "K",  "L",  "M",  "N",  "NG", "OW", "OY", "P",
                       "R",  "S",  "SH", "T",  "TH", "UH", "UW", "V",  "W",  "Y",  "Z",  "ZH"};

bool context::operator<(const context &o) const {
    if (curr != o.curr) return curr < o.curr;
    if (prev != o.prev) return prev < o.prev;
    return next < o.next;
}
context context::operator++() {
    // odometer over (prev, curr, next) with next fastest; wraps to (SIL, SIL, SIL)
    int n = (int)next + 1, c = (int)curr, p = (int)prev;
    if (n == NUM_PH) {
        n = 0;
        if (++c == NUM_PH) {
            c = 0;
            if (++p == NUM_PH) p = 0;
        }
    }
    prev = (phone)p;
    curr = (phone)c;
    next = (phone)n;
    return *this;
}
ties::ties(mode m) {
    for (int i = 0; i < NUM_PH; i++) c[(phone)i] = m == NULL_CONTEXT ? SIL : (phone)i;
    if (m == PHONE_CLASSES) {
        static const struct {
            phone rep;
            phone members[10];
            int n;
        } classes[] = {
            {T, {B, D, G, K, P, T}, 6},                                   // stops
            {N, {M, N, NG}, 3},                                           // nasals
            {SH, {CH, DH, F, JH, S, SH, TH, V, Z, ZH}, 10},               // fricatives
            {L, {L, R, W, Y}, 4},                                         // liquids
            {AE, {AE, EH, IH, IY}, 4},                                    // front vowels
            {AA, {AA, AH, AO, ER}, 4},                                    // central vowels
            {AW, {AW, OW, UH, UW}, 4},                                    // back vowels
            {OY, {AY, EY, OY}, 3},                                        // diphthongs
            {HH, {HH}, 1},
        };
        c.clear();  // the reference leaves SIL unmapped in this mode: map::operator[] default-inserts SIL (= 0)
        for (const auto &k : classes)
            for (int i = 0; i < k.n; i++) c[k.members[i]] = k.rep;
    }
}
context ties::operator()(context ctx) { return context(c[ctx.prev], ctx.curr, c[ctx.next]); }
}  // namespace phone

phone::phone findPhone(string w) {
    size_t d = w.find_first_of("0123456789");
    if (d != string::npos) w.erase(d, 1);
    for (int i = 0; i < phone::NUM_PH; i++)
        if (w == phone::names[i]) return (phone::phone)i;
    if (w == "SIL") return phone::SIL;
    throw srx_error(SRX_ERR_INVALID, "unknown phone name '" + w + "'");
}

// ============================================================================================ gmm
gaussian::gaussian() : weight(1), logwt(0), zetasum(0) {
    for (int i = 0; i < FV_LEN; i++) {
        mean[i] = 0;
        var[i] = 1;
    }
}
static float uniform_rand(float lo, float hi) { return lo + (hi - lo) * (float)srx::next_random(); }
void gaussian::initialize(featurevec &mu, featurevec &sigma2) {
    for (int i = 0; i < FV_LEN; i++) {
        float sd = std::sqrt(sigma2[i]);
        float K = 1.5f;
        mean[i] = mu[i] + uniform_rand(-K * sd, K * sd);
        var[i] = sigma2[i] * uniform_rand(0.8f, 1.2f);
    }
}
void gaussian::set_wt(float wt) {
    weight = wt;
    logwt = std::log(wt);
}
logprob_t gaussian::operator()(featurevec &fv) {
    double m[FV_LEN], v[FV_LEN];
    for (int i = 0; i < FV_LEN; i++) {
        m[i] = (double)mean[i];
        v[i] = (double)var[i];
    }
    float x[3 * NPARAMS], out = 0;
    for (int i = 0; i < 3 * NPARAMS; i++) x[i] = fv[i];
    check(srx_gaussian_logprob(srx::context(), x, 3 * NPARAMS, 1, FV_LEN, m, v, logwt, &out));
    return out;
}
logprob_t gaussian::logprob(featurevec &fv) { return (*this)(fv); }
prob_t gaussian::prob(featurevec &fv) { return expl((prob_t)(*this)(fv)); }
void gaussian::clear() {
    set_wt(1);
    for (int i = 0; i < FV_LEN; i++) {
        mean[i] = 0;
        var[i] = 0;
    }
}
void gaussian::divide_means() {
    for (int i = 0; i < FV_LEN; i++) mean[i] /= zetasum;
}
void gaussian::divide_variances() {
    for (int i = 0; i < FV_LEN; i++) var[i] /= zetasum;
}
void gaussian::blend(gaussian &g, float f) {
    for (int i = 0; i < FV_LEN; i++) {
        mean[i] = mean[i] * f + g.mean[i] * (1 - f);
        var[i] = var[i] * f + g.var[i] * (1 - f);
    }
    set_wt(weight * f + g.weight * (1 - f));
}
logprob_t gmm::operator()(featurevec &fv) { return gaussians[0](fv); }
void gmm::clear() {
    for (gaussian &g : gaussians) g.clear();
}
void gmm::sum_zetas() {
    zetasum = 0;
    for (gaussian &g : gaussians) zetasum += g.zetasum;
}
void gmm::clear_zetas() {
    zetasum = 0;
    for (gaussian &g : gaussians) g.zetasum = 0;
}
void gmm::divide_means() {
    for (gaussian &g : gaussians) g.divide_means();
}
void gmm::divide_variances() {
    for (gaussian &g : gaussians) g.divide_variances();
}
void gmm::blend(gmm *o, float f) {
    for (size_t i = 0; i < gaussians.size(); i++) gaussians[i].blend(o->gaussians[i], f);
}
gmm::gmm(int n) : gaussians(*new vector<gaussian>()), zetasum(0) {
    for (int i = 0; i < n; i++) gaussians.push_back(gaussian());
}
gmm::gmm(gmm *g) : gaussians(*new vector<gaussian>(g->gaussians)), zetasum(0) {}
void gmm::initialize(featurevec &mu, featurevec &var) {
    for (gaussian &g : gaussians) {
        g.initialize(mu, var);
        g.set_wt(1.0f / gaussians.size());
    }
}

acoustic_model::acoustic_model(phone::ties &t, int nmix) : ties(t), dev(nullptr), host_dirty(true) {
    phone::context start(phone::SIL);
    phone::context c = start;
    do {
        phone::context rep = t(c);
        if (mixtures.count(rep) == 0) mixtures[rep] = new gmm(nmix);
        ++c;
    } while (start < c);
    build_index();
}
acoustic_model::acoustic_model(acoustic_model *a) : ties(a->ties), dev(nullptr), host_dirty(true) {
    a->download();
    for (acm_iter it = a->begin(); it != a->end(); it++) mixtures[it->first] = new gmm(it->second);
    build_index();
}
acoustic_model::~acoustic_model() {
    if (dev) srx_model_destroy(dev);
    for (acm_iter it = begin(); it != end(); it++) {
        delete &it->second->gaussians;
        delete it->second;
    }
}
void acoustic_model::build_index() {
    index.clear();
    int i = 0;
    for (acm_iter it = begin(); it != end(); it++) index[it->first] = i++;
}
int acoustic_model::num_mix() const { return mixtures.empty() ? 0 : (int)mixtures.begin()->second->gaussians.size(); }
int acoustic_model::tied_index(phone::context ph) {
    auto it = index.find(ties(ph));
    if (it == index.end()) throw srx_error(SRX_ERR_INVALID, "acoustic_model: context has no tied state");  // reference: UB
    return it->second;
}
acm_iter acoustic_model::begin() { return mixtures.begin(); }
acm_iter acoustic_model::end() { return mixtures.end(); }
void acoustic_model::initialize(featurevec &mu, featurevec &var) {
    download();
    for (acm_iter it = begin(); it != end(); it++) it->second->initialize(mu, var);
    host_dirty = true;
}
// The EM bookkeeping of src/gmm.cpp:188-229 on the host mirror (long double, as the reference); the device copy follows on
// the next use.  The zeta sums live on the host objects only: hmm::train keeps its own on the device.
void acoustic_model::clear() {
    download();
    for (acm_iter it = begin(); it != end(); it++) it->second->clear();
    host_dirty = true;
}
void acoustic_model::sum_zetas() {
    prob_t zetamax = 0;
    acm_iter best = end();
    for (acm_iter it = begin(); it != end(); it++) {
        it->second->sum_zetas();
        if (zetamax < it->second->zetasum) {
            zetamax = it->second->zetasum;
            best = it;
        }
    }
    if (srx::verbose() && best != end())   // (the reference dereferences `best` even when no sum was positive)
        std::cout << "Maximum zetasum: " << (double)zetamax << " for " << phone::names[best->first.curr] << std::endl;
}
void acoustic_model::clear_zetas() {
    for (acm_iter it = begin(); it != end(); it++) it->second->clear_zetas();
}
void acoustic_model::divide_means() {
    download();
    for (acm_iter it = begin(); it != end(); it++) it->second->divide_means();
    host_dirty = true;
}
void acoustic_model::divide_variances() {
    download();
    for (acm_iter it = begin(); it != end(); it++) it->second->divide_variances();
    host_dirty = true;
}
void acoustic_model::blend(acoustic_model *a, float f) {
    download();
    a->download();
    for (acm_iter it = begin(); it != end(); it++) it->second->blend(a->mixtures[it->first], f);
    host_dirty = true;
}
srx_model *acoustic_model::device() {
    const int G = num_states(), M = num_mix();
    if (!dev) {
        check(srx_model_create(srx::context(), G, M, FV_LEN, &dev));
        host_dirty = true;
    }
    if (host_dirty) {
        vector<double> mean((size_t)G * M * FV_LEN), var(mean.size());
        vector<float> w((size_t)G * M);
        size_t o = 0;
        for (acm_iter it = begin(); it != end(); it++)
            for (const gaussian &g : it->second->gaussians) {
                for (int i = 0; i < FV_LEN; i++) {
                    mean[o * FV_LEN + i] = (double)g.mean[i];
                    var[o * FV_LEN + i] = (double)g.var[i];
                }
                w[o++] = g.weight;
            }
        check(srx_model_set(dev, mean.data(), var.data(), w.data()));
        host_dirty = false;
    }
    return dev;
}
void acoustic_model::download() {
    if (!dev || host_dirty) return;  // nothing newer on the device
    const int G = num_states(), M = num_mix();
    vector<double> mean((size_t)G * M * FV_LEN), var(mean.size());
    vector<float> w((size_t)G * M);
    check(srx_model_get(dev, mean.data(), var.data(), w.data()));
    size_t o = 0;
    for (acm_iter it = begin(); it != end(); it++)
        for (gaussian &g : it->second->gaussians) {
            for (int i = 0; i < FV_LEN; i++) {
                g.mean[i] = mean[o * FV_LEN + i];
                g.var[i] = var[o * FV_LEN + i];
            }
            g.set_wt(w[o++]);
        }
}
vector<float> acoustic_model::score(const vector<featurevec *> &fvs, bool logsumexp, bool exact) {
    vector<float> x;
    pack_features(fvs, x);
    const int G = num_states();
    vector<float> out(fvs.size() * (size_t)G);
    if (fvs.empty()) return out;
    check(srx_gmm_score(srx::context(), device(), logsumexp ? SRX_MIX_LOGSUMEXP : SRX_MIX_COMPONENT0,
                        exact ? SRX_SCORE_EXACT : SRX_SCORE_MFMA, x.data(), 3 * NPARAMS, (int64_t)fvs.size(), out.data()));
    return out;
}
logprob_t acoustic_model::operator()(featurevec &fv, phone::context ph) {
    vector<featurevec *> one(1, &fv);
    vector<float> s = score(one, g_mix_mode == SRX_MIX_LOGSUMEXP, true);
    return s[(size_t)tied_index(ph)];
}

// ============================================================================================ langmodel
unigram_model::unigram_model(long count) : total_count(count), logtotal((float)std::log((double)count)) {}   // src/langmodel.cpp:12-15
unigram_model::unigram_model(std::istream &in, long count) : total_count(count), logtotal((float)std::log((double)count)) {
    string w;
    long c;
    while (in >> w >> c) put(w, c);   // (the reference's `while (in.good())` reads the last pair twice: same table)
}
void unigram_model::put(string w, long count) {
    float logprob = (float)(std::log((double)count) - (double)logtotal);
    probs[w] = logprob;
}
float unigram_model::get(string w) const {
    std::map<string, float>::const_iterator it = probs.find(w);
    return it != probs.end() ? it->second : -logtotal;   // an unseen word counts as seen once
}

bigram_model::bigram_model(unigram_model *u, long count) : total_count(count), logtotal((float)std::log((double)count)), uni(u) {}
bigram_model::bigram_model(std::istream &in, unigram_model *u, long count)
    : total_count(count), logtotal((float)std::log((double)count)), uni(u) {
    string w1, w2;
    long c;
    while (in >> w1 >> w2 >> c) put(w1, w2, c);
}
void bigram_model::put(string w1, string w2, long count) {   // src/langmodel.cpp:31-34
    probs[w1 + "^" + w2] = (float)(std::log((double)count) - (double)logtotal);
}
float bigram_model::get(string w1, string w2) const {        // src/langmodel.cpp:37-44
    std::map<string, float>::const_iterator it = probs.find(w1 + "^" + w2);
    return it != probs.end() ? it->second - uni->get(w1) : uni->get(w2);
}

// ============================================================================================ hmm
hmm_state::hmm_state(phspec p) : ph(p), ctx(phone::context(p.first)), word(nullptr) {}
float hmm_state::get_logprob(int i) const { return tr[i].second; }
void hmm_state::set_prob(int i, float p) { tr[i].second = std::log(p); }
float hmm_state::get_adv_prob() const { return std::exp(tr[1].second); }
void hmm_state::set_adv_prob(float p) {
    float avail = tr.size() == 3 ? (1 - SIL_SKIP_P) : 1;
    set_prob(0, (1 - p) * avail);
    set_prob(1, p * avail);
}

float state_model::operator()(hmm_state *state) {
    if (ties == NULL) return state->get_logprob(1);
    phone::context c = (*ties)(state->ctx);
    if (counts.count(c) == 0) return 0.5f;
    return probsums[c] / counts[c];
}
void state_model::augment(hmm_state *state) {
    if (ties == NULL) return;
    phone::context c = (*ties)(state->ctx);
    probsums[c] += state->get_adv_prob();
    counts[c] += 1;
}

hmm::hmm(vector<phone::phone> ph, acoustic_model *ac) : acm(ac), temp_acm(NULL), sm(new state_model(NULL)) {
    const int S = (int)ph.size();
    if (S == 0) throw srx_error(SRX_ERR_INVALID, "hmm: empty pronunciation");  // reference: size()-1 underflows
    vector<int32_t> phones(ph.begin(), ph.end()), ntr((size_t)S), ctx3((size_t)3 * S);
    vector<float> tr((size_t)3 * S);
    int rc = srx_hmm_build_training(phones.data(), S, ntr.data(), tr.data(), ctx3.data());
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    states.reserve((size_t)S);
    for (int i = 0; i < S; i++) states.push_back(hmm_state(phspec(ph[i], phone::MIDDLE)));
    for (int i = 0; i < S; i++) {
        for (int k = 0; k < ntr[i]; k++) states[i].tr.push_back(transition(&states[i + k], tr[3 * i + k]));
        states[i].ctx = phone::context((phone::phone)ctx3[3 * i], (phone::phone)ctx3[3 * i + 1], (phone::phone)ctx3[3 * i + 2]);
    }
    for (int i = 0; i < S; i++) idx_map[&states[i]] = i;
}
hmm::hmm(acoustic_model *ac, state_model *smo) : acm(ac), temp_acm(new acoustic_model(ac)), sm(smo) {}

// src/hmm.cpp:92-129: the topology comes from srx_hmm_build_lexicon (words in the map's order, first pronunciation each)
hmm::hmm(pronlex &pr, unigram_model &uni, acoustic_model *ac, state_model *smo)
    : acm(ac), temp_acm(new acoustic_model(ac)), sm(smo), general_topology(true) {
    vector<int32_t> phone_off(1, 0), phones;
    vector<float> logp;
    vector<const string *> word_ptr;
    for (std::map<string, pronlist>::iterator it = pr.pr.begin(); it != pr.pr.end(); it++) {
        const vector<phone::phone> &p = it->second.pronunciations.at(0);
        if (p.empty()) continue;   // (the reference would index states out of order for an empty pronunciation)
        for (phone::phone q : p) phones.push_back((int32_t)q);
        phone_off.push_back((int32_t)phones.size());
        logp.push_back(uni.get(it->first));
        word_ptr.push_back(&it->first);
    }
    const int W = (int)logp.size();
    int64_t ns = 0, nt = 0;
    int rc = srx_hmm_build_lexicon(W, phone_off.data(), phones.data(), logp.data(), (float)LMSF, nullptr, nullptr, nullptr, nullptr, nullptr, &ns, &nt);
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    vector<int32_t> tr_off((size_t)ns + 1), tr_tgt((size_t)nt), ctx3((size_t)ns * 3), word_of((size_t)ns);
    vector<float> tr_lp((size_t)nt);
    rc = srx_hmm_build_lexicon(W, phone_off.data(), phones.data(), logp.data(), (float)LMSF, tr_off.data(), tr_tgt.data(), tr_lp.data(),
                               ctx3.data(), word_of.data(), &ns, &nt);
    if (rc != SRX_OK) throw srx_error(rc, srx_last_error(nullptr));
    states.reserve((size_t)ns);
    for (int64_t i = 0; i < ns; i++) {
        hmm_state st(phspec((phone::phone)ctx3[(size_t)i * 3 + 1], i == 0 ? phone::BEGIN : phone::MIDDLE));
        st.ctx = phone::context((phone::phone)ctx3[(size_t)i * 3], (phone::phone)ctx3[(size_t)i * 3 + 1], (phone::phone)ctx3[(size_t)i * 3 + 2]);
        st.word = word_of[(size_t)i] >= 0 ? word_ptr[(size_t)word_of[(size_t)i]] : nullptr;
        states.push_back(st);
    }
    for (int64_t i = 0; i < ns; i++)
        for (int32_t t = tr_off[(size_t)i]; t < tr_off[(size_t)i + 1]; t++)
            states[(size_t)i].tr.push_back(transition(&states[(size_t)tr_tgt[(size_t)t]], tr_lp[(size_t)t]));
    for (int64_t i = 0; i < ns; i++) idx_map[&states[(size_t)i]] = (int)i;
}
hmm::~hmm() { srx_decode_plan_destroy(dplan); }

void hmm::ensure_dplan(const float *tr_lp_now) {
    srx_ctx *ctx = srx::context();
    if (dplan && tr_lp_now) {
        check(srx_decode_plan_set_transitions(dplan, tr_lp_now));
        return;
    }
    vector<int32_t> tr_off(1, 0), tr_tgt, tied;
    vector<float> tr_lp;
    for (hmm_state &st : states) {
        for (transition &t : st.tr) {
            if (!dplan) tr_tgt.push_back(indexOf(t.first));
            tr_lp.push_back(t.second);
        }
        if (!dplan) {
            tr_off.push_back((int32_t)tr_tgt.size());
            tied.push_back(acm->tied_index(st.ctx));
        }
    }
    if (dplan) {
        check(srx_decode_plan_set_transitions(dplan, tr_lp.data()));
        return;
    }
    check(srx_decode_plan_create(ctx, (int)states.size(), tr_off.data(), tr_tgt.data(), tr_lp_now ? tr_lp_now : tr_lp.data(), tied.data(), &dplan));
}

path hmm::run_graph(vector<featurevec *> &fvs, float &prevp, float beam_width, int final_mode) {
    path result;
    const int64_t T = (int64_t)fvs.size();
    if (T == 0 || states.empty()) {
        if (srx::verbose()) std::cout << "NO PATH" << std::endl;
        return result;
    }
    srx_ctx *ctx = srx::context();
    // the lexicon HMM's image is refreshed by update_states; a (wide) training HMM's transition scores move with every
    // train_transition_probabilities, so its image takes them from the states on every call
    if (!dplan || !general_topology) ensure_dplan(nullptr);
    vector<float> x;
    pack_features(fvs, x);
    Dev<float> d_x;
    d_x.upload(x);
    const int G = acm->num_states();
    Dev<float> d_e((size_t)T * G), d_final(1);
    check(srx_gmm_score_dev(ctx, acm->device(), g_mix_mode, g_score_path, d_x.p, 3 * NPARAMS, T, d_e.p, G, 1));
    Dev<int32_t> d_path((size_t)T), d_status(1), d_fs(1);
    const int64_t fo[2] = {0, T};
    check(srx_viterbi_graph_dev(ctx, dplan, d_e.p, G, 1, fo, 1, beam_width, final_mode, d_path.p, d_final.p, d_fs.p, d_status.p));
    int32_t status = 0;
    d_status.download(&status, 1);
    if (status == 3)   // not a property of the utterance: the frontier sweep's store is too small for this beam
        throw srx_error(SRX_ERR_NOMEM, "hmm::viterbi: the beam keeps more states than the decode workspace holds (srx_decode_plan_set_sweep)");
    if (status != 0) {
        if (srx::verbose()) std::cout << "NO PATH" << std::endl;
        return result;
    }
    float fin = 0;
    d_final.download(&fin, 1);
    float nlp = fin / fvs.size();
    if (srx::verbose()) std::cout << "Viterbi path has normalized logprob " << nlp << "\tdelta: " << nlp - prevp << std::endl;
    prevp = nlp;
    vector<int32_t> p = d_path.get();
    for (int64_t t = 0; t < T; t++) result.push_back(&states[(size_t)p[(size_t)t]]);
    return result;
}

path hmm::decode(vector<featurevec *> fvs, float &prevp, float beam_width) { return run_graph(fvs, prevp, beam_width, 1); }

vector<string> hmm::transcript(const path &p) {
    vector<string> words;
    hmm_state *prev = nullptr;
    for (hmm_state *s : p) {
        if (s != prev && s->word) {
            const int i = indexOf(s);
            const bool first = i >= 1 && states[(size_t)i - 1].word != s->word;   // the state after the start state or after another word's silence
            if (first) words.push_back(*s->word);
        }
        prev = s;
    }
    return words;
}
int hmm::indexOf(hmm_state *s) { return idx_map[s]; }

namespace {
// device image of a set of training HMMs + their frames
struct DeviceBatch {
    vector<int64_t> foff;
    vector<int32_t> soff, tied, ntr;
    vector<float> trlogp;
    Dev<int64_t> d_foff;
    Dev<int32_t> d_soff, d_tied, d_ntr;
    Dev<float> d_tr;
    srx_hmm_batch c;
    int max_states = 1;

    void add(hmm &h, int64_t nframes) {
        if (foff.empty()) foff.push_back(0);
        if (soff.empty()) soff.push_back(0);
        foff.push_back(foff.back() + nframes);
        const int S = (int)h.states.size();
        soff.push_back(soff.back() + S);
        if (S > max_states) max_states = S;
        for (int s = 0; s < S; s++) {
            hmm_state &st = h.states[s];
            tied.push_back(h.acm->tied_index(st.ctx));
            ntr.push_back((int32_t)st.tr.size());
            for (int k = 0; k < 3; k++) trlogp.push_back(k < (int)st.tr.size() ? st.tr[k].second : 0.0f);
        }
    }
    void upload() {
        if (foff.empty()) foff.push_back(0);   // an empty shard: zero utterances, every array still allocated
        if (soff.empty()) soff.push_back(0);
        if (tied.empty()) { d_tied.alloc(1); d_ntr.alloc(1); d_tr.alloc(3); }
        d_foff.upload(foff);
        d_soff.upload(soff);
        d_tied.upload(tied);
        d_ntr.upload(ntr);
        d_tr.upload(trlogp);
        c.n_utt = (int)foff.size() - 1;
        c.frame_off_dev = d_foff.p;
        c.state_off_dev = d_soff.p;
        c.tied_dev = d_tied.p;
        c.ntr_dev = d_ntr.p;
        c.trlogp_dev = d_tr.p;
        c.max_states = max_states;
        c.total_frames = foff.back();
    }
    // device transition log-probs -> the hmm_state objects
    void store_transitions(vector<hmm *> &hs) {
        trlogp = d_tr.get();
        for (size_t u = 0; u < hs.size(); u++)
            for (size_t s = 0; s < hs[u]->states.size(); s++) {
                hmm_state &st = hs[u]->states[s];
                for (size_t k = 0; k < st.tr.size(); k++) st.tr[k].second = trlogp[((size_t)soff[u] + s) * 3 + k];
            }
    }
};
}  // namespace

path hmm::viterbi(vector<featurevec *> fvs, float &prevp, float beam_width) {
    // srx_viterbi_batch_dev keeps an utterance's states in the registers of one wavefront (512 at most); the reference has
    // no limit (src/hmm.cpp:56-88): a wider training HMM goes through the general kernel, which gives the same score and path
    if (general_topology || states.size() > (size_t)SRX_VITERBI_MAX_STATES) return run_graph(fvs, prevp, beam_width, 0);
    path result;
    const int64_t T = (int64_t)fvs.size();
    if (T == 0 || states.empty()) {
        if (srx::verbose()) std::cout << "NO PATH" << std::endl;
        return result;
    }
    srx_ctx *ctx = srx::context();
    vector<float> x;
    pack_features(fvs, x);
    Dev<float> d_x;
    d_x.upload(x);
    const int G = acm->num_states();
    Dev<float> d_e((size_t)T * G);
    check(srx_gmm_score_dev(ctx, acm->device(), g_mix_mode, g_score_path, d_x.p, 3 * NPARAMS, T, d_e.p, G, 1));
    DeviceBatch b;
    b.add(*this, T);
    b.upload();
    Dev<int32_t> d_path((size_t)T), d_status(1);
    Dev<float> d_final(1);
    check(srx_viterbi_batch_dev(ctx, &b.c, d_e.p, G, 1, beam_width, d_path.p, d_final.p, d_status.p));
    int32_t status = 0;
    d_status.download(&status, 1);
    if (status != 0) {
        if (srx::verbose()) std::cout << "NO PATH" << std::endl;
        return result;
    }
    float fin = 0;
    d_final.download(&fin, 1);
    float nlp = fin / fvs.size();
    if (srx::verbose())
        std::cout << "Viterbi path has normalized logprob " << nlp << "\tdelta: " << nlp - prevp << std::endl;
    prevp = nlp;
    vector<int32_t> p = d_path.get();
    for (int64_t t = 0; t < T; t++) result.push_back(&states[(size_t)p[(size_t)t]]);
    return result;
}

void hmm::train_transition_probabilities(path p) {
    if (p.empty()) return;
    srx_ctx *ctx = srx::context();
    DeviceBatch b;
    b.add(*this, (int64_t)p.size());
    b.upload();
    vector<int32_t> idx;
    for (hmm_state *s : p) idx.push_back(indexOf(s));
    Dev<int32_t> d_path, d_status(1);
    d_path.upload(idx);
    check(srx_memset(ctx, d_status.p, 0, sizeof(int32_t)));
    check(srx_train_transitions_dev(ctx, &b.c, d_path.p, d_status.p));
    vector<hmm *> me(1, this);
    b.store_transitions(me);
}

void hmm::train(vector<utterance> &ut) { train(ut, EM_ITERS); }

void hmm::train(vector<utterance> &ut, int iterations) {
    if (!temp_acm) throw srx_error(SRX_ERR_INVALID, "hmm::train: this hmm was not constructed as a trainer");
    srx_ctx *ctx = srx::context();
    srx_comm *comm = srx::communicator();
    if (ut.empty() && srx_comm_world(comm) == 1) return;   // (a rank with an empty shard still takes part in the exchanges)
    // ---- corpus -> device, once ----
    vector<float> x;
    DeviceBatch b;
    vector<hmm *> hs;
    for (utterance &u : ut) {
        pack_features(u.features, x);
        b.add(*u.hmm, (int64_t)u.features.size());
        hs.push_back(u.hmm);
    }
    b.upload();
    const int64_t T = b.c.total_frames;
    const int ld = 3 * NPARAMS;
    Dev<float> d_x;
    if (x.empty()) d_x.alloc(ld);
    else d_x.upload(x);
    // every utterance HMM keeps the acoustic_model it was built with (src/utterance.cpp:20) while this object
    // swaps acm / temp_acm each iteration (src/hmm.cpp:344-346): Viterbi runs under the utterances' model
    acoustic_model *vit_model = ut.empty() ? acm : ut[0].hmm->acm;
    const int G = vit_model->num_states(), M = acm->num_mix();
    // Emissions: the split-f16 and exact score paths look up only each utterance's own states, as hmm::viterbi does
    // (src/hmm.cpp:166), into compact per-utterance blocks; the f32 matrix-core path fills the dense [G][T] matrix.
    struct PlanGuard {
        srx_align_plan *p = nullptr;
        ~PlanGuard() { srx_align_plan_destroy(p); }
    } plan;
    // the fused per-utterance scoring keeps a 256-entry state table per workgroup: wider batches take the full matrix
    const bool compact = g_score_path != SRX_SCORE_MFMA && b.max_states <= 256;
    if (compact) {
        check(srx_align_plan_create(ctx, b.foff.data(), b.soff.data(), b.c.n_utt, &plan.p));
        check(srx_align_plan_cache_frames(plan.p, 1));   // d_x is fixed for all iterations: frame operands are built once
    }
    const size_t n_emis = compact ? (size_t)srx_align_plan_emission_count(plan.p) : (size_t)T * G;
    Dev<float> d_e(n_emis ? n_emis : 1), d_final((size_t)b.c.n_utt), d_logp((size_t)T * M), d_shift((size_t)G * M);
    Dev<int32_t> d_path((size_t)T), d_status((size_t)b.c.n_utt), d_gid((size_t)T);
    Dev<double> d_s1((size_t)G * M * (1 + FV_LEN)), d_s2((size_t)G * M * FV_LEN);
    // the one-pass statistics where the shape allows (M <= 16, D <= 39, G <= 16383: srx_em_fused_len is 0 otherwise)
    const int64_t n12 = g_em_path != 0 ? srx_em_fused_len(acm->device()) : 0;
    Dev<double> d_s12((size_t)(n12 > 0 ? n12 : 1));
    Dev<int32_t> d_under(1);
    vector<float> prevp(ut.size(), 0.0f);
    uint64_t scored_version = 0;   // version of the utterances' model that d_e was scored under (0: never)
    for (int it = 1; it <= iterations; it++) {
        if (srx::verbose()) std::cout << "----------------------Iteration " << it << "-------------------------" << std::endl;
        // Viterbi paths for each utterance (src/hmm.cpp:257-267), emissions state-major [G][T]
        // The utterances' model object is rewritten only every second iteration (it is acm in one and temp_acm in the next,
        // src/hmm.cpp:344-346): while its parameters have not changed (srx_model_version) the emission scores in d_e are the
        // ones a new pass would produce, so iterations 2k+1 and 2k+2 share one scoring pass; the alignment still moves, because
        // the transition probabilities do (src/hmm.cpp:269-271).
        if (T == 0) {
            // an empty shard: nothing to align, the statistics below contribute zeros to the exchanges
        } else {
            srx_model *vm = vit_model->device();
            const bool rescore = srx_model_version(vm) != scored_version;
            if (compact) {
                if (rescore) check(srx_gmm_score_align_dev(ctx, vm, g_mix_mode, g_score_path, d_x.p, ld, &b.c, plan.p, d_e.p));
                check(srx_viterbi_align_dev(ctx, &b.c, plan.p, d_e.p, 1e10f, d_path.p, d_final.p, d_status.p));
            } else {
                if (rescore) check(srx_gmm_score_dev(ctx, vm, g_mix_mode, g_score_path, d_x.p, ld, T, d_e.p, 1, T));
                check(srx_viterbi_batch_dev(ctx, &b.c, d_e.p, 1, T, 1e10f, d_path.p, d_final.p, d_status.p));
            }
            scored_version = srx_model_version(vm);
            if (b.max_states > SRX_VITERBI_MAX_STATES) {
                // utterances wider than srx_viterbi_batch_dev takes (status 2): one by one through the general kernel, on the
                // same emission matrix, with the transition scores the batch holds now
                vector<int32_t> st = d_status.get();
                vector<float> tr;
                for (size_t u = 0; u < ut.size(); u++) {
                    if (st[u] != 2) continue;
                    if (tr.empty()) tr = b.d_tr.get();
                    hmm &h = *ut[u].hmm;
                    vector<float> lp;
                    for (size_t s = 0; s < h.states.size(); s++)
                        for (size_t k = 0; k < h.states[s].tr.size(); k++) lp.push_back(tr[((size_t)b.soff[u] + s) * 3 + k]);
                    h.ensure_dplan(lp.data());
                    const int64_t fo[2] = {b.foff[u], b.foff[u + 1]};
                    check(srx_viterbi_graph_dev(ctx, h.dplan, d_e.p, 1, T, fo, 1, 1e10f, 0, d_path.p, d_final.p + u, nullptr, d_status.p + u));
                }
            }
        }
        if (srx::verbose()) {
            vector<float> fin = d_final.get();
            vector<int32_t> st = d_status.get();
            for (size_t u = 0; u < ut.size(); u++) {
                if (st[u] != 0) {
                    std::cout << "NO PATH" << std::endl;
                    continue;
                }
                float nlp = fin[u] / ut[u].features.size();
                std::cout << "Viterbi path has normalized logprob " << nlp << "\tdelta: " << nlp - prevp[u] << std::endl;
                prevp[u] = nlp;
            }
        }
        // transition probabilities of the utterance HMMs (src/hmm.cpp:269-271)
        if (T > 0) check(srx_train_transitions_dev(ctx, &b.c, d_path.p, d_status.p));
        // mixture parameters (src/hmm.cpp:272-341): statistics under acm, rebuilt into temp_acm
        if (T > 0) check(srx_em_gather_gid_dev(ctx, &b.c, d_path.p, d_status.p, T, d_gid.p));
        srx_model *A = acm->device(), *B = temp_acm->device();
        // `ut` is this rank's shard of the corpus: the sums of the reference's three sequential loops over ALL utterances
        // (src/hmm.cpp:275, 298, 319) are completed across the ranks at the three points below (world 1: no-ops); every
        // rank then rebuilds the identical temp_acm
        bool done = false;
        if (n12 > 0) {
            const bool given = g_em_path == 1;
            if (given) {
                check(srx_em_shift_dev(ctx, A, d_x.p, ld, d_gid.p, T, d_logp.p, d_shift.p));
                check(srx_allreduce_max_f32_dev(comm, d_shift.p, (int64_t)G * M));
            }
            check(srx_em_fused_dev(ctx, A, d_x.p, ld, d_gid.p, T, given ? d_logp.p : nullptr, given ? d_shift.p : nullptr, d_s12.p));
            check(srx_allreduce_sum_f64_dev(comm, d_s12.p, n12));
            check(srx_em_fused_finish_dev(ctx, A, B, d_s12.p, given ? d_shift.p : nullptr, GMM_BLEND_FAC, given ? nullptr : d_under.p));
            done = given;
            if (!given) {
                // formed from the SUMMED statistics: every rank reads the same count and takes the same branch
                int32_t lost = 0;
                d_under.download(&lost, 1);
                done = lost == 0;
                if (!done && srx::verbose())
                    std::cout << lost << " components lost to f64 underflow: this iteration through the data-shifted passes" << std::endl;
            }
        }
        if (!done) {
            check(srx_em_shift_dev(ctx, A, d_x.p, ld, d_gid.p, T, d_logp.p, d_shift.p));
            check(srx_allreduce_max_f32_dev(comm, d_shift.p, (int64_t)G * M));
            check(srx_em_pass1_dev(ctx, A, d_x.p, ld, d_gid.p, T, d_logp.p, d_shift.p, d_s1.p));
            check(srx_allreduce_sum_f64_dev(comm, d_s1.p, (int64_t)G * M * (1 + FV_LEN)));
            check(srx_em_means_dev(ctx, B, d_s1.p, d_shift.p));
            check(srx_em_pass2_dev(ctx, A, B, d_x.p, ld, d_gid.p, T, d_logp.p, d_shift.p, d_s2.p));
            check(srx_allreduce_sum_f64_dev(comm, d_s2.p, (int64_t)G * M * FV_LEN));
            check(srx_em_finish_dev(ctx, A, B, d_s1.p, d_s2.p, GMM_BLEND_FAC));
        }
        // swap references (src/hmm.cpp:344-346)
        acoustic_model *t = acm;
        acm = temp_acm;
        temp_acm = t;
    }
    check(srx_sync(ctx));
    b.store_transitions(hs);
    acm->download();
    temp_acm->download();
    // lexicon-HMM parameters from the final paths (src/hmm.cpp:350-357)
    if (sm && sm->ties) {
        // classes are numbered by the tied context itself, (prev * 40 + curr) * 40 + next, so that every rank sums into
        // the same table and the tables add up across ranks
        const int C = phone::NUM_PH * phone::NUM_PH * phone::NUM_PH;
        vector<int32_t> cls;
        for (hmm *h : hs)
            for (hmm_state &st : h->states) {
                phone::context c = (*sm->ties)(st.ctx);
                const int id = ((int)c.prev * phone::NUM_PH + (int)c.curr) * phone::NUM_PH + (int)c.next;
                cls.push_back(st.tr.size() >= 2 ? id : -1);  // get_adv_prob reads tr[1]: absent on a last state (UB in the reference)
            }
        Dev<int32_t> d_cls, d_cnt((size_t)C);
        Dev<double> d_sum((size_t)C);
        if (cls.empty()) d_cls.alloc(1);
        else d_cls.upload(cls);
        check(srx_state_model_dev(ctx, &b.c, d_cls.p, d_path.p, d_status.p, C, d_sum.p, d_cnt.p));
        check(srx_allreduce_sum_f64_dev(comm, d_sum.p, C));
        check(srx_allreduce_sum_i32_dev(comm, d_cnt.p, C));
        vector<double> sum = d_sum.get();
        vector<int32_t> cnt = d_cnt.get();
        for (int c = 0; c < C; c++)
            if (cnt[c] > 0) {
                phone::context k((phone::phone)(c / (phone::NUM_PH * phone::NUM_PH)), (phone::phone)(c / phone::NUM_PH % phone::NUM_PH),
                                 (phone::phone)(c % phone::NUM_PH));
                sm->probsums[k] += (float)sum[c];
                sm->counts[k] += cnt[c];
            }
    }
    update_states();
}

void hmm::update_states() {
    for (size_t i = 1; i < states.size(); i++) {
        float p = (*sm)(&states[i]);
        if (states[i].tr.size() >= 2) states[i].set_adv_prob(p);   // (every lexicon state has two or three; a one-transition state would be UB in the reference)
    }
    if (dplan) {   // the device image follows
        vector<float> tr_lp;
        for (hmm_state &st : states)
            for (transition &t : st.tr) tr_lp.push_back(t.second);
        check(srx_decode_plan_set_transitions(dplan, tr_lp.data()));
    }
}

// ============================================================================================ utterance
utterance::utterance(std::istream &in, string path_base, acoustic_model &acm) : hmm(nullptr) {
    string mfcc_fname;
    in >> mfcc_fname;
    std::getline(in, text);
    std::istringstream ss(text);
    string tok;
    while (ss >> tok) pronunciation.push_back(findPhone(tok));
    hmm = new class hmm(pronunciation, &acm);
    features = read_features(path_base + mfcc_fname, 2);
}
utterance::utterance(const string &t, const vector<phone::phone> &pron, const vector<featurevec *> &fv, acoustic_model &acm)
    : text(t), pronunciation(pron), hmm(new class hmm(pron, &acm)), features(fv) {}

namespace {
// sum over this rank's frames, completed across the ranks: out[0..FV_LEN) the column sums (mu == nullptr) or the sums of
// squared f32 differences to mu; returns the global frame count
int64_t corpus_sums(vector<utterance> &ut, const float *mu, vector<double> &sums) {
    vector<float> x;
    for (utterance &u : ut) pack_features(u.features, x);
    const int64_t T = (int64_t)(x.size() / (3 * NPARAMS));
    srx_ctx *ctx = srx::context();
    Dev<float> d_x, d_mu(FV_LEN);
    if (x.empty()) d_x.alloc(3 * NPARAMS);
    else d_x.upload(x);
    if (mu) d_mu.upload(mu, FV_LEN);
    Dev<double> d_s(FV_LEN + 1);
    check(srx_corpus_sums_dev(ctx, d_x.p, 3 * NPARAMS, T, FV_LEN, mu ? d_mu.p : nullptr, d_s.p));
    const double t_local = (double)T;
    check(srx_memcpy_h2d(ctx, d_s.p + FV_LEN, &t_local, sizeof(double)));
    check(srx_allreduce_sum_f64_dev(srx::communicator(), d_s.p, FV_LEN + 1));
    check(srx_sync(ctx));
    sums = d_s.get();
    return (int64_t)sums[FV_LEN];
}
}  // namespace

// mean(ut) / variance(ut, mu), src/utterance.cpp:32-65, over the utterances of ALL ranks (each rank passes its shard).
featurevec mean(vector<utterance> &ut) {
    vector<double> s;
    const int64_t T = corpus_sums(ut, nullptr, s);
    featurevec mu;
    for (int j = 0; j < FV_LEN; j++) mu[j] = (float)s[j] / (float)(int)T;
    return mu;
}
featurevec variance(vector<utterance> &ut, featurevec &mu) {
    vector<double> s;
    float m[FV_LEN];
    for (int j = 0; j < FV_LEN; j++) m[j] = mu[j];
    const int64_t T = corpus_sums(ut, m, s);
    featurevec var;
    for (int j = 0; j < FV_LEN; j++) var[j] = (float)(s[j] / (double)T);
    return var;
}
