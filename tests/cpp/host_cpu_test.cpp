// CPU-only walk through the parts of the host layer that never touch the GPU -- the text front end, WAV reading and
// writing, the topology builders, the unigram table, sharding, the host-side model bookkeeping -- built with
// -fsanitize=address,undefined (speechrecognition_amd/host/Makefile: target `sanitize`) and fed the awkward inputs a corpus
// walk meets: empty files, lines ending in blanks, truncated and malformed WAV headers, words the lexicon does not hold.
// Prints "ok <what>" lines; any sanitizer report or failed check ends it with a non-zero status
// (tests/test_host_sanitizers.py).  Usage: host_cpu_test <golden lexicon dir> <scratch dir>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <sstream>

#include "../../include/srx.h"
#include "../../speechrecognition_amd/host/srx_host.hpp"

using namespace std;

#define CHECK(cond)                                                              \
    do {                                                                         \
        if (!(cond)) {                                                           \
            fprintf(stderr, "CHECK failed at line %d: %s\n", __LINE__, #cond);   \
            exit(1);                                                             \
        }                                                                        \
    } while (0)

static void put_file(const string &name, const string &bytes) {
    ofstream f(name.c_str(), ios::binary);
    f.write(bytes.data(), (streamsize)bytes.size());
}

static string le16(int v) { return string(1, (char)(v & 255)) + string(1, (char)((v >> 8) & 255)); }
static string le32(long v) { return le16((int)(v & 65535)) + le16((int)((v >> 16) & 65535)); }

static string wav_bytes(int channels, int bps, int sps, const vector<int> &samples, long claimed_data = -1, const string &extra_chunk = "") {
    string data;
    for (int v : samples) data += bps == 8 ? string(1, (char)(v & 255)) : le16(v);
    const long dlen = claimed_data < 0 ? (long)data.size() : claimed_data;
    string fmt = le16(1) + le16(channels) + le32(sps) + le32((long)sps * channels * bps / 8) + le16(channels * bps / 8) + le16(bps);
    string body = string("WAVE") + "fmt " + le32((long)fmt.size()) + fmt + extra_chunk + "data" + le32(dlen) + data;
    return string("RIFF") + le32((long)body.size()) + body;
}

template <class F>
static const char *wav_error(F &&f) {
    try {
        f();
    } catch (wav_io_exception &e) {
        return e.what();
    }
    return nullptr;
}

int main(int argc, char **argv) {
    if (argc < 3) return 2;
    setvbuf(stdout, NULL, _IONBF, 0);
    const string lex = string(argv[1]) + "/", tmp = string(argv[2]) + "/";

    // ---- utils ----
    put_file(tmp + "empty.txt", "");
    put_file(tmp + "lines.txt", "one\n\n  two  \nlast-without-newline");
    // src/utils.cpp:40-51 pushes the first getline's result before it looks at the stream: an empty or missing file gives one
    // empty line (the reference pushes an uninitialised buffer for a missing file), and a last line without a newline is lost
    CHECK(readlines(tmp + "empty.txt").size() == 1 && readlines(tmp + "empty.txt")[0].empty());
    CHECK(readlines((tmp + "lines.txt").c_str()).size() == 3);
    CHECK(readlines(tmp + "does-not-exist.txt").size() == 1);
    CHECK(fabsf(hamming_window(0, 320) - 1.0f) < 1e-6f);   // the reference's "+" form peaks at the edges (src/utils.cpp:21-23)
    CHECK(rectangular_window(5, 7) == 1.0f);
    puts("ok utils");

    // ---- WAV: round trips and every way a header can be wrong ----
    {
        vector<int> s16;
        for (int i = 0; i < 1000; i++) s16.push_back((int)(12000 * sin(i * 0.05)) & 65535);
        put_file(tmp + "m16.wav", wav_bytes(1, 16, 16000, s16));
        Clip c = read_wav(tmp + "m16.wav");
        CHECK(c.length() == 1000);
        write_wav(c, 16000, 16, tmp + "m16_again.wav");
        Clip d = read_wav(tmp + "m16_again.wav");
        CHECK(d.length() == 1000);
        for (int i = 0; i < 1000; i++) CHECK(fabsf(c.buf->data()[i] - d.buf->data()[i]) <= 1.0f / 32768 + 1e-7f);
        write_wav(c, 16000, 8, tmp + "m8.wav");
        CHECK(read_wav(tmp + "m8.wav").length() == 1000);
        vector<int> st;
        for (int i = 0; i < 600; i++) st.push_back((i * 37) & 65535);
        put_file(tmp + "st16.wav", wav_bytes(2, 16, 16000, st));
        CHECK(read_wav(tmp + "st16.wav").length() == 300);          // channels are averaged or the first is taken: one value per frame
        put_file(tmp + "extra.wav", wav_bytes(1, 16, 16000, s16, -1, string("LIST") + le32(6) + "abcdef"));
        const char *e_extra = wav_error([&] { CHECK(read_wav(tmp + "extra.wav").length() >= 0); });
        (void)e_extra;                                               // a chunk before "data": read or refused, never overrun
        put_file(tmp + "zero.wav", wav_bytes(1, 16, 16000, vector<int>()));
        const char *e_zero = wav_error([&] { CHECK(read_wav(tmp + "zero.wav").length() == 0); });
        (void)e_zero;
        // a data chunk that claims more than the file holds, and the 0xFFFFFFFF of a streamed file: cut to what is there
        put_file(tmp + "short.wav", wav_bytes(1, 16, 16000, s16, 1000000));
        CHECK(read_wav(tmp + "short.wav").length() == 1000);
        put_file(tmp + "stream.wav", wav_bytes(1, 16, 16000, s16, 0xFFFFFFFFL));
        CHECK(read_wav(tmp + "stream.wav").length() == 1000);
        // truncated at every length up to the full header, and two wrong magic numbers
        const string full = wav_bytes(1, 16, 16000, s16);
        for (size_t n = 0; n <= 60; n += 3) {
            put_file(tmp + "trunc.wav", full.substr(0, n));
            (void)wav_error([&] { (void)read_wav(tmp + "trunc.wav"); });
        }
        string bad = full;
        bad[0] = 'X';
        put_file(tmp + "notriff.wav", bad);
        CHECK(wav_error([&] { (void)read_wav(tmp + "notriff.wav"); }) != nullptr);
        bad = full;
        bad[8] = 'X';
        put_file(tmp + "notwave.wav", bad);
        CHECK(wav_error([&] { (void)read_wav(tmp + "notwave.wav"); }) != nullptr);
        bad = full;
        bad[20] = 3;   // format tag: not PCM
        put_file(tmp + "notpcm.wav", bad);
        CHECK(wav_error([&] { (void)read_wav(tmp + "notpcm.wav"); }) != nullptr);
        CHECK(wav_error([&] { (void)read_wav(tmp + "missing.wav"); }) != nullptr);
    }
    puts("ok wav");

    // ---- Clip views (host side only) ----
    {
        ClipArrayBuffer ab(1000);
        for (int i = 0; i < 1000; i++) ab[i] = (float)i;
        Clip whole(ab);
        CHECK(whole.length() == 1000);
        Clip fr(whole, 0, 320);
        int frames = 1;
        while (fr.slide(160)) frames++;
        CHECK(frames == 5);                  // 1 + floor((1000 - 320 - 1) / 160): the slide that fails has already moved the view
        CHECK(fr.start == 800 && fr[0] == 800.0f);
    }
    puts("ok clip");

    // ---- text front end on the golden lexicon and on lines a PROMPTS file really holds ----
    {
        pronouncer pr((lex + "main.txt").c_str(), (lex + "prefixes.txt").c_str(), (lex + "suffixes.txt").c_str());
        for (const string &line : readlines(lex + "lines.txt")) {
            try {
                vector<phone::phone> p = pr.pronounce(line);
                CHECK(p.size() < 100000);
            } catch (int) {
            }
        }
        const char *odd[] = {"", " ", "   \t ", "A", "a", "ZZZZQQQQXXXX", "UN", "UNUNUNUNUNUNUNUN", "ING", "S", "'S", "HELLO  WORLD ", "  LEADING",
                             "TRAILING   ", "MIXED case Words", "DIGITS 123 45", "PUNCT, UATION. HERE!", "A-B-C", "\xc3\xa9\xc3\xa8", "RE", "RERERERE",
                             "EDEDED", "LYLYLY"};
        for (const char *w : odd) {
            try {
                (void)pr.pronounce(string(w));
            } catch (int) {
            }
            try {
                (void)pr.pronounce_word(string(w));
            } catch (int) {
            }
        }
        string longword(5000, 'A');
        try {
            (void)pr.pronounce_word(longword);
        } catch (int) {
        }
        // a lexicon stream with blank lines, comment-like lines, a line of blanks only, a word without phones, unknown phones
        // an unknown phone name is an error here (the reference's findPhone falls off its end: src/pronounce.cpp:16-26)
        bool threw = false;
        try {
            istringstream bad_in("CAT  K AE1 T\nBAD  QQ ZZ\n");
            pronlex bad_px(bad_in);
        } catch (srx_error &) {
            threw = true;
        }
        CHECK(threw);
        istringstream in("\n;;; comment\nCAT  K AE1 T\nDOG  D AO1 G   \n   \nEMPTY\nCAT(2)  K AE2 T\nLAST  L AE1 S T");
        pronlex px(in);
        CHECK(px.get("CAT") != NULL);
        CHECK(px.get("NOPE") == NULL);
        px.add("NEW", vector<phone::phone>(1, findPhone("K")));
        CHECK(px.get("NEW") != NULL);
        istringstream none("");
        pronlex empty_lex(none);
        CHECK(empty_lex.get("CAT") == NULL);
        for (const char *name : {"", "NOT-A-PHONE", "1", "AE12"}) {
            threw = false;
            try {
                (void)findPhone(name);
            } catch (srx_error &) {
                threw = true;
            }
            CHECK(threw);
        }
        CHECK(findPhone("AE1") == phone::AE && findPhone("SIL") == phone::SIL);
    }
    puts("ok text");

    // ---- ties / contexts ----
    {
        phone::ties nul(phone::ties::NULL_CONTEXT), pc(phone::ties::PHONE_CLASSES), id(phone::ties::IDENT);
        const phone::phone k = findPhone("K"), ae = findPhone("AE"), t = findPhone("T");
        phone::context c(k, ae, t);
        (void)nul(c);
        (void)pc(c);
        (void)id(c);
    }
    puts("ok ties");

    // ---- unigram table ----
    {
        istringstream in("the 500\ncat 20\n\nmalformed\ndog notanumber\n  spaced   7  \nlast 1");
        unigram_model uni(in, 1000);
        unigram_model none(1000);
        none.put("word", 10);
        // bigram with unigram back-off (src/langmodel.cpp:31-44): log P(w2 | w1) = log c(w1 w2)/N - log c(w1)/N, else the unigram of w2
        istringstream bin("the cat 10\nthe dog 5\nbroken line\n");
        bigram_model bi(bin, &uni, 1000);
        const float want = (float)(log(10.0) - (double)(float)log(1000.0)) - uni.get("the");
        CHECK(fabsf(bi.get("the", "cat") - want) <= 1e-6f);
        CHECK(bi.get("cat", "the") == uni.get("the"));                 // never seen: the unigram of w2
        CHECK(bi.get("cat", "zebra") == -(float)log(1000.0));          // ... which for an unseen word is -logtotal
        bigram_model empty(&none, 1000);
        CHECK(empty.get("a", "word") == none.get("word"));
    }
    puts("ok unigram");

    // ---- topology builders through the C ABI (host only) ----
    {
        for (int S : {1, 2, 3, 11, 256}) {
            vector<int32_t> ph((size_t)S), ntr((size_t)S), ctx((size_t)S * 3);
            vector<float> tr((size_t)S * 3);
            for (int i = 0; i < S; i++) ph[(size_t)i] = i % 3 == 1 ? 0 : 1 + i % 20;   // silences in between: skip transitions
            CHECK(srx_hmm_build_training(ph.data(), S, ntr.data(), tr.data(), ctx.data()) == SRX_OK);
            CHECK(ntr[(size_t)S - 1] == 1);
        }
        CHECK(srx_hmm_build_training(nullptr, 3, nullptr, nullptr, nullptr) != SRX_OK);
        for (int W : {0, 1, 2, 50}) {
            vector<int32_t> off((size_t)W + 1, 0), phs;
            vector<float> lp((size_t)W, -2.0f);
            for (int w = 0; w < W; w++) {
                const int len = 1 + w % 5;
                for (int i = 0; i < len; i++) phs.push_back(1 + (w + i) % 30);
                off[(size_t)w + 1] = (int32_t)phs.size();
            }
            if (phs.empty()) phs.push_back(0);
            int64_t ns = 0, nt = 0;
            CHECK(srx_hmm_build_lexicon(W, off.data(), phs.data(), lp.data(), 5.0f, nullptr, nullptr, nullptr, nullptr, nullptr, &ns, &nt) == SRX_OK);
            vector<int32_t> tro((size_t)ns + 1), trt((size_t)nt), cx((size_t)ns * 3), wo((size_t)ns);
            vector<float> trl((size_t)nt);
            CHECK(srx_hmm_build_lexicon(W, off.data(), phs.data(), lp.data(), 5.0f, tro.data(), trt.data(), trl.data(), cx.data(), wo.data(), &ns,
                                        &nt) == SRX_OK);
            CHECK(tro[(size_t)ns] == nt && tro[0] == 0);
            for (int64_t q = 0; q < nt; q++) CHECK(trt[(size_t)q] >= 0 && trt[(size_t)q] < ns);
        }
        {   // a word without phones is refused, not built
            int32_t off[3] = {0, 0, 1}, phs[1] = {3};
            float lp[2] = {-1, -1};
            int64_t ns = 0, nt = 0;
            CHECK(srx_hmm_build_lexicon(2, off, phs, lp, 5.0f, nullptr, nullptr, nullptr, nullptr, nullptr, &ns, &nt) != SRX_OK);
        }
    }
    puts("ok topology");

    // ---- filterbank and frame counts through the C ABI (host only) ----
    {
        for (int bins : {256, 320, 400, 768}) {
            vector<srx_filter> fb(48);
            int bin_max = 0;
            CHECK(srx_mel_filterbank(7200.0f, 48, bins, bins == 768 ? 44100 : 16000, fb.data(), &bin_max) == SRX_OK);
            CHECK(bin_max > 0 && bin_max <= bins);
        }
        srx_mfcc_cfg cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.window_len = 320;
        cfg.frame_shift = 160;
        cfg.n_ceps = 13;
        int64_t so[5] = {0, 321, 321 + 5000, 321 + 5000 + 480, 321 + 5000 + 480 + 16000}, fo[5];
        CHECK(srx_mfcc_frame_offsets(&cfg, so, 4, fo) == SRX_OK);
        CHECK(fo[1] == 1 && fo[4] > fo[3]);
        int64_t bad[2] = {0, 320};
        CHECK(srx_mfcc_frame_offsets(&cfg, bad, 1, fo) != SRX_OK);   // a bare window: the reference reads one past it
        CHECK(srx_mfcc_frame_offsets(&cfg, so, 0, fo) == SRX_OK);    // no utterances
    }
    puts("ok filterbank");

    // ---- sharding ----
    {
        vector<double> costs;
        for (int i = 0; i < 1000; i++) costs.push_back(200 + (i * 7919) % 600);
        for (int world : {1, 2, 3, 8}) {
            vector<char> seen(costs.size(), 0);
            for (int r = 0; r < world; r++)
                for (int u : srx::shard_utterances(costs, world, r)) {
                    CHECK(u >= 0 && u < (int)costs.size() && !seen[(size_t)u]);
                    seen[(size_t)u] = 1;
                }
            for (char c : seen) CHECK(c);
        }
        CHECK(srx::shard_utterances(vector<double>(), 4, 2).empty());
        vector<double> few(3, 1.0);
        size_t total = 0;
        for (int r = 0; r < 8; r++) total += srx::shard_utterances(few, 8, r).size();
        CHECK(total == 3);
    }
    puts("ok sharding");

    // ---- id rendezvous of srx::init_distributed: a file of an earlier launch must not be taken for this one's ----
    {
        const string file = tmp + "srx.id";
        unsigned char old_id[SRX_COMM_ID_BYTES], new_id[SRX_COMM_ID_BYTES], got[SRX_COMM_ID_BYTES];
        memset(old_id, 0x11, sizeof old_id);
        memset(new_id, 0x22, sizeof new_id);
        setenv("MASTER_PORT", "29500", 1);
        setenv("WORLD_SIZE", "2", 1);
        const uint64_t old_nonce = srx::launch_nonce();
        setenv("MASTER_PORT", "29517", 1);
        const uint64_t nonce = srx::launch_nonce();
        CHECK(nonce != old_nonce && nonce == srx::launch_nonce());
        setenv("SRX_COMM_NONCE", "job-42", 1);
        CHECK(srx::launch_nonce() != nonce);
        unsetenv("SRX_COMM_NONCE");
        srx::publish_comm_id(file, old_nonce, old_id);                      // the stale file of a crashed launch
        CHECK(!srx::await_comm_id(file, nonce, got, 50));                   // not ours: keep waiting (here: time out)
        put_file(file, string((const char *)old_id, sizeof old_id));        // a bare id, as an older build wrote it
        CHECK(!srx::await_comm_id(file, nonce, got, 30));
        put_file(file, string("SRXCOMM1"));                                  // a partial file
        CHECK(!srx::await_comm_id(file, nonce, got, 30));
        srx::publish_comm_id(file, nonce, new_id);                           // rank 0 of this launch replaces it
        CHECK(srx::await_comm_id(file, nonce, got, 1000) && !memcmp(got, new_id, sizeof got));
        remove(file.c_str());
        CHECK(!srx::await_comm_id(file, nonce, got, 20));                   // no file at all
        bool threw = false;
        try {
            srx::publish_comm_id(tmp + "no/such/dir/srx.id", nonce, new_id);
        } catch (srx_error &e) {
            threw = e.code == SRX_ERR_IO;
        }
        CHECK(threw);
        unsetenv("MASTER_PORT");
        unsetenv("WORLD_SIZE");
    }
    puts("ok rendezvous");

    // ---- host-side state bookkeeping ----
    {
        featurevec a, b;
        for (int i = 0; i < 39; i++) a[i] = (float)i, b[i] = 1.0f;
        a += b;
        a *= 0.5f;
        hmm_state st(phspec(findPhone("K"), phone::BEGIN));
        (void)st;
    }
    puts("ok state");
    puts("all ok");
    return 0;
}
