// srx -- headless driver: the run-time-flag equivalent of the reference's compile-time-selected main()
// (src/main.cpp:13-15): --mode mfcc is the MFCC block (src/main.cpp:62-92), --mode train the TRAIN block
// (src/main.cpp:25-61), --mode align one Viterbi pass over the same corpus.  The GUI block is not built.
//
//   srx --mode mfcc  --voxforge DIR/ [--list _list]  [--window 320 --shift 160] [--batch 256] [--verbose]
//   srx --mode train --voxforge DIR/ [--list _train] [--iters 80] [--mix 1] [--seed 1] [--out model.txt]
//   srx --mode align --voxforge DIR/ [--list _train] [--model model.txt] [--out paths.txt]
//   srx --mode decode --voxforge DIR/ --lexicon cmu.txt --model model.txt [--unigram FILE --unigram-total N] [--beam 200] [--states]
//        free decoding over the lexicon HMM (src/hmm.cpp:92-129) with the recognition end condition the reference has
//        commented out (src/hmm.cpp:196-209): one line per utterance, "<index> <normalised logprob> WORD WORD ..."
//        (--states: the state indices of the path instead of the words)
//   RANK=r WORLD_SIZE=n LOCAL_RANK=g SRX_COMM_FILE=/shared/path srx --mode train ...    one process per GPU: every rank
//        loads its share of the utterances, statistics are all-reduced with RCCL, rank 0 writes --out
//   srx --mode pronounce --lexicon cmu.txt [--prefixes prefixes.txt --suffixes suffixes.txt]   (text lines on stdin)
//
// train/align take the same --lexicon/--prefixes/--suffixes: with a lexicon the transcripts are read from
// <folder>etc/PROMPTS ("<mfc path relative to DIR> WORD WORD ...") and pronounced as the reference does
// (src/main.cpp:27, 43); without one, from <folder>etc/PHONES ("<mfc path relative to DIR> PH PH ...").
//
// Corpus layout (README:26-39 of the reference): DIR/<list file> names one submission folder per line;
// <folder>etc/PROMPTS has one "<path>/<wavname> words..." line per recording; audio in <folder>wav/<wavname>.wav,
// features go to <folder>mfc/<wavname>.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iostream>
#include <algorithm>
#include <sstream>

#include "srx_host.hpp"

using namespace std;

static string arg(int argc, char **argv, const char *name, const char *def) {
    for (int i = 1; i + 1 < argc; i++)
        if (!strcmp(argv[i], name)) return argv[i + 1];
    return def;
}
static bool flag(int argc, char **argv, const char *name) {
    for (int i = 1; i < argc; i++)
        if (!strcmp(argv[i], name)) return true;
    return false;
}

static int run_mfcc(const string &dir, const string &list, unsigned wl, unsigned hop, int batch) {
    filterbank fbank = *mel_filterbank(7200, 48, wl, 16000);  // src/main.cpp:64
    vector<string> folders = readlines(dir + list);
    vector<Clip> clips;
    vector<string> outs;
    long nfiles = 0, nfail = 0;
    auto flush = [&]() {
        if (clips.empty()) return;
        try {
            write_features_batch(clips, wl, hop, fbank, outs, /*normalize_volume=*/true);
        } catch (int e) {
            if (e == 1) cout << "IO ERROR" << endl;  // src/main.cpp:85-89
            nfail += (long)clips.size();
        }
        for (Clip &c : clips) delete c.buf;  // src/main.cpp:84
        clips.clear();
        outs.clear();
    };
    for (size_t i = 0; i < folders.size(); i++) {
        if (folders[i].empty()) continue;
        if (srx::verbose()) cout << "Working on folder " << i << endl;
        vector<string> prompts = readlines(dir + folders[i] + "etc/PROMPTS");
        for (size_t j = 0; j < prompts.size(); j++) {
            if (prompts[j].empty()) continue;
            size_t s = prompts[j].find_last_of('/');
            size_t e = prompts[j].find_first_of(' ');
            if (s == string::npos) s = 0;
            string wavname = prompts[j].substr(s, e == string::npos ? string::npos : e - s);
            string wav_file = dir + folders[i] + "wav/" + wavname + ".wav";
            try {
                Clip wav = read_wav(wav_file);
                if (wav.length() <= (int)wl) {
                    cout << "skipping " << wav_file << ": " << wav.length() << " samples <= window" << endl;
                    delete wav.buf;
                    nfail++;
                    continue;
                }
                clips.push_back(wav);
                outs.push_back(dir + folders[i] + "mfc/" + wavname);
                nfiles++;
            } catch (wav_io_exception &ex) {
                cout << ex.what() << ": " << wav_file << endl;
                nfail++;
            }
            if ((int)clips.size() >= batch) flush();
        }
    }
    flush();
    cout << "wrote features for " << nfiles - nfail << " of " << nfiles << " recordings" << endl;
    return 0;
}

// One utterance per PROMPTS / PHONES line (src/main.cpp:41-50).  Under a multi-rank launch every rank walks the same
// lists, prices each line (frames from the 4-byte header of its .mfc file x states of its pronunciation), takes its
// share by srx::shard_utterances and loads only that.
static void load_corpus(const string &dir, const string &list, acoustic_model &acm, vector<utterance> &ut, pronouncer *pron) {
    vector<string> folders = readlines(dir + list);
    vector<string> lines;
    for (size_t i = 0; i < folders.size(); i++) {
        if (folders[i].empty()) continue;
        ifstream prompts((dir + folders[i] + (pron ? "etc/PROMPTS" : "etc/PHONES")).c_str());
        string line;
        while (getline(prompts, line))
            if (line.find_first_not_of(" \t\r") != string::npos) lines.push_back(line);
    }
    vector<char> mine(lines.size(), 1);
    if (srx::world() > 1) {
        vector<double> costs(lines.size(), 0.0);
        for (size_t k = 0; k < lines.size(); k++) {
            istringstream in(lines[k]);
            string mfc, text;
            in >> mfc;
            getline(in, text);
            double states = 1, frames = 0;
            try {
                if (pron) states = (double)pron->pronounce(text).size();
                else {
                    istringstream ss(text);
                    string tok;
                    states = 0;
                    while (ss >> tok) states++;
                }
            } catch (int) {
                states = 0;   // unpronounceable: the constructor below throws for it on its owner, as in the reference
            }
            if (FILE *f = fopen((dir + mfc).c_str(), "rb")) {
                int32_t n = 0;
                if (fread(&n, sizeof n, 1, f) == 1 && n > 0) frames = n;
                fclose(f);
            }
            costs[k] = frames * states;
        }
        fill(mine.begin(), mine.end(), 0);
        for (int k : srx::shard_utterances(costs, srx::world(), srx::rank())) mine[(size_t)k] = 1;
    }
    for (size_t k = 0; k < lines.size(); k++) {
        if (!mine[k]) continue;
        istringstream in(lines[k]);
        try {
            ut.push_back(pron ? utterance(in, dir, acm, *pron) : utterance(in, dir, acm));
        } catch (int ex) {
            cout << "Caught exception " << ex << endl;  // src/main.cpp:44-48
        }
    }
    if (srx::world() > 1) cout << "rank " << srx::rank() << " of " << srx::world() << ": ";
    cout << "Got " << ut.size() << " utterances" << (srx::world() > 1 ? " of " + to_string(lines.size()) : string()) << endl;
}

static void dump_model(acoustic_model &acm, const string &fname) {
    FILE *out = fopen(fname.c_str(), "w");
    if (!out) throw 1;
    fprintf(out, "srx-model %d %d %d\n", acm.num_states(), acm.num_mix(), FV_LEN);
    for (acm_iter it = acm.begin(); it != acm.end(); it++) {
        fprintf(out, "state %d %d %d\n", (int)it->first.prev, (int)it->first.curr, (int)it->first.next);
        for (gaussian &g : it->second->gaussians) {
            fprintf(out, "w %.9g\nmean", g.weight);
            for (int i = 0; i < FV_LEN; i++) fprintf(out, " %.17g", (double)g.mean[i]);
            fprintf(out, "\nvar");
            for (int i = 0; i < FV_LEN; i++) fprintf(out, " %.17g", (double)g.var[i]);
            fprintf(out, "\n");
        }
    }
    fclose(out);
}

static bool load_model(acoustic_model &acm, const string &fname) {
    ifstream in(fname.c_str());
    string tag;
    int G, M, D;
    if (!(in >> tag >> G >> M >> D) || tag != "srx-model" || G != acm.num_states() || M != acm.num_mix() || D != FV_LEN) return false;
    // numbers go through strtod: a state that saw no frames holds nan (0/0, as in the reference), which
    // operator>> does not read back
    auto num = [&in]() {
        string t;
        in >> t;
        return strtod(t.c_str(), nullptr);
    };
    for (acm_iter it = acm.begin(); it != acm.end(); it++) {
        int p, c, n;
        in >> tag >> p >> c >> n;
        for (gaussian &g : it->second->gaussians) {
            in >> tag;
            g.set_wt((float)num());
            in >> tag;
            for (int i = 0; i < FV_LEN; i++) g.mean[i] = num();
            in >> tag;
            for (int i = 0; i < FV_LEN; i++) g.var[i] = num();
        }
    }
    acm.mark_host_dirty();
    return in.good() || in.eof();
}

int main(int argc, char **argv) {
    const string mode = arg(argc, argv, "--mode", "");
    string dir = arg(argc, argv, "--voxforge", "");
    srx::set_verbose(flag(argc, argv, "--verbose"));
    const string lexicon = arg(argc, argv, "--lexicon", "");
    if (mode.empty() || (dir.empty() && mode != "pronounce") || (mode == "pronounce" && lexicon.empty())) {
        cerr << "usage: srx --mode mfcc|train|align|decode --voxforge DIR/ [options] | --mode pronounce --lexicon FILE"
                "  (see the head of srx_main.cpp)" << endl;
        return 2;
    }
    if (!dir.empty() && dir[dir.size() - 1] != '/') dir += '/';
    try {
        pronouncer *pron = nullptr;
        if (!lexicon.empty())   // src/main.cpp:27; a missing affix file reads as an empty lexicon
            pron = new pronouncer(lexicon.c_str(), arg(argc, argv, "--prefixes", "/dev/null").c_str(),
                                  arg(argc, argv, "--suffixes", "/dev/null").c_str());
        if (mode == "pronounce") {
            string line;
            while (getline(cin, line)) {
                try {
                    vector<phone::phone> ph = pron->pronounce(line);
                    for (size_t i = 0; i < ph.size(); i++) cout << (i ? " " : "") << phone::names[ph[i]];
                    cout << endl;
                } catch (int ex) {
                    cout << "!" << ex << endl;
                }
            }
            return 0;
        }
        if (mode == "mfcc")
            return run_mfcc(dir, arg(argc, argv, "--list", "_list"), (unsigned)atoi(arg(argc, argv, "--window", "320").c_str()),
                            (unsigned)atoi(arg(argc, argv, "--shift", "160").c_str()), atoi(arg(argc, argv, "--batch", "256").c_str()));
        if (mode == "train" || mode == "align" || mode == "decode") {
            srx::init_distributed();   // RANK / WORLD_SIZE / LOCAL_RANK of a launcher; a plain run is world 1
            srx::seed_random((unsigned)atoi(arg(argc, argv, "--seed", "1").c_str()));
            phone::ties acm_ties(phone::ties::NULL_CONTEXT);   // src/main.cpp:34-36
            phone::ties hmm_ties(phone::ties::PHONE_CLASSES);
            acoustic_model acm(acm_ties, atoi(arg(argc, argv, "--mix", "1").c_str()));
            vector<utterance> ut;
            load_corpus(dir, arg(argc, argv, "--list", "_train"), acm, ut, pron);
            if (ut.empty() && srx::world() == 1) return 1;
            const string model_in = arg(argc, argv, "--model", "");
            if (!model_in.empty()) {
                if (!load_model(acm, model_in)) {
                    cerr << "cannot read model " << model_in << endl;
                    return 1;
                }
            } else {
                featurevec mu = mean(ut);               // src/main.cpp:52-54
                featurevec var = variance(ut, mu);
                acm.initialize(mu, var);
            }
            if (mode == "train") {
                hmm h(&acm, new state_model(&hmm_ties));    // src/main.cpp:58 (trainer object; no lexicon topology)
                h.train(ut, atoi(arg(argc, argv, "--iters", "80").c_str()));
                const string out = arg(argc, argv, "--out", "");
                if (!out.empty() && (srx::rank() == 0 || flag(argc, argv, "--dump-every-rank"))) dump_model(*h.acm, out);   // every rank holds the same model
                cout << "trained " << h.acm->num_states() << " tied states on " << ut.size() << " utterances" << endl;
            } else if (mode == "decode") {
                if (!pron) {
                    cerr << "--mode decode needs --lexicon" << endl;
                    return 2;
                }
                // src/main.cpp:29-31, 58: the unigram table (absent: every word counts as seen once) and the lexicon HMM
                const long total = atol(arg(argc, argv, "--unigram-total", "111452435771").c_str());
                const string uni_file = arg(argc, argv, "--unigram", "");
                ifstream lm(uni_file.c_str());
                unigram_model uni = uni_file.empty() ? unigram_model(total) : unigram_model(lm, total);
                hmm h(pron->main, uni, &acm, new state_model(&hmm_ties));
                const float beam = (float)atof(arg(argc, argv, "--beam", "1e10").c_str());
                const bool want_states = flag(argc, argv, "--states");
                const string out = arg(argc, argv, "--out", "");
                FILE *f = out.empty() ? stdout : fopen(out.c_str(), "w");
                if (!f) throw 1;
                fprintf(f, "lexicon-hmm %zu states\n", h.states.size());
                for (size_t u = 0; u < ut.size(); u++) {
                    float prevp = 0;
                    path p = h.decode(ut[u].features, prevp, beam);
                    fprintf(f, "%zu %.9g", u, p.empty() ? 0.0 : (double)prevp);
                    if (want_states)
                        for (hmm_state *s : p) fprintf(f, " %d", h.indexOf(s));
                    else
                        for (const string &w : h.transcript(p)) fprintf(f, " %s", w.c_str());
                    fprintf(f, "\n");
                }
                if (f != stdout) fclose(f);
            } else {
                const string out = arg(argc, argv, "--out", "");
                FILE *f = out.empty() ? stdout : fopen(out.c_str(), "w");
                if (!f) throw 1;
                for (size_t u = 0; u < ut.size(); u++) {
                    float prevp = 0;
                    path p = ut[u].hmm->viterbi(ut[u].features, prevp);
                    fprintf(f, "%zu %.9g", u, p.empty() ? 0.0 : (double)prevp);
                    for (hmm_state *s : p) fprintf(f, " %d", ut[u].hmm->indexOf(s));
                    fprintf(f, "\n");
                }
                if (f != stdout) fclose(f);
            }
            return 0;
        }
        cerr << "unknown mode " << mode << endl;
        return 2;
    } catch (srx_error &e) {
        cerr << "srx: " << e.what() << " (code " << e.code << ")" << endl;
        return 3;
    } catch (int e) {
        cerr << "srx: exception " << e << endl;
        return 4;
    }
}
