// Exercises the single-call surface of speechrecognition_amd/host/srx_host.hpp the way the reference's
// callers use it and prints the results as "key v v v ..." lines; tests/test_gpu_host_cpp.py compares
// them with the oracle.  Usage: host_api_test <workdir>   (reads <workdir>/in.wav, writes <workdir>/out.mfc)
#include <cstdio>
#include <iostream>

#include "../../speechrecognition_amd/host/srx_host.hpp"

using namespace std;

static void dump(const char *key, const float *v, size_t n) {
    printf("%s", key);
    for (size_t i = 0; i < n; i++) printf(" %.9g", (double)v[i]);
    printf("\n");
}

int main(int argc, char **argv) {
    if (argc < 2) return 2;
    const string dir = string(argv[1]) + "/";
    try {
        // ---- the MFCC block of src/main.cpp:62-92 on one file, call by call ----
        filterbank fbank = *mel_filterbank(7200, 48, 320, 16000);
        Clip wav = read_wav(dir + "in.wav");
        printf("nsamples %d\n", wav.length());
        printf("rms %.17g\n", wav.rmsAmplitude());
        wav.normalizeVolume(1);
        printf("rms_normalized %.17g\n", wav.rmsAmplitude());
        dump("normalized_head", wav.buf->data(), 16);
        // one frame through mfcc() before write_features touches the samples
        Clip frame(wav, 160 * 3, 160 * 3 + 320);
        float *one = mfcc(frame, fbank, NULL);
        dump("mfcc_frame3_nopreemph", one, NPARAMS);
        delete[] one;
        write_features(wav, 320, 160, fbank, dir + "out.mfc");
        dump("preemph_head", wav.buf->data(), 16);   // write_features pre-emphasises in place
        bool threw = false;
        try {
            write_features(wav, 320, 160, fbank, dir + "no/such/dir/out.mfc");
        } catch (int e) {
            threw = e == 1;
        }
        printf("throws_1_on_bad_path %d\n", threw ? 1 : 0);
        try {
            read_wav(dir + "missing.wav");
            printf("missing_wav nothrow\n");
        } catch (wav_io_exception &e) {
            printf("missing_wav %s\n", e.what());
        }
        vector<featurevec *> fvs = read_features(dir + "out.mfc", 2);
        printf("nframes %zu\n", fvs.size());
        dump("feat0", fvs[0]->coeffs, NPARAMS);
        dump("feat0_delta", fvs[0]->deltas, NPARAMS);
        dump("feat5_delta2", fvs[5]->delta2s, NPARAMS);

        // ---- acoustic model + training HMM + Viterbi (src/main.cpp:34-36, src/utterance.cpp:20) ----
        phone::ties acm_ties(phone::ties::NULL_CONTEXT);
        acoustic_model acm(acm_ties, 1);
        printf("tied_states %d\n", acm.num_states());
        phone::ties pc(phone::ties::PHONE_CLASSES), id(phone::ties::IDENT);
        acoustic_model acm_pc(pc, 1);
        printf("tied_states_phone_classes %d\n", acm_pc.num_states());
        // deterministic parameters: mean_g = feature of frame 7g, var = 1 + 0.01 g
        int g = 0;
        for (acm_iter it = acm.begin(); it != acm.end(); it++, g++)
            for (int i = 0; i < FV_LEN; i++) {
                it->second->gaussians[0].mean[i] = (*fvs[(7 * g) % fvs.size()])[i];
                it->second->gaussians[0].var[i] = 1.0 + 0.01 * g;
            }
        acm.mark_host_dirty();
        namespace P = ::phone;   // (clang: after a `using namespace phone`, `phone::` is ambiguous between the namespace and its enum)
        vector<P::phone> pron = {P::SIL, P::HH, P::AH, P::L, P::OW, P::SIL, P::W, P::ER, P::L, P::D, P::SIL};
        hmm h(pron, &acm);
        printf("nstates %zu\n", h.states.size());
        printf("ntr");
        for (hmm_state &s : h.states) printf(" %zu", s.tr.size());
        printf("\n");
        float e = acm(*fvs[3], h.states[2].ctx);
        printf("score_f3_s2 %.9g\n", (double)e);
        float prevp = 0;
        path p = h.viterbi(fvs, prevp, 1e10);
        printf("prevp %.9g\n", (double)prevp);
        printf("path");
        for (hmm_state *s : p) printf(" %d", h.indexOf(s));
        printf("\n");
        h.train_transition_probabilities(p);
        printf("tr");
        for (hmm_state &s : h.states)
            for (transition &t : s.tr) printf(" %.9g", (double)t.second);
        printf("\n");
        vector<utterance> uts;
        uts.push_back(utterance("hello world", pron, fvs, acm));
        featurevec mu = mean(uts);
        featurevec var = variance(uts, mu);
        dump("corpus_mean", mu.coeffs, FV_LEN);
        dump("corpus_var", var.coeffs, FV_LEN);
        // ---- state_model + lexicon HMM + update_states (src/hmm.cpp:39-51, 92-129, 361-366; src/main.cpp:58) ----
        phone::ties hmm_ties(phone::ties::PHONE_CLASSES);
        state_model smod(&hmm_ties);
        printf("aug");   // every path frame whose state has an advance transition: tied class triple + get_adv_prob()
        for (hmm_state *s : p)
            if (s->tr.size() >= 2) {
                smod.augment(s);
                phone::context c = hmm_ties(s->ctx);
                printf(" %d %d %d %.9g", (int)c.prev, (int)c.curr, (int)c.next, (double)s->get_adv_prob());
            }
        printf("\n");
        pronlex lx;
        lx.add("HELLO", {P::HH, P::AH, P::L, P::OW});
        lx.add("WORLD", {P::W, P::ER, P::L, P::D});
        lx.add("A", {P::AH});
        lx.add("LOW", {P::L, P::OW});
        unigram_model uni(1000);
        uni.put("HELLO", 50);
        uni.put("WORLD", 20);
        printf("uni %.9g %.9g %.9g\n", (double)uni.get("HELLO"), (double)uni.get("WORLD"), (double)uni.get("A"));
        hmm lex(lx, uni, &acm, &smod);
        printf("lex_nstates %zu\n", lex.states.size());
        printf("lex_ctx");
        for (hmm_state &s : lex.states) {
            phone::context c = hmm_ties(s.ctx);
            printf(" %d %d %d", (int)c.prev, (int)c.curr, (int)c.next);
        }
        printf("\nlex_ntr");
        for (hmm_state &s : lex.states) printf(" %zu", s.tr.size());
        printf("\nlex_tr_before");
        for (hmm_state &s : lex.states)
            for (transition &t : s.tr) printf(" %.9g", (double)t.second);
        printf("\n");
        float lp2 = 0;
        path dp = lex.decode(fvs, lp2, 300.0f);      // builds the device image before update_states refreshes it
        lex.update_states();
        printf("lex_tr_after");
        for (hmm_state &s : lex.states)
            for (transition &t : s.tr) printf(" %.9g", (double)t.second);
        printf("\n");
        float lp3 = 0;
        path dp2 = lex.decode(fvs, lp3, 300.0f);
        printf("lex_decode_prevp %.9g %.9g\n", (double)lp2, (double)lp3);
        printf("lex_path_after");
        for (hmm_state *s : dp2) printf(" %d", lex.indexOf(s));
        printf("\nlex_words_after");
        for (const string &w : lex.transcript(dp2)) printf(" %s", w.c_str());
        printf("\n");
        vector<featurevec *> few(fvs.begin(), fvs.begin() + 4);
        path none = h.viterbi(few, prevp, 1e10);   // 4 frames cannot reach state 10
        printf("no_path_len %zu\n", none.size());
        // ---- a training HMM wider than the alignment kernel's 512 states (the reference has no limit, src/hmm.cpp:56-88):
        // hmm::viterbi takes it through the general kernel; the second call sees the re-estimated transitions ----
        vector<P::phone> widepron(1, P::SIL);
        for (int i = 0; i < 598; i++) widepron.push_back((P::phone)(1 + (i * 7) % 39));
        widepron.push_back(P::SIL);
        hmm hw(widepron, &acm);
        vector<featurevec *> longf;
        for (int r = 0; r < 10; r++) longf.insert(longf.end(), fvs.begin(), fvs.end());
        float pw = 0;
        path wp = hw.viterbi(longf, pw, 1e10);
        printf("wide_nstates %zu\nwide_prevp %.9g\nwide_path", hw.states.size(), (double)pw);
        for (hmm_state *s : wp) printf(" %d", hw.indexOf(s));
        printf("\n");
        hw.train_transition_probabilities(wp);
        float pw2 = 0;
        path wp2 = hw.viterbi(longf, pw2, 1e10);
        printf("wide_prevp2 %.9g\nwide_path2", (double)pw2);
        for (hmm_state *s : wp2) printf(" %d", hw.indexOf(s));
        printf("\n");
        // ---- the per-frame functions of clip.hpp / mfcc.hpp / gmm.hpp, call by call ----
        Clip fr(wav, 160 * 5, 160 * 5 + 320);           // pre-emphasised samples by now
        dump("fr_samples", wav.buf->data() + 160 * 5, 320);
        fr.window = hamming_window;
        float spec[144];
        fr.partial_dft(spec, 144);                      // src/clip.cpp:174-194, as mfcc() calls it (src/mfcc.cpp:223)
        dump("pd_spec", spec, 144);
        float acc[144];
        for (int i = 0; i < 144; i++) acc[i] = 0.001f * i;
        fr.partial_dft(acc, 144, true);                 // stack
        dump("pd_stack", acc, 144);
        Clip small(wav, 1000, 1000 + 64);               // rectangular window, every bin: the mirrored writes
        dump("small_samples", wav.buf->data() + 1000, 64);
        float full[64];
        small.dft(full);
        dump("dft64", full, 64);
        float *filt = fbank(spec);                      // filterbank::operator(), src/mfcc.cpp:161-167
        dump("fb", filt, 48);
        printf("fb7 %.9g\n", (double)fbank.filters[7](spec));   // filter::operator(), src/mfcc.cpp:143-149
        log(filt, fbank.length());                      // src/utils.hpp:55-60
        dump("logfb", filt, 48);
        float *cc = dct(filt, fbank.length(), NPARAMS); // src/mfcc.cpp:205-214
        dump("dct", cc, NPARAMS);
        delete[] cc;
        delete[] filt;
        filterbank *lin = linear_filterbank(100, 50, 3, 200, 320, 16000);   // src/mfcc.cpp:171-177 (steps by n)
        printf("linfb");
        for (filter &q : lin->filters) printf(" %.9g %.9g %.9g", (double)q.center, (double)q.width, (double)q.height);
        printf("\n");
        // compute_deltas(vector) and the per-vector members, src/mfcc.cpp:73-102
        vector<featurevec *> cp;
        for (int i = 0; i < 6; i++) cp.push_back(new featurevec(fvs[i]->coeffs));
        compute_deltas(cp);
        dump("cd0_delta", cp[0]->deltas, NPARAMS);
        dump("cd3_delta2", cp[3]->delta2s, NPARAMS);
        featurevec lone(cp[2]->coeffs);
        lone.compute_deltas(cp, 2);
        lone.compute_delta2s(cp, 5);
        dump("lone_delta", lone.deltas, NPARAMS);
        dump("lone_delta2", lone.delta2s, NPARAMS);
        printf("dist %.9g\n", (double)cp[1]->dist(*cp[4]));
        // gaussian / gmm as objects of their own, src/gmm.cpp:41-59, 99-100
        gaussian gz;
        for (int i = 0; i < FV_LEN; i++) {
            gz.mean[i] = (*fvs[9])[i];
            gz.var[i] = 0.5 + 0.1 * i;
        }
        gz.set_wt(0.3f);
        printf("gauss %.9g %.9g %.17Lg\n", (double)gz(*fvs[4]), (double)gz.logprob(*fvs[4]), gz.prob(*fvs[4]));
        gmm gm(2);
        gm.gaussians[0] = gz;
        printf("gmm0 %.9g\n", (double)gm(*fvs[4]));
        // EM bookkeeping on the objects, src/gmm.cpp:61-88, 119-154, 188-229
        gaussian ga = gz, gb;
        ga.zetasum = 4;
        ga.divide_means();
        ga.divide_variances();
        printf("div %.17Lg %.17Lg\n", ga.mean[3], ga.var[3]);
        gb.blend(ga, 0.25f);
        printf("blend %.17Lg %.17Lg %.9g %.9g\n", gb.mean[3], gb.var[3], (double)gb.weight, (double)gb.logwt);
        gm.gaussians[0].zetasum = 1.5;
        gm.gaussians[1].zetasum = 2.25;
        gm.sum_zetas();
        printf("zsum %.17Lg\n", gm.zetasum);
        gm.clear_zetas();
        gm.clear();
        printf("cleared %.17Lg %.17Lg %.17Lg %.9g\n", gm.zetasum, gm.gaussians[0].mean[0], gm.gaussians[0].var[0], (double)gm.gaussians[0].weight);
        acoustic_model acm2(&acm);
        for (acm_iter it = acm2.begin(); it != acm2.end(); it++) it->second->gaussians[0].zetasum = 2;
        acm2.sum_zetas();
        acm2.divide_means();
        acm2.divide_variances();
        acm2.blend(&acm, 0.5f);       // 0.5 * (param / 2) + 0.5 * param
        printf("acm_blend %.17Lg %.17Lg\n", acm2.begin()->second->gaussians[0].mean[2] / acm.begin()->second->gaussians[0].mean[2],
               acm2.begin()->second->gaussians[0].var[2] / acm.begin()->second->gaussians[0].var[2]);
        printf("acm_blend_score %.9g\n", (double)acm2(*fvs[3], h.states[2].ctx));   // the device copy follows the host edit
        acm2.clear_zetas();
        acm2.clear();
        printf("acm_cleared %.17Lg %.9g\n", acm2.begin()->second->gaussians[0].var[5], (double)acm2.begin()->second->gaussians[0].weight);
        delete wav.buf;
    } catch (srx_error &e) {
        printf("srx_error %d %s\n", e.code, e.what());
        return 3;
    }
    return 0;
}
