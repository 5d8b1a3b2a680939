"""The C++ host layer (speechrecognition_amd/host: the reference's class surface over the C ABI) and the
headless driver `srx` (run-time-flag equivalent of src/main.cpp:25-92), run as the reference's callers would
run them and checked against the oracle.  Binaries are prebuilt by __graft_entry__.build() (make -C
speechrecognition_amd/host); nothing is compiled here."""
import os
import struct
import subprocess
import wave

import numpy as np
import pytest

from speechrecognition_amd import synth

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "speechrecognition_amd", "lib")
if os.environ.get("SRX_TEST_BIN_DIR"):   # e.g. gpurun_tmp/san: the host-sanitized build of tools/build_host_asan.sh
    LIB = os.path.abspath(os.environ["SRX_TEST_BIN_DIR"])
SRX = os.path.join(LIB, "srx")
API_TEST = os.path.join(LIB, "srx_host_api_test")

PHONE_NAMES = ["SIL", "AA", "AE", "AH", "AO", "AW", "AY", "B", "CH", "D", "DH", "EH", "ER", "EY", "F", "G", "HH", "IH", "IY",
               "JH", "K", "L", "M", "N", "NG", "OW", "OY", "P", "R", "S", "SH", "T", "TH", "UH", "UW", "V", "W", "Y", "Z", "ZH"]


def _write_wav(path, x, channels=1, bits=16):
    """x float in [-1,1] (mono) -> PCM file; returns the samples as read_wav (src/wav.cpp:36-53) will see them."""
    with wave.open(path, "wb") as w:
        w.setnchannels(channels)
        w.setsampwidth(bits // 8)
        w.setframerate(16000)
        if bits == 16:
            q = np.round(x * 32767).astype(np.int16)
            if channels == 2:
                r = np.roll(q, 7)
                w.writeframes(np.stack([q, r], axis=1).tobytes())
                return ((q.astype(np.float32) / np.float32(32768)) + (r.astype(np.float32) / np.float32(32768))) / np.float32(2)
            w.writeframes(q.tobytes())
            return q.astype(np.float32) / np.float32(32768)
        q = np.round((x / 2 + 0.5) * 255).astype(np.uint8)
        w.writeframes(q.tobytes())
        return q.astype(np.float32) / np.float32(255)


def _read_mfc(path):
    raw = open(path, "rb").read()
    n = struct.unpack("<i", raw[:4])[0]
    c = np.frombuffer(raw[4:], np.float32).reshape(n, 13)
    return c


def _frame_rel_err(got, ref):
    scale = np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1e-30)
    return (np.abs(got - ref) / scale).max()


def _run(cmd, env_extra=None, cwd=None):
    env = dict(os.environ)
    env.update(env_extra or {})
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, cwd=cwd, timeout=110)
    assert r.returncode == 0, (cmd, r.returncode, r.stdout[-2000:], r.stderr[-2000:])
    return r.stdout


def _kv(out):
    d = {}
    for ln in out.splitlines():
        parts = ln.split()
        if parts:
            d[parts[0]] = parts[1:]
    return d


def test_host_api_calls_match_oracle(oracle, tmp_path):
    assert os.path.exists(API_TEST), "run __graft_entry__.build() first"
    x = synth.utterance_audio(21, 16000)
    samples = _write_wav(str(tmp_path / "in.wav"), x)
    d = _kv(_run([API_TEST, str(tmp_path)], {"SRX_SCORE_PATH": "exact"}))
    f = lambda k: np.array([float(v) for v in d[k]])
    assert int(d["nsamples"][0]) == 16000
    rms = oracle.lib().orc_rms_amplitude(samples, len(samples))
    assert abs(f("rms")[0] - rms) <= 1e-10 * rms
    xn = oracle.normalize_volume(samples)
    assert abs(f("rms_normalized")[0] - 1.0) < 1e-6
    np.testing.assert_allclose(f("normalized_head"), xn[:16], rtol=3e-7)
    fb, _ = oracle.mel_filterbank(7200, 48, 320, 16000)
    one = np.zeros(13, np.float32)
    oracle.lib().orc_mfcc_frame(np.ascontiguousarray(xn[480:800]), 320, fb.ctypes.data, 48, 13, one, None)
    assert np.abs(f("mfcc_frame3_nopreemph") - one).max() <= 1e-4 * np.abs(one).max()
    ref_ceps, xpre = oracle.write_features(xn, 320, 160, fb, 13)
    np.testing.assert_allclose(f("preemph_head"), xpre[:16], rtol=2e-6, atol=1e-9)
    got = _read_mfc(str(tmp_path / "out.mfc"))
    assert got.shape == ref_ceps.shape and _frame_rel_err(got, ref_ceps) <= 1e-4
    assert d["throws_1_on_bad_path"] == ["1"]
    assert " ".join(d["missing_wav"]) == "Could not open WAV file"
    feats = oracle.read_features(got, 2)       # smoothing + deltas are bit-exact given the same cepstra
    assert int(d["nframes"][0]) == len(feats)
    np.testing.assert_allclose(f("feat0"), feats[0, :13], rtol=1e-7)
    np.testing.assert_allclose(f("feat0_delta"), feats[0, 13:26], rtol=1e-7)
    np.testing.assert_allclose(f("feat5_delta2"), feats[5, 26:], rtol=1e-6, atol=1e-9)
    assert d["tied_states"] == ["40"] and d["tied_states_phone_classes"] == ["4000"]
    assert d["nstates"] == ["11"] and d["ntr"] == "2 2 2 2 3 2 2 2 2 2 1".split()
    # model: state g (= phone id under NULL_CONTEXT ties) has mean = feats[7g % n], var = 1 + 0.01 g
    n = len(feats)
    mean = np.stack([feats[(7 * g) % n, :13] for g in range(40)]).astype(np.float64)[:, None, :]
    var = np.stack([np.full(13, 1.0 + 0.01 * g) for g in range(40)])[:, None, :]
    lw = np.zeros((40, 1), np.float32)
    emis = oracle.gmm_score(feats, mean, var, lw, 13, 0)
    ph = np.array([0, 16, 3, 21, 25, 0, 36, 12, 21, 9, 0], np.int32)
    assert abs(f("score_f3_s2")[0] - emis[3, ph[2]]) <= 1e-6 * max(1, abs(emis[3, ph[2]]))
    ntr, tr, _ = oracle.build_training_hmm(ph)
    r = oracle.viterbi(emis, ph, ntr, tr)
    assert r["status"] == 0
    assert [int(v) for v in d["path"]] == list(r["path"])
    assert abs(f("prevp")[0] - r["prevp"]) <= 1e-6 * abs(r["prevp"])
    tr2 = oracle.train_transitions(r["path"], ntr, tr)
    want = np.concatenate([tr2[s, :ntr[s]] for s in range(11)])
    np.testing.assert_allclose(f("tr"), want, rtol=3e-6, atol=1e-7)
    assert d["no_path_len"] == ["0"]
    # ---- 600 states: beyond srx_viterbi_batch_dev (512), through the general kernel; then again after re-estimation ----
    phw = np.array([0] + [1 + (i * 7) % 39 for i in range(598)] + [0], np.int32)
    ntrw, trw, _ = oracle.build_training_hmm(phw)
    emis_long = np.tile(emis, (10, 1))
    rw = oracle.viterbi(emis_long, phw, ntrw, trw)
    assert d["wide_nstates"] == ["600"] and rw["status"] == 0
    assert [int(v) for v in d["wide_path"]] == list(rw["path"])
    assert abs(f("wide_prevp")[0] - rw["prevp"]) <= 1e-6 * abs(rw["prevp"])
    trw2 = oracle.train_transitions(rw["path"], ntrw, trw)
    rw2 = oracle.viterbi(emis_long, phw, ntrw, trw2)
    assert rw2["status"] == 0 and [int(v) for v in d["wide_path2"]] == list(rw2["path"])
    assert abs(f("wide_prevp2")[0] - rw2["prevp"]) <= 1e-6 * abs(rw2["prevp"]) and rw2["prevp"] != rw["prevp"]
    # ---- state_model::augment / operator() and hmm::update_states on the lexicon HMM (a24; src/hmm.cpp:39-51, 361-366) ----
    aug = np.array([float(v) for v in d["aug"]]).reshape(-1, 4)
    adv_path = np.exp(tr2[r["path"], 1].astype(np.float32)).astype(np.float32)[ntr[r["path"]] >= 2]
    np.testing.assert_allclose(aug[:, 3], adv_path, rtol=2e-6)      # get_adv_prob() of the re-estimated path states, in path order
    keys = [tuple(int(v) for v in row[:3]) for row in aug]
    classes = {k: i for i, k in enumerate(dict.fromkeys(keys))}
    ps, cnt = oracle.state_model(np.array([classes[k] for k in keys], np.int32), aug[:, 3].astype(np.float32), len(classes) + 1)
    np.testing.assert_allclose(f("uni"), [np.float32(np.log(50.0) - np.float32(np.log(1000.0))), np.float32(np.log(20.0) - np.float32(np.log(1000.0))),
                                          -np.float32(np.log(1000.0))], rtol=1e-6)       # src/langmodel.cpp:16-28
    # the lexicon the C++ constructor was given, through the oracle's constructor (words in map order: A, HELLO, LOW, WORLD)
    words = [np.array(w, np.int32) for w in ([3], [16, 3, 21, 25], [21, 25], [36, 12, 21, 9])]
    uni_lp = np.array([f("uni")[2], f("uni")[0], f("uni")[2], f("uni")[1]], np.float32)
    L = oracle.build_lexicon_hmm(words, uni_lp)
    assert int(d["lex_nstates"][0]) == L["n_states"] == 1 + 2 + 5 + 3 + 5
    assert [int(v) for v in d["lex_ntr"]] == list(np.diff(L["tr_off"]))
    np.testing.assert_allclose(f("lex_tr_before"), L["tr_logp"], rtol=1e-6)
    lex_keys = [tuple(int(v) for v in f("lex_ctx")[3 * i:3 * i + 3]) for i in range(L["n_states"])]
    cls = np.array([classes.get(k, len(classes)) for k in lex_keys], np.int32)      # a class never augmented: count 0 -> 0.5
    assert (cls < len(classes)).any() and (cls == len(classes)).any()
    want_tr = oracle.update_states(L["tr_off"], L["tr_logp"], cls, ps, cnt)
    np.testing.assert_allclose(f("lex_tr_after"), want_tr, rtol=3e-6, atol=1e-7)
    assert np.array_equal(want_tr[:4], L["tr_logp"][:4]) and not np.allclose(want_tr, L["tr_logp"])   # the start state keeps its LM scores
    # decoding with the refreshed transitions (device image follows update_states): the oracle on the same exact-path emissions
    rl = oracle.viterbi_graph(emis, L["ctx"][:, 1].copy(), L["tr_off"], L["tr_tgt"], f("lex_tr_after").astype(np.float32), 300.0, 0, 1)
    assert rl["status"] == 0 and [int(v) for v in d["lex_path_after"]] == list(rl["path"])
    assert abs(f("lex_decode_prevp")[1] - float(rl["final"]) / len(feats)) <= 1e-6 * abs(float(rl["final"]) / len(feats))
    assert f("lex_decode_prevp")[0] != f("lex_decode_prevp")[1]
    # ---- the per-frame functions, call by call, on the samples the C++ side printed (bit-exact unless noted) ----
    f32 = lambda k: np.array([np.float32(v) for v in d[k]], np.float32)
    fr = f32("fr_samples")
    ham = np.array([oracle.hamming(k, 320) for k in range(160)], np.float32)
    spec = oracle.partial_dft_win(fr, ham, 320, 144)
    assert f32("pd_spec").tobytes() == spec.tobytes()
    acc0 = (np.float32(0.001) * np.arange(144, dtype=np.float32)).astype(np.float32)
    assert f32("pd_stack").tobytes() == oracle.partial_dft_win(fr, ham, 320, 144, stack=True, out=acc0).tobytes()
    assert f32("dft64").tobytes() == oracle.partial_dft_win(f32("small_samples"), None, 64, 64).tobytes()
    filt = oracle.filterbank_apply(fb, spec)
    assert f32("fb").tobytes() == filt.tobytes() and f32("fb7")[0] == filt[7]
    lf = oracle.log1p(filt)
    got_lf = f32("logfb")
    assert np.abs(got_lf.view(np.int32).astype(np.int64) - lf.view(np.int32).astype(np.int64)).max() <= 1   # device logf vs glibc
    assert f32("dct").tobytes() == oracle.dct(got_lf, 13).tobytes()
    lin = f32("linfb").reshape(3, 3)       # linear_filterbank steps by n = 3 Hz, not by `step` (src/mfcc.cpp:174)
    np.testing.assert_allclose(lin[:, 0], (np.float32(100) + np.arange(3, dtype=np.float32) * 3) / np.float32(16000) * np.float32(320), rtol=1e-6)
    cd = oracle.compute_deltas(feats[:6, :13])
    assert f32("cd0_delta").tobytes() == cd[0, 13:26].tobytes() and f32("cd3_delta2").tobytes() == cd[3, 26:].tobytes()
    assert f32("lone_delta").tobytes() == cd[2, 13:26].tobytes() and f32("lone_delta2").tobytes() == cd[5, 26:].tobytes()
    assert abs(f("dist")[0] - np.sqrt(((feats[1, :13] - feats[4, :13]) ** 2).sum())) <= 1e-5 * f("dist")[0]
    gmean, gvar = feats[9, :13].astype(np.float64), 0.5 + 0.1 * np.arange(13)
    glw = oracle.set_wt(np.array([0.3], np.float32))[0]
    gref = oracle.gaussian_logprob(feats[4, :13], gmean, gvar, glw)
    g = f("gauss")
    assert abs(g[0] - gref) <= 1e-6 * max(1, abs(gref)) and g[0] == g[1] and f("gmm0")[0] == g[0]
    assert abs(g[2] - np.exp(np.longdouble(np.float32(g[0])))) <= 1e-15 * g[2]
    np.testing.assert_allclose(f("div"), [gmean[3] / 4, gvar[3] / 4], rtol=1e-15)
    w_bl = np.float32(np.float32(1) * np.float32(0.25) + np.float32(0.3) * np.float32(0.75))
    np.testing.assert_allclose(f("blend"), [0 * 0.25 + gmean[3] / 4 * 0.75, 1 * 0.25 + gvar[3] / 4 * 0.75, w_bl, np.log(w_bl)], rtol=1e-6)
    assert f("zsum")[0] == 3.75 and list(f("cleared")) == [0, 0, 0, 1]
    np.testing.assert_allclose(f("acm_blend"), [0.75, 0.75], rtol=1e-15)
    e_bl = oracle.gmm_score(feats[3:4], mean * 0.75, var * 0.75, lw, 13, 0)[0, ph[2]]
    assert abs(f("acm_blend_score")[0] - e_bl) <= 1e-6 * max(1, abs(e_bl))
    assert list(f("acm_cleared")) == [0, 1]
    mu, var = oracle.corpus_mean_var(feats, 13)   # reference: f32 running sums; device: f64 sums
    np.testing.assert_allclose(f("corpus_mean"), mu, rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(f("corpus_var"), var, rtol=1e-4)


def _make_corpus(root, rng):
    """A Voxforge-shaped tree (README:26-39): _list / _train, <folder>etc/PROMPTS, wav/, mfc/."""
    folders = ["spk1-20260101-aaa/", "spk2-20260102-bbb/"]
    samples = {}
    for fi, fo in enumerate(folders):
        for sub in ("etc", "wav", "mfc"):
            os.makedirs(os.path.join(root, fo, sub))
        lines = []
        for k in range(3):
            name = "a%02d%d" % (fi, k)
            L = int(rng.integers(12000, 30000))
            x = synth.utterance_audio(100 * fi + k, L)
            kind = (fi * 3 + k) % 3
            samples[(fo, name)] = _write_wav(os.path.join(root, fo, "wav", name + ".wav"), x,
                                             channels=2 if kind == 1 else 1, bits=8 if kind == 2 else 16)
            lines.append("%s/mfc/%s SOME WORDS HERE" % (fo.rstrip("/"), name))
        open(os.path.join(root, fo, "etc", "PROMPTS"), "w").write("\n".join(lines) + "\n")
    for lst in ("_list", "_train"):
        open(os.path.join(root, lst), "w").write("\n".join(folders) + "\n")
    return folders, samples


def _oracle_train(oracle, feats, phone_strings, mean0, var0, w0, iters, D):
    """hmm::train (src/hmm.cpp:249-347) with the oracle, including the acm / temp_acm ping-pong: the
    utterance HMMs keep pointing at object O while the trainer swaps (acm, temp_acm) = (O, C), (C, O), ...  Returns the
    trainer's acm after `iters` iterations as [mean, var, w]."""
    allf = np.concatenate(feats)
    O = [mean0.copy(), var0.copy(), w0.copy()]
    Cm = [mean0.copy(), var0.copy(), w0.copy()]
    acm, temp = O, Cm
    hmms = [list(oracle.build_training_hmm(ph)[:2]) for ph in phone_strings]
    gidx = [ph.astype(np.int32) for ph in phone_strings]      # tied index = phone id under NULL_CONTEXT ties
    for _ in range(iters):
        gids = []
        for u, fv in enumerate(feats):
            e = oracle.gmm_score(fv, O[0], O[1], oracle.set_wt(O[2]), D, 0)
            r = oracle.viterbi(e, gidx[u], hmms[u][0], hmms[u][1])
            assert r["status"] == 0
            hmms[u][1] = oracle.train_transitions(r["path"], hmms[u][0], hmms[u][1])
            gids.append(gidx[u][r["path"]])
        Bm, Bv, Bw, _, _ = oracle.em_statistics(allf, np.concatenate(gids), D, acm[0], acm[1], acm[2], 0.2)
        temp[0][...] = np.asarray(Bm, np.float64)
        temp[1][...] = np.asarray(Bv, np.float64)
        temp[2][...] = Bw
        acm, temp = temp, acm
    return acm


def test_headless_driver_trains_on_an_utterance_wider_than_the_alignment_kernel(oracle, tmp_path):
    """hmm::train with a 600-phone utterance in the corpus (the reference has no state limit, src/hmm.cpp:56-88; the
    alignment kernel holds 512): that utterance is aligned by the general kernel inside the batch loop, with the batch's
    current transition scores, and the trained model matches the oracle's at the bars of the faithful statistics."""
    assert os.path.exists(SRX), "run __graft_entry__.build() first"
    rng = np.random.default_rng(12)
    root = str(tmp_path / "vox") + "/"
    fo = "spk1-20260101-aaa/"
    for sub in ("etc", "wav", "mfc"):
        os.makedirs(os.path.join(root, fo, sub))
    lens = {"wide": 150000, "b": 20000, "c": 26000}
    for k, (name, L) in enumerate(sorted(lens.items())):
        _write_wav(os.path.join(root, fo, "wav", name + ".wav"), synth.utterance_audio(300 + k, L))
    open(os.path.join(root, fo, "etc", "PROMPTS"), "w").write("".join("%smfc/%s X\n" % (fo, n) for n in sorted(lens)))
    for lst in ("_list", "_train"):
        open(os.path.join(root, lst), "w").write(fo + "\n")
    _run([SRX, "--mode", "mfcc", "--voxforge", root])
    phones = {"wide": np.array([0] + [1 + (i * 11) % 39 for i in range(598)] + [0], np.int32),
              "b": synth.phone_string(rng, n_words=3), "c": synth.phone_string(rng, n_words=3)}
    open(os.path.join(root, fo, "etc", "PHONES"), "w").write(
        "".join("%smfc/%s %s\n" % (fo, n, " ".join(PHONE_NAMES[p] for p in phones[n])) for n in sorted(lens)))
    order = sorted(lens)
    feats = [oracle.read_features(_read_mfc(os.path.join(root, fo, "mfc", n)), 2) for n in order]
    assert len(feats[2]) > 700        # "wide" sorts last; enough frames for its 600 states
    allf = np.concatenate(feats)
    G, D = 40, 13
    mu, sd = allf[:, :D].mean(0), allf[:, :D].std(0)
    mean0 = (mu + sd * rng.standard_normal((G, 1, D)) * 0.7).astype(np.float64)
    var0 = (sd ** 2 * rng.uniform(0.8, 1.2, (G, 1, D))).astype(np.float64)
    w0 = np.ones((G, 1), np.float32)
    with open(root + "init.txt", "w") as f:
        f.write("srx-model %d 1 %d\n" % (G, D))
        for g in range(G):
            f.write("state 0 %d 0\nw 1\nmean %s\nvar %s\n" % (g, " ".join(repr(float(v)) for v in mean0[g, 0]),
                                                           " ".join(repr(float(v)) for v in var0[g, 0])))
    iters = 2
    env = {"SRX_SCORE_PATH": "exact", "SRX_EM_PATH": "faithful"}
    _run([SRX, "--mode", "train", "--voxforge", root, "--iters", str(iters), "--model", root + "init.txt", "--out", root + "trained.txt"], env)
    acm = _oracle_train(oracle, feats, [phones[n] for n in order], mean0, var0, w0, iters, D)
    got = open(root + "trained.txt").read().split("\n")
    gm = np.array([[float(v) for v in ln.split()[1:]] for ln in got if ln.startswith("mean")])
    gv = np.array([[float(v) for v in ln.split()[1:]] for ln in got if ln.startswith("var")])
    rm, rv = acm[0][:, 0, :], acm[1][:, 0, :]
    assert np.array_equal(np.isnan(gm), np.isnan(rm))
    ok = ~np.isnan(rm)
    assert ok.sum() == 13 * 40        # the wide utterance visits every phone
    np.testing.assert_allclose(gm[ok], rm[ok], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gv[ok], rv[ok], rtol=1e-7, atol=1e-9)
    # alignment of the wide utterance under the trained model: hmm::viterbi's own fallback
    a = _run([SRX, "--mode", "align", "--voxforge", root, "--model", root + "trained.txt"], env)
    rows = [ln.split() for ln in a.splitlines() if ln and ln[0].isdigit()]
    assert len(rows) == 3 and len(rows[2]) == 2 + len(feats[2]) and int(rows[2][-1]) == 599


def test_headless_driver_mfcc_train_align(oracle, tmp_path):
    assert os.path.exists(SRX), "run __graft_entry__.build() first"
    rng = np.random.default_rng(3)
    root = str(tmp_path / "vox") + "/"
    os.makedirs(root)
    folders, samples = _make_corpus(root, rng)
    out = _run([SRX, "--mode", "mfcc", "--voxforge", root, "--batch", "4"])
    assert "wrote features for 6 of 6" in out
    fb, _ = oracle.mel_filterbank(7200, 48, 320, 16000)
    ceps = {}
    for (fo, name), x in samples.items():
        got = _read_mfc(os.path.join(root, fo, "mfc", name))
        ref, _ = oracle.write_features(oracle.normalize_volume(x), 320, 160, fb, 13)
        assert got.shape == ref.shape, (fo, name)
        assert _frame_rel_err(got, ref) <= 1e-4, (fo, name)
        ceps[(fo, name)] = got
    # ---- TRAIN mode: phone strings stand in for the text front end ----
    keys = sorted(ceps)
    phones = {}
    for fo in folders:
        lines = []
        for (f2, name) in keys:
            if f2 != fo:
                continue
            ph = synth.phone_string(rng, n_words=3)
            phones[(f2, name)] = ph
            lines.append("%smfc/%s %s" % (fo, name, " ".join(PHONE_NAMES[p] for p in ph)))
        open(os.path.join(root, fo, "etc", "PHONES"), "w").write("\n".join(lines) + "\n")
    order = [k for fo in folders for k in keys if k[0] == fo]
    feats = [oracle.read_features(ceps[k], 2) for k in order]
    allf = np.concatenate(feats)
    G, D = 40, 13
    mu, sd = allf[:, :D].mean(0), allf[:, :D].std(0)
    mean0 = (mu + sd * rng.standard_normal((G, 1, D)) * 0.7).astype(np.float64)
    var0 = (sd ** 2 * rng.uniform(0.8, 1.2, (G, 1, D))).astype(np.float64)
    w0 = np.ones((G, 1), np.float32)
    with open(root + "init.txt", "w") as f:
        f.write("srx-model %d 1 %d\n" % (G, D))
        for g in range(G):
            f.write("state 0 %d 0\nw 1\nmean %s\nvar %s\n" % (g, " ".join(repr(float(v)) for v in mean0[g, 0]),
                                                           " ".join(repr(float(v)) for v in var0[g, 0])))
    iters = 3
    # SRX_EM_PATH=faithful: the five statistics steps that reproduce every f32 rounding of the reference (tight bars below);
    # the driver's default, the one-pass statistics on the f64 matrix cores, is compared further down at its own bars
    env = {"SRX_SCORE_PATH": "exact", "SRX_EM_PATH": "faithful"}
    _run([SRX, "--mode", "train", "--voxforge", root, "--iters", str(iters), "--model", root + "init.txt", "--out",
          root + "trained.txt"], env)
    # ---- the same three iterations with the oracle, including the acm / temp_acm ping-pong:
    # the utterance HMMs keep pointing at object O while the trainer swaps (acm, temp_acm) = (O, C), (C, O), ...
    acm = _oracle_train(oracle, feats, [phones[k] for k in order], mean0, var0, w0, iters, D)
    gidx = [phones[k].astype(np.int32) for k in order]      # tied index = phone id under NULL_CONTEXT ties
    got = open(root + "trained.txt").read().split("\n")
    gm = np.array([[float(v) for v in ln.split()[1:]] for ln in got if ln.startswith("mean")])
    gv = np.array([[float(v) for v in ln.split()[1:]] for ln in got if ln.startswith("var")])
    rm, rv = acm[0][:, 0, :], acm[1][:, 0, :]
    assert np.array_equal(np.isnan(gm), np.isnan(rm))
    ok = ~np.isnan(rm)
    assert ok.sum() > 13 * 10
    np.testing.assert_allclose(gm[ok], rm[ok], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(gv[ok], rv[ok], rtol=1e-7, atol=1e-9)
    # ---- the default statistics path of hmm::train (srx_em_fused_dev: log-densities and sums on the f64 matrix cores, one
    # exchange) and the variance-fold-only form: three iterations against the same oracle.  Bars: means
    # within 1e-4 sigma and variances within 1e-4 relative after three iterations for the former; 1e-6 for the latter.
    for em_path, bar in (("fused", 1e-4), ("fold", 1e-6)):
        fn = root + "trained_%s.txt" % em_path
        _run([SRX, "--mode", "train", "--voxforge", root, "--iters", str(iters), "--model", root + "init.txt", "--out", fn],
             {"SRX_SCORE_PATH": "exact", "SRX_EM_PATH": em_path})
        txt = open(fn).read().split("\n")
        fm = np.array([[float(v) for v in ln.split()[1:]] for ln in txt if ln.startswith("mean")])
        fv_ = np.array([[float(v) for v in ln.split()[1:]] for ln in txt if ln.startswith("var")])
        assert np.array_equal(np.isnan(fm), np.isnan(rm))
        dm = (np.abs(fm - rm)[ok] / np.sqrt(rv[ok])).max()
        dv = (np.abs(fv_ - rv)[ok] / rv[ok]).max()
        assert dm <= bar and dv <= bar, (em_path, dm, dv)
    # mean/variance initialisation path (src/main.cpp:52-54; random jitter, so only that it runs and writes a model)
    _run([SRX, "--mode", "train", "--voxforge", root, "--iters", "1", "--out", root + "rand.txt"], env)
    assert open(root + "rand.txt").readline().split() == ["srx-model", "40", "1", "13"]

    # ---- the same runs as a TWO-RANK launch of the C++ driver: each rank loads its share of the utterances, hmm::train /
    # mean / variance complete their sums through the communicator of the C ABI (srx_comm_*).  Both ranks sit on this
    # box's one GPU, which RCCL refuses, so the exchange goes through the host rehearsal transport (SRX_COMM_SHM); the
    # RCCL calls themselves are exercised by test_comm_rccl_single_rank below and by the driver's multi-GPU bench.
    def nums(fn, tag):
        return np.array([[float(v) for v in ln.split()[1:]] for ln in open(fn).read().split("\n") if ln.startswith(tag)])

    def launch2(args, outs, env=env):
        shm = "/srx_t_%d_%d" % (os.getpid(), rng.integers(1 << 30))
        ps = []
        for r in range(2):
            e2 = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", SRX_COMM_SHM=shm, **env)
            ps.append(subprocess.Popen([SRX] + args + ["--out", outs[r], "--dump-every-rank"], env=e2, stdout=subprocess.PIPE,
                                       stderr=subprocess.STDOUT, text=True))
        texts = []
        for p_ in ps:
            try:
                o, _ = p_.communicate(timeout=110)
            except subprocess.TimeoutExpired:
                for q in ps:
                    q.kill()
                raise
            texts.append(o)
        for r, p_ in enumerate(ps):
            assert p_.returncode == 0, (r, texts[r][-2000:])
        return texts

    outs = [root + "trained_r0.txt", root + "trained_r1.txt"]
    texts = launch2(["--mode", "train", "--voxforge", root, "--iters", str(iters), "--model", root + "init.txt"], outs)
    got_n = [int(t.split("Got ")[1].split()[0]) for t in texts]
    assert sum(got_n) == len(order) and min(got_n) >= 1, texts
    assert open(outs[0]).read() == open(outs[1]).read()          # every rank ends with the identical model
    for tag, ref in (("mean", rm), ("var", rv)):
        g2 = nums(outs[0], tag)
        assert np.array_equal(np.isnan(g2), np.isnan(ref))
        np.testing.assert_allclose(g2[ok], ref[ok], rtol=1e-7, atol=1e-9)     # the single-process oracle
    np.testing.assert_allclose(nums(outs[0], "mean")[ok], gm[ok], rtol=1e-9, atol=1e-12)   # and the one-rank run of the driver
    # the driver's default statistics (one pass, ONE exchange per iteration) across two ranks: identical on both, and the
    # one-rank model of the same path up to the order in which the two shards' sums are added
    outs = [root + "fused_r0.txt", root + "fused_r1.txt"]
    launch2(["--mode", "train", "--voxforge", root, "--iters", str(iters), "--model", root + "init.txt"], outs,
            env={"SRX_SCORE_PATH": "exact", "SRX_EM_PATH": "fused"})
    assert open(outs[0]).read() == open(outs[1]).read()
    for tag in ("mean", "var"):
        one, two = nums(root + "trained_fused.txt", tag), nums(outs[0], tag)
        assert np.array_equal(np.isnan(one), np.isnan(two))
        np.testing.assert_allclose(two[ok], one[ok], rtol=1e-9, atol=1e-12)
    outs = [root + "rand_r0.txt", root + "rand_r1.txt"]
    launch2(["--mode", "train", "--voxforge", root, "--iters", "1"], outs)
    assert open(outs[0]).read() == open(outs[1]).read()
    for tag in ("mean", "var"):
        a1, a2 = nums(root + "rand.txt", tag), nums(outs[0], tag)
        m_ = ~np.isnan(a1)
        assert np.array_equal(np.isnan(a1), np.isnan(a2))
        np.testing.assert_allclose(a2[m_], a1[m_], rtol=1e-6, atol=1e-8)      # same seed, same jitter; sums added in another order
    # ---- ALIGN mode under the initial model ----
    out = _run([SRX, "--mode", "align", "--voxforge", root, "--model", root + "init.txt"], env)
    rows = [ln.split() for ln in out.splitlines() if ln and ln[0].isdigit()]
    assert len(rows) == len(order)
    for u, row in enumerate(rows):
        e = oracle.gmm_score(feats[u], mean0, var0, oracle.set_wt(w0), D, 0)
        ntr, tr, _ = oracle.build_training_hmm(phones[order[u]])
        r = oracle.viterbi(e, gidx[u], ntr, tr)
        assert [int(v) for v in row[2:]] == list(r["path"]), u
        assert abs(float(row[1]) - r["prevp"]) <= 1e-6 * abs(r["prevp"])


def test_headless_driver_prompts_through_the_lexicon(ctx, oracle, tmp_path):
    """--lexicon: transcripts from etc/PROMPTS pronounced by the host layer's pronouncer (src/main.cpp:27, 43;
    src/utterance.cpp:14-28) give the alignment that the same phone strings give through etc/PHONES; an
    unpronounceable line drops its utterance (src/main.cpp:44-48)."""
    from oracle import pronounce as P
    lex = [os.path.join(ROOT, "tests", "golden", "lexicon", n + ".txt") for n in ("main", "prefixes", "suffixes")]
    pr = P.load(*lex)
    rng = np.random.default_rng(5)
    root = str(tmp_path / "vox") + "/"
    os.makedirs(root)
    folders, samples = _make_corpus(root, rng)
    _run([SRX, "--mode", "mfcc", "--voxforge", root])
    texts = ["THE QUICK DOG", "WALKING DOGS", "UNHAPPY CAT", "HOUSEBOAT", "MOTHER-IN-LAW", "BEWALKED A CAT"]
    k = 0
    for fo in folders:
        prompts, phones = [], []
        for (f2, name) in sorted(samples):
            if f2 != fo:
                continue
            prompts.append("%smfc/%s %s" % (fo, name, texts[k]))
            try:
                phones.append("%smfc/%s %s" % (fo, name, " ".join(PHONE_NAMES[i] for i in pr.pronounce(texts[k]))))
            except P.Unpronounceable:
                pass
            k += 1
        open(os.path.join(root, fo, "etc", "PROMPTS"), "w").write("\n".join(prompts) + "\n")
        open(os.path.join(root, fo, "etc", "PHONES"), "w").write("\n".join(phones) + "\n")
    env = {"SRX_SCORE_PATH": "exact"}
    _run([SRX, "--mode", "train", "--voxforge", root, "--iters", "2", "--seed", "4", "--out", root + "m.txt"], env)
    a = _run([SRX, "--mode", "align", "--voxforge", root, "--model", root + "m.txt"], env)
    b = _run([SRX, "--mode", "align", "--voxforge", root, "--model", root + "m.txt", "--lexicon", lex[0], "--prefixes", lex[1],
              "--suffixes", lex[2]], env)
    rows_a = [ln for ln in a.splitlines() if ln and ln[0].isdigit()]
    rows_b = [ln for ln in b.splitlines() if ln and ln[0].isdigit()]
    assert len(rows_a) == 5 and rows_a == rows_b
    assert "Caught exception 1" in b and "Got 5 utterances" in b
    # and training from the text gives the model that training from the phone strings gives
    _run([SRX, "--mode", "train", "--voxforge", root, "--iters", "2", "--seed", "4", "--out", root + "m2.txt", "--lexicon", lex[0],
          "--prefixes", lex[1], "--suffixes", lex[2]], env)
    def nums(fn):
        return np.array([float(v) for ln in open(fn) if ln[:4] in ("mean", "var ") for v in ln.split()[1:]])
    m1, m2 = nums(root + "m.txt"), nums(root + "m2.txt")
    assert m1.size == 40 * 26 and np.isfinite(m1).sum() >= 26 * 10
    np.testing.assert_allclose(m2, m1, rtol=1e-9, equal_nan=True)   # summation order inside the device sums is not fixed

    # ---- DECODE mode: the lexicon HMM of src/hmm.cpp:92-129 built by the C++ constructor from the same lexicon file, free
    # decoding with the recognition end condition (src/hmm.cpp:196-209).  Checked against the oracle's lexicon constructor
    # and general-topology Viterbi on the emissions the same exact-path scoring kernel gives (parity unpinned, see
    # tests/test_gpu_decode.py).  The trained model holds NaN rows for phones the corpus never visits (reference behaviour);
    # decoding is run under a model whose every row is finite: the trained rows, and the corpus mean / variance elsewhere.
    from speechrecognition_amd import SRX_MIX_COMPONENT0, SRX_SCORE_EXACT, host
    txt = open(root + "m.txt").read().split("\n")
    gm = np.array([[float(v) for v in ln.split()[1:]] for ln in txt if ln.startswith("mean")]).reshape(40, 1, 13)
    gv = np.array([[float(v) for v in ln.split()[1:]] for ln in txt if ln.startswith("var")]).reshape(40, 1, 13)
    gw = np.array([float(ln.split()[1]) for ln in txt if ln.startswith("w ")], np.float32).reshape(40, 1)
    bad = np.isnan(gm).any(axis=(1, 2)) | np.isnan(gv).any(axis=(1, 2)) | np.isnan(gw[:, 0])
    assert bad.any() and (~bad).sum() >= 10
    gm[bad], gv[bad], gw[bad] = np.nanmean(gm[~bad], axis=0), np.nanmean(gv[~bad], axis=0) * 4.0, 1.0
    with open(root + "mfin.txt", "w") as f:
        f.write("srx-model 40 1 13\n")
        for g_ in range(40):
            f.write("state 0 %d 0\nw %r\nmean %s\nvar %s\n" % (g_, float(gw[g_, 0]), " ".join(repr(float(v)) for v in gm[g_, 0]),
                                                               " ".join(repr(float(v)) for v in gv[g_, 0])))
    out = _run([SRX, "--mode", "decode", "--voxforge", root, "--model", root + "mfin.txt", "--lexicon", lex[0], "--prefixes", lex[1],
                "--suffixes", lex[2], "--beam", "250", "--states", "--unigram-total", "1000"], env)
    rows = [ln.split() for ln in out.splitlines() if ln and ln[0].isdigit()]
    outw = _run([SRX, "--mode", "decode", "--voxforge", root, "--model", root + "mfin.txt", "--lexicon", lex[0], "--prefixes", lex[1],
                 "--suffixes", lex[2], "--beam", "250", "--unigram-total", "1000"], env)
    rows_w = [ln.split() for ln in outw.splitlines() if ln and ln[0].isdigit()]
    model = host.AcousticModel(ctx, 40, 1, 13)
    model.set(gm, gv, gw)

    def check_decode(pr, out, rows, rows_w, beam):
        # the lexicon as the reference's map holds it: spelling -> first pronunciation, sorted by spelling
        # (through the oracle's lexicon parser, which restates the reference's quirks: alternates "(1)" fold into the word and
        #  only the first listed pronunciation is used, src/hmm.cpp:96; a line ending in blanks repeats its last phone)
        entries = {w: np.array(p[0], np.int32) for w, p in pr.main.items() if len(p[0]) > 0}
        spell = sorted(entries)
        uni = np.full(len(spell), -np.float32(np.log(1000.0)), np.float32)        # an empty unigram table: -logtotal for every word
        L = oracle.build_lexicon_hmm([entries[w] for w in spell], uni)
        assert ("lexicon-hmm %d states" % L["n_states"]) in out
        tied = L["ctx"][:, 1].copy()                       # NULL_CONTEXT ties: the tied state of a context is its phone
        used = [(f2, name) for fo_ in folders for (f2, name) in sorted(samples) if f2 == fo_]
        assert len(rows) == 5 == len(rows_w)
        n_words_seen = 0
        ui = 0
        for (f2, name), text in zip(used, texts):
            try:
                pr.pronounce(text)
            except P.Unpronounceable:
                continue                                   # the driver skipped this utterance ("Caught exception 1")
            feats = oracle.read_features(_read_mfc(os.path.join(root, f2, "mfc", name)), 2)
            e = model.score(feats, SRX_MIX_COMPONENT0, SRX_SCORE_EXACT)
            r = oracle.viterbi_graph(e, tied, L["tr_off"], L["tr_tgt"], L["tr_logp"], beam, 0, 1)
            row = rows[ui]
            if r["status"] != 0:
                assert len(row) == 2
            else:
                assert [int(v) for v in row[2:]] == list(r["path"]), ui
                assert abs(float(row[1]) - float(r["final"]) / len(feats)) <= 1e-6 * abs(float(r["final"]) / len(feats))
                # the word line: a word is counted when the path enters its first state
                first = np.flatnonzero((L["word_of"] >= 0) & (np.roll(L["word_of"], 1) != L["word_of"]))
                p_ = r["path"]
                enter = [p_[0]] + [b_ for a_, b_ in zip(p_[:-1], p_[1:]) if a_ != b_]
                want = [spell[L["word_of"][s_]] for s_ in enter if s_ in first]
                assert rows_w[ui][2:] == want, ui
                n_words_seen += len(want)
            ui += 1
        assert ui == 5
        return n_words_seen, L["n_states"]

    assert check_decode(pr, out, rows, rows_w, 250.0)[0] >= 5

    # ---- the same at the size of the reference's own lexicon (src/main.cpp:27, 58: res/lexicon/cmu.txt, 133,860 lines, about
    # 10^6 HMM states; the file does not travel to the GPU box, so the extra words are synthetic, with its length distribution,
    # spelled to sort after the real ones).  hmm::decode goes through the frontier sweep here (srx_decode_frontier.hip): bounded
    # memory -- the dense sweep would need 4 * frames * states bytes of back-pointers, 0.75 GB for the longest utterance.
    big = root + "big_lexicon.txt"
    with open(big, "w") as f:
        f.write(open(lex[0]).read())
        lens = np.clip(rng.poisson(5.4, 133860 - 26) + 1, 1, 20)
        names = [n for n in PHONE_NAMES[1:40]]
        for k, ln in enumerate(lens):
            f.write("ZZW%06d  %s\n" % (k, " ".join(names[i] for i in rng.integers(0, len(names), int(ln)))))
    prb = P.load(big, lex[1], lex[2])
    cmd = [SRX, "--mode", "decode", "--voxforge", root, "--model", root + "mfin.txt", "--lexicon", big, "--prefixes", lex[1],
           "--suffixes", lex[2], "--beam", "200", "--unigram-total", "1000"]
    out_b = _run(cmd + ["--states"], env)
    outw_b = _run(cmd, env)
    rows_b = [ln.split() for ln in out_b.splitlines() if ln and ln[0].isdigit()]
    rows_wb = [ln.split() for ln in outw_b.splitlines() if ln and ln[0].isdigit()]
    seen, n_states = check_decode(prb, out_b, rows_b, rows_wb, 200.0)
    assert n_states > 900000 and seen >= 5
    model.close()


def test_comm_rccl_single_rank_and_noop(ctx):
    """srx_comm_* through the C ABI: the world-1 communicator is a no-op; with SRX_COMM_FORCE_RCCL=1 a ONE-rank RCCL
    communicator is built from srx_comm_unique_id and the three all-reduces run through ncclAllReduce on the context's
    stream (identity on one rank: what is checked is the library lookup, the ABI and the stream ordering with the
    kernels around them -- two ranks on one GPU are refused by RCCL, tools/probe/rccl_same_device.py)."""
    from speechrecognition_amd import host
    rng = np.random.default_rng(0)
    a = rng.standard_normal(100_003)
    c0 = host.Comm(ctx)
    d = ctx.array(a)
    c0.allreduce_sum_f64(d)
    assert c0.bytes_reduced() == 0 and np.array_equal(d.get(), a)
    c0.close()
    os.environ["SRX_COMM_FORCE_RCCL"] = "1"
    try:
        c1 = host.Comm(ctx, 0, 1, host.Comm.unique_id())
    finally:
        del os.environ["SRX_COMM_FORCE_RCCL"]
    f = ctx.array(a.astype(np.float32))
    i = ctx.array(np.arange(1000, dtype=np.int32))
    c1.allreduce_sum_f64(d)
    c1.allreduce_max_f32(f)
    c1.allreduce_sum_i32(i)
    ctx.sync()
    assert np.array_equal(d.get(), a) and np.array_equal(f.get(), a.astype(np.float32)) and np.array_equal(i.get(), np.arange(1000))
    assert c1.bytes_reduced() == 100_003 * 12 + 4000
    c1.close()
