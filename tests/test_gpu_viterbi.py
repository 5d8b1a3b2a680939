"""Parity of batched Viterbi forced alignment (hmm::viterbi, src/hmm.cpp:150-224) and
train_transition_probabilities (src/hmm.cpp:229-243).

Bar: final score bit-exact always; path bit-exact against the oracle run with the same fixed tie
order (descending source index).  The reference's own order is unordered_map iteration order, so
its path is only defined up to exact ties; the oracle's tie counter reports how many occurred."""
import numpy as np
import pytest

from speechrecognition_amd import host, synth

pytestmark = pytest.mark.gpu


def _make_batch(rng, n_utt, G, t_lo, t_hi, max_states=None, words=None):
    phones, ntrs, trs, tieds, Ts = [], [], [], [], []
    for _ in range(n_utt):
        ph = synth.phone_string(rng, words, max_states)
        ntr, tr, _ = host.build_training_hmm(ph)
        phones.append(ph)
        ntrs.append(ntr)
        trs.append(tr)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
        Ts.append(int(rng.integers(t_lo, t_hi)))
    fo = np.zeros(n_utt + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    return phones, ntrs, trs, tieds, fo


def _check(ctx, oracle, ntrs, trs, tieds, fo, emis, beam=1e10, state_major=False):
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    if state_major:   # [G, T] layout, the one srx_gmm_score_dev writes fastest
        T = emis.shape[0]
        Tp = (T + 3) // 4 * 4 if state_major == "aligned" else T     # 16-byte aligned rows: four frames per load
        buf = np.full((emis.shape[1], Tp), np.nan, np.float32)
        buf[:, :T] = emis.T
        emis_d = ctx.array(buf)
        path_d, final_d, status_d = hb.viterbi(emis_d, 1, Tp, beam)
    else:
        emis_d = ctx.array(emis)
        path_d, final_d, status_d = hb.viterbi(emis_d, emis.shape[1], 1, beam)
    ctx.sync()
    path, final, status = path_d.get(), final_d.get(), status_d.get()
    for u in range(len(ntrs)):
        e = emis[fo[u]:fo[u + 1]]
        r = oracle.viterbi(e, tieds[u], ntrs[u], trs[u], beam, 0)
        assert status[u] == r["status"], (u, status[u], r["status"])
        if r["status"] == 0:
            assert final[u].tobytes() == r["final"].tobytes(), (u, final[u], r["final"])
            assert np.array_equal(path[fo[u]:fo[u + 1]], r["path"]), u
    return hb, path_d, status_d, path, status


@pytest.mark.parametrize("words,spl", [(4, 1), (20, 2), (34, 4)])
def test_viterbi_bit_exact(ctx, oracle, words, spl):
    rng = np.random.default_rng(20 + spl)
    G = 50
    _, ntrs, trs, tieds, fo = _make_batch(rng, 24, G, 200, 800, max_states=64 * spl, words=words)
    mx = max(len(n) for n in ntrs)
    assert (mx > 64 * (spl // 2)) or spl == 1
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 20).astype(np.float32)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, state_major=True)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, state_major="aligned")     # four frames per load


def test_viterbi_mixed_widths_in_one_batch(ctx, oracle):
    """Utterances with at most 64, at most 128 and more than 128 states in one batch: the kernel is built for the
    widest and each utterance takes the body of its own width (one, two or four states per lane); every emission
    layout, with and without a beam, and the compact per-utterance layout of srx_viterbi_align_dev."""
    rng = np.random.default_rng(77)
    G = 60
    parts = [_make_batch(rng, 6, G, 40, 300, max_states=64 * spl, words=w) for w, spl in ((3, 1), (20, 2), (34, 4), (5, 1))]
    ntrs = [x for p in parts for x in p[1]]
    trs = [x for p in parts for x in p[2]]
    tieds = [x for p in parts for x in p[3]]
    Ts = np.concatenate([np.diff(p[4]) for p in parts])
    order = rng.permutation(len(ntrs))
    ntrs, trs, tieds, Ts = [ntrs[i] for i in order], [trs[i] for i in order], [tieds[i] for i in order], Ts[order]
    widths = [len(n) for n in ntrs]
    assert min(widths) <= 64 and any(64 < w <= 128 for w in widths) and max(widths) > 128
    fo = np.zeros(len(ntrs) + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 20).astype(np.float32)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, state_major="aligned")
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, beam=80.0, state_major="aligned")
    # compact layout: utterance u owns [T_u][S_u] at emis_off[u]
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    hb.align_plan()
    comp = np.empty(hb.emis_count, np.float32)
    for u in range(len(ntrs)):
        comp[hb.emis_off[u]:hb.emis_off[u + 1]] = emis[fo[u]:fo[u + 1]][:, tieds[u]].ravel()
    path_d, final_d, status_d = hb.viterbi_align(ctx.array(comp))
    ctx.sync()
    path, final, status = path_d.get(), final_d.get(), status_d.get()
    for u in range(len(ntrs)):
        r = oracle.viterbi(emis[fo[u]:fo[u + 1]], tieds[u], ntrs[u], trs[u], 1e10, 0)
        assert status[u] == r["status"], u
        if r["status"] == 0:
            assert final[u].tobytes() == r["final"].tobytes(), u
            assert np.array_equal(path[fo[u]:fo[u + 1]], r["path"]), u
    hb.close()


def test_viterbi_wide_utterances_eight_states_per_lane(ctx, oracle):
    """More than 256 states per utterance (the reference has no limit, src/hmm.cpp:56-88): up to
    SRX_VITERBI_MAX_STATES = 512 the alignment kernel takes eight states per lane, bit-exact in every layout and with a
    beam; a still wider utterance gets status 2 there and the same score and path from the general kernel
    (srx_viterbi_graph_dev), which is what hmm::viterbi / hmm::train of the host layer fall back to."""
    rng = np.random.default_rng(88)
    G = 45
    sizes = [300, 512, 257, 40, 130, 600]
    ntrs, trs, tieds, Ts, phs = [], [], [], [], []
    for S in sizes:
        ph = rng.integers(1, 40, S).astype(np.int32)
        ph[rng.random(S) < 0.12] = 0           # optional silences: three-transition states
        ph[0] = ph[-1] = 0
        ntr, tr, _ = host.build_training_hmm(ph)
        phs.append(ph); ntrs.append(ntr); trs.append(tr)
        tieds.append(rng.integers(0, G, S).astype(np.int32))
        Ts.append(S + int(rng.integers(30, 200)))
    fo = np.zeros(len(sizes) + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 20).astype(np.float32)
    n_ok = len(sizes) - 1
    # the five that fit: every layout, and a beam
    _check(ctx, oracle, ntrs[:n_ok], trs[:n_ok], tieds[:n_ok], fo[:n_ok + 1], emis)
    _check(ctx, oracle, ntrs[:n_ok], trs[:n_ok], tieds[:n_ok], fo[:n_ok + 1], emis, state_major=True)
    _check(ctx, oracle, ntrs[:n_ok], trs[:n_ok], tieds[:n_ok], fo[:n_ok + 1], emis, state_major="aligned")
    _check(ctx, oracle, ntrs[:n_ok], trs[:n_ok], tieds[:n_ok], fo[:n_ok + 1], emis, beam=150.0, state_major="aligned")
    # all six: the 600-state utterance is refused (status 2), the others are as before
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    assert hb.max_states == 600
    emis_d = ctx.array(emis)
    path_d, final_d, status_d = hb.viterbi(emis_d, G, 1, 1e10)
    ctx.sync()
    path, final, status = path_d.get(), final_d.get(), status_d.get()
    assert list(status) == [0, 0, 0, 0, 0, 2]
    for u in range(n_ok):
        r = oracle.viterbi(emis[fo[u]:fo[u + 1]], tieds[u], ntrs[u], trs[u], 1e10, 0)
        assert final[u].tobytes() == r["final"].tobytes() and np.array_equal(path[fo[u]:fo[u + 1]], r["path"]), u
    # ... and goes through the general kernel on the same emission matrix
    u = n_ok
    S = sizes[u]
    off = np.zeros(S + 1, np.int32)
    off[1:] = np.cumsum(ntrs[u])
    tgt = np.concatenate([np.arange(s, s + ntrs[u][s]) for s in range(S)]).astype(np.int32)
    lp = np.concatenate([trs[u][s, :ntrs[u][s]] for s in range(S)]).astype(np.float32)
    g = host.GraphHmm(ctx, off, tgt, lp, tieds[u])
    r = oracle.viterbi(emis[fo[u]:fo[u + 1]], tieds[u], ntrs[u], trs[u], 1e10, 0)
    for sweep in (host.DECODE_DENSE, host.DECODE_FRONTIER):
        g.set_sweep(sweep)
        p, f, fs, st = g.viterbi(emis_d, G, 1, fo[u:u + 2], 1e10, 0)
        assert st[0] == 0 == r["status"] and f[0].tobytes() == r["final"].tobytes()
        assert np.array_equal(p[fo[u]:fo[u + 1]], r["path"])
    g.close()
    # train_transition_probabilities on the wide batch (its per-state counters are sized by max_states)
    hb.train_transition_probabilities(path_d, status_d)
    ctx.sync()
    got, so = hb.trlogp(), hb.state_off
    for u in range(n_ok):
        want = oracle.train_transitions(path[fo[u]:fo[u + 1]], ntrs[u], trs[u])
        np.testing.assert_allclose(got[so[u]:so[u + 1]], want, rtol=3e-6, atol=1e-7)
    np.testing.assert_array_equal(got[so[n_ok]:so[n_ok + 1]], trs[n_ok])      # status 2: untouched
    hb.close()


def test_viterbi_positive_scores_and_short_T(ctx, oracle):
    """Log-densities are routinely positive (log(2/pi) constant, src/gmm.cpp:43); T barely reaching the
    last state, T too short (NO PATH, src/hmm.cpp:212-215), T = 1."""
    rng = np.random.default_rng(31)
    G = 40
    ph = np.array([0, 5, 6, 0, 7, 8, 9, 0], np.int32)
    ntr, tr, _ = host.build_training_hmm(ph)
    Ts = [1, 3, 5, 6, 7, 8, 40]
    fo = np.zeros(len(Ts) + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    n = len(Ts)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 3 + 12).astype(np.float32)
    tieds = [rng.integers(0, G, len(ph)).astype(np.int32) for _ in range(n)]
    _, _, _, path, status = _check(ctx, oracle, [ntr] * n, [tr] * n, tieds, fo, emis)
    assert status[0] == 1 and status[-1] == 0 and status.sum() >= 2
    _check(ctx, oracle, [ntr] * n, [tr] * n, tieds, fo, emis, state_major="aligned")   # utterances starting at every alignment


def test_viterbi_finite_beam(ctx, oracle):
    rng = np.random.default_rng(41)
    G = 30
    _, ntrs, trs, tieds, fo = _make_batch(rng, 16, G, 100, 400, words=5)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 6 - 10).astype(np.float32)
    for beam in (1e9, 200.0, 60.0, 25.0):
        _check(ctx, oracle, ntrs, trs, tieds, fo, emis, beam)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, 60.0, state_major="aligned")


def test_train_transition_probabilities(ctx, oracle):
    rng = np.random.default_rng(51)
    G = 40
    _, ntrs, trs, tieds, fo = _make_batch(rng, 12, G, 150, 500, words=6)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 15).astype(np.float32)
    hb, path_d, status_d, path, status = _check(ctx, oracle, ntrs, trs, tieds, fo, emis)
    for it in range(2):  # twice: the second update starts from non-default probabilities
        hb.train_transition_probabilities(path_d, status_d)
        ctx.sync()
        got = hb.trlogp()
        so = hb.state_off
        for u in range(len(ntrs)):
            trs[u] = oracle.train_transitions(path[fo[u]:fo[u + 1]], ntrs[u], trs[u])
            np.testing.assert_allclose(got[so[u]:so[u + 1]], trs[u], rtol=3e-6, atol=1e-7)


def test_viterbi_nonfinite_and_huge_emissions_follow_the_reference_pruning_test(ctx, oracle):
    """The reference keeps `logp < current_best_logp - beam_width` at its default width of 1e10 (src/hmm.cpp:165):
    a source at -inf (zero-weight component) or 1e10 below the frame's best is skipped, and a frame whose emissions
    are all -inf leaves `next` empty -> NO PATH (status 1), not a path of garbage back-pointers.  Every layout."""
    rng = np.random.default_rng(61)
    G = 30
    _, ntrs, trs, tieds, fo = _make_batch(rng, 10, G, 60, 200, words=4)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 20).astype(np.float32)
    ninf = np.float32(-np.inf)
    # utt 0: one frame with every emission -inf -> NO PATH
    emis[fo[0] + 17, :] = ninf
    # utt 1: one state's column -inf for a stretch (the path has to route around / through it)
    emis[fo[1] + 5:fo[1] + 12, tieds[1][2]] = ninf
    # utt 2: the first frame all -inf -> NO PATH
    emis[fo[2], :] = ninf
    # utt 3: the last frame all -inf
    emis[fo[4] - 1, :] = ninf
    # utt 4: finite but absurd magnitudes: a frame 3e10 below everything (pruned at 1e10 in the reference) ...
    emis[fo[4] + 9, :] = np.float32(-3e10)
    # utt 5: ... and a single state 3e10 below the others for some frames
    emis[fo[5] + 20:fo[5] + 26, tieds[5][1]] = np.float32(-3e10)
    # utt 6: -inf everywhere except one frame
    emis[fo[6] + 1:fo[7], :] = ninf
    _, _, _, _, status = _check(ctx, oracle, ntrs, trs, tieds, fo, emis)
    assert status[0] == 1 and status[2] == 1 and status[7] == 0 and status[8] == 0 and status[9] == 0
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, state_major=True)
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, state_major="aligned")
    _check(ctx, oracle, ntrs, trs, tieds, fo, emis, beam=50.0)


def test_viterbi_nan_emissions_take_the_pruning_body(ctx, oracle):
    """a tied state whose parameters became 0/0 (never visited in training) scores NaN on every frame -- reference
    behaviour.  A NaN delta used to slip past the no-beam body's "largest |delta|" record (fmaxf drops NaN operands) and the
    utterance was never repeated through the body that applies the reference's pruning test.  Now it is: the result is the
    one the beam body gives, which is the oracle's with the same visiting order (a NaN candidate is kept only where it
    arrives first, src/hmm.cpp:171-183: `logp > next[target]` is false for NaN on either side)."""
    rng = np.random.default_rng(62)
    G = 30
    _, ntrs, trs, tieds, fo = _make_batch(rng, 6, G, 60, 200, words=4)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 20).astype(np.float32)
    emis[fo[1]:fo[2], tieds[1][3]] = np.nan          # one tied state NaN throughout an utterance
    emis[fo[3] + 10:fo[3] + 14, tieds[3][1]] = np.nan  # and for a few frames only
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    emis_d = ctx.array(emis)
    res = {}
    for beam in (1e10, 1e9):                          # no-beam body (with its repeat) and the pruning body itself
        p, f, s = hb.viterbi(emis_d, G, 1, beam)
        ctx.sync()
        res[beam] = (p.get(), f.get(), s.get())
    assert np.array_equal(res[1e10][2], res[1e9][2])
    for u in range(len(ntrs)):
        if res[1e10][2][u] == 0:
            a, b = res[1e10][1][u], res[1e9][1][u]
            assert a.tobytes() == b.tobytes() or (np.isnan(a) and np.isnan(b)), u
            assert np.array_equal(res[1e10][0][fo[u]:fo[u + 1]], res[1e9][0][fo[u]:fo[u + 1]]), u
    for u in (0, 2, 4, 5):                            # the untouched utterances: the oracle, as ever
        r = oracle.viterbi(emis[fo[u]:fo[u + 1]], tieds[u], ntrs[u], trs[u], 1e10, 0)
        assert res[1e10][2][u] == r["status"] == 0 and res[1e10][1][u].tobytes() == r["final"].tobytes()
        assert np.array_equal(res[1e10][0][fo[u]:fo[u + 1]], r["path"])
    report = {}
    for u in (1, 3):                                  # the NaN utterances against the oracle: recorded, see the assertion below
        r = oracle.viterbi(emis[fo[u]:fo[u + 1]], tieds[u], ntrs[u], trs[u], 1e10, 0)
        same = res[1e10][2][u] == r["status"] and (r["status"] != 0 or np.array_equal(res[1e10][0][fo[u]:fo[u + 1]], r["path"]))
        report[u] = bool(same)
    assert all(report.values()), report
    hb.close()
