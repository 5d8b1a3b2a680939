"""The lexicon HMM (hmm::hmm(pronlex&, unigram_model&, ...), src/hmm.cpp:92-129) and hmm::viterbi on it (src/hmm.cpp:150-224)
with a finite beam and both end conditions -- SURVEY.md section 8 f3.

PARITY UNPINNED: the reference's main() never runs Viterbi on its lexicon HMM, the selection of the best final state is
commented out there (src/hmm.cpp:196-209) and its unigram file is not in the repository.  The checker is the oracle's
restatement of the same loops on an arbitrary transition list (orc_viterbi_graph), which reproduces the training-HMM
oracle bit for bit on training topologies (tests/test_oracle_pins.py).  Bar: topology arrays identical; score bit-exact;
path identical (the oracle visits sources in descending index, the kernel's tie rule).  Both walks of srx_viterbi_graph_dev are
held to it: the dense sweep and the frontier sweep (srx_decode_plan_set_sweep), the latter up to the size of the
reference's own lexicon (src/main.cpp:58 reads the 133,860 lines of res/lexicon/cmu.txt; the file does not travel to the GPU
box, so the words here are synthetic with its length distribution)."""
import numpy as np
import pytest

from speechrecognition_amd import host, synth

pytestmark = pytest.mark.gpu

G = 40   # NULL_CONTEXT tying: the emission column of a state is its phone


def _lexicon(rng, n_words):
    words = [rng.integers(1, 40, int(rng.integers(1, 7))).astype(np.int32) for _ in range(n_words)]
    counts = rng.integers(1, 5000, n_words)
    uni = (np.log(counts) - np.float32(np.log(counts.sum() * 3.0))).astype(np.float32)   # unigram_model::put, src/langmodel.cpp:16-19
    return words, uni


def _emissions_for(rng, lex, words, T_per_word, n_utt):
    """Frames that roughly follow a random word sequence, so that paths visit many words."""
    emis, fo = [], [0]
    for _ in range(n_utt):
        seq = rng.integers(0, len(words), int(rng.integers(1, 5)))
        rows = []
        for w in seq:
            for ph in list(words[w]) + [0]:
                n = int(rng.integers(2, T_per_word))
                e = rng.standard_normal((n, G)).astype(np.float32) * 2 - 12
                e[:, ph] += 9.0
                rows.append(e)
        e = np.concatenate(rows)
        emis.append(e)
        fo.append(fo[-1] + len(e))
    return np.concatenate(emis), np.array(fo, np.int64)


@pytest.mark.parametrize("sweep", [host.DECODE_DENSE, host.DECODE_FRONTIER])
@pytest.mark.parametrize("n_words", [1, 10, 60, 500])
def test_lexicon_hmm_topology_and_decode_bit_exact(ctx, oracle, n_words, sweep):
    rng = np.random.default_rng(100 + n_words)
    words, uni = _lexicon(rng, n_words)
    a = host.build_lexicon_hmm(words, uni)
    b = oracle.build_lexicon_hmm(words, uni)
    assert a["n_states"] == b["n_states"] == 1 + sum(len(w) + 1 for w in words)
    for k in ("tr_off", "tr_tgt", "tr_logp", "ctx", "word_of"):
        assert a[k].tobytes() == b[k].tobytes(), k
    # the start state fans out to every word with LMSF * unigram (src/hmm.cpp:108); the last phone of a word has three
    # transitions at ln 1/3, the terminating silence two at ln 1/2 (src/hmm.cpp:110-122)
    assert a["tr_off"][1] == n_words and np.array_equal(a["tr_logp"][:n_words], np.float32(5.0) * uni)
    tied = a["ctx"][:, 1].copy()
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    hmm.set_sweep(sweep)
    emis, fo = _emissions_for(rng, a, words, 9, 6)
    emis_d = ctx.array(emis)
    for final_mode in (1, 0):
        for beam in (1e10, 400.0, 60.0):
            path, final, fstate, status = hmm.viterbi(emis_d, G, 1, fo, beam, final_mode)
            n_ok = 0
            for u in range(len(fo) - 1):
                r = oracle.viterbi_graph(emis[fo[u]:fo[u + 1]], tied, a["tr_off"], a["tr_tgt"], a["tr_logp"], beam, 0, final_mode)
                assert status[u] == r["status"], (final_mode, beam, u)
                if r["status"] == 0:
                    n_ok += 1
                    assert final[u].tobytes() == r["final"].tobytes(), (final_mode, beam, u, final[u], r["final"])
                    assert fstate[u] == r["final_state"]
                    assert np.array_equal(path[fo[u]:fo[u + 1]], r["path"]), (final_mode, beam, u)
            if final_mode == 1 and beam >= 400.0:
                assert n_ok == len(fo) - 1
    # state-major emissions take the same path
    p2, f2, s2, st2 = hmm.viterbi(ctx.array(np.ascontiguousarray(emis.T)), 1, len(emis), fo, 1e10, 1)
    p1, f1, s1, st1 = hmm.viterbi(emis_d, G, 1, fo, 1e10, 1)
    assert np.array_equal(p1, p2) and f1.tobytes() == f2.tobytes() and np.array_equal(st1, st2)
    hmm.close()


def test_graph_viterbi_equals_alignment_kernel_on_training_hmms(ctx, oracle):
    """The general kernel on a training topology gives the alignment kernel's (and the training oracle's) score and path."""
    rng = np.random.default_rng(7)
    for _ in range(4):
        ph = synth.phone_string(rng, 4)
        ntr, tr, _ = host.build_training_hmm(ph)
        S = len(ph)
        off = np.zeros(S + 1, np.int32)
        off[1:] = np.cumsum(ntr)
        tgt = np.concatenate([np.arange(s, s + ntr[s]) for s in range(S)]).astype(np.int32)
        lp = np.concatenate([tr[s, :ntr[s]] for s in range(S)]).astype(np.float32)
        T = int(rng.integers(60, 200))
        emis = (rng.standard_normal((T, G)) * 4 - 20).astype(np.float32)
        fo = np.array([0, T], np.int64)
        hmm = host.GraphHmm(ctx, off, tgt, lp, ph)
        path, final, fstate, status = hmm.viterbi(ctx.array(emis), G, 1, fo, 1e10, 0)
        r = oracle.viterbi(emis, ph, ntr, tr)
        hb = host.HmmBatch(ctx, fo, [ntr], [tr], [ph.astype(np.int32)])
        p_d, f_d, s_d = hb.viterbi(ctx.array(emis), G)
        ctx.sync()
        assert status[0] == 0 == r["status"] and fstate[0] == S - 1
        assert final[0].tobytes() == r["final"].tobytes() == f_d.get()[0].tobytes()
        assert np.array_equal(path, r["path"]) and np.array_equal(path, p_d.get())
        hmm.close()


def test_update_states_on_lexicon_transitions(ctx, oracle):
    """hmm::update_states (src/hmm.cpp:361-366): pooled advance probabilities into the lexicon HMM's transitions, then
    decoding with them (srx_decode_plan_set_transitions)."""
    rng = np.random.default_rng(3)
    words, uni = _lexicon(rng, 30)
    a = host.build_lexicon_hmm(words, uni)
    tied = a["ctx"][:, 1].copy()
    C_ = 40
    probsum = rng.uniform(5, 50, C_).astype(np.float32)
    count = rng.integers(0, 100, C_).astype(np.int32)
    count[5] = 0
    new_tr = oracle.update_states(a["tr_off"], a["tr_logp"], tied, probsum, count)
    assert not np.array_equal(new_tr, a["tr_logp"]) and np.array_equal(new_tr[:30], a["tr_logp"][:30])   # start state untouched
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    hmm.set_transitions(new_tr)
    emis, fo = _emissions_for(rng, a, words, 8, 3)
    path, final, fstate, status = hmm.viterbi(ctx.array(emis), G, 1, fo, 300.0, 1)
    for u in range(len(fo) - 1):
        r = oracle.viterbi_graph(emis[fo[u]:fo[u + 1]], tied, a["tr_off"], a["tr_tgt"], new_tr, 300.0, 0, 1)
        assert status[u] == r["status"] == 0 and final[u].tobytes() == r["final"].tobytes()
        assert np.array_equal(path[fo[u]:fo[u + 1]], r["path"])
    hmm.close()


@pytest.mark.parametrize("sweep", [host.DECODE_DENSE, host.DECODE_FRONTIER])
def test_graph_viterbi_edge_cases(ctx, oracle, sweep):
    """Ragged and degenerate inputs: an empty batch, an utterance of zero frames (NO PATH), one frame, a beam that prunes
    everything but the best, and the compiled end condition on a lexicon HMM (the path must end in the LAST word's silence)."""
    rng = np.random.default_rng(9)
    words, uni = _lexicon(rng, 12)
    a = host.build_lexicon_hmm(words, uni)
    tied = a["ctx"][:, 1].copy()
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    hmm.set_sweep(sweep)
    emis, _ = _emissions_for(rng, a, words, 6, 2)
    emis_d = ctx.array(emis)
    p, f, s, st = hmm.viterbi(emis_d, G, 1, np.array([0], np.int64), 1e10, 1)          # no utterances
    assert len(st) == 1 and len(f) == 1                                               # (buffers are padded to one element; untouched)
    T = len(emis)
    fo = np.array([0, 0, 1, 1, T], np.int64)                                           # T = 0, 1, 0, T-1
    for final_mode in (0, 1):
        for beam in (1e10, 0.0):
            p, f, s, st = hmm.viterbi(emis_d, G, 1, fo, beam, final_mode)
            for u in range(4):
                e = emis[fo[u]:fo[u + 1]]
                if len(e) == 0:
                    assert st[u] == 1                                                  # nothing to decode: the reference's curr holds only the start state
                    continue
                r = oracle.viterbi_graph(e, tied, a["tr_off"], a["tr_tgt"], a["tr_logp"], beam, 0, final_mode)
                assert st[u] == r["status"], (final_mode, beam, u)
                if r["status"] == 0:
                    assert f[u].tobytes() == r["final"].tobytes() and s[u] == r["final_state"]
                    assert np.array_equal(p[fo[u]:fo[u + 1]], r["path"])
    hmm.close()


def _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, beam, final_mode, expect_ok=None):
    path, final, fstate, status = hmm.viterbi(emis_d, G, 1, fo, beam, final_mode)
    n_ok = 0
    for u in range(len(fo) - 1):
        r = oracle.viterbi_graph(emis[fo[u]:fo[u + 1]], tied, a["tr_off"], a["tr_tgt"], a["tr_logp"], beam, 0, final_mode)
        assert status[u] == r["status"], (final_mode, beam, u, status[u], r["status"])
        if r["status"] == 0:
            n_ok += 1
            assert final[u].tobytes() == r["final"].tobytes(), (final_mode, beam, u, final[u], r["final"])
            assert fstate[u] == r["final_state"]
            assert np.array_equal(path[fo[u]:fo[u + 1]], r["path"]), (final_mode, beam, u)
    if expect_ok is not None:
        assert n_ok == expect_ok, (final_mode, beam, n_ok)
    return path


def test_graph_with_mixed_fanout_both_sweeps(ctx, oracle):
    """A random graph whose sources have 1..80 transitions (both sides of the frontier sweep's in-thread / chain split at 33
    and of the dense sweep's at 4), repeated targets within a source included."""
    rng = np.random.default_rng(77)
    n = 400
    deg = rng.choice([1, 2, 3, 5, 20, 32, 33, 34, 80], n)
    off = np.zeros(n + 1, np.int32)
    off[1:] = np.cumsum(deg)
    tgt = rng.integers(0, n, off[-1]).astype(np.int32)
    lp = (-rng.uniform(0.1, 3.0, off[-1])).astype(np.float32)
    tied = rng.integers(0, G, n).astype(np.int32)
    a = {"tr_off": off, "tr_tgt": tgt, "tr_logp": lp}
    T = 40
    emis = (rng.standard_normal((3 * T, G)) * 2 - 3).astype(np.float32)
    fo = np.array([0, T, 2 * T, 3 * T], np.int64)
    emis_d = ctx.array(emis)
    for sweep in (host.DECODE_DENSE, host.DECODE_FRONTIER):
        hmm = host.GraphHmm(ctx, off, tgt, lp, tied)
        hmm.set_sweep(sweep)
        for final_mode in (1, 0):
            for beam in (1e10, 30.0, 5.0):
                _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, beam, final_mode)
        hmm.close()


@pytest.mark.parametrize("n_words", [20000, 133860])
def test_frontier_sweep_at_reference_lexicon_size_bit_exact(ctx, oracle, n_words):
    """20,000 words, and 133,860 -- the line count of the reference's res/lexicon/cmu.txt, which
    src/main.cpp:58 turns into one HMM of about 10^6 states.  Word lengths follow that lexicon (mean 6.4 phones, up to 20).
    The start state's fan-out is one serial f32 chain over every word, as the reference carries `logp` (src/hmm.cpp:164-169);
    with a beam of 200 only the head of that chain survives src/hmm.cpp:165, which is what bounds the frontier."""
    rng = np.random.default_rng(n_words)
    lens = np.clip(rng.poisson(5.4, n_words) + 1, 1, 20)
    words = [rng.integers(1, 40, int(k)).astype(np.int32) for k in lens]
    counts = rng.integers(1, 5000, n_words)
    uni = (np.log(counts) - np.float32(np.log(counts.sum() * 3.0))).astype(np.float32)
    a = host.build_lexicon_hmm(words, uni)
    assert a["n_states"] == 1 + int(lens.sum()) + n_words
    tied = a["ctx"][:, 1].copy()
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)      # AUTO: too many states for the dense sweep
    # two utterances that spell words from the head of the lexicon (the only ones the carried logp leaves reachable)
    emis, fo = [], [0]
    for seq in ([0, 2, 1], [3, 0]):
        rows = []
        for w in seq:
            for ph in list(words[w]) + [0]:
                e = rng.standard_normal((int(rng.integers(2, 6)), G)).astype(np.float32) * 2 - 12
                e[:, ph] += 9.0
                rows.append(e)
        emis.append(np.concatenate(rows))
        fo.append(fo[-1] + len(emis[-1]))
    emis, fo = np.concatenate(emis), np.array(fo, np.int64)
    emis_d = ctx.array(emis)
    _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 200.0, 1, expect_ok=2)     # the beam VERDICT r02 item 5 names
    _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 200.0, 0)
    if n_words == 20000:
        _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 1e10, 1, expect_ok=2)  # nothing pruned: every layer in the store
        _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 1e10, 0)
    # a workspace limit too small for the unpruned layers: status 3, not a fault
    hmm.set_sweep(host.DECODE_FRONTIER, 64 << 20 if n_words == 20000 else 400 << 20)
    _, _, _, st = hmm.viterbi(emis_d, G, 1, fo, 1e10, 1)
    assert list(st) == [3, 3]
    path, final, _, st = hmm.viterbi(emis_d, G, 1, fo, 200.0, 1)
    assert list(st) == [0, 0]
    hmm.close()


def test_dense_sweep_in_utterance_groups(ctx, oracle):
    """The dense sweep's back-pointers (4 * frames * states bytes) are bounded by the plan's workspace limit: the utterances go
    in groups that fit half of it.  Same results as in one launch."""
    rng = np.random.default_rng(5)
    words, uni = _lexicon(rng, 40)
    a = host.build_lexicon_hmm(words, uni)
    tied = a["ctx"][:, 1].copy()
    emis, fo = _emissions_for(rng, a, words, 7, 9)
    emis_d = ctx.array(emis)
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    per_utt = int(np.diff(fo).max()) * a["n_states"] * 4
    hmm.set_sweep(host.DECODE_DENSE, 2 * 2 * per_utt + 1024)         # two of the longest utterances per group at most
    _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 300.0, 1, expect_ok=9)
    _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 1e10, 0)
    hmm.set_sweep(host.DECODE_AUTO, 2 * per_utt // 2)                # not even one fits: AUTO goes to the frontier sweep
    _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 300.0, 1, expect_ok=9)
    hmm.close()


def test_frontier_sweep_more_utterances_than_fit_in_flight(ctx, oracle):
    """The frontier sweep keeps at most 32 utterances in flight (fewer under a small workspace limit) and takes a batch in
    groups: 70 utterances, results equal to the dense sweep's for all of them and to the oracle's for a sample."""
    rng = np.random.default_rng(17)
    words, uni = _lexicon(rng, 25)
    a = host.build_lexicon_hmm(words, uni)
    tied = a["ctx"][:, 1].copy()
    emis, fo = _emissions_for(rng, a, words, 6, 70)
    emis_d = ctx.array(emis)
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    hmm.set_sweep(host.DECODE_DENSE)
    ref = hmm.viterbi(emis_d, G, 1, fo, 250.0, 1)
    for limit in (0, 3 << 20):                                    # default limit (groups of 32), and one that carries only a few
        hmm.set_sweep(host.DECODE_FRONTIER, limit)
        got = hmm.viterbi(emis_d, G, 1, fo, 250.0, 1)
        assert np.array_equal(got[3], ref[3]) and (got[3] == 0).sum() >= 60
        ok = ref[3] == 0
        assert got[1][ok].tobytes() == ref[1][ok].tobytes() and np.array_equal(got[2][ok], ref[2][ok])
        for u in np.flatnonzero(ok):
            assert np.array_equal(got[0][fo[u]:fo[u + 1]], ref[0][fo[u]:fo[u + 1]]), u
    for u in (0, 33, 69):
        r = oracle.viterbi_graph(emis[fo[u]:fo[u + 1]], tied, a["tr_off"], a["tr_tgt"], a["tr_logp"], 250.0, 0, 1)
        assert ref[3][u] == r["status"]
        if r["status"] == 0:
            assert ref[1][u].tobytes() == r["final"].tobytes() and np.array_equal(ref[0][fo[u]:fo[u + 1]], r["path"])
    hmm.close()


@pytest.mark.parametrize("shift", [0.0, 25.0, 130.0])
def test_frontier_chain_early_stop_is_exact(ctx, oracle, shift):
    """The start state's serial chain stops once its running value is below next frame's threshold and provably cannot come back
    up (srx_decode_frontier.hip, k_fr_chain).  Three regimes on a 20,000-word lexicon (start transitions -94 .. -51): emission
    scores all negative (every step goes down: the chain stops after its first tile), some positive but smaller than the
    smallest language-model penalty (steps still go down, by less), and positive beyond it (the running value climbs along the
    lexicon: the test never passes and the whole chain runs).  Bit-exact against the oracle in all three, both end conditions."""
    n_words = 20000
    rng = np.random.default_rng(31)
    lens = np.clip(rng.poisson(5.4, n_words) + 1, 1, 20)
    words = [rng.integers(1, 40, int(k)).astype(np.int32) for k in lens]
    counts = rng.integers(1, 5000, n_words)
    uni = (np.log(counts) - np.float32(np.log(counts.sum() * 3.0))).astype(np.float32)
    a = host.build_lexicon_hmm(words, uni)
    assert a["tr_logp"][:n_words].max() < -45 and a["tr_logp"][:n_words].min() > -100
    tied = a["ctx"][:, 1].copy()
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    emis, fo = [], [0]
    for seq in ([0, 2, 1], [3, 0], [5]):
        rows = []
        for w in seq:
            for ph in list(words[w]) + [0]:
                e = rng.standard_normal((int(rng.integers(2, 6)), G)).astype(np.float32) * 2 - 12 + np.float32(shift)
                e[:, ph] += 9.0
                rows.append(e)
        emis.append(np.concatenate(rows))
        fo.append(fo[-1] + len(emis[-1]))
    emis, fo = np.concatenate(emis), np.array(fo, np.int64)
    emis_d = ctx.array(emis)
    for beam in (200.0, 40.0):
        _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, beam, 1, expect_ok=3 if beam == 200.0 else None)
        _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, beam, 0)
    hmm.close()


def test_frontier_early_stop_follows_set_transitions(ctx, oracle):
    """srx_decode_plan_set_transitions refreshes what the chain's early stop relies on (the suffix maxima of the start state's
    transition scores).  Here the new start transitions are half as negative as the old ones and the emission scores sit between
    the two: with stale maxima the chain would stop where it must not.  Bit-exact against the oracle before and after."""
    n_words = 3000
    rng = np.random.default_rng(8)
    lens = np.clip(rng.poisson(4.0, n_words) + 1, 1, 12)
    words = [rng.integers(1, 40, int(k)).astype(np.int32) for k in lens]
    counts = rng.integers(1, 5000, n_words)
    uni = (np.log(counts) - np.float32(np.log(counts.sum() * 3.0))).astype(np.float32)
    a = host.build_lexicon_hmm(words, uni)
    tied = a["ctx"][:, 1].copy()
    lo, hi = float(a["tr_logp"][:n_words].min()), float(a["tr_logp"][:n_words].max())
    assert hi < -40
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    hmm.set_sweep(host.DECODE_FRONTIER)
    rows = []
    for w in (1, 0, 4):
        for ph in list(words[w]) + [0]:
            e = rng.standard_normal((int(rng.integers(2, 6)), G)).astype(np.float32) * 2 + np.float32(-hi * 0.75)   # positive, below |hi|
            e[:, ph] += 4.0
            rows.append(e)
    emis = np.concatenate(rows)
    fo = np.array([0, len(emis)], np.int64)
    emis_d = ctx.array(emis)
    assert emis.max() < -hi and emis.max() > -hi * 0.5 + 5
    _check_against_oracle(oracle, hmm, a, tied, emis, emis_d, fo, 150.0, 1)
    # new scores: the first tile of words still drags the running value down (below the threshold: the stale tables, whose
    # suffix maximum is `hi`, would let the chain stop there), the rest are half as negative as before -- under these emissions
    # the value climbs back and the best words are late in the lexicon
    new_tr = a["tr_logp"].copy()
    new_tr[:1024] = np.float32(hi * 0.9)
    new_tr[1024:n_words] *= np.float32(0.5)
    hmm.set_transitions(new_tr)
    b = dict(a, tr_logp=new_tr)
    path = _check_against_oracle(oracle, hmm, b, tied, emis, emis_d, fo, 150.0, 1)
    assert a["word_of"][path].max() >= 1024               # the path does use words behind the first tile
    _check_against_oracle(oracle, hmm, b, tied, emis, emis_d, fo, 150.0, 0)
    hmm.close()
