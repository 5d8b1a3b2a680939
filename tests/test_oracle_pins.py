"""Pins for the CPU oracle (oracle/srx_oracle.c).  Runs without a GPU.

The reference ships no tests, golden vectors or fixtures, and cannot be compiled in this image (every
translation unit on the path includes OpenFrameworks' ofMain.h, which is absent; DESIGN.md "Oracle").
PARITY UNPINNED: the only reference-derived anchors are the scalar facts SURVEY.md section 4 recorded
from the reference (filterbank constants, frame count, training-HMM topology, the log(2/pi) constant,
the accumulate form of Viterbi).  They are checked here, next to closed-form / independent-numpy
restatements of each stage and the committed regression vectors under tests/golden/ (which were made
by the oracle itself, see tools/make_golden.py: they detect drift, they do not pin parity).
"""
import math
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- SURVEY section 4 facts
def test_pin_mel_filterbank_constants(oracle):
    """mel_filterbank(7200,48,320,16000): 48 filters, last centre 136.375397 bins, width 7.256612,
    bin_max 144 (src/mfcc.cpp:190-201,220-221)."""
    fb, bin_max = oracle.mel_filterbank(7200.0, 48, 320, 16000)
    assert len(fb) == 48 and bin_max == 144
    assert abs(float(fb[47]["center"]) - 136.375397) < 5e-6
    assert abs(float(fb[47]["width"]) - 7.256612) < 5e-7
    # structure: centres increase, width = distance to the lower neighbour, heights normalise the triangle
    c, w, h = fb["center"], fb["width"], fb["height"]
    assert (np.diff(c) > 0).all()
    np.testing.assert_allclose(w[1:], np.diff(c), rtol=2e-5)
    for i in range(48):
        s = sum(max(0.0, 1.0 - abs(float(c[i]) - b) / float(w[i])) for b in range(int(c[i] - w[i]), int(c[i] + w[i]) + 1))
        assert abs(float(h[i]) * s - 1.0) < 1e-5


def test_pin_frame_count(oracle):
    """5 s at 16 kHz, 320/160 -> 498 frames (SURVEY.md sections 3.1, 4); n = 1 + floor((L-wl-1)/hop)."""
    assert oracle.num_frames(80000, 320, 160) == 498
    for L in (321, 322, 480, 481, 482, 640, 641, 16000, 79999, 80001):
        assert oracle.num_frames(L, 320, 160) == 1 + (L - 320 - 1) // 160
    assert oracle.num_frames(80000, 400, 160) == 1 + (80000 - 401) // 160


def test_pin_training_hmm_topology(oracle):
    """SIL HH AH L OW SIL W ER L D SIL -> 11 states; 2 transitions at ln 1/2, the state before an interior
    SIL has 3 at ln 1/3 (self, SIL, skip), the last state a self loop at 0 (SURVEY.md section 4;
    src/hmm.cpp:56-88, src/hmm.hpp:25-26)."""
    SIL, HH, AH, L, OW, W, ER, D = 0, 16, 3, 21, 25, 36, 12, 9
    ph = [SIL, HH, AH, L, OW, SIL, W, ER, L, D, SIL]
    ntr, tr, ctx = oracle.build_training_hmm(ph)
    assert len(ntr) == 11
    assert list(ntr) == [2, 2, 2, 2, 3, 2, 2, 2, 2, 2, 1]
    lh, lt = np.float32(-0.6931471805), np.float32(-1.09861228866)
    for s in range(10):
        want = lt if ntr[s] == 3 else lh
        assert (tr[s, :ntr[s]] == want).all()
    assert tr[10, 0] == 0.0
    assert list(ctx[0]) == [SIL, SIL, HH] and list(ctx[4]) == [L, OW, SIL] and list(ctx[10]) == [D, SIL, SIL]
    # the state before the FINAL SIL has no skip (i + 2 < S fails)
    assert ntr[9] == 2


def test_pin_log_2_over_pi_constant(oracle):
    """gaussian::operator() uses log(M_2_PI) = log(2/pi) (src/gmm.cpp:43): at x = mu,
    var = 1, weight = 1 the log-density is -D/2 * log(2/pi) > 0."""
    for D in (13, 39):
        x = np.zeros((1, 39), np.float32)
        got = oracle.gmm_score(x, np.zeros((1, 1, D)), np.ones((1, 1, D)), np.zeros((1, 1), np.float32), D, 0)[0, 0]
        assert got > 0
        assert abs(got - (-0.5 * D * math.log(2 / math.pi))) < 1e-5


def test_pin_viterbi_accumulates_across_transitions(oracle):
    """SURVEY.md fact 10 (src/hmm.cpp:164-169): the k-th transition's candidate carries the log-probs and
    emissions of transitions 0..k-1 of the same source.  Hand-computed 3-state case."""
    ph = [0, 5, 0]
    ntr, tr, _ = oracle.build_training_hmm(ph)
    assert list(ntr) == [2, 2, 1]
    e = np.array([[-1.0, -2.0, -30.0], [-50.0, -0.5, -4.0], [-50.0, -60.0, -0.25]], np.float32)
    r = oracle.viterbi(e, None, ntr, tr)
    lh = np.float32(-0.6931471805)
    f32 = np.float32
    # frame 0 from state 0: to 0: (0+lh)+e00 ; to 1: that + lh + e01
    d0 = f32(f32(0 + lh) + e[0, 0])
    d1 = f32(f32(d0 + lh) + e[0, 1])
    # frame 1: state1 <- max(self from 1, advance from 0 (accumulated through 0's self candidate))
    c00 = f32(f32(d0 + lh) + e[1, 0])
    c01 = f32(f32(c00 + lh) + e[1, 1])
    c11 = f32(f32(d1 + lh) + e[1, 1])
    c12 = f32(f32(c11 + lh) + e[1, 2])
    n1 = max(c01, c11)
    n2 = c12
    # frame 2
    c11b = f32(f32(n1 + lh) + e[2, 1])
    c12b = f32(f32(c11b + lh) + e[2, 2])
    c22b = f32(f32(n2 + 0.0) + e[2, 2])
    final = max(c12b, c22b)
    assert r["status"] == 0
    assert r["final"].tobytes() == f32(final).tobytes()
    # textbook Viterbi would give a different number
    tb = max(f32(f32(n1 + lh) + e[2, 2]), c22b)
    assert f32(tb) != r["final"]


# --------------------------------------------------------------------------- independent restatements
def _np_partial_dft(frame, N, nb):
    """Mathematical form of Clip::partial_dft in f64: first N/2 samples, 0.54+0.46cos window, 1/sqrt(N)."""
    n = np.arange(N // 2)
    w = 0.54 + 0.46 * np.cos(2 * np.pi * n / (N - 1))
    y = frame[:N // 2].astype(np.float64) * w
    k = np.arange(nb)[:, None]
    z = (y[None, :] * np.exp(-2j * np.pi * k * n[None, :] / N)).sum(axis=1)
    return np.abs(z) / math.sqrt(N)


@pytest.mark.parametrize("N,nb", [(320, 144), (400, 180), (64, 20)])
def test_partial_dft_against_numpy(oracle, N, nb):
    rng = np.random.default_rng(N)
    frame = rng.standard_normal(N).astype(np.float32)
    got = oracle.partial_dft(frame, N, nb)
    ref = _np_partial_dft(frame, N, nb)
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-5 * np.abs(ref).max())
    # equivalent statement: length-N FFT of the windowed half-frame zero-padded
    n = np.arange(N // 2)
    y = np.zeros(N)
    y[:N // 2] = frame[:N // 2] * (0.54 + 0.46 * np.cos(2 * np.pi * n / (N - 1)))
    np.testing.assert_allclose(ref, np.abs(np.fft.fft(y))[:nb] / math.sqrt(N), atol=1e-9)


def test_hamming_is_the_reference_s_plus_form(oracle):
    """hamming_window = 0.54 PLUS 0.46 cos (src/utils.cpp:21-23): 1.0 at x = 0, ~0.08 mid-window."""
    L = oracle.lib()
    assert L.orc_hamming(0, 320) == 1.0
    assert abs(L.orc_hamming(160, 320) - 0.08) < 1e-3
    assert abs(L.orc_hamming(319, 320) - 1.0) < 1e-6


def test_preemphasis_and_normalize(oracle):
    x = np.array([1.0, 2.0, 4.0, 8.0], np.float32)
    y = oracle.preemphasis(x, 0.5)
    assert list(y) == [1.5, 1.5, 3.0, 6.0]  # y[i] = x[i] - a x[i-1] on ORIGINAL left neighbours; y[0] = the NEW y[1]
    x = np.array([3.0, 4.0], np.float32)
    y = oracle.normalize_volume(x, 1.0)  # rms is an L2 norm: no 1/N (src/clip.cpp:157-164)
    np.testing.assert_allclose(y, [0.6, 0.8], rtol=1e-6)


def test_preemphasis_first_sample_is_new_second(oracle):
    x = np.array([5.0, 2.0, 7.0], np.float32)
    y = oracle.preemphasis(x, 0.96)
    assert y[0] == y[1] == np.float32(2.0) - np.float32(0.96) * np.float32(5.0)


def test_dct_and_log_forms(oracle):
    """dct uses cos(j*i*pi/len) with index i (not i + 1/2) and log is log(x+1)."""
    import ctypes as C
    L = oracle.lib()
    L.orc_dct.argtypes = [np.ctypeslib.ndpointer(np.float32), C.c_int, C.c_int, np.ctypeslib.ndpointer(np.float32)]
    f = np.random.default_rng(0).random(48).astype(np.float32)
    out = np.zeros(13, np.float32)
    L.orc_dct(f, 48, 13, out)
    i = np.arange(48)
    ref = np.array([(f * np.cos(j * i * np.pi / 48)).sum() for j in range(13)])
    np.testing.assert_allclose(out, ref, atol=2e-5)
    assert abs(out[0] - f.sum()) < 1e-4


def test_read_features_smoothing_and_deltas(oracle):
    """Triangular 1,2,3,2,1 smoothing renormalised by the in-range integer weight sum; clamped central
    differences; n = 1 gives 0/0 = NaN deltas."""
    raw = np.arange(1, 7, dtype=np.float32)[:, None] * np.ones((1, 13), np.float32)
    f = oracle.read_features(raw, 2)
    assert f.shape == (6, 39)
    # interior of a linear ramp is unchanged by a symmetric smoother
    np.testing.assert_allclose(f[2:4, :13], raw[2:4], rtol=1e-6)
    # edges: weights (3,2,1)/6 on rows 0,1,2
    assert abs(f[0, 0] - (3 * 1 + 2 * 2 + 1 * 3) / 6) < 1e-6
    assert abs(f[1, 0] - (2 * 1 + 3 * 2 + 2 * 3 + 1 * 4) / 8) < 1e-6
    # deltas: clamped at the ends (divide by 1), central elsewhere (divide by 2)
    assert abs(f[0, 13] - (f[1, 0] - f[0, 0])) < 1e-6
    assert abs(f[3, 13] - (f[4, 0] - f[2, 0]) / 2) < 1e-6
    assert abs(f[3, 26] - (f[4, 13] - f[2, 13]) / 2) < 1e-6
    one = oracle.read_features(raw[:1], 2)
    assert np.isnan(one[0, 13:]).all() and one[0, 0] == 1.0


def test_gaussian_logprob_against_f64(oracle):
    rng = np.random.default_rng(3)
    G, M, D = 7, 3, 13
    mean = rng.standard_normal((G, M, D))
    var = rng.uniform(0.5, 1.5, (G, M, D))
    w = rng.uniform(0.5, 1.5, (G, M)).astype(np.float32)
    w /= w.sum(axis=1, keepdims=True)
    lw = oracle.set_wt(w)
    x = rng.standard_normal((50, 39)).astype(np.float32)
    xx = x[:, None, None, :D].astype(np.float64)
    lp = -0.5 * (math.log(2 / math.pi) + np.log(var) + (xx - mean) ** 2 / var).sum(axis=-1) + lw
    np.testing.assert_allclose(oracle.gmm_score(x, mean, var, lw, D, 0), lp[:, :, 0], rtol=2e-6, atol=2e-5)
    lse = np.log(np.exp(lp).sum(axis=-1))
    np.testing.assert_allclose(oracle.gmm_score(x, mean, var, lw, D, 1), lse, rtol=2e-6, atol=2e-5)


def test_viterbi_path_properties_and_no_path(oracle):
    rng = np.random.default_rng(9)
    ph = [0, 3, 4, 0, 5, 6, 7, 0]
    ntr, tr, _ = oracle.build_training_hmm(ph)
    S = len(ph)
    e = (rng.standard_normal((60, S)) * 3).astype(np.float32)
    r = oracle.viterbi(e, None, ntr, tr)
    p = r["path"]
    assert r["status"] == 0 and p[-1] == S - 1 and p[0] in (0, 1)
    d = np.diff(p)
    assert ((d >= 0) & (d <= 2)).all()
    for t in np.nonzero(d == 2)[0]:
        assert ntr[p[t]] == 3        # a skip only leaves a state that has the SIL-skip transition
    assert r["prevp"] == np.float32(r["final"] / np.float32(60))
    # both tie orders give the same score
    assert oracle.viterbi(e, None, ntr, tr, order=1)["final"].tobytes() == r["final"].tobytes()
    # too few frames to reach the last state -> NO PATH (src/hmm.cpp:212-215)
    assert oracle.viterbi(e[:3], None, ntr, tr)["status"] == 1


def test_train_transitions_closed_form(oracle):
    ph = [0, 3, 0, 4, 0]
    ntr, tr, _ = oracle.build_training_hmm(ph)
    path = np.array([0, 0, 0, 0, 1, 2, 2, 3, 3, 3, 4, 4], np.int32)
    new = oracle.train_transitions(path, ntr, tr)
    # state 0: run of 4 -> p = 0.2*min(0.55, 0.25) + 0.8*exp(tr[1]); it has 3 transitions (next is SIL? no: state 1 is phone 3)
    for s, cnt in ((0, 4), (1, 1), (2, 2), (3, 3)):
        adv = math.exp(float(tr[s, 1]))
        p = 0.2 * min(0.55, 1.0 / cnt) + 0.8 * adv
        avail = 0.7 if ntr[s] == 3 else 1.0
        assert abs(new[s, 0] - math.log((1 - p) * avail)) < 1e-5
        assert abs(new[s, 1] - math.log(p * avail)) < 1e-5
        if ntr[s] == 3:
            assert new[s, 2] == tr[s, 2]   # tr[2] is never updated (src/hmm.cpp:26-31)
    assert (new[4] == tr[4]).all()          # the last run is never re-estimated


def test_em_statistics_single_component_closed_form(oracle):
    """M = 1: zeta cancels in the mean/variance ratios only if constant per state -- it is not (zeta =
    w * p(x)), so compare with the f64 weighted-moment formula."""
    rng = np.random.default_rng(5)
    G, M, D, T = 4, 1, 13, 400
    mean = rng.standard_normal((G, M, D))
    var = rng.uniform(0.5, 1.5, (G, M, D))
    w = np.ones((G, M), np.float32)
    gid = rng.integers(0, G, T).astype(np.int32)
    x = np.zeros((T, 39), np.float32)
    x[:, :D] = (mean[gid, 0] + rng.standard_normal((T, D))).astype(np.float32)
    Bm, Bv, Bw, _, zs = oracle.em_statistics(x, gid, D, mean, var, w, 0.2)
    lp = oracle.gmm_score(x, mean, var, np.zeros((G, M), np.float32), D, 0).astype(np.float64)
    for g in range(G):
        sel = gid == g
        z = np.exp(lp[sel, g])
        xs = x[sel, :D].astype(np.float64)
        mu = (z[:, None] * xs).sum(0) / z.sum()
        v = (z[:, None] * (xs - mu) ** 2).sum(0) / z.sum()
        np.testing.assert_allclose(np.asarray(Bm[g, 0], np.float64), 0.2 * mu + 0.8 * mean[g, 0], rtol=1e-6)
        np.testing.assert_allclose(np.asarray(Bv[g, 0], np.float64), 0.2 * v + 0.8 * var[g, 0], rtol=1e-5)
        assert abs(float(zs[g, 0]) - z.sum()) <= 1e-6 * z.sum()
    np.testing.assert_allclose(Bw, 1.0, rtol=1e-6)


# --------------------------------------------------------------------------- committed regression vectors
GOLDEN = os.path.join(HERE, "golden", "oracle_vectors.npz")


@pytest.mark.skipif(not os.path.exists(GOLDEN), reason="tests/golden/oracle_vectors.npz not generated")
def test_oracle_matches_committed_vectors(oracle):
    """tests/golden/oracle_vectors.npz was written by tools/make_golden.py FROM THE ORACLE (the reference
    cannot run here): a drift detector across compilers / libm versions / boxes, not a parity pin."""
    from tools import make_golden
    fresh = make_golden.generate(oracle)
    with np.load(GOLDEN) as z:
        assert set(z.files) == set(fresh)
        for k in z.files:
            a, b = z[k], fresh[k]
            assert a.shape == b.shape and a.dtype == b.dtype, k
            assert a.tobytes() == b.tobytes(), k


def test_graph_viterbi_restatement_equals_training_viterbi_restatement(oracle):
    """orc_viterbi_graph (arbitrary transition lists: the checker of the lexicon-HMM decode) reproduces orc_viterbi -- the
    restatement that SURVEY.md section 4's accumulate-semantics fact pins -- bit for bit on training topologies, with and
    without a beam and in both visiting orders."""
    import numpy as np
    from speechrecognition_amd import synth
    rng = np.random.default_rng(5)
    for trial in range(12):
        ph = synth.phone_string(rng, int(rng.integers(1, 6)))
        ntr, tr, _ = oracle.build_training_hmm(ph)
        S = len(ph)
        off = np.zeros(S + 1, np.int32)
        off[1:] = np.cumsum(ntr)
        tgt = np.concatenate([np.arange(s, s + ntr[s]) for s in range(S)]).astype(np.int32)
        lp = np.concatenate([tr[s, :ntr[s]] for s in range(S)]).astype(np.float32)
        T = int(rng.integers(3, 150))
        emis = (rng.standard_normal((T, 40)) * 5 - 15).astype(np.float32)
        for beam in (1e10, 40.0):
            for order in (0, 1):
                a = oracle.viterbi(emis, ph, ntr, tr, beam, order)
                b = oracle.viterbi_graph(emis, ph, off, tgt, lp, beam, order, 0)
                assert a["status"] == b["status"]
                if a["status"] == 0:
                    assert a["final"].tobytes() == b["final"].tobytes() and np.array_equal(a["path"], b["path"])
