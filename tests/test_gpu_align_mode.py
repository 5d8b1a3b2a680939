"""Alignment-mode emission scoring (srx_gmm_score_align_dev): each utterance's frames against its own states only --
the scores the reference computes lazily inside hmm::viterbi (src/hmm.cpp:166).

Bar: bit-identical to the dense matrix of srx_gmm_score_dev on the same score path (same arithmetic, different data
movement), hence within 1e-4 of the oracle; Viterbi on the compact emissions gives the same bits as on the dense matrix."""
import numpy as np
import pytest

from speechrecognition_amd import SRX_MIX_COMPONENT0, SRX_MIX_LOGSUMEXP, SRX_SCORE_EXACT, host, synth
from speechrecognition_amd._capi import SRX_SCORE_MFMA_F16X3

pytestmark = pytest.mark.gpu


def _batch(rng, n_utt, G, t_lo, t_hi, words=None):
    ntrs, trs, tieds, Ts = [], [], [], []
    for _ in range(n_utt):
        ph = synth.phone_string(rng, words)
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr)
        trs.append(tr)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
        Ts.append(int(rng.integers(t_lo, t_hi)))
    fo = np.zeros(n_utt + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    return ntrs, trs, tieds, fo


def _compact_from_dense(dense, hb, tieds):
    """dense [T, G] -> the compact layout (utterance u: [T_u][S_u] at emis_off[u])."""
    out = np.empty(hb.emis_count, np.float32)
    for u in range(hb.n_utt):
        a, b = int(hb.frame_off[u]), int(hb.frame_off[u + 1])
        blk = dense[a:b][:, tieds[u]]              # [T_u, S_u]
        out[hb.emis_off[u]:hb.emis_off[u + 1]] = blk.ravel()
    return out


@pytest.mark.parametrize("G,M,D,mode", [(2000, 16, 39, SRX_MIX_LOGSUMEXP), (40, 1, 13, SRX_MIX_COMPONENT0),
                                        (70, 4, 39, SRX_MIX_LOGSUMEXP), (70, 3, 13, SRX_MIX_LOGSUMEXP),
                                        (300, 16, 39, SRX_MIX_COMPONENT0)])
def test_align_scores_equal_dense_scores_bitwise(ctx, oracle, G, M, D, mode):
    rng = np.random.default_rng(G + M)
    ntrs, trs, tieds, fo = _batch(rng, 23, G, 1, 300)      # ragged: 1-frame utterances, partial wave blocks, odd tile counts
    T = int(fo[-1])
    feats = (rng.standard_normal((T, 39)) * 1.3).astype(np.float32)
    feats[5] = 1e4                                          # an outlier frame: re-evaluated exactly on both paths
    mean, var, w = synth.random_model(G, M, D, 3)
    model = host.AcousticModel(ctx, G, M, D)
    model.set(mean, var, w)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    feats_d = ctx.array(feats)
    got = hb.score_align(model, feats_d, 39, mix_mode=mode).get()
    assert got.shape == (sum(len(t) * (fo[u + 1] - fo[u]) for u, t in enumerate(tieds)),)
    dense = model.score(feats, mode, SRX_SCORE_MFMA_F16X3)
    ref = _compact_from_dense(dense, hb, tieds)
    assert got.tobytes() == ref.tobytes()
    # and the oracle, on two utterances
    lw = oracle.set_wt(w)
    for u in (0, 7):
        a, b = int(fo[u]), int(fo[u + 1])
        o = oracle.gmm_score(feats[a:b], mean, var, lw, D, mode)[:, tieds[u]].ravel()
        g = got[hb.emis_off[u]:hb.emis_off[u + 1]]
        fin = np.isfinite(o)
        if not fin.all():
            # the outlier frame underflows to -inf in the reference's (dead) log-sum-exp: the library takes the sum
            # relative to the largest component there -- finite, checked against that form
            assert mode == SRX_MIX_LOGSUMEXP and u == 0
            o = oracle.gmm_score_shifted(feats[a:b], mean, var, lw, D)[:, tieds[u]].ravel()
            assert np.isfinite(o).all() and np.isfinite(g).all()
        assert (np.abs(g - o) <= 1e-4 * np.maximum(np.abs(o), 1.0)).all()
    # the frame-operand cache (srx_align_plan_cache_frames): same bits, also on the second call (cache hit) and after the
    # model changed (the cache is kept while the model's centre / scale vector is, rebuilt when it is not)
    hb.cache_frames(True)
    for rep in range(2):
        assert hb.score_align(model, feats_d, 39, mix_mode=mode).get().tobytes() == got.tobytes(), rep
    model.set(mean + 0.01, var * 1.02, w)                   # a small move: centre / scale are kept, operands reused
    moved = hb.score_align(model, feats_d, 39, mix_mode=mode).get()
    assert moved.tobytes() == _compact_from_dense(model.score(feats, mode, SRX_SCORE_MFMA_F16X3), hb, tieds).tobytes()
    model.set(mean * 40.0 + 7.0, var * 900.0, w)           # a large move: new centre / scale, operands rebuilt
    moved = hb.score_align(model, feats_d, 39, mix_mode=mode).get()
    assert moved.tobytes() == _compact_from_dense(model.score(feats, mode, SRX_SCORE_MFMA_F16X3), hb, tieds).tobytes()
    hb.cache_frames(False)
    model.set(mean, var, w)
    assert hb.score_align(model, feats_d, 39, mix_mode=mode).get().tobytes() == got.tobytes()
    # exact path in the compact layout
    ex = hb.score_align(model, feats_d, 39, mix_mode=mode, path=SRX_SCORE_EXACT).get()
    exd = _compact_from_dense(model.score(feats, mode, SRX_SCORE_EXACT), hb, tieds)
    assert ex.tobytes() == exd.tobytes()
    hb.close()
    model.close()


def test_align_table_follows_the_centre_scale_vector_in_force(ctx):
    """the per-state coefficient table of the alignment kernel is scaled with the model's centre / scale vector,
    which ensure_packed16 KEEPS from one model version to the next while it still suits the model and may REPLACE without a
    change of the model's version (a dense score in the other mix mode whose coefficients no longer fit f16 with the kept
    one).  The table's validity is keyed on that vector's epoch as well as on (version, mode).  Here: one model, contracted
    to 0.3 of its size (kept vector in force: the scales are within the 4x window), then sharpened 50x in one component
    (coefficients 50x larger: what still fits is mode dependent), with mix modes and dense / alignment scoring alternating
    -- alignment scores equal the dense ones bit for bit at every step, whichever vector each step ended up with."""
    rng = np.random.default_rng(99)
    G, M, D = 64, 4, 39
    ntrs, trs, tieds, fo = _batch(rng, 9, G, 20, 200)
    T = int(fo[-1])
    feats = (rng.standard_normal((T, 39)) * 0.5).astype(np.float32)
    feats_d = ctx.array(feats)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    mean, var, w = synth.random_model(G, M, D, 3)
    model = host.AcousticModel(ctx, G, M, D)
    other = {SRX_MIX_COMPONENT0: SRX_MIX_LOGSUMEXP, SRX_MIX_LOGSUMEXP: SRX_MIX_COMPONENT0}

    def check(mode):
        got = hb.score_align(model, feats_d, 39, mix_mode=mode).get()
        ref = _compact_from_dense(model.score(feats, mode, SRX_SCORE_MFMA_F16X3), hb, tieds)
        assert got.tobytes() == ref.tobytes(), mode

    def alternate():
        for mode in (SRX_MIX_COMPONENT0, SRX_MIX_LOGSUMEXP, SRX_MIX_COMPONENT0, SRX_MIX_LOGSUMEXP):
            check(mode)                                                   # builds / reuses the alignment table of `mode`
            model.score(feats, other[mode], SRX_SCORE_MFMA_F16X3)         # a dense score in the other mode: may replace the vector
            check(mode)                                                   # same model version, same mode: the table must follow

    model.set(mean, var, w)
    alternate()
    mean2, var2 = mean * 0.3, var * 0.09
    model.set(mean2, var2, w)
    alternate()
    var3 = var2.copy()
    var3[:, 1, :] /= 50.0
    model.set(mean2, var3, w)
    alternate()
    model.close()
    hb.close()


def test_model_f16_cannot_hold_is_scored_exactly(ctx):
    rng = np.random.default_rng(1)
    G, M, D = 30, 2, 13
    ntrs, trs, tieds, fo = _batch(rng, 5, G, 20, 90)
    feats = (rng.standard_normal((int(fo[-1]), 39))).astype(np.float32)
    mean, var, w = synth.random_model(G, M, D, 4)
    var[3, 1, 2] = 1e-7                                     # a coefficient beyond f16
    model = host.AcousticModel(ctx, G, M, D)
    model.set(mean, var, w)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    got = hb.score_align(model, ctx.array(feats), 39, mix_mode=SRX_MIX_LOGSUMEXP).get()
    ref = _compact_from_dense(model.score(feats, SRX_MIX_LOGSUMEXP, SRX_SCORE_EXACT), hb, tieds)
    assert got.tobytes() == ref.tobytes()
    hb.close()
    model.close()


def test_viterbi_on_compact_emissions_matches_dense(ctx, oracle):
    rng = np.random.default_rng(2)
    G, M, D = 2000, 16, 39
    ntrs, trs, tieds, fo = _batch(rng, 60, G, 150, 700)
    T = int(fo[-1])
    feats = (rng.standard_normal((T, 39)) * 1.3).astype(np.float32)
    mean, var, w = synth.random_model(G, M, D, 777)
    model = host.AcousticModel(ctx, G, M, D)
    model.set(mean, var, w)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    feats_d = ctx.array(feats)
    emis_c = hb.score_align(model, feats_d, 39, mix_mode=SRX_MIX_LOGSUMEXP)
    p1, f1, s1 = hb.viterbi_align(emis_c)
    dense_d = ctx.empty((G, T), np.float32)
    model.score_dev(feats_d, 39, T, dense_d, 1, T, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA_F16X3)
    p2, f2, s2 = hb.viterbi(dense_d, 1, T)
    ctx.sync()
    assert (s1.get() == 0).all() and np.array_equal(s1.get(), s2.get())
    assert f1.get().tobytes() == f2.get().tobytes()
    assert np.array_equal(p1.get(), p2.get())
    # oracle on the compact emissions of a few utterances
    ec = emis_c.get()
    for u in (0, 11, 59):
        S, Tu = len(tieds[u]), int(fo[u + 1] - fo[u])
        e = ec[hb.emis_off[u]:hb.emis_off[u + 1]].reshape(Tu, S).copy()      # [T_u, S_u], column = state
        r = oracle.viterbi(e, np.arange(S, dtype=np.int32), ntrs[u], trs[u], 1e10, 0)
        assert r["status"] == 0 and f1.get()[u].tobytes() == r["final"].tobytes()
        assert np.array_equal(p1.get()[fo[u]:fo[u + 1]], r["path"])
    hb.close()
    model.close()


def test_plan_rejects_a_block_beyond_32_bit_offsets(ctx):
    """The alignment kernel indexes inside an utterance's [T_u][S_u] block with 32 bits; a block of 2^31 entries or more
    is refused at plan creation with a message, not truncated (offsets only: nothing of that size is allocated)."""
    from speechrecognition_amd._capi import SrxError
    ph = np.zeros(40, np.int32)
    ntr, tr, _ = host.build_training_hmm(ph)
    fo = np.array([0, 1 << 26], np.int64)                 # 2^26 frames x 40 states = 2.7e9 entries
    hb = host.HmmBatch(ctx, fo, [ntr], [tr], [np.zeros(len(ph), np.int32)])
    with pytest.raises(SrxError) as ei:
        hb.align_plan()
    assert "frames" in str(ei.value) and "states" in str(ei.value)
    fo_ok = np.array([0, 1 << 20], np.int64)
    hb2 = host.HmmBatch(ctx, fo_ok, [ntr], [tr], [np.zeros(len(ph), np.int32)])
    hb2.align_plan()
    assert hb2.emis_count == (1 << 20) * len(ph)
    hb2.close()
