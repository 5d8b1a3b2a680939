"""Degenerate inputs through the C ABI: empty batches, zero-frame utterances inside a batch, one-frame utterances,
one-state HMMs.  The reference's loops simply do nothing on an empty container (`for i < ut.size()`, src/hmm.cpp:258;
`for i < fvs.size()`, src/hmm.cpp:158) -- here a zero-size grid is a launch error unless the entry point returns first,
so every entry point is called with nothing to do, and then with the smallest thing it can do, against the oracle."""
import ctypes as C

import numpy as np
import pytest

from speechrecognition_amd import _capi, host

pytestmark = pytest.mark.gpu


def _model(ctx, rng, G, M, D):
    m = host.AcousticModel(ctx, G, M, D)
    mean = rng.normal(0, 1, (G, M, D))
    var = rng.uniform(0.5, 2.0, (G, M, D))
    w = rng.uniform(0.1, 1, (G, M)).astype(np.float32)
    w /= w.sum(1, keepdims=True)
    m.set(mean, var, w)
    return m, mean, var, w


def test_mfcc_and_features_with_no_utterances(ctx):
    cfg = host.mfcc_config()
    fb, _ = host.mel_filterbank()
    plan = host.MfccPlan(ctx, cfg, fb)
    for path in (host.MFCC_F32, host.MFCC_F16X3):
        plan.set_path(path)
        ceps, fo = plan.write_features(np.zeros(0, np.float32), np.zeros(1, np.int64))
        assert ceps.shape == (0, 13) and fo.tolist() == [0]
    spec, fo = plan.partial_dft(np.zeros(0, np.float32), np.zeros(1, np.int64))
    assert spec.shape[0] == 0
    feats = host.read_features(ctx, np.zeros((0, 13), np.float32), np.zeros(1, np.int64))
    assert feats.shape == (0, 39)
    plan.close()


def test_in_place_clip_ops_with_no_utterances(ctx):
    assert host.normalize_volume(ctx, np.zeros(0, np.float32), np.zeros(1, np.int64)).shape == (0,)
    assert host.preemphasis(ctx, np.zeros(0, np.float32), np.zeros(1, np.int64)).shape == (0,)


def test_mfcc_one_frame_utterances_beside_long_ones(ctx, oracle):
    """The shortest utterance the boundary accepts (one sample more than a window: `slide` reads one past the window,
    src/clip.cpp:160-168, so a bare window is rejected, tests/test_gpu_mfcc.py::test_short_utterance_rejected) before,
    between and after ordinary ones."""
    rng = np.random.default_rng(5)
    cfg = host.mfcc_config()
    fb, _ = host.mel_filterbank()
    lens = [321, 5000, 321, 480, 8000, 321]
    so = np.zeros(len(lens) + 1, np.int64)
    so[1:] = np.cumsum(lens)
    x = rng.normal(0, 0.1, so[-1]).astype(np.float32)
    plan = host.MfccPlan(ctx, cfg, fb)
    for path in (host.MFCC_F32, host.MFCC_F16X3):
        plan.set_path(path)
        ceps, fo = plan.write_features(x, so)
        assert (np.diff(fo) == [oracle.num_frames(n, 320, 160) for n in lens]).all()
        assert np.diff(fo)[0] == 1 and np.diff(fo)[2] == 1 and np.diff(fo)[5] == 1
        for u in range(len(lens)):
            ref, _ = oracle.write_features(oracle.normalize_volume(x[so[u]:so[u + 1]]), 320, 160, fb, 13)
            got = ceps[fo[u]:fo[u + 1]]
            assert got.shape == ref.shape
            assert np.all(np.abs(got - ref) <= 1e-4 * np.maximum(np.abs(ref).max(axis=1, keepdims=True), 1.0)), (path, u)
    plan.close()


@pytest.mark.parametrize("path", [_capi.SRX_SCORE_EXACT, _capi.SRX_SCORE_MFMA, _capi.SRX_SCORE_MFMA_F16X3])
def test_gmm_score_with_no_frames_and_one_frame(ctx, oracle, path):
    rng = np.random.default_rng(7)
    G, M, D = 37, 4, 39
    m, mean, var, w = _model(ctx, rng, G, M, D)
    for mode in (_capi.SRX_MIX_COMPONENT0, _capi.SRX_MIX_LOGSUMEXP):
        out = m.score(np.zeros((0, D), np.float32), mode, path)
        assert out.shape == (0, G)
        x = rng.normal(0, 1, (1, D)).astype(np.float32)
        out = m.score(x, mode, path)
        ref = oracle.gmm_score(x, mean, var, oracle.set_wt(w), D, mode)
        assert np.all(np.abs(out - ref) <= 1e-4 * np.maximum(np.maximum(np.abs(out), np.abs(ref)), 1.0))
    m.close()


def test_viterbi_with_no_utterances(ctx):
    c = _capi.srx_hmm_batch(0, None, None, None, None, None, 1, 0)
    emis = ctx.zeros((1,), np.float32)
    path, fin, st = ctx.zeros((1,), np.int32), ctx.zeros((1,), np.float32), ctx.zeros((1,), np.int32)
    assert ctx.lib.srx_viterbi_batch_dev(ctx.h, C.byref(c), emis.ptr, 1, 1, 1e10, path.ptr, fin.ptr, st.ptr) == 0
    assert ctx.lib.srx_train_transitions_dev(ctx.h, C.byref(c), path.ptr, st.ptr) == 0
    gid = ctx.zeros((1,), np.int32)
    assert ctx.lib.srx_em_gather_gid_dev(ctx.h, C.byref(c), path.ptr, st.ptr, 0, gid.ptr) == 0
    ctx.sync()


def test_viterbi_zero_frame_one_frame_and_one_state_utterances(ctx, oracle):
    """Inside one batch: an utterance with no frames (status NO PATH, as the reference's empty `layer`), one frame, a
    one-state HMM, and ordinary ones."""
    rng = np.random.default_rng(11)
    G = 20
    specs = [(5, 0), (1, 1), (1, 7), (3, 1), (6, 40), (5, 0), (9, 300)]      # (states, frames)
    ntrs, trs, tieds, Ts = [], [], [], []
    for S, T in specs:
        ph = rng.integers(1, 30, S).astype(np.int32)
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr)
        trs.append(tr)
        tieds.append(rng.integers(0, G, S).astype(np.int32))
        Ts.append(T)
    fo = np.zeros(len(specs) + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    emis = rng.normal(-20, 5, (int(fo[-1]), G)).astype(np.float32)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    for layout in ("frame", "state", "align"):
        if layout == "frame":
            path_d, final_d, status_d = hb.viterbi(ctx.array(emis), G, 1)
        elif layout == "state":
            path_d, final_d, status_d = hb.viterbi(ctx.array(np.ascontiguousarray(emis.T)), 1, emis.shape[0])
        else:
            hb.align_plan()
            comp = np.zeros(max(hb.emis_count, 1), np.float32)
            for u, (S, T) in enumerate(specs):
                comp[hb.emis_off[u]:hb.emis_off[u + 1]] = emis[fo[u]:fo[u + 1]][:, tieds[u]].reshape(-1)
            path_d, final_d, status_d = hb.viterbi_align(ctx.array(comp))
        ctx.sync()
        path, final, status = path_d.get(), final_d.get(), status_d.get()
        for u, (S, T) in enumerate(specs):
            if T == 0:
                assert status[u] != 0, (layout, u)      # nothing to align: never reported as a path
                continue
            r = oracle.viterbi(emis[fo[u]:fo[u + 1]], tieds[u], ntrs[u], trs[u], 1e10, 0)
            assert status[u] == r["status"], (layout, u, status[u], r["status"])
            if r["status"] == 0:
                assert final[u].tobytes() == r["final"].tobytes(), (layout, u)
                assert np.array_equal(path[fo[u]:fo[u + 1]], r["path"]), (layout, u)
    hb.close()


def test_alignment_scoring_with_zero_frame_utterances(ctx, oracle):
    """srx_gmm_score_align_dev on a batch whose first, middle and last utterances have no frames, and on an empty batch."""
    rng = np.random.default_rng(13)
    G, M, D = 30, 16, 39
    m, mean, var, w = _model(ctx, rng, G, M, D)
    specs = [(4, 0), (5, 70), (3, 0), (8, 1), (6, 130), (2, 0)]
    ntrs, trs, tieds = [], [], []
    for S, T in specs:
        ntr, tr, _ = host.build_training_hmm(rng.integers(1, 30, S).astype(np.int32))
        ntrs.append(ntr)
        trs.append(tr)
        tieds.append(rng.integers(0, G, S).astype(np.int32))
    fo = np.zeros(len(specs) + 1, np.int64)
    fo[1:] = np.cumsum([t for _, t in specs])
    feats = rng.normal(0, 1, (int(fo[-1]), D)).astype(np.float32)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    feats_d = ctx.array(feats)
    for cached in (False, True):
        hb.cache_frames(cached)
        emis_d = hb.score_align(m, feats_d, D, mix_mode=_capi.SRX_MIX_LOGSUMEXP)
        ctx.sync()
        comp = emis_d.get()
        ref = oracle.gmm_score(feats, mean, var, oracle.set_wt(w), D, _capi.SRX_MIX_LOGSUMEXP)
        for u, (S, T) in enumerate(specs):
            got = comp[hb.emis_off[u]:hb.emis_off[u + 1]].reshape(T, S)
            exp = ref[fo[u]:fo[u + 1]][:, tieds[u]]
            assert np.all(np.abs(got - exp) <= 1e-4 * np.maximum(np.maximum(np.abs(got), np.abs(exp)), 1.0)), (cached, u)
    hb.close()
    # nothing at all
    h = C.c_void_p()
    z64, z32 = np.zeros(1, np.int64), np.zeros(1, np.int32)
    assert ctx.lib.srx_align_plan_create(ctx.h, z64.ctypes.data_as(C.c_void_p), z32.ctypes.data_as(C.c_void_p), 0, C.byref(h)) == 0
    assert ctx.lib.srx_align_plan_emission_count(h) == 0
    c = _capi.srx_hmm_batch(0, None, None, None, None, None, 1, 0)
    one = ctx.zeros((1,), np.float32)
    assert ctx.lib.srx_gmm_score_align_dev(ctx.h, m.h, _capi.SRX_MIX_LOGSUMEXP, _capi.SRX_SCORE_MFMA_F16X3, one.ptr, D, C.byref(c), h,
                                           one.ptr) == 0
    ctx.sync()
    ctx.lib.srx_align_plan_destroy(h)
    m.close()


def test_statistics_with_no_frames_leave_a_defined_model(ctx, oracle):
    """No frame reaches any Gaussian: the reference divides 0/0 (src/hmm.cpp:290-292) and blends NaN into every parameter;
    the same happens here -- the call succeeds and the oracle's NaN pattern is reproduced."""
    rng = np.random.default_rng(17)
    G, M, D = 6, 2, 5
    A, mean, var, w = _model(ctx, rng, G, M, D)
    B = host.AcousticModel(ctx, G, M, D)
    B.copy_from(A)
    feats_d = ctx.zeros((1, D), np.float32)
    gid_d = ctx.zeros((1,), np.int32)
    host.em_statistics(ctx, A, B, feats_d, D, gid_d, 0)
    m2, v2, w2 = B.get()
    rm, rv, rw, _, _ = oracle.em_statistics(np.zeros((1, D), np.float32)[:0], np.zeros(0, np.int32), D, mean, var, w, 0.2)
    assert np.array_equal(np.isnan(m2), np.isnan(rm.astype(np.float64)))
    assert np.array_equal(np.isnan(v2), np.isnan(rv.astype(np.float64)))
    assert np.array_equal(np.isnan(w2), np.isnan(rw))
    ok = ~np.isnan(rm.astype(np.float64))
    assert np.allclose(m2[ok], rm.astype(np.float64)[ok], rtol=1e-9)
    A.close()
    B.close()


def test_corpus_sums_and_collectives_with_nothing_to_do(ctx):
    """mean(ut) / variance(ut, mu) over an empty corpus (src/utterance.cpp:32-65: the loops do nothing and the division
    is the caller's), and the exchange step on empty buffers."""
    D = 39
    f = ctx.zeros((1, D), np.float32)
    s = ctx.zeros((2 * D,), np.float64)
    # the one-call form needs the mean of the first pass inside the second: 0/0 on an empty corpus (the reference's
    # `mean` returns NaN there) -- rejected with a message rather than computed
    assert ctx.lib.srx_corpus_mean_var_dev(ctx.h, f.ptr, D, 0, D, s.ptr) == _capi.SRX_ERR_INVALID
    assert b"srx_corpus_mean_var_dev" in ctx.lib.srx_last_error(ctx.h)
    # the two-call form (what a rank with an empty shard runs before the all-reduce): zeros
    s1 = ctx.empty((D,), np.float64)
    assert ctx.lib.srx_corpus_sums_dev(ctx.h, None, D, 0, D, None, s1.ptr) == 0
    ctx.sync()
    assert (s1.get() == 0).all()
    comm = host.Comm(ctx)
    for fn, dt in ((comm.allreduce_max_f32, np.float32), (comm.allreduce_sum_f64, np.float64), (comm.allreduce_sum_i32, np.int32)):
        fn(ctx.zeros((0,), dt))
    ctx.sync()
    assert comm.bytes_reduced() == 0
    comm.close()
