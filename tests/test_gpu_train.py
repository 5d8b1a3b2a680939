"""Parity of the embedded-training statistics (hmm::train, src/hmm.cpp:274-341) against the oracle.

The reference accumulates in 80-bit long double; the GPU in f64 with a per-state log shift.  Means,
variances and weights are ratios of sums, compared at rtol 1e-9 (far inside the 1e-4 bar)."""
import numpy as np
import pytest

from speechrecognition_amd import host, synth

pytestmark = pytest.mark.gpu


def _run(ctx, oracle, G, M, D, T, seed, far=False):
    rng = np.random.default_rng(seed)
    mean, var, w = synth.random_model(G, M, D, seed)
    gid = rng.integers(0, G, T).astype(np.int32)
    gid[gid == 3] = 4                      # tied state 3 is never visited -> NaN row (reference behaviour)
    feats = np.zeros((T, 39), np.float32)
    comp = rng.integers(0, M, T)
    feats[:, :D] = (mean[gid, comp] + rng.standard_normal((T, D)) * np.sqrt(var[gid, comp])).astype(np.float32)
    if far:
        feats[:, :D] += 25.0               # log-densities near -4000: f64 exp() underflows without the shift
    A = host.AcousticModel(ctx, G, M, D)
    B = host.AcousticModel(ctx, G, M, D)
    A.set(mean, var, w)
    B.copy_from(A)
    host.em_statistics(ctx, A, B, ctx.array(feats), 39, ctx.array(gid), T, 0.2)
    gm, gv, gw = B.get()
    rm, rv, rw, rlw, zs = oracle.em_statistics(feats, gid, D, mean, var, w, 0.2)
    rm, rv = rm.astype(np.float64), rv.astype(np.float64)
    assert np.array_equal(np.isnan(gm), np.isnan(rm))
    assert np.isnan(rm[3]).all() and not np.isnan(rm[4]).any()
    ok = ~np.isnan(rm)
    np.testing.assert_allclose(gm[ok], rm[ok], rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(gv[ok], rv[ok], rtol=1e-9, atol=1e-12)
    okw = ~np.isnan(rw)
    np.testing.assert_allclose(gw[okw], rw[okw], rtol=3e-7)
    A.close()
    B.close()


def test_em_statistics_reference_shape(ctx, oracle):
    _run(ctx, oracle, 40, 1, 13, 6000, 1)


def test_em_statistics_mixtures(ctx, oracle):
    _run(ctx, oracle, 12, 4, 13, 5000, 2)


def test_em_statistics_underflow_guard(ctx, oracle):
    _run(ctx, oracle, 10, 2, 13, 3000, 3, far=True)


def test_em_statistics_hbm_table(ctx, oracle):
    """Table too large for LDS privatisation -> global f64 atomics path."""
    _run(ctx, oracle, 300, 2, 39, 4000, 4)


def _run_fused(ctx, oracle, G, M, D, T, seed, given_logp, far=0.0, offset=0.0):
    """The one-pass statistics (srx_em_fused_dev) against the long-double oracle.  Returns the worst deviations:
    (means in units of the component's standard deviation, variances relative, weights relative, underflow count)."""
    rng = np.random.default_rng(seed)
    mean, var, w = synth.random_model(G, M, D, seed)
    mean += offset                         # the whole model (and its data) away from the origin
    gid = rng.integers(0, G, T).astype(np.int32)
    gid[gid == 3] = 4                      # tied state 3 is never visited -> NaN row (reference behaviour)
    feats = np.zeros((T, 39), np.float32)
    comp = rng.integers(0, M, T)
    feats[:, :D] = (mean[gid, comp] + rng.standard_normal((T, D)) * np.sqrt(var[gid, comp])).astype(np.float32)
    feats[:, :D] += far
    A = host.AcousticModel(ctx, G, M, D)
    B = host.AcousticModel(ctx, G, M, D)
    A.set(mean, var, w)
    B.copy_from(A)
    _, under = host.em_statistics_fused(ctx, A, B, ctx.array(feats), 39, ctx.array(gid), T, 0.2, given_logp)
    gm, gv, gw = B.get()
    rm, rv, rw, rlw, zs = oracle.em_statistics(feats, gid, D, mean, var, w, 0.2)
    rm, rv = rm.astype(np.float64), rv.astype(np.float64)
    A.close()
    B.close()
    if under:
        return None, None, None, under
    assert np.array_equal(np.isnan(gm), np.isnan(rm))
    assert np.isnan(rm[3]).all() and not np.isnan(rm[4]).any()
    ok = ~np.isnan(rm)
    # B = 0.2 new + 0.8 A (gaussian::blend): deviations of the NEW mean / variance are 5x what B shows, and are measured
    # against the new standard deviation / variance (the oracle's, un-blended)
    # The variance is formed as S2/S0 - (S1/S0)^2 around the state's centre c: beside the bar on the statistics themselves it
    # carries the f64 cancellation term 16 * 2^-53 * (vnew + (mean_new - c)^2), which only shows when a component's new mean
    # is far from c in units of its new spread (a component held by one distant frame); that term is taken out before the
    # relative deviation is formed.
    vnew = (rv - 0.8 * var) / 0.2
    mnew = (rm - 0.8 * mean) / 0.2
    c = mean.mean(axis=1, keepdims=True)
    cancel = 16 * 2.0 ** -53 * (vnew + (mnew - c) ** 2)
    dm = (np.abs(gm - rm)[ok] * 5 / np.sqrt(vnew[ok])).max()
    dv = (np.maximum(np.abs(gv - rv) * 5 - cancel, 0)[ok] / vnew[ok]).max()
    okw = ~np.isnan(rw)
    dw = (np.abs(gw - rw)[okw] / np.maximum(rw[okw], 1e-30)).max() * 5
    return dm, dv, dw, under


@pytest.mark.parametrize("G,M,D,T,seed", [(40, 1, 13, 6000, 1), (12, 4, 13, 5000, 2), (300, 2, 39, 4000, 4), (50, 16, 39, 60000, 5)])
def test_em_fused_given_logp_folds_only_the_variance_pass(ctx, oracle, G, M, D, T, seed):
    """zeta from the rounding-faithful log-densities: means and weights are the five-step form's sums (1e-9), the variances
    come from S2/S0 - (S1/S0)^2 in f64 around the state's centre instead of the reference's sum of f32-rounded differences
    (src/hmm.cpp:331).  Bar on a new variance: 2^-23 = 1.2e-7 relative -- the reference rounds every difference to f32 (2^-24
    relative), so every squared term of its sum is off by at most 2^-23 of itself and so is the sum of those non-negative
    terms; a component with a handful of frames comes close to that bound, one with hundreds averages it down (VERDICT r2
    measured 7e-9 at 150 frames per state) -- plus the 1e-9 the five-step form is held to.  Means: 1e-9 (in sigmas)."""
    dm, dv, dw, under = _run_fused(ctx, oracle, G, M, D, T, seed, True)
    assert under == 0
    assert dm <= 1e-9 and dv <= 2.0 ** -23 + 1e-9 and dw <= 2e-6, (dm, dv, dw)


@pytest.mark.parametrize("G,M,D,T,seed", [(40, 1, 13, 6000, 1), (12, 4, 13, 5000, 2), (300, 2, 39, 4000, 4), (50, 16, 39, 60000, 5)])
def test_em_fused_matrix_core_log_densities(ctx, oracle, G, M, D, T, seed):
    """Log-densities on the f64 matrix cores, without the f32 roundings gaussian::operator() applies to every term
    (src/gmm.cpp:44-47): zeta differs from the reference's by the reference's own rounding noise.  Bar:
    new means within 1e-5 sigma, new variances within 1e-5 relative after one iteration."""
    dm, dv, dw, under = _run_fused(ctx, oracle, G, M, D, T, seed, False)
    assert under == 0
    assert dm <= 1e-5 and dv <= 1e-5 and dw <= 1e-4, (dm, dv, dw)


def test_em_fused_means_far_from_the_origin(ctx, oracle):
    """Component means hundreds of standard deviations from the origin: S2/S0 - (S1/S0)^2 would cancel without the centring."""
    dm, dv, dw, under = _run_fused(ctx, oracle, 20, 4, 39, 8000, 7, False, offset=500.0)
    assert under == 0 and dm <= 1e-5 and dv <= 1e-5, (dm, dv)
    dm, dv, dw, under = _run_fused(ctx, oracle, 20, 4, 39, 8000, 7, True, offset=500.0)
    assert under == 0 and dm <= 1e-9 and dv <= 2.0 ** -23 + 1e-9, (dm, dv)


def test_em_fused_reports_underflow_instead_of_returning_garbage(ctx, oracle):
    """Every frame 25 units from every mean: log-densities near -4000 relative to the peak, beyond f64's exp().  The
    matrix-core form counts the components it lost (the caller then runs the five-step form, which shifts by the data's own
    maximum: test_em_statistics_underflow_guard); with given log-densities and their shift nothing is lost."""
    *_, under = _run_fused(ctx, oracle, 10, 2, 13, 3000, 3, False, far=25.0)
    assert under > 0
    dm, dv, dw, under = _run_fused(ctx, oracle, 10, 2, 13, 3000, 3, True, far=25.0)
    assert under == 0 and dm <= 1e-9 and dv <= 2.0 ** -23 + 1e-9, (dm, dv)


def test_gather_gid_skips_no_path(ctx):
    ph = np.array([0, 5, 6, 0], np.int32)
    ntr, tr, _ = host.build_training_hmm(ph)
    fo = np.array([0, 2, 12], np.int64)
    tied = np.array([7, 8, 9, 7], np.int32)
    hb = host.HmmBatch(ctx, fo, [ntr, ntr], [tr, tr], [tied, tied])
    emis = ctx.array(np.zeros((12, 10), np.float32))
    p, f, s = hb.viterbi(emis, 10)
    gid = host.gather_gid(ctx, hb, p, s).get()
    st = s.get()
    assert st[0] == 1 and st[1] == 0
    assert (gid[:2] == -1).all() and set(gid[2:]) <= {7, 8, 9}


def test_state_model_augment_against_oracle(ctx, oracle):
    """a24: state_model::augment over every frame of every final path (src/hmm.cpp:46-51, 350-356) and
    state_model::operator() (:39-44).  The reference keeps f32 running sums in path order, the kernel f64 atomics:
    counts are exact; sums agree with the oracle's f32 running sums at rtol 2e-5 (f32 accumulation error of <= 1e4
    terms) and with an f64 sum of the same f32 terms at 1e-6 (device expf vs libm expf, <= 1 ulp per term).
    The last state of a training HMM has one transition: get_adv_prob() reads tr[1] out of bounds there in the
    reference (UB); those frames carry class -1 and are skipped on both sides."""
    rng = np.random.default_rng(71)
    G, C_ = 40, 23
    ntrs, trs, tieds, Ts, clss = [], [], [], [], []
    for u in range(30):
        ph = synth.phone_string(rng, 5)
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr); trs.append(tr)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
        Ts.append(int(rng.integers(4 if u == 0 else 120, 5 if u == 0 else 400)))     # utterance 0: too short -> NO PATH
        c = rng.integers(0, C_ - 1, len(ph)).astype(np.int32)     # class C_-1 is never seen -> operator() gives 0.5
        c[-1] = -1
        clss.append(c)
    fo = np.zeros(len(Ts) + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    emis = (rng.standard_normal((int(fo[-1]), G)) * 4 - 15).astype(np.float32)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    path_d, final_d, status_d = hb.viterbi(ctx.array(emis), G)
    for _ in range(2):
        hb.train_transition_probabilities(path_d, status_d)
    ctx.sync()
    path, status, tr = path_d.get(), status_d.get(), hb.trlogp()
    assert status[0] == 1 and (status[1:] == 0).all()
    ps, cnt = hb.state_model(np.concatenate(clss), path_d, status_d, C_)
    # the same frames, in path order, for the oracle
    cls_t, adv_t = [], []
    for u in range(1, len(Ts)):
        st = path[fo[u]:fo[u + 1]]
        c = clss[u][st]
        a = np.exp(tr[hb.state_off[u] + st, 1].astype(np.float32)).astype(np.float32)
        keep = c >= 0
        cls_t.append(c[keep]); adv_t.append(a[keep])
    cls_t, adv_t = np.concatenate(cls_t), np.concatenate(adv_t)
    rps, rcnt = oracle.state_model(cls_t, adv_t, C_)
    assert np.array_equal(cnt, rcnt) and cnt[C_ - 1] == 0 and cnt.sum() == len(cls_t)
    np.testing.assert_allclose(ps, rps.astype(np.float64), rtol=2e-5)
    np.testing.assert_allclose(ps, np.bincount(cls_t, adv_t.astype(np.float64), C_), rtol=1e-6)
    # state_model::operator(): pooled mean, 0.5 where nothing was seen
    pooled = np.where(cnt > 0, ps / np.maximum(cnt, 1), 0.5)
    ref = np.where(rcnt > 0, rps / np.maximum(rcnt, 1).astype(np.float32), np.float32(0.5))
    np.testing.assert_allclose(pooled, ref, rtol=2e-5)
    assert pooled[C_ - 1] == 0.5


@pytest.mark.parametrize("emissions", ["compact", "dense"])
def test_score_reuse_across_the_model_swap_changes_nothing(ctx, emissions):
    """hmm::train's utterance HMMs score under main()'s acoustic_model object (src/utterance.cpp:20), which the acm / temp_acm
    swap (src/hmm.cpp:344-346) rewrites only every second iteration: the training shard skips the scoring pass while that
    object's version has not moved.  Four iterations with the reuse on and off: two scoring passes against four, the same
    alignments bit for bit; the models agree to the last bits that the statistics' LDS atomics leave undetermined."""
    from speechrecognition_amd import parallel
    rng = np.random.default_rng(41)
    G, M, D, n_utt = 40, 2, 13, 6
    phones = [synth.phone_string(rng, n_words=3) for _ in range(n_utt)]
    lens = rng.integers(120, 300, n_utt)
    fo = np.zeros(n_utt + 1, np.int64)
    fo[1:] = np.cumsum(lens)
    mean, var, w = synth.random_model(G, M, D, 9)
    var *= 2.0
    feats = np.zeros((int(fo[-1]), 39), np.float32)
    for u in range(n_utt):
        seg = np.sort(rng.integers(0, len(phones[u]), lens[u]))
        feats[fo[u]:fo[u + 1], :D] = (mean[phones[u][seg], 0] + rng.standard_normal((lens[u], D)) * 1.2).astype(np.float32)
    results = []
    for reuse in (True, False):
        hm = [host.build_training_hmm(p) for p in phones]
        hb = host.HmmBatch(ctx, fo, [h[0] for h in hm], [h[1] for h in hm], [p.astype(np.int32) for p in phones])
        X, Y = host.AcousticModel(ctx, G, M, D), host.AcousticModel(ctx, G, M, D)
        X.set(mean, var, w)
        Y.copy_from(X)
        shard = parallel.TrainingShard(ctx, hb, ctx.array(feats), 39, G, M, D, None, 1, 2 if emissions == "compact" else 0,
                                       emissions=emissions)
        shard.reuse_scores = reuse
        paths = []
        for k in range(4):
            A, B = (X, Y) if k % 2 == 0 else (Y, X)
            v0 = X.version()
            shard.iteration(X, A, B, 0.2)
            ctx.sync()
            assert (X.version() != v0) == (k % 2 == 1)      # X is rebuilt in the iterations where it is temp_acm
            paths.append(shard.path.get()[:hb.total_frames].copy())
        assert shard.scoring_passes == (2 if reuse else 4)
        results.append((paths, X.get(), Y.get(), hb.trlogp()))
        hb.close()
        X.close()
        Y.close()
    (p1, x1, y1, t1), (p2, x2, y2, t2) = results
    for a, b in zip(p1, p2):
        assert np.array_equal(a, b)
    assert t1.tobytes() == t2.tobytes()
    for m1, m2 in ((x1, x2), (y1, y2)):
        for a, b in zip(m1, m2):
            ok = ~np.isnan(a)
            assert np.array_equal(np.isnan(a), np.isnan(b))
            np.testing.assert_allclose(b[ok], a[ok], rtol=1e-10)
