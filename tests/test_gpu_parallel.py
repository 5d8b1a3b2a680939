"""Two ranks (gloo, both on cuda:0) run parallel.TrainingShard.iteration -- alignment, transition re-estimation,
statistics through libsrx, the three exchanges -- for two iterations with the acm / temp_acm ping-pong, and must
end with the single-process result of the oracle (src/hmm.cpp:254-347).  RCCL itself needs one GPU per rank and is
exercised by the driver's multi-GPU bench; the collective calls are the same torch.distributed.all_reduce."""
import numpy as np
import pytest

from speechrecognition_amd import synth
from tests.test_parallel_cpu import run_world

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("M,mix_mode,emissions", [(1, 0, "dense"), (2, 1, "dense"), (2, 1, "compact")])
def test_training_iteration_world2_matches_single_process_oracle(oracle, tmp_path, M, mix_mode, emissions):
    rng = np.random.default_rng(21 + M)
    G, D, iters = 40, 13, 2
    n_utt = 7
    phones = [synth.phone_string(rng, n_words=3) for _ in range(n_utt)]
    lens = rng.integers(120, 300, n_utt)
    frame_off = np.zeros(n_utt + 1, np.int64)
    frame_off[1:] = np.cumsum(lens)
    T = int(frame_off[-1])
    mean, var, w = synth.random_model(G, M, D, 9)
    var *= 2.0
    feats = np.zeros((T, 39), np.float32)
    for u in range(n_utt):   # frames drawn around the utterance's own phone sequence so that alignments are non-trivial
        seg = np.sort(rng.integers(0, len(phones[u]), lens[u]))
        g = phones[u][seg]
        feats[frame_off[u]:frame_off[u + 1], :D] = (mean[g, 0] + rng.standard_normal((lens[u], D)) * 1.2).astype(np.float32)
    np.savez(tmp_path / "in.npz", shape=[G, M, D], mean=mean, var=var, w=w, feats=feats, frame_off=frame_off, iters=iters,
             mix_mode=mix_mode, emissions=emissions, **{"ph%d" % u: phones[u] for u in range(n_utt)})
    outs = run_world("gpu", tmp_path, 2)
    assert sorted(np.concatenate([o["mine"] for o in outs]).tolist()) == list(range(n_utt))
    assert all(len(o["mine"]) > 0 for o in outs)
    # ---- single-process oracle ----
    O = [mean.copy(), var.copy(), w.copy()]
    Cm = [mean.copy(), var.copy(), w.copy()]
    A, B = O, Cm
    hm = [list(oracle.build_training_hmm(p)[:2]) for p in phones]
    paths = [None] * n_utt
    for _ in range(iters):
        gids = []
        for u in range(n_utt):
            fv = feats[frame_off[u]:frame_off[u + 1]]
            e = oracle.gmm_score(fv, O[0], O[1], oracle.set_wt(O[2]), D, mix_mode)
            r = oracle.viterbi(e, phones[u].astype(np.int32), hm[u][0], hm[u][1])
            assert r["status"] == 0
            paths[u] = r["path"]
            hm[u][1] = oracle.train_transitions(r["path"], hm[u][0], hm[u][1])
            gids.append(phones[u][r["path"]].astype(np.int32))
        Bm, Bv, Bw, _, _ = oracle.em_statistics(feats, np.concatenate(gids), D, A[0], A[1], A[2], 0.2)
        B[0][...], B[1][...], B[2][...] = np.asarray(Bm, np.float64), np.asarray(Bv, np.float64), Bw
        A, B = B, A
    rm, rv, rw = A
    for o in outs:
        assert np.array_equal(np.isnan(o["mean"]), np.isnan(rm))
        ok = ~np.isnan(rm)
        # after the first iteration the models differ in the last bits (f64 vs 80-bit accumulation), which can move an
        # f32 log-density of the second iteration by one ulp and with it one frame's zeta by ~1e-6 relative
        np.testing.assert_allclose(o["mean"][ok], rm[ok], rtol=1e-6, atol=2e-7)
        np.testing.assert_allclose(o["var"][ok], rv[ok], rtol=1e-6, atol=2e-7)
        okw = ~np.isnan(rw)
        np.testing.assert_allclose(o["w"][okw], rw[okw], rtol=1e-6)
        assert int(o["bytes_reduced"]) == iters * (G * M * 4 + G * M * (1 + D) * 8 + G * M * D * 8)
        # this rank's last alignments and transition log-probs equal the single-process ones
        got_path, got_tr = o["path"], o["trlogp"]
        pos = spos = 0
        for u in o["mine"]:
            n, S = len(paths[u]), len(phones[u])
            assert np.array_equal(got_path[pos:pos + n], paths[u]), u
            np.testing.assert_allclose(got_tr[spos:spos + S], hm[u][1], rtol=5e-6, atol=1e-7)
            pos, spos = pos + n, spos + S
    assert outs[0]["mean"].tobytes() == outs[1]["mean"].tobytes()   # identical on every rank


def test_rccl_exchange_one_rank_equals_no_exchange(tmp_path):
    """The exchange as the driver's multi-GPU run performs it -- torch.distributed "nccl" (= RCCL) on device tensors, enqueued
    on the context's stream between the producer and the consumer kernels -- with the one rank a one-GPU box allows (RCCL
    refuses two ranks on one device): an all-reduce over one rank is the identity, so two iterations must end with the
    alignments and statistics of the run without collectives, and the byte count must be the three buffers' per iteration."""
    rng = np.random.default_rng(33)
    G, M, D, iters, n_utt = 40, 2, 13, 2, 6
    phones = [synth.phone_string(rng, n_words=3) for _ in range(n_utt)]
    lens = rng.integers(120, 300, n_utt)
    frame_off = np.zeros(n_utt + 1, np.int64)
    frame_off[1:] = np.cumsum(lens)
    mean, var, w = synth.random_model(G, M, D, 9)
    var *= 2.0
    feats = np.zeros((int(frame_off[-1]), 39), np.float32)
    for u in range(n_utt):
        seg = np.sort(rng.integers(0, len(phones[u]), lens[u]))
        feats[frame_off[u]:frame_off[u + 1], :D] = (mean[phones[u][seg], 0] + rng.standard_normal((lens[u], D)) * 1.2).astype(np.float32)
    outs = {}
    for mode in ("gpu", "gpu_rccl"):
        work = tmp_path / mode
        work.mkdir()
        np.savez(work / "in.npz", shape=[G, M, D], mean=mean, var=var, w=w, feats=feats, frame_off=frame_off, iters=iters,
                 mix_mode=1, emissions="compact", **{"ph%d" % u: phones[u] for u in range(n_utt)})
        outs[mode] = run_world(mode, work, 1)[0]
    a, b = outs["gpu"], outs["gpu_rccl"]
    assert int(a["bytes_reduced"]) == 0
    assert int(b["bytes_reduced"]) == iters * (G * M * 4 + G * M * (1 + D) * 8 + G * M * D * 8)
    for k in ("trlogp", "path"):
        assert a[k].tobytes() == b[k].tobytes(), k
    # (the statistics of a table this small are summed with LDS atomics, whose order differs from run to run: last bits)
    for k, tol in (("mean", 1e-12), ("var", 1e-12), ("w", 1e-6)):
        assert np.array_equal(np.isnan(a[k]), np.isnan(b[k])), k
        ok = ~np.isnan(a[k])
        np.testing.assert_allclose(b[k][ok], a[k][ok], rtol=tol, atol=0, err_msg=k)


def _corpus(seed, G, M, D, n_utt):
    rng = np.random.default_rng(seed)
    phones = [synth.phone_string(rng, n_words=3) for _ in range(n_utt)]
    lens = rng.integers(120, 300, n_utt)
    frame_off = np.zeros(n_utt + 1, np.int64)
    frame_off[1:] = np.cumsum(lens)
    mean, var, w = synth.random_model(G, M, D, 9)
    var *= 2.0
    feats = np.zeros((int(frame_off[-1]), 39), np.float32)
    for u in range(n_utt):
        seg = np.sort(rng.integers(0, len(phones[u]), lens[u]))
        feats[frame_off[u]:frame_off[u + 1], :D] = (mean[phones[u][seg], 0] + rng.standard_normal((lens[u], D)) * 1.2).astype(np.float32)
    return phones, frame_off, mean, var, w, feats


def _oracle_training(oracle, phones, frame_off, mean, var, w, feats, D, mix_mode, iters):
    """hmm::train's iterations in one process (src/hmm.cpp:254-347), Viterbi always under the first model object."""
    n_utt = len(phones)
    O = [mean.copy(), var.copy(), w.copy()]
    Cm = [mean.copy(), var.copy(), w.copy()]
    A, B = O, Cm
    hm = [list(oracle.build_training_hmm(p)[:2]) for p in phones]
    paths = [None] * n_utt
    for _ in range(iters):
        gids = []
        for u in range(n_utt):
            fv = feats[frame_off[u]:frame_off[u + 1]]
            e = oracle.gmm_score(fv, O[0], O[1], oracle.set_wt(O[2]), D, mix_mode)
            r = oracle.viterbi(e, phones[u].astype(np.int32), hm[u][0], hm[u][1])
            assert r["status"] == 0
            paths[u] = r["path"]
            hm[u][1] = oracle.train_transitions(r["path"], hm[u][0], hm[u][1])
            gids.append(phones[u][r["path"]].astype(np.int32))
        Bm, Bv, Bw, _, _ = oracle.em_statistics(feats, np.concatenate(gids), D, A[0], A[1], A[2], 0.2)
        B[0][...], B[1][...], B[2][...] = np.asarray(Bm, np.float64), np.asarray(Bv, np.float64), Bw
        A, B = B, A
    return A, paths


@pytest.mark.parametrize("em_path,iters", [("fold", 2), ("fused", 3)])
def test_fused_statistics_world2_over_the_library_communicator(oracle, tmp_path, em_path, iters):
    """Two ranks sharing cuda:0, the one-pass statistics (srx_em_fused_dev) and their single exchange through srx_comm_* (the
    host rehearsal transport: RCCL refuses two ranks on one device), against the single-process long-double oracle.
    Bars: "fold" -- only the variance formula differs -- as the five-step form over two iterations;
    "fused" -- log-densities without the reference's f32 roundings -- means within 1e-4 sigma and variances within 1e-4
    relative after three iterations.  Alignments must be the oracle's either way."""
    G, M, D, n_utt, mix_mode = 40, 2, 13, 7, 1
    phones, frame_off, mean, var, w, feats = _corpus(23, G, M, D, n_utt)
    np.savez(tmp_path / "in.npz", shape=[G, M, D], mean=mean, var=var, w=w, feats=feats, frame_off=frame_off, iters=iters,
             mix_mode=mix_mode, emissions="compact", em_path=em_path, exchange="srx", **{"ph%d" % u: phones[u] for u in range(n_utt)})
    outs = run_world("gpu", tmp_path, 2)
    (rm, rv, rw), paths = _oracle_training(oracle, phones, frame_off, mean, var, w, feats, D, mix_mode, iters)
    n12 = G * M * 16 * ((D + 1 + 7) // 8)
    for o in outs:
        assert int(o["fallbacks"]) == 0
        assert np.array_equal(np.isnan(o["mean"]), np.isnan(rm))
        ok = ~np.isnan(rm)
        sig = np.sqrt(rv[ok])
        dm = (np.abs(o["mean"] - rm)[ok] / sig).max()
        dv = (np.abs(o["var"] - rv)[ok] / rv[ok]).max()
        if em_path == "fold":
            assert dm <= 1e-6 and dv <= 1e-6, (dm, dv)
            assert int(o["bytes_reduced"]) == iters * (G * M * 4 + n12 * 8)
        else:
            assert dm <= 1e-4 and dv <= 1e-4, (dm, dv)
            assert int(o["bytes_reduced"]) == iters * n12 * 8          # ONE exchange per iteration
        okw = ~np.isnan(rw)
        np.testing.assert_allclose(o["w"][okw], rw[okw], rtol=1e-4)
        pos = 0
        for u in o["mine"]:
            n = len(paths[u])
            assert np.array_equal(o["path"][pos:pos + n], paths[u]), u
            pos += n
    assert outs[0]["mean"].tobytes() == outs[1]["mean"].tobytes()


def test_library_communicator_one_rank_rccl_equals_no_exchange(tmp_path):
    """srx_comm_create with one forced RCCL rank (SRX_COMM_FORCE_RCCL): ncclAllReduce on the context's stream between the
    fused statistics kernel and its finish.  The result must be the run without any collective, bit for bit (the segmented
    reduction is deterministic), and the byte count one buffer per iteration."""
    G, M, D, n_utt, iters = 40, 2, 13, 6, 2
    phones, frame_off, mean, var, w, feats = _corpus(35, G, M, D, n_utt)
    outs = {}
    for mode, exch in (("gpu", "torch"), ("gpu_rccl", "srx")):
        work = tmp_path / mode
        work.mkdir()
        np.savez(work / "in.npz", shape=[G, M, D], mean=mean, var=var, w=w, feats=feats, frame_off=frame_off, iters=iters,
                 mix_mode=1, emissions="compact", em_path="fused", exchange=exch, **{"ph%d" % u: phones[u] for u in range(n_utt)})
        outs[mode] = run_world(mode, work, 1)[0]
    a, b = outs["gpu"], outs["gpu_rccl"]
    assert int(a["bytes_reduced"]) == 0
    assert int(b["bytes_reduced"]) == iters * G * M * 16 * ((D + 1 + 7) // 8) * 8
    for k in ("trlogp", "path", "mean", "var", "w"):
        assert a[k].tobytes() == b[k].tobytes(), k
