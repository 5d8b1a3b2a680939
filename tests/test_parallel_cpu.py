"""Multi-GPU host logic without a GPU: the utterance sharder and the three-exchange protocol of one EM
iteration (speechrecognition_amd/parallel.py) over a world_size-2 gloo group, against the single-process
oracle (the reference's sequential loops, src/hmm.cpp:274-341)."""
import os
import socket
import subprocess
import sys

import numpy as np

from speechrecognition_amd import parallel, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def run_world(mode, work, world=2, timeout=240):
    port = free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "_dist_worker.py"), mode, str(work)],
                                      env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = []
    for p in procs:
        try:
            o, _ = p.communicate(timeout=timeout)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        outs.append(o)
    for r, p in enumerate(procs):
        assert p.returncode == 0, (r, outs[r][-3000:])
    return [np.load(os.path.join(str(work), "out%d.npz" % r)) for r in range(world)]


def test_shard_utterances_balanced_deterministic_complete():
    rng = np.random.default_rng(0)
    costs = rng.integers(200, 800, 1000) * rng.integers(15, 85, 1000)
    for world in (1, 2, 4, 8):
        parts = parallel.shard_utterances(costs, world)
        assert len(parts) == world
        allu = np.concatenate(parts)
        assert sorted(allu.tolist()) == list(range(1000))          # every utterance exactly once
        loads = np.array([costs[p].sum() for p in parts], np.float64)
        assert loads.max() / loads.mean() < 1.01                   # LPT on 1000 items: within 1 %
        again = parallel.shard_utterances(costs, world)
        assert all((a == b).all() for a, b in zip(parts, again))
    # ragged: more ranks than utterances, empty corpus
    parts = parallel.shard_utterances([5.0, 1.0, 3.0], 8)
    assert sorted(np.concatenate(parts).tolist()) == [0, 1, 2] and sum(len(p) == 0 for p in parts) == 5
    assert all(len(p) == 0 for p in parallel.shard_utterances([], 4))


def test_exchange_is_noop_without_process_group():
    import torch
    ex = parallel.Exchange()
    assert not ex.active
    t = torch.arange(4, dtype=torch.float64)
    assert ex.allreduce_sum(t) is t and ex.bytes_reduced == 0


def test_em_exchange_protocol_world2_gloo(oracle, tmp_path):
    rng = np.random.default_rng(11)
    G, M, D = 12, 3, 13
    mean, var, w = synth.random_model(G, M, D, 5)
    lens = rng.integers(40, 160, 9)
    frame_off = np.zeros(len(lens) + 1, np.int64)
    frame_off[1:] = np.cumsum(lens)
    T = int(frame_off[-1])
    gid = rng.integers(0, G, T).astype(np.int32)
    gid[gid == 7] = 8                          # tied state 7 never visited -> NaN row on every rank
    comp = rng.integers(0, M, T)
    feats = np.zeros((T, 39), np.float32)
    feats[:, :D] = (mean[gid, comp] + rng.standard_normal((T, D)) * np.sqrt(var[gid, comp])).astype(np.float32)
    feats[: lens[0], :D] += 20.0               # one utterance far out: its rank's local maximum differs -> the MAX exchange matters
    np.savez(tmp_path / "in.npz", shape=[G, M, D], mean=mean, var=var, w=w, feats=feats, gid=gid, frame_off=frame_off)
    outs = run_world("cpu", tmp_path, 2)
    assert sum(int(o["nframes"]) for o in outs) == T and all(int(o["nframes"]) > 0 for o in outs)
    rm, rv, rw, _, _ = oracle.em_statistics(feats, gid, D, mean, var, w, 0.2)
    rm, rv = np.asarray(rm, np.float64), np.asarray(rv, np.float64)
    for o in outs:                              # every rank ends with the same model, equal to the single-process one
        assert np.array_equal(np.isnan(o["mean"]), np.isnan(rm)) and np.isnan(rm[7]).all()
        ok = ~np.isnan(rm)
        # the numpy worker evaluates log-densities in f64, the reference rounds them to f32 (src/gmm.cpp:41-49):
        # zeta differs by ~1e-5 relative; this test is about the protocol (the kernels' numerics: test_gpu_train.py)
        np.testing.assert_allclose(o["mean"][ok], rm[ok], rtol=5e-5, atol=1e-7)
        np.testing.assert_allclose(o["var"][ok], rv[ok], rtol=1e-4, atol=1e-7)
        okw = ~np.isnan(rw)
        np.testing.assert_allclose(o["w"][okw], rw[okw], rtol=1e-4)
        # bytes on the wire per iteration: shift f32 + stats1 f64 + stats2 f64
        assert int(o["bytes_reduced"]) == G * M * 4 + G * M * (1 + D) * 8 + G * M * D * 8
    assert outs[0]["mean"].tobytes() == outs[1]["mean"].tobytes()
