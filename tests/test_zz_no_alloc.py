"""srx_viterbi_batch_noalloc_dev (include/srx.h): the entry point that allocates nothing.  Same kernel as
srx_viterbi_batch_dev with the back-pointer words in the library's fixed per-GPU device array; called through the C
ABI and checked against srx_viterbi_batch_dev bit for bit (scores, paths, status: integer/byte bar), on two calls in a
row, the second with a smaller input (the array keeps the first call's contents), then beyond the limit.
Collected last (file name): pytest -x would otherwise hide every test after a failure here."""
import ctypes as C

import numpy as np
import pytest

from speechrecognition_amd import host, synth
from speechrecognition_amd._capi import SRX_ERR_INVALID

pytestmark = pytest.mark.gpu


def _batch(ctx, rng, n_utt, G, t_lo, t_hi, words):
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
    return host.HmmBatch(ctx, fo, ntrs, trs, tieds)


def _both(ctx, hb, emis, layout, beam):
    T, G = emis.shape
    if layout == "state-major":
        Tp = (T + 3) // 4 * 4
        buf = np.zeros((G, Tp), np.float32)
        buf[:, :T] = emis.T
        e_d, st_t, st_g = ctx.array(buf), 1, Tp
    else:
        e_d, st_t, st_g = ctx.array(emis), G, 1
    outs = []
    for fn in (hb.viterbi, hb.viterbi_noalloc):
        # outputs pre-filled with a pattern: what a call leaves untouched (NO PATH rows) must agree too
        path_d = ctx.array(np.full(max(T, 1), -7, np.int32))
        final_d = ctx.array(np.full(hb.n_utt, -3.0, np.float32))
        status_d = ctx.array(np.full(hb.n_utt, 9, np.int32))
        fn(e_d, st_t, st_g, beam, path_d, final_d, status_d)
        ctx.sync()
        outs.append((path_d.get(), final_d.get(), status_d.get()))
    (p0, f0, s0), (p1, f1, s1) = outs
    assert np.array_equal(s0, s1)
    assert f0.tobytes() == f1.tobytes()
    assert np.array_equal(p0, p1)
    return s0


def test_noalloc_viterbi_matches_allocating_entry_point_twice_second_call_smaller(ctx):
    rng = np.random.default_rng(5)
    G = 64
    assert ctx.lib.srx_viterbi_noalloc_max_bytes() == 32 << 20
    # call 1: 300 utterances, up to two states per lane; call 2: smaller batch, narrower utterances, other layout
    big = _batch(ctx, rng, 300, G, 100, 700, words=14)
    e_big = (rng.standard_normal((big.total_frames, G)) * 4 - 20).astype(np.float32)
    e_big[big.frame_off[3]:big.frame_off[3] + 2, :] = -np.inf          # one NO PATH utterance
    s = _both(ctx, big, e_big, "state-major", 1e10)
    assert s[3] == 1 and (np.delete(s, 3) == 0).all()
    small = _batch(ctx, rng, 17, G, 5, 90, words=3)
    e_small = (rng.standard_normal((small.total_frames, G)) * 4 - 20).astype(np.float32)
    _both(ctx, small, e_small, "frame-major", 1e10)
    _both(ctx, small, e_small, "state-major", 40.0)
    _both(ctx, big, e_big, "frame-major", 1e10)


def test_noalloc_viterbi_rejects_sizes_beyond_the_fixed_scratch(ctx):
    """No kernel is launched: only the host-side size check runs (the pointers are never dereferenced)."""
    rng = np.random.default_rng(6)
    hb = _batch(ctx, rng, 4, 8, 50, 60, words=3)
    c = hb.c
    saved = c.total_frames
    c.total_frames = 40_000_000          # 2.5 M words-rows x 256 B = 640 MB of back-pointer words
    dummy = ctx.zeros((16,), np.float32)
    rc = ctx.lib.srx_viterbi_batch_noalloc_dev(ctx.h, C.byref(c), dummy.ptr, 8, 1, 1e10, dummy.ptr, dummy.ptr, dummy.ptr)
    c.total_frames = saved
    assert rc == SRX_ERR_INVALID
    msg = ctx.lib.srx_last_error(ctx.h).decode()
    assert "fixed scratch" in msg and "srx_viterbi_batch_dev" in msg
