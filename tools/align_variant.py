#!/usr/bin/env python3
"""Kernel A/B helper: times k_gmm_f16_align (cached frame operands) of one or more builds of libsrx on 10,000 utterances,
without looking at the results (diagnostic builds may compute garbage).  Development tool only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import _capi, synth  # noqa: E402


def run(path, n_utt):
    _capi._lib = None
    _capi.LIB_PATH = path
    from speechrecognition_amd import host
    G, M, D = 2000, 16, 39
    ctx = host.Context(0)
    rng = np.random.default_rng(4)
    ntrs, trs, tieds, Ts = [], [], [], []
    for _ in range(n_utt):
        ph = synth.phone_string(rng, int(rng.integers(4, 13)))
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr); trs.append(tr)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
        t_ = int(rng.integers(200, 801))
        q_ = int(os.environ.get("SRX_T_QUANTUM", "0"))     # e.g. 256: every utterance fills whole workgroups (packing bound)
        Ts.append(max(q_, (t_ + q_ // 2) // q_ * q_) if q_ else t_)
    fo = np.zeros(n_utt + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    T = int(fo[-1])
    feats = (rng.standard_normal((T, D)) * 1.3).astype(np.float32)
    A = host.AcousticModel(ctx, G, M, D)
    A.set(*synth.random_model(G, M, D, 777))
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    hb.align_plan()
    hb.cache_frames(True)
    feats_d = ctx.array(feats)
    emis = ctx.empty((hb.emis_count,), np.float32)
    from speechrecognition_amd._capi import SRX_SCORE_MFMA_F16X3
    hb.score_align(A, feats_d, D, emis, 1, SRX_SCORE_MFMA_F16X3)
    ctx.sync()
    ctx.profile_enable(True)
    for _ in range(5):
        hb.score_align(A, feats_d, D, emis, 1, SRX_SCORE_MFMA_F16X3)
    ms, k = ctx.profile_query("k_gmm_f16_align")
    ctx.close()
    print("  frames %d, %.4f ns per frame" % (T, ms / k * 1e6 / T))
    return ms / k


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print("%-32s k_gmm_f16_align %.4f ms" % (os.path.basename(p), run(os.path.abspath(p), 10000)), flush=True)
