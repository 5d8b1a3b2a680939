#!/usr/bin/env python3
"""Kernel A/B helper: times k_em_fused (one-pass EM statistics) of one or more builds of libsrx on a synthetic alignment of
20 M frames (state runs of ~12 frames, 2,000 states x 16 components x 39 dims), without looking at the results (diagnostic
builds may compute garbage).  Development tool only.
  hipcc ... -DSRX_FUSED_EXP_<X> -c srx_train_seg.hip -o /tmp/v.o && link against the other objects -> python tools/em_variant.py lib1.so lib2.so"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import _capi, synth  # noqa: E402


def run(path, T):
    _capi._lib = None
    _capi.LIB_PATH = path
    from speechrecognition_amd import host
    G, M, D = 2000, 16, 39
    ctx = host.Context(0)
    rng = np.random.default_rng(4)
    runs = rng.integers(4, 21, T // 10)
    gid = np.repeat(rng.integers(0, G, len(runs)).astype(np.int32), runs)[:T]
    T = len(gid)
    feats = (rng.standard_normal((T, D)) * 1.3).astype(np.float32)
    A, B = host.AcousticModel(ctx, G, M, D), host.AcousticModel(ctx, G, M, D)
    A.set(*synth.random_model(G, M, D, 777))
    B.copy_from(A)
    feats_d, gid_d = ctx.array(feats), ctx.array(gid)
    host.em_statistics_fused(ctx, A, B, feats_d, D, gid_d, T)
    ctx.profile_enable(True)
    for _ in range(5):
        host.em_statistics_fused(ctx, A, B, feats_d, D, gid_d, T)
    ms, k = ctx.profile_query("k_em_fused")
    ctx.close()
    return ms / k, T


if __name__ == "__main__":
    for p in sys.argv[1:]:
        ms, T = run(os.path.abspath(p), 20_000_000)
        print("%-28s k_em_fused %.3f ms  (%.4f ns per frame, %d frames)" % (os.path.basename(os.path.dirname(os.path.abspath(p))) + "/" + os.path.basename(p), ms, ms * 1e6 / T, T), flush=True)
