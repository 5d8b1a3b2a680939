#!/usr/bin/env python3
"""Kernel A/B helper: times k_gmm_mfma of one or more builds of libsrx (paths on the command line) on the
bench shape (T=499,674, G=2000, M=16, D=39), interleaved in one process.  Development tool only."""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import _capi, synth  # noqa: E402


def run(path, feats, model_np, T, G, M, D, reps):
    _capi._lib = None
    _capi.LIB_PATH = path
    from speechrecognition_amd import host
    ctx = host.Context(0)
    model = host.AcousticModel(ctx, G, M, D)
    model.set(*model_np)
    f = ctx.array(feats)
    out = ctx.empty((G, T), np.float32)
    model.score_dev(f, D, T, out, 1, T, 1, 0)
    ctx.sync()
    ctx.profile_enable(True)
    for _ in range(reps):
        model.score_dev(f, D, T, out, 1, T, 1, 0)
    ms, n = ctx.profile_query("k_gmm_mfma")
    chk = float(out.get()[::97, ::1013].sum())
    ctx.close()
    return ms / n, chk


def main():
    T, G, M, D = 499674, 2000, 16, 39
    rng = np.random.default_rng(0)
    feats = (rng.standard_normal((T, D)) * 1.3).astype(np.float32)
    model_np = synth.random_model(G, M, D, 777)
    for rnd in range(1):
        for p in sys.argv[1:]:
            ms, chk = run(os.path.abspath(p), feats, model_np, T, G, M, D, 1)
            print("%-28s %.3f ms  (%.1f%% of 157.3 TF)  checksum %.4f" % (os.path.basename(p), ms, 5.056e6 * T / ms / 1e9 / 157.3 * 100, chk), flush=True)


if __name__ == "__main__":
    main()
