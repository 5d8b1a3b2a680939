#!/usr/bin/env python3
"""Kernel A/B helper: times the GMM scoring kernels of one or more builds of libsrx (paths on the command line) on
the bench shape (T=499,674, G=2000, M=16, D=39), in one process.  Development tool only.
  python tools/gmm_variant.py [--path 0|2] lib1.so [lib2.so ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import _capi, synth  # noqa: E402


def run(path, feats, model_np, T, G, M, D, reps, score_path):
    _capi._lib = None
    _capi.LIB_PATH = path
    from speechrecognition_amd import host
    ctx = host.Context(0)
    model = host.AcousticModel(ctx, G, M, D)
    model.set(*model_np)
    f = ctx.array(feats)
    out = ctx.empty((G, T), np.float32)
    model.score_dev(f, D, T, out, 1, T, 1, score_path)
    ctx.sync()
    ctx.profile_enable(True)
    for _ in range(reps):
        model.score_dev(f, D, T, out, 1, T, 1, score_path)
    res = {}
    for k in ("k_gmm_mfma", "k_gmm_prep", "k_gmm_f16", "k_prep16", "k_gmm_fixup16"):
        ms, n = ctx.profile_query(k)
        if n:
            res[k] = ms / n
    chk = float(out.get()[::97, ::1013].sum())
    ctx.close()
    return res, chk


def main():
    args = sys.argv[1:]
    score_path = 0
    if args and args[0] == "--path":
        score_path = int(args[1])
        args = args[2:]
    T, G, M, D = int(os.environ.get("SRX_T", "499674")), int(os.environ.get("SRX_G", "2000")), 16, 39
    rng = np.random.default_rng(0)
    feats = (rng.standard_normal((T, D)) * 1.3).astype(np.float32)
    model_np = synth.random_model(G, M, D, 777)
    for p in args:
        res, chk = run(os.path.abspath(p), feats, model_np, T, G, M, D, int(os.environ.get("SRX_REPS", "5")), score_path)
        print("%-28s %s checksum %.4f" % (os.path.basename(p), " ".join("%s %.3f ms" % kv for kv in res.items()), chk), flush=True)


if __name__ == "__main__":
    main()
