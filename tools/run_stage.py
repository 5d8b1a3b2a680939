#!/usr/bin/env python3
"""Runs one stage of the pipeline a few times (for rocprofv3 --pmc passes): gmm | mfcc | viterbi."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA, host, synth  # noqa: E402

stage = sys.argv[1] if len(sys.argv) > 1 else "gmm"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
T = int(sys.argv[3]) if len(sys.argv) > 3 else 499674
ctx = host.Context(0)
rng = np.random.default_rng(0)
if stage == "gmm":
    G, M, D = 2000, 16, 39
    model = host.AcousticModel(ctx, G, M, D)
    model.set(*synth.random_model(G, M, D, 777))
    feats = ctx.array((rng.standard_normal((T, D)) * 1.3).astype(np.float32))
    out = ctx.empty((G, T), np.float32)
    for _ in range(reps):
        model.score_dev(feats, D, T, out, 1, T, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA)
    ctx.sync()
elif stage == "mfcc":
    x, soff = synth.corpus_audio(1000, 1)
    fb, _ = host.mel_filterbank(7200, 48, 400, 16000)
    cfg = host.mfcc_config(400, 160, 13)
    plan = host.MfccPlan(ctx, cfg, fb)
    foff = host.frame_offsets(cfg, soff)
    Tn = int(foff[-1])
    xd, sd, fd = ctx.array(x), ctx.array(soff), ctx.array(foff)
    sc = ctx.empty((1000,), np.float32)
    ceps = ctx.empty((Tn, 13), np.float32)
    for _ in range(reps):
        plan.write_features_dev(xd, sd, 1000, fd, Tn, sc, ceps)
    ctx.sync()
print("done", stage)
