#!/usr/bin/env python3
"""Kernel A/B helper: times the split-f16 MFCC kernel of one or more builds of libsrx (paths on the command line) on the
bench shape (1,000 utterances, 400/160), in one process, and prints a checksum.  Development tool only."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import _capi, synth  # noqa: E402


def run(path, x, soff, reps):
    _capi._lib = None
    _capi.LIB_PATH = path
    from speechrecognition_amd import host
    ctx = host.Context(0)
    fb, _ = host.mel_filterbank(7200.0, 48, 400, 16000)
    cfg = host.mfcc_config(400, 160, 13)
    plan = host.MfccPlan(ctx, cfg, fb, host.MFCC_F16X3)
    foff = host.frame_offsets(cfg, soff)
    T, n = int(foff[-1]), len(soff) - 1
    x_d, soff_d, foff_d = ctx.array(x), ctx.array(soff), ctx.array(foff)
    scale_d, ceps_d = ctx.empty((n,), np.float32), ctx.empty((T, 13), np.float32)
    plan.write_features_dev(x_d, soff_d, n, foff_d, T, scale_d, ceps_d)
    ctx.sync()
    ctx.profile_enable(True)
    for _ in range(reps):
        plan.write_features_dev(x_d, soff_d, n, foff_d, T, scale_d, ceps_d)
    ms, k = ctx.profile_query("k_mfcc_f16")
    chk = float(np.abs(ceps_d.get()).sum())
    ctx.close()
    return ms / k, chk


def main():
    x, soff = synth.corpus_audio(1000, 1)
    for p in sys.argv[1:]:
        ms, chk = run(os.path.abspath(p), x, soff, 10)
        print("%-32s k_mfcc_f16 %.4f ms  checksum %.6f" % (os.path.basename(p), ms, chk), flush=True)


if __name__ == "__main__":
    main()
