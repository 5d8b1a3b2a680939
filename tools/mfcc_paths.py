#!/usr/bin/env python3
"""Times the MFCC paths (split-f16 matrix cores, f32 matrix cores, LDS FFT) of one or more builds of libsrx on the bench
shape (1,000 utterances, 400/160 and 320/160), interleaved in one process, with the deviation of each from the f32 path.
  python tools/mfcc_paths.py [libsrx.so ...]     Development / measurement tool."""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import _capi, synth  # noqa: E402


def run(lib, x, soff, wl, reps, rounds):
    _capi._lib = None
    _capi.LIB_PATH = lib
    from speechrecognition_amd import host
    ctx = host.Context(0)
    fb, _ = host.mel_filterbank(7200.0, 48, wl, 16000)
    cfg = host.mfcc_config(wl, 160, 13)
    foff = host.frame_offsets(cfg, soff)
    T, n = int(foff[-1]), len(soff) - 1
    x_d, soff_d, foff_d = ctx.array(x), ctx.array(soff), ctx.array(foff)
    scale_d, ceps_d = ctx.empty((n,), np.float32), ctx.empty((T, 13), np.float32)
    paths = {"f16x3": (host.MFCC_F16X3, "k_mfcc_f16"), "fft": (host.MFCC_FFT, "k_mfcc_fft"), "f32": (host.MFCC_F32, "k_mfcc_tile")}
    plans = {k: host.MfccPlan(ctx, cfg, fb, v[0]) for k, v in paths.items()}
    out, ref = {}, None
    for k, p in plans.items():
        p.write_features_dev(x_d, soff_d, n, foff_d, T, scale_d, ceps_d)
        ctx.sync()
        c = ceps_d.get()
        if k == "f32":
            ref = c
        out[k] = {"ceps": c, "ms": []}
    for _ in range(rounds):
        for k, p in plans.items():
            ctx.profile_enable(True)
            for _ in range(reps):
                p.write_features_dev(x_d, soff_d, n, foff_d, T, scale_d, ceps_d)
            ms, cnt = ctx.profile_query(paths[k][1])
            ctx.profile_enable(False)
            out[k]["ms"].append(ms / cnt)
    res = {"frames": T, "window": wl}
    sc = np.abs(ref).max(axis=1, keepdims=True)
    for k in plans:
        res[k] = {"ms_median": float(np.median(out[k]["ms"])), "ms_min": float(np.min(out[k]["ms"])),
                  "max_frame_rel_dev_from_f32_path": float((np.abs(out[k]["ceps"] - ref) / sc).max())}
    for p in plans.values():
        p.close()
    ctx.close()
    return res


def main():
    libs = sys.argv[1:] or [_capi.LIB_PATH]
    x, soff = synth.corpus_audio(1000, 1)
    for lib in libs:
        for wl in (400, 320):
            print(os.path.basename(os.path.dirname(os.path.abspath(lib))), json.dumps(run(os.path.abspath(lib), x, soff, wl, 10, 3)), flush=True)


if __name__ == "__main__":
    main()
