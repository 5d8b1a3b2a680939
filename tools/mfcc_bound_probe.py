#!/usr/bin/env python3
"""Measures the constant K of the element-wise MFCC bound |dc_j| <= 1e-4 |c_j| + K 2^-24 sum_i |cos(j i pi / n)| (test_gpu_mfcc.py)
for every path and both windows on the parity corpus.  Development tool (uses the test oracle): python tools/mfcc_bound_probe.py"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402
from speechrecognition_amd import host, synth  # noqa: E402


def main():
    O.build()
    ctx = host.Context(0)
    out = []
    for normalize in (True, False):
        for wl in (320, 400):
            lens = [16000, 23456, 5000]
            off = np.zeros(len(lens) + 1, np.int64)
            off[1:] = np.cumsum(lens)
            x = np.concatenate([synth.utterance_audio(100 + i, n) for i, n in enumerate(lens)])
            x[:16000] = (0.5 * np.random.default_rng(1).standard_normal(16000)).clip(-1, 1)
            nf = 48
            fb, _ = host.mel_filterbank(7200, nf, wl, 16000)
            refs = []
            for u in range(len(lens)):
                xu = x[off[u]:off[u + 1]]
                if normalize:
                    xu = O.normalize_volume(xu)
                refs.append(O.write_features(xu, wl, 160, fb, 13)[0])
            ref = np.concatenate(refs)
            cosabs = np.abs(np.cos(np.arange(13)[:, None] * np.arange(nf)[None, :] * np.pi / nf)).sum(axis=1)
            for name, path in (("f16x3", host.MFCC_F16X3), ("f32", host.MFCC_F32), ("fft", host.MFCC_FFT)):
                plan = host.MfccPlan(ctx, host.mfcc_config(wl, 160, normalize=normalize), fb, path)
                got, _ = plan.write_features(x, off)
                d = np.abs(got.astype(np.float64) - ref)
                k = (d - 1e-4 * np.abs(ref)) / (2.0 ** -24 * cosabs[None, :])
                out.append({"normalize": normalize, "window": wl, "path": name, "K_needed_max": float(k.max()),
                            "K_needed_p999": float(np.quantile(k, 0.999)), "max_abs_c": float(np.abs(ref).max())})
                plan.close()
    print(json.dumps(out, indent=1))
    ctx.close()


if __name__ == "__main__":
    main()
