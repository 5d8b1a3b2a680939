#!/usr/bin/env python3
"""Measures how far the MFCC paths are from an ELEMENT-WISE 1e-4 relative bar (the tests hold the bar per frame,
against the frame's largest |cepstrum|): distribution of |got - ref| / max(|ref|, floor) for several floors.
  python tools/mfcc_elementwise.py > gpurun_out/mfcc_elementwise.json"""
import json
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    import torch
    torch.cuda.init()
    from oracle import oracle as O
    from speechrecognition_amd import host, synth
    O.build()
    ctx = host.Context(0)
    out = {}
    for wl in (320, 400):
        lens = [16000, 23456, 40000, 5000]
        off = np.zeros(len(lens) + 1, np.int64)
        off[1:] = np.cumsum(lens)
        x = np.concatenate([synth.utterance_audio(100 + i, n) for i, n in enumerate(lens)])
        rng = np.random.default_rng(1)
        x[off[0]:off[1]] = (0.5 * rng.standard_normal(16000)).clip(-1, 1)
        fb, _ = host.mel_filterbank(7200, 48, wl, 16000)
        ref = np.concatenate([O.write_features(O.normalize_volume(x[off[u]:off[u + 1]]), wl, 160, fb, 13)[0]
                              for u in range(len(lens))])
        scale = np.abs(ref).max(axis=1, keepdims=True)
        for name, path in (("f16x3", host.MFCC_F16X3), ("f32", host.MFCC_F32)):
            plan = host.MfccPlan(ctx, host.mfcc_config(wl, 160), fb, path)
            got, _ = plan.write_features(x, off)
            d = np.abs(got - ref)
            rec = {"frames": int(len(ref)), "per_frame_scale_max": float((d / scale).max()),
                   "abs_err_max": float(d.max()), "abs_ref_median": float(np.median(np.abs(ref))),
                   "abs_ref_min": float(np.abs(ref).min())}
            for fl_name, floor in (("pure_relative", np.full_like(ref, 1e-30)), ("floor_1e-3_of_frame_scale", 1e-3 * scale),
                                   ("floor_1e-2_of_frame_scale", 1e-2 * scale), ("floor_abs_1e-2", np.full_like(ref, 1e-2)),
                                   ("floor_abs_1e-1", np.full_like(ref, 1e-1))):
                r = d / np.maximum(np.abs(ref), floor)
                rec[fl_name] = {"max": float(r.max()), "p99.9": float(np.quantile(r, 0.999)), "median": float(np.median(r)),
                                "frac_within_1e-4": float((r <= 1e-4).mean())}
            out[f"wl{wl}_{name}"] = rec
            plan.close()
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
