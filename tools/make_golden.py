#!/usr/bin/env python3
"""Writes tests/golden/oracle_vectors.npz: small input/output vectors of every stage of the hot path.

PROVENANCE: the outputs are produced by the CPU oracle (oracle/srx_oracle.c), NOT by the reference --
the reference cannot be compiled in this image (it needs OpenFrameworks' ofMain.h, absent) and holds no
fixtures of its own.  The vectors freeze the oracle's behaviour so that drift (compiler, libm, box) is
detected and so that the GPU parity tests have fixed known answers; they do not pin parity with the
reference.  Inputs come from the 32-bit LCG of SURVEY.md section 8d, so they can be regenerated anywhere.

  python tools/make_golden.py        # (re)writes the file
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tests", "golden", "oracle_vectors.npz")


def lcg_uniform(seed, n):
    """s = s*1664525 + 1013904223 (mod 2^32); u = ((s >> 8) & 0xFFFFFF) / 2^24."""
    out = np.empty(n, np.float64)
    s = seed & 0xFFFFFFFF
    for i in range(n):
        s = (s * 1664525 + 1013904223) & 0xFFFFFFFF
        out[i] = ((s >> 8) & 0xFFFFFF) / float(1 << 24)
    return out


def lcg_normal(seed, n):
    u = lcg_uniform(seed, 2 * n).reshape(2, n)
    return np.sqrt(-2.0 * np.log(np.maximum(u[0], 2.0 ** -24))) * np.cos(2 * np.pi * u[1])


def config0_wav(n=80000, seed=12345):
    """BASELINE configs[0]: 0.3 sin(2 pi 440 t) + 0.2 sin(2 pi 1800 t) + 0.1 (u - 1/2), 16-bit PCM round trip
    (s / 32768.0f, src/wav.cpp:41-44)."""
    t = np.arange(n) / 16000.0
    x = 0.3 * np.sin(2 * np.pi * 440 * t) + 0.2 * np.sin(2 * np.pi * 1800 * t) + 0.1 * (lcg_uniform(seed, n) - 0.5)
    return (np.round(x * 32767).astype(np.int16).astype(np.float32) / np.float32(32768.0)).astype(np.float32)


PHONES_HELLO_WORLD = np.array([0, 16, 3, 21, 25, 0, 36, 12, 21, 9, 0], np.int32)  # SIL HH AH L OW SIL W ER L D SIL


def model_ref(G=40, M=1, D=13, seed=4242):
    mean = lcg_normal(seed, G * M * D).reshape(G, M, D) * 2.0
    var = 0.5 + lcg_uniform(seed + 1, G * M * D).reshape(G, M, D)
    w = (0.5 + lcg_uniform(seed + 2, G * M)).reshape(G, M)
    w = (w / w.sum(axis=1, keepdims=True)).astype(np.float32)
    return mean, var, w


def generate(O):
    v = {}
    # ---- filterbanks
    for name, args in (("fb320", (7200.0, 48, 320, 16000)), ("fb400", (7200.0, 48, 400, 16000)),
                       ("fb768", (7200.0, 48, 768, 44100))):
        fb, bm = O.mel_filterbank(*args)
        v[name] = np.stack([fb["center"], fb["width"], fb["height"]], axis=1).astype(np.float32)
        v[name + "_bin_max"] = np.array([bm], np.int32)
    fb320, _ = O.mel_filterbank(7200.0, 48, 320, 16000)
    fb400, _ = O.mel_filterbank(7200.0, 48, 400, 16000)
    # ---- MFCC: configs[0] clip at the reference's settings; short clips at the edge lengths
    x = config0_wav()
    xn = O.normalize_volume(x, 1.0)
    ceps, xpre = O.write_features(xn, 320, 160, fb320, 13)
    v["cfg0_ceps"] = ceps
    v["cfg0_preemph_head"] = xpre[:64].copy()
    v["cfg0_feats"] = O.read_features(ceps, 2)
    v["cfg0_spec_frame7"] = O.partial_dft(xpre[7 * 160:7 * 160 + 320], 320, 144)
    for L in (321, 481, 640):
        xs = (lcg_uniform(L, L) - 0.5).astype(np.float32)
        c, _ = O.write_features(O.normalize_volume(xs, 1.0), 320, 160, fb320, 13)
        v["short%d_ceps" % L] = c
    xs = (lcg_uniform(99, 4000) - 0.5).astype(np.float32)
    v["wl400_ceps"], _ = O.write_features(O.normalize_volume(xs, 1.0), 400, 160, fb400, 13)
    # ---- GMM scores: reference-native shape (G=40, M=1, D=13) and a mixture shape
    feats = v["cfg0_feats"][:128]
    mean, var, w = model_ref()
    # put the model where the features live so scores are not all hugely negative
    mu_f = feats[:, :13].mean(axis=0)
    sd_f = feats[:, :13].std(axis=0) + 0.1
    mean = mean * sd_f + mu_f
    var = var * sd_f ** 2
    v["gmm_mean"], v["gmm_var"], v["gmm_w"] = mean, var, w
    lw = O.set_wt(w)
    emis = O.gmm_score(feats, mean, var, lw, 13, 0)
    v["gmm_scores_M1"] = emis
    m4, v4, w4 = model_ref(6, 4, 39, 777)
    f39 = (lcg_normal(31, 64 * 39).reshape(64, 39) * 1.3).astype(np.float32)
    v["gmm4_mean"], v["gmm4_var"], v["gmm4_w"], v["gmm4_feats"] = m4, v4, w4, f39
    v["gmm4_scores_lse"] = O.gmm_score(f39, m4, v4, O.set_wt(w4), 39, 1)
    v["gmm4_scores_c0"] = O.gmm_score(f39, m4, v4, O.set_wt(w4), 39, 0)
    # ---- Viterbi on the hello-world HMM over the first 128 frames, emissions = the scores above
    ntr, tr, ctx = O.build_training_hmm(PHONES_HELLO_WORLD)
    v["hmm_ntr"], v["hmm_tr"], v["hmm_ctx"] = ntr, tr, ctx
    r = O.viterbi(emis, PHONES_HELLO_WORLD, ntr, tr)
    assert r["status"] == 0
    v["vit_path"] = r["path"]
    v["vit_final"] = np.array([r["final"], r["prevp"]], np.float32)
    v["vit_nties"] = np.array([r["nties"]], np.int64)
    v["vit_tr_after"] = O.train_transitions(r["path"], ntr, tr)
    # ---- one EM statistics half-iteration from that alignment
    gid = PHONES_HELLO_WORLD[r["path"]].astype(np.int32)
    Bm, Bv, Bw, Blw, zs = O.em_statistics(feats, gid, 13, mean, var, w, 0.2)
    v["em_gid"] = gid
    v["em_mean"] = np.asarray(Bm, np.float64)
    v["em_var"] = np.asarray(Bv, np.float64)
    v["em_w"] = Bw
    return v


def main():
    sys.path.insert(0, ROOT)
    from oracle import oracle as O
    O.build()
    v = generate(O)
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    np.savez_compressed(OUT, **v)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", len(v), "arrays")


if __name__ == "__main__":
    main()
