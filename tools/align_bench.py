#!/usr/bin/env python3
"""Times BASELINE configs 3/4 shapes on one GPU: forced alignment of N utterances (dense vs alignment-mode emission
scoring) and one embedded-training iteration (parallel.TrainingShard).  Development / measurement tool.
  python tools/align_bench.py [--utts 10000] [--reps 3] [--no-dense]"""
import argparse
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--utts", type=int, default=10000)
    ap.add_argument("--reps", type=int, default=3)
    ap.add_argument("--no-dense", action="store_true")
    ap.add_argument("--em-path", default="faithful,fold,fused", help="statistics paths to time (comma separated)")
    args = ap.parse_args()
    import torch
    torch.cuda.init()
    from speechrecognition_amd import _capi
    if os.environ.get("SRX_LIB"):
        _capi.LIB_PATH = os.path.abspath(os.environ["SRX_LIB"])   # A/B a variant build
    from speechrecognition_amd import SRX_MIX_LOGSUMEXP, host, parallel, synth
    from speechrecognition_amd._capi import SRX_SCORE_MFMA_F16X3
    G, M, D = 2000, 16, 39
    ctx = host.Context(0)
    rng = np.random.default_rng(4)
    ntrs, trs, tieds, Ts = [], [], [], []
    for _ in range(args.utts):
        ph = synth.phone_string(rng, int(rng.integers(4, 13)))
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr); trs.append(tr)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
        Ts.append(int(rng.integers(200, 801)))
    fo = np.zeros(args.utts + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    T = int(fo[-1])
    mean, var, w = synth.random_model(G, M, D, 777)
    feats = np.empty((T, D), np.float32)
    for a in range(0, T, 1 << 20):
        b = min(T, a + (1 << 20))
        feats[a:b] = rng.standard_normal((b - a, D), dtype=np.float32) * np.float32(1.3)
    A = host.AcousticModel(ctx, G, M, D); A.set(mean, var, w)
    B = host.AcousticModel(ctx, G, M, D); B.copy_from(A)
    hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
    feats_d = ctx.array(feats)
    mean_S = float(np.mean([len(t) for t in tieds]))
    out = {"utterances": args.utts, "frames": T, "mean_states": mean_S, "G": G, "M": M, "D": D}

    def timed(fn):
        fn(); ctx.sync()
        t0 = time.perf_counter()
        for _ in range(args.reps):
            fn()
        ctx.sync()
        return (time.perf_counter() - t0) / args.reps * 1e3

    ctx.profile_enable(True)
    emis_c = ctx.empty((hb.align_plan() and hb.emis_count,), np.float32)
    path_d = ctx.zeros((T,), np.int32); final_d = ctx.zeros((args.utts,), np.float32); status_d = ctx.zeros((args.utts,), np.int32)
    out["align_mode_ms"] = timed(lambda: (hb.score_align(A, feats_d, D, emis_c, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA_F16X3),
                                          hb.viterbi_align(emis_c, 1e10, path_d, final_d, status_d)))
    out["emis_compact_GB"] = hb.emis_count * 4 / 1e9
    kern = {}
    for k in ("k_gmm_f16_align", "k_viterbi", "k_gmm_fixup16_align"):
        ms, n = ctx.profile_query(k)
        if n:
            kern[k] = ms / n
    out["align_mode_kernels_ms"] = kern
    flop = 2.0 * (2 * D + 1) * M * sum(len(t) * (fo[u + 1] - fo[u]) for u, t in enumerate(tieds))
    out["align_scoring_algorithmic_TFLOPs"] = flop / (kern["k_gmm_f16_align"] * 1e-3) / 1e12
    assert (status_d.get() == 0).all()
    if not args.no_dense:
        emis = ctx.empty((G, T), np.float32)
        out["dense_ms"] = timed(lambda: (A.score_dev(feats_d, D, T, emis, 1, T, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA_F16X3),
                                         hb.viterbi(emis, 1, T, 1e10, path_d, final_d, status_d)))
        out["emis_dense_GB"] = G * T * 4 / 1e9
        del emis
    ctx.profile_enable(False)
    for em_path in args.em_path.split(","):
        shard = parallel.TrainingShard(ctx, hb, feats_d, D, G, M, D, mix_mode=SRX_MIX_LOGSUMEXP, score_path=SRX_SCORE_MFMA_F16X3,
                                       emissions="compact", em_path=em_path)
        shard.reuse_scores = False   # one model object, never rewritten here: time the iteration WITH its scoring pass
        shard.iteration(A, A, B, 0.2)   # first launches load their code objects: keep that out of the event pairs
        ctx.sync()
        ctx.profile_enable(True)
        ms_it = timed(lambda: shard.iteration(A, A, B, 0.2))
        kern = {}
        for k in ("k_gmm_f16_align", "k_viterbi", "k_train_transitions", "k_gather_gid", "k_fill_f32", "k_em_logp", "k_em_accum1",
                  "k_em_means", "k_em_accum2", "k_em_finish", "k_em_csort_hist", "k_em_csort_scan", "k_em_csort_scatter",
                  "k_em_item_scan", "k_em_logp_seg", "k_em_logp_skipped", "k_em_accum_seg1", "k_em_accum_seg2", "k_em_seg_reduce",
                  "k_em_fused", "k_em_fused_finish", "k_model_lconst"):
            ms, n = ctx.profile_query(k)
            if n:
                kern[k] = ms / (args.reps + 1)     # per iteration (timed() runs one untimed pass first)
        ctx.profile_enable(False)
        out["training_" + em_path] = {"iteration_ms": ms_it, "frames_per_s": T / (ms_it * 1e-3), "kernels_ms_per_iteration": kern,
                                       "statistics_ms": sum(v for k, v in kern.items() if k.startswith(("k_em_", "k_fill", "k_model_lconst"))),
                                       "fallbacks": shard.fallbacks}
        del shard
    out["alignment_frames_per_s"] = T / (out["align_mode_ms"] * 1e-3)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
