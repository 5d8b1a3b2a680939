#!/usr/bin/env python3
"""Times hmm::viterbi (src/hmm.cpp:150-224) on a lexicon HMM of the size src/main.cpp:58 builds (133,860 words, about 10^6
states; synthetic words with the length distribution of res/lexicon/cmu.txt) for one 5 s utterance (500 frames), through the
frontier sweep of srx_viterbi_graph_dev, at several beams.  Development / measurement tool:  python tools/decode_bench.py"""
import json
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from speechrecognition_amd import host  # noqa: E402

G = 40


def main():
    n_words, T = 133860, 500
    rng = np.random.default_rng(1)
    lens = np.clip(rng.poisson(5.4, n_words) + 1, 1, 20)
    words = [rng.integers(1, 40, int(k)).astype(np.int32) for k in lens]
    counts = rng.integers(1, 5000, n_words)
    uni = (np.log(counts) - np.float32(np.log(counts.sum() * 3.0))).astype(np.float32)
    a = host.build_lexicon_hmm(words, uni)
    tied = a["ctx"][:, 1].copy()
    ctx = host.Context(0)
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], tied)
    rows = []
    while sum(len(r) for r in rows) < T:          # frames that spell words from the head of the lexicon
        w = int(rng.integers(0, 8))
        for ph in list(words[w]) + [0]:
            e = rng.standard_normal((int(rng.integers(3, 9)), G)).astype(np.float32) * 2 - 12
            e[:, ph] += 9.0
            rows.append(e)
    emis = np.concatenate(rows)[:T]
    emis_d = ctx.array(emis)
    fo = np.array([0, T], np.int64)
    out = {"what": "one 500-frame utterance on a 133,860-word lexicon HMM, frontier sweep, MI355X", "n_states": int(a["n_states"]),
           "n_transitions": int(a["tr_off"][-1]), "dense_sweep_backpointer_bytes": int(4 * T * a["n_states"]), "runs": []}
    for beam in (200.0, 1000.0, 1e10):
        hmm.set_sweep(host.DECODE_FRONTIER, 8 << 30)
        hmm.viterbi(emis_d, G, 1, fo, beam, 1)
        ctx.profile_enable(True)
        t0 = time.perf_counter()
        path, final, fstate, status = hmm.viterbi(emis_d, G, 1, fo, beam, 1)
        wall = time.perf_counter() - t0
        ms, cnt = ctx.profile_query("k_viterbi_frontier")
        ctx.profile_enable(False)
        out["runs"].append({"beam": beam, "status": int(status[0]), "device_ms": ms / max(cnt, 1), "wall_ms_with_copies": wall * 1e3,
                            "final_logp": float(final[0]), "distinct_states_on_path": int(len(set(path.tolist())))})
    print(json.dumps(out))
    hmm.close()
    ctx.close()


if __name__ == "__main__":
    main()
