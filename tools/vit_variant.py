#!/usr/bin/env python3
"""Kernel A/B helper: times k_viterbi of one or more builds of libsrx (paths on the command line) on the bench shape
(1,000 utterances, dense state-major emissions) and on 10,000 utterances with compact emissions.  Development tool.
  python tools/vit_variant.py lib1.so [lib2.so ...]"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

torch.cuda.init()
torch.manual_seed(5)
from speechrecognition_amd import _capi, synth  # noqa: E402


def workload(n, seed):
    rng = np.random.default_rng(seed)
    ntrs, trs, tieds, Ts = [], [], [], []
    from speechrecognition_amd import host
    for _ in range(n):
        ph = synth.phone_string(rng, int(rng.integers(4, 13)))
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr); trs.append(tr)
        tieds.append(rng.integers(0, 2000, len(ph)).astype(np.int32))
        Ts.append(int(rng.integers(200, 801)))
    fo = np.zeros(n + 1, np.int64)
    fo[1:] = np.cumsum(Ts)
    return fo, ntrs, trs, tieds


def run(path, reps=10):
    _capi._lib = None
    _capi.LIB_PATH = path
    from speechrecognition_amd import host
    ctx = host.Context(0)
    res = {}
    G = 2000
    for name, n in (("dense_1k", 1000), ("compact_10k", 10000)):
        fo, ntrs, trs, tieds = workload(n, 4)
        T = int(fo[-1])
        hb = host.HmmBatch(ctx, fo, ntrs, trs, tieds)
        path_d = ctx.zeros((T,), np.int32)
        fin = ctx.zeros((n,), np.float32)
        st = ctx.zeros((n,), np.int32)
        if name.startswith("dense"):
            Tp = (T + 3) // 4 * 4
            torch.manual_seed(5)
            e = torch.randn((G, Tp), device="cuda", dtype=torch.float32) * 3 - 60
            ep = host.DevArray.wrap(ctx, e.data_ptr(), (G, Tp), np.float32, owner=e)
            fn = lambda: hb.viterbi(ep, 1, Tp, 1e10, path_d, fin, st)
        else:
            hb.align_plan()
            cnt = hb.emis_count
            ep = ctx.array((np.random.default_rng(1).standard_normal(cnt, dtype=np.float32) * 3 - 60))
            fn = lambda: hb.viterbi_align(ep, 1e10, path_d, fin, st)
        fn(); ctx.sync()
        ctx.profile_enable(True)
        for _ in range(reps):
            fn()
        ms, k = ctx.profile_query("k_viterbi")
        ctx.profile_enable(False)
        res[name] = ms / k
        if os.environ.get("SRX_VIT_TIMING"):
            pth = path_d.get()
            Ts = np.diff(fo)
            tk = np.stack([pth[fo[:-1] + i] for i in range(3)], 1).astype(np.float64) * 0.01   # us at 100 MHz
            u = int(np.argmax(Ts))
            res[name + "_phases_us_longest(T=%d)" % Ts[u]] = tk[u].round(1).tolist()
            res[name + "_phases_us_mean"] = tk.mean(0).round(1).tolist()
            res[name + "_us_per_frame_loop_bt"] = [float((tk[:, 1] / Ts).mean().round(4)), float((tk[:, 2] / Ts).mean().round(4))]
        res[name + "_chk"] = int(path_d.get()[::101].sum()) + int(st.get().sum()) * 1000003
        hb.close()
    ctx.close()
    return res


if __name__ == "__main__":
    for p in sys.argv[1:]:
        print(os.path.basename(p), run(os.path.abspath(p)), flush=True)
