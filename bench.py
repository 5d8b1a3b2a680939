#!/usr/bin/env python3
"""Benchmark of the MI355X acoustic pipeline (BASELINE.json metric: audio frames/sec through
MFCC + GMM scoring + Viterbi forced alignment).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...   (driver, N > 1)

One step = one pass of the whole hot path over one resident batch: BASELINE.json configs[1]'s corpus
(1,000 synthetic Voxforge-shaped utterances, 2-8 s, 16 kHz) at the BASELINE shapes (25 ms / 10 ms
frames, 13 cepstra + deltas = 39 dims, 2,000 tied states x 16 diagonal Gaussians), i.e.
   samples -> MFCC -> smoothing + deltas -> dense GMM log-likelihoods [T x 2000] -> Viterbi alignment.
Inputs are resident in HBM before the timed region.  With N > 1 every rank owns its own 1,000
utterances (weak scaling, no data-path collective: utterances are independent).

Rank 0 prints ONE JSON line; `roofline` describes the dominant kernel (GMM scoring, f32 MFMA) timed
with events on the launch stream; `cpu_baseline` times the CPU oracle (a port of the reference's
arithmetic, test infrastructure) on a bounded sample of the same workload on this host.
"""
import argparse
import json
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WL, HOP, NCEPS, NFILT, FEND, SPS = 400, 160, 13, 48, 7200.0, 16000
G, M, D = 2000, 16, 39
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: f32-input MFMA (= vector rate)
PEAK_F16_MFMA_TFLOPS = 2500.0  # same guide: dense BF16/F16 MFMA, ~2.5 PFLOP/s
PEAK_HBM_GBS = 8000.0


_T0 = time.perf_counter()


def progress(msg):
    """One line on stderr per stage (rank 0): the JSON line on stdout stays the only thing there."""
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench %6.1f s] %s" % (time.perf_counter() - _T0, msg), file=sys.stderr, flush=True)


def build_workload(n_utt, seed):
    from speechrecognition_amd import host, synth
    x, soff = synth.corpus_audio(n_utt, seed)
    rng = np.random.default_rng(seed + 99)
    ntrs, trs, tieds = [], [], []
    for _ in range(n_utt):
        ph = synth.phone_string(rng)
        ntr, tr, _ = host.build_training_hmm(ph)
        ntrs.append(ntr)
        trs.append(tr)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
    return x, soff, ntrs, trs, tieds


def pmc_traffic(kernel, T):
    """HBM bytes per launch of `kernel` from the committed PMC pass (profiles/pmc_traffic.json, written by
    tools/pmc_traffic.py from separate `rocprofv3 --pmc` runs of this same workload; the counters cannot be
    collected from inside this process).  None when no pass is on record for this frame count."""
    try:
        with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
            rec = json.load(f).get(kernel)
        if rec and int(rec.get("frames", -1)) == int(T):
            return rec["hbm_bytes_per_launch"]
    except (OSError, ValueError):
        pass
    return None


def cpu_baseline(x, soff, ntrs, trs, tieds, model, budget_s=20.0):
    """Times the oracle (kind "port") end to end on a bounded sample of the same utterances."""
    from oracle import oracle as O
    O.build()
    mean, var, w = model
    lw = O.set_wt(w)
    fb, _ = O.mel_filterbank(FEND, NFILT, WL, SPS)
    mean_l = np.ascontiguousarray(mean, np.longdouble)
    var_l = np.ascontiguousarray(var, np.longdouble)
    # threads = this job's CPU share: the GPU box exports OMP_NUM_THREADS = 16 for a one-GPU lease while os.cpu_count()
    # reports every core of the host (SURVEY 8d asks for hardware_concurrency(); a pool that size is not this job's to use)
    threads = int(min(os.cpu_count() or 1, int(os.environ.get("OMP_NUM_THREADS", "16"))))
    cpu_model = "unknown"
    try:
        with open("/proc/cpuinfo") as f:
            for ln in f:
                if ln.startswith("model name"):
                    cpu_model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    # calibrate on one short utterance, single thread
    order = np.argsort(np.diff(soff))
    u0 = int(order[0])
    t0 = time.perf_counter()
    n0, _, _ = O.pipeline_utterance(x[soff[u0]:soff[u0 + 1]], WL, HOP, fb, NCEPS, 0.96, 2, mean_l, var_l, lw, D, 1,
                                    tieds[u0], ntrs[u0], trs[u0])
    dt0 = time.perf_counter() - t0
    per_frame_state = dt0 / (n0 * len(ntrs[u0]))
    # pick utterances (shortest first) until ~budget_s of wall time across the thread pool
    chosen, est = [], 0.0
    for u in order[1:]:
        u = int(u)
        cost = per_frame_state * ((soff[u + 1] - soff[u]) / HOP) * len(ntrs[u])
        if chosen and (est + cost) / threads > budget_s:
            break
        chosen.append(u)
        est += cost
        if len(chosen) >= 40 * threads:
            break
    frames = [0] * len(chosen)

    def work(k0):
        for k in range(k0, len(chosen), threads):
            u = chosen[k]
            n, _, _ = O.pipeline_utterance(x[soff[u]:soff[u + 1]], WL, HOP, fb, NCEPS, 0.96, 2, mean_l, var_l, lw, D, 1,
                                           tieds[u], ntrs[u], trs[u])
            frames[k] = n

    t0 = time.perf_counter()
    ths = [threading.Thread(target=work, args=(i,)) for i in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt = time.perf_counter() - t0
    # The pipeline above scores each utterance's own ~40 states (what the reference does, lazily inside Viterbi); the GPU
    # step scores all G states of every frame.  Like for like: the oracle's dense scoring (long-double Gaussians, all
    # G x M components) on a small sample of frames, same threads -> the CPU rate of the step the GPU runs.
    nd = 16 * threads
    fd = np.zeros((nd, 39), np.float32)
    fd[:, :D] = np.random.default_rng(3).standard_normal((nd, D)).astype(np.float32) * 1.3
    part = np.array_split(np.arange(nd), threads)

    def dense(k):
        if len(part[k]):
            O.gmm_score(fd[part[k]], mean_l, var_l, lw, D, 1)

    t1 = time.perf_counter()
    ths = [threading.Thread(target=dense, args=(i,)) for i in range(threads)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    dt_dense = time.perf_counter() - t1
    per_frame_pipeline = dt / max(sum(frames), 1)
    per_frame_dense = dt_dense / nd
    return {
        "like_for_like": {"value": 1.0 / (per_frame_pipeline + per_frame_dense), "unit": "frames/s",
                          "what": f"same pipeline with ALL {G} states x {M} components scored per frame, as the GPU step does: dense "
                                  f"scoring timed on {nd} frames ({dt_dense:.1f} s, {1.0 / per_frame_dense:.0f} frames/s on {threads} threads) "
                                  "added to the per-frame time of the pipeline above"},
        "value": sum(frames) / dt, "unit": "frames/s", "cores": threads, "kind": "port",
        "cpu_model": cpu_model, "host_logical_cpus": os.cpu_count(), "threads_used": threads,
        "single_thread": {"value": n0 / dt0, "unit": "frames/s", "what": "the same pipeline on one utterance, one thread (calibration run)"},
        "sample": f"{len(chosen)} shortest of the 1000 utterances ({sum(frames)} frames) in {dt:.1f} s; oracle/srx_oracle.c "
                  f"(naive half-window DFT in f64, long-double Gaussians of the utterance's own states only, "
                  f"accumulating Viterbi); single-thread calibration {n0 / dt0:.0f} frames/s",
    }


def alignment_corpus(n_utt, seed):
    """BASELINE configs 3/4: T ~ U{200..800} frames, 4-12 words x 2-6 phones with SIL between, tied ids
    uniform in [0, G).  Deterministic in (n_utt, seed): every rank generates the same descriptors and keeps its shard."""
    from speechrecognition_amd import synth
    rng = np.random.default_rng(seed)
    phones, tieds, Ts = [], [], np.empty(n_utt, np.int64)
    for u in range(n_utt):
        ph = synth.phone_string(rng, int(rng.integers(4, 13)))
        phones.append(ph)
        tieds.append(rng.integers(0, G, len(ph)).astype(np.int32))
        Ts[u] = int(rng.integers(200, 801))
    return phones, tieds, Ts


def build_shard(ctx, torch, host, parallel, phones, tieds, Ts, rank, world, seed):
    """This rank's share of the corpus (greedy balance of frames x states), resident on its GPU: HMM batch + features
    ~ N(0, 1.3^2) generated on the device (config 3's features)."""
    costs = Ts * np.array([len(p) for p in phones], np.int64)
    mine = parallel.shard_utterances(costs, world)[rank]
    hm = [host.build_training_hmm(phones[u]) for u in mine]
    fo = np.zeros(len(mine) + 1, np.int64)
    fo[1:] = np.cumsum(Ts[mine])
    hb = host.HmmBatch(ctx, fo, [h[0] for h in hm], [h[1] for h in hm], [tieds[u] for u in mine])
    T = int(fo[-1])
    gen = torch.Generator(device="cuda")
    gen.manual_seed(seed + rank)
    feats_t = torch.randn((max(T, 1), D), device="cuda", generator=gen) * 1.3
    feats_d = host.DevArray.wrap(ctx, feats_t.data_ptr(), (max(T, 1), D), np.float32, feats_t)
    flop = 2.0 * (2 * D + 1) * M * float(sum(len(phones[u]) * Ts[u] for u in mine))
    return hb, feats_d, T, flop


def reduce_max_sum(torch, dist, backend, vals_max, vals_sum):
    tm = torch.tensor(vals_max, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    ts = torch.tensor(vals_sum, dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if dist is not None:
        dist.all_reduce(tm, op=dist.ReduceOp.MAX)
        dist.all_reduce(ts, op=dist.ReduceOp.SUM)
    return tm.tolist(), ts.tolist()


def training_leg(ctx, torch, dist, backend, rank, world, n_total, iters, warm):
    """BASELINE config 5: one embedded-training iteration (hmm::train, src/hmm.cpp:254-347) over n_total utterances
    split across the ranks (STRONG scaling: the corpus is fixed, each rank holds 1/world of it): alignment-mode scoring,
    Viterbi, transition re-estimation, statistics, the three all-reduces (RCCL under nccl), blend.  Timed with a barrier
    and a device synchronisation on both sides; the slowest rank counts."""
    from speechrecognition_amd import SRX_MIX_LOGSUMEXP, host, parallel, synth
    from speechrecognition_amd._capi import SRX_SCORE_MFMA_F16X3
    phones, tieds, Ts = alignment_corpus(n_total, 4)
    hb, feats_d, T, flop = build_shard(ctx, torch, host, parallel, phones, tieds, Ts, rank, world, 1000)
    # the reference's two model objects: X is main()'s acm -- the one every utterance HMM scores under (src/utterance.cpp:20)
    # -- and Y the lexicon HMM's temp_acm; hmm::train reads one and rebuilds the other, swapping each iteration
    # (src/hmm.cpp:344-346), so X is rewritten every second iteration and iterations 2k+1, 2k+2 align under the same scores
    X, Y = (host.AcousticModel(ctx, G, M, D) for _ in range(2))
    X.set(*synth.random_model(G, M, D, 777))
    Y.copy_from(X)
    # The exchange goes through the library's own communicator (srx_comm_*: ncclAllReduce on the context's stream) -- what
    # hmm::train / `srx --mode train` of the C++ layer use.  Rank 0 draws the RCCL id, torch.distributed's store hands it round.
    # SRX_BENCH_BACKEND=gloo (ranks sharing a GPU, which RCCL refuses): the host rehearsal transport of the same communicator.
    # SRX_BENCH_EXCHANGE=torch selects torch.distributed's all_reduce instead (round 2's form).
    force = os.environ.get("SRX_BENCH_FORCE_GROUP", "0") == "1"
    if os.environ.get("SRX_BENCH_EXCHANGE", "srx") == "torch":
        ex = parallel.Exchange(force=force)
    elif dist is not None and world > 1 and backend != "nccl":
        ex = parallel.SrxExchange(ctx, rank, world, shm_name="/srx_bench_%s" % os.environ.get("MASTER_PORT", "0"))
    else:
        def broadcast_id(raw):
            box = [raw]
            dist.broadcast_object_list(box, src=0)
            return box[0]
        ex = parallel.SrxExchange(ctx, rank, world, broadcast_id=broadcast_id, force_rccl=force and world == 1)
    em_path = os.environ.get("SRX_BENCH_EM_PATH", "fused")
    shard = parallel.TrainingShard(ctx, hb, feats_d, D, G, M, D, ex, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA_F16X3, emissions="compact",
                                   em_path=em_path)
    iters += iters & 1          # whole swap cycles
    warm += warm & 1
    state = {"k": 0}

    def one_iteration():
        A, B = (X, Y) if state["k"] % 2 == 0 else (Y, X)
        shard.iteration(X, A, B, 0.2)
        state["k"] += 1

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    progress("training shard resident (%d frames on this rank); warm-up" % T)
    for _ in range(warm):
        one_iteration()
    barrier()
    # first every iteration scored (what the reference does, lazily inside its Viterbi), then the product's behaviour: the
    # scoring pass skipped while the utterances' model object has not been rewritten
    progress("training: %d timed iterations, every one scored" % iters)
    shard.reuse_scores = False
    t0 = time.perf_counter()
    for _ in range(iters):
        one_iteration()
    barrier()
    dt_all = time.perf_counter() - t0
    progress("training: %d timed iterations" % iters)
    shard.reuse_scores = True
    shard._scored_key = None
    shard.scoring_passes = 0
    shard.time_collectives = True
    ex.bytes_reduced = 0
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(iters):
        one_iteration()
    barrier()
    dt = time.perf_counter() - t0
    coll = {k: float(np.mean(v)) for k, v in shard.collective_ms().items()}
    kern = {}
    score_ms_per_launch = None
    for k in ("k_gmm_f16_align", "k_viterbi", "k_train_transitions", "k_gather_gid", "k_em_csort_hist", "k_em_csort_scan", "k_em_csort_scatter", "k_em_logp_seg",
              "k_em_accum_seg1", "k_em_accum_seg2", "k_em_seg_reduce", "k_em_means", "k_em_finish", "k_em_fused", "k_em_peak",
              "k_em_fused_finish", "rccl_allreduce_max", "rccl_allreduce_sum"):
        ms, n = ctx.profile_query(k)
        if n:
            kern[k] = ms / iters
            if k == "k_gmm_f16_align":
                score_ms_per_launch = ms / n
    ctx.profile_enable(False)
    assert (shard.status.get()[:hb.n_utt] == 0).all(), "alignment failed for some utterance"
    (dt_max, dt_all_max), (frames_all, flop_all) = reduce_max_sum(torch, dist, backend, [dt, dt_all], [float(T), flop])
    ms_it = dt_max / iters * 1e3
    n12 = G * M * 16 * ((D + 1 + 7) // 8)
    per_iter_bytes = {"faithful": {"max_shift_f32": G * M * 4, "sum_stats1_f64": G * M * (1 + D) * 8, "sum_stats2_f64": G * M * D * 8},
                      "fold": {"max_shift_f32": G * M * 4, "sum_stats12_f64": n12 * 8},
                      "fused": {"sum_stats12_f64": n12 * 8}}[em_path]
    out = {"workload": f"configs[4]: one embedded-training iteration over {n_total} utterances split across {world} GPU(s) "
                       f"(alignment-mode scoring + Viterbi + transitions + statistics [{em_path}] + {len(per_iter_bytes)} all-reduce(s) + blend)",
           "statistics_path": em_path, "statistics_fallbacks_to_five_step_form": shard.fallbacks,
           "exchange": {"through": type(ex).__name__, "transport": ex.transport,
                        "note": "SrxExchange = srx_comm_* of the C ABI (ncclAllReduce on the context's stream; 'host' = the "
                                "shared-memory rehearsal for ranks sharing a GPU; 'none' = one rank, every call a no-op)"},
           "scaling": "strong", "utterances_total": n_total, "frames_total": frames_all, "iterations_timed": iters,
           "ms_per_iteration": ms_it, "frames_per_s": frames_all / (ms_it * 1e-3),
           "scoring_passes_per_iteration": shard.scoring_passes / iters,
           "ms_per_iteration_every_iteration_scored": dt_all_max / iters * 1e3,
           "note": "acm / temp_acm swap as in src/hmm.cpp:344-346: the utterances' model object is rewritten every second "
                   "iteration, so two consecutive iterations align under identical emission scores and share one scoring pass; "
                   "ms_per_iteration is the mean over whole swap cycles, ..._every_iteration_scored the same loop with the "
                   "reuse switched off",
           "rank0_kernels_ms_per_iteration": kern, "rank0_collectives_ms": coll,
           "collective_bytes_per_iteration": per_iter_bytes,
           "bytes_reduced_per_iteration_counted": ex.counted_bytes() / max(iters, 1),
           "k_gmm_f16_align_ms_per_launch_rank0": score_ms_per_launch,
           "align_scoring_algorithmic_TFLOPs_rank0": (flop / (score_ms_per_launch * 1e-3) / 1e12) if score_ms_per_launch else None}
    hb.close()
    for m_ in (X, Y):
        m_.close()
    if hasattr(ex, "close"):
        ex.close()
    return out


def decode_leg(ctx, n_words, T):
    """hmm::viterbi with a beam on the lexicon HMM of src/hmm.cpp:92-129 at the size src/main.cpp:58 builds it
    (133,860 words: the line count of res/lexicon/cmu.txt; synthetic words with its length distribution), one utterance of T
    frames with synthetic emission scores that spell words from the head of the lexicon.  Rank 0 only: decoding one utterance
    does not shard."""
    from speechrecognition_amd import host
    rng = np.random.default_rng(1)
    lens = np.clip(rng.poisson(5.4, n_words) + 1, 1, 20)
    words = [rng.integers(1, 40, int(k)).astype(np.int32) for k in lens]
    counts = rng.integers(1, 5000, n_words)
    uni = (np.log(counts) - np.float32(np.log(counts.sum() * 3.0))).astype(np.float32)
    a = host.build_lexicon_hmm(words, uni)
    hmm = host.GraphHmm(ctx, a["tr_off"], a["tr_tgt"], a["tr_logp"], a["ctx"][:, 1].copy())
    rows = []
    while sum(len(r) for r in rows) < T:
        w = int(rng.integers(0, 8))
        for ph in list(words[w]) + [0]:
            e = rng.standard_normal((int(rng.integers(3, 9)), 40)).astype(np.float32) * 2 - 12
            e[:, ph] += 9.0
            rows.append(e)
    emis_d = ctx.array(np.concatenate(rows)[:T])
    fo = np.array([0, T], np.int64)
    out = {"workload": f"free decoding of one {T}-frame utterance over a {n_words}-word lexicon HMM (frontier sweep of srx_viterbi_graph_dev), "
                       "recognition end condition (best active state)",
           "n_states": int(a["n_states"]), "n_transitions": int(a["tr_off"][-1]), "runs": []}
    for beam in (200.0, 1000.0):      # (unpruned: tools/decode_bench.py -- the start state's whole serial chain every frame, 1.1 s)
        hmm.viterbi(emis_d, 40, 1, fo, beam, 1)                # (SRX_DECODE_AUTO: the frontier sweep at this size; default workspace limit)
        ctx.profile_enable(True)
        path, final, fstate, status = hmm.viterbi(emis_d, 40, 1, fo, beam, 1)
        ms, cnt = ctx.profile_query("k_viterbi_frontier")
        ctx.profile_enable(False)
        assert status[0] == 0, "decode failed"
        out["runs"].append({"beam": beam, "device_ms": ms / max(cnt, 1), "x_real_time": T * 0.010 / (ms / max(cnt, 1) * 1e-3),
                            "final_logp": float(final[0])})
    assert out["runs"][0]["final_logp"] == out["runs"][1]["final_logp"]      # here the beam prunes nothing the best path needs
    hmm.close()
    return out


def alignment_leg(ctx, torch, dist, backend, rank, world, n_total, reps):
    """BASELINE config 3/4: forced alignment of n_total utterances split across the ranks (strong scaling): alignment-mode
    scoring + Viterbi, no collective (utterances are independent)."""
    from speechrecognition_amd import SRX_MIX_LOGSUMEXP, host, parallel, synth
    from speechrecognition_amd._capi import SRX_SCORE_MFMA_F16X3
    phones, tieds, Ts = alignment_corpus(n_total, 6)
    hb, feats_d, T, _ = build_shard(ctx, torch, host, parallel, phones, tieds, Ts, rank, world, 2000)
    model = host.AcousticModel(ctx, G, M, D)
    model.set(*synth.random_model(G, M, D, 777))
    hb.align_plan()
    emis = ctx.empty((max(hb.emis_count, 1),), np.float32)
    path_d, final_d, status_d = ctx.zeros((max(T, 1),), np.int32), ctx.zeros((hb.n_utt,), np.float32), ctx.zeros((hb.n_utt,), np.int32)

    def once():
        hb.score_align(model, feats_d, D, emis, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA_F16X3)
        hb.viterbi_align(emis, 1e10, path_d, final_d, status_d)

    once()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        once()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    assert (status_d.get() == 0).all()
    (dt_max,), (frames_all,) = reduce_max_sum(torch, dist, backend, [dt], [float(T)])
    hb.close()
    model.close()
    return {"workload": f"configs[3]: forced alignment of {n_total} utterances split across {world} GPU(s)", "scaling": "strong",
            "utterances_total": n_total, "frames_total": frames_all, "ms_per_pass": dt_max / reps * 1e3,
            "frames_per_s": frames_all * reps / dt_max}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--utts", type=int, default=1000)
    ap.add_argument("--no-cpu", action="store_true", help="skip the cpu_baseline leg")
    ap.add_argument("--frame-major", action="store_true", help="emission matrix as [T][G] instead of [G][T]")
    ap.add_argument("--train-utts", type=int, default=100000, help="config-5 training leg: utterances in the whole corpus (0: skip)")
    ap.add_argument("--train-iters", type=int, default=5)
    ap.add_argument("--align-utts", type=int, default=10000, help="config-3/4 alignment leg: utterances in the whole corpus (0: skip)")
    ap.add_argument("--decode-words", type=int, default=133860, help="decode leg: words in the lexicon HMM (0: skip)")
    ap.add_argument("--score-path", choices=["f16x3", "f32"], default="f16x3",
                    help="GMM scoring kernel: split-f16 MFMA (default) or f32-input MFMA")
    args = ap.parse_args()

    # Rank 0's stdout carries ONE JSON line and nothing else: libraries that greet on stdout (RCCL prints a version banner when
    # a communicator is created outside torch) are sent to stderr for the whole run, and the line goes out through the saved
    # descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)

    import torch
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # SRX_BENCH_BACKEND=gloo is the rehearsal of the N>1 control flow on a box with fewer GPUs than ranks (ranks then
    # share devices and the two scalar reductions go through host memory); the driver's runs use nccl (= RCCL).
    backend = os.environ.get("SRX_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local %= max(torch.cuda.device_count(), 1)
    # SRX_BENCH_FORCE_GROUP=1: a process group (and live all-reduces) even with one rank -- the rehearsal of the RCCL code
    # path of the N>1 run (device tensors, collectives on the context's stream, event pairs) on a one-GPU box
    force_group = os.environ.get("SRX_BENCH_FORCE_GROUP", "0") == "1"
    if world > 1 or force_group:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        os.environ.setdefault("RANK", str(rank))
        os.environ.setdefault("WORLD_SIZE", str(world))
        torch.cuda.set_device(local)
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local))
        else:
            dist.init_process_group(backend)
    else:
        dist = None
        torch.cuda.set_device(local)
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE {world}"

    from speechrecognition_amd import SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA, host, synth
    from speechrecognition_amd._capi import SRX_SCORE_MFMA_F16X3
    score_path = SRX_SCORE_MFMA_F16X3 if args.score_path == "f16x3" else SRX_SCORE_MFMA
    gmm_kernel, gmm_prep = ("k_gmm_f16", "k_prep16") if args.score_path == "f16x3" else ("k_gmm_mfma", "k_gmm_prep")
    mfcc_kernel = "k_mfcc_f16" if args.score_path == "f16x3" else "k_mfcc_tile"
    stream = torch.cuda.Stream()   # a real (non-null) stream: libsrx launches on it and the events below time it
    torch.cuda.set_stream(stream)
    ctx = host.Context(local, stream.cuda_stream)

    progress("process group up, building the headline workload")
    # ---- workload, resident in HBM ----
    x, soff, ntrs, trs, tieds = build_workload(args.utts, seed=1 + rank)
    fb, _ = host.mel_filterbank(FEND, NFILT, WL, SPS)
    cfg = host.mfcc_config(WL, HOP, NCEPS)
    plan = host.MfccPlan(ctx, cfg, fb, host.MFCC_F16X3 if args.score_path == "f16x3" else host.MFCC_F32)
    foff = host.frame_offsets(cfg, soff)
    T = int(foff[-1])
    n_utt = args.utts
    model_np = synth.random_model(G, M, D, 777)
    model = host.AcousticModel(ctx, G, M, D)
    model.set(*model_np)
    hb = host.HmmBatch(ctx, foff, ntrs, trs, tieds)
    x_d, soff_d = ctx.array(x), ctx.array(soff)
    scale_d = ctx.empty((n_utt,), np.float32)
    ceps_d = ctx.empty((T, NCEPS), np.float32)
    feats_d = ctx.empty((T, 3 * NCEPS), np.float32)
    # emission matrix layout: state-major [G][Tp] (default; rows 16-byte aligned: the alignment kernel then fetches four
    # frames per load) or frame-major [T][G]
    Tp = (T + 3) // 4 * 4
    emis_d = ctx.empty((G, Tp), np.float32)
    ST_T, ST_G = (G, 1) if args.frame_major else (1, Tp)
    path_d = ctx.zeros((T,), np.int32)
    final_d = ctx.zeros((n_utt,), np.float32)
    status_d = ctx.zeros((n_utt,), np.int32)

    ev = lambda: torch.cuda.Event(enable_timing=True)
    stage_ms = {"mfcc": 0.0, "features": 0.0, "gmm": 0.0, "viterbi": 0.0}

    def step(timed):
        e = [ev() for _ in range(5)] if timed else None
        if timed: e[0].record(stream)
        plan.write_features_dev(x_d, soff_d, n_utt, hb.frame_off_d, T, scale_d, ceps_d)
        if timed: e[1].record(stream)
        host.read_features_dev(ctx, 2, NCEPS, ceps_d, hb.frame_off_d, n_utt, T, feats_d)
        if timed: e[2].record(stream)
        model.score_dev(feats_d, 3 * NCEPS, T, emis_d, ST_T, ST_G, SRX_MIX_LOGSUMEXP, score_path)
        if timed: e[3].record(stream)
        hb.viterbi(emis_d, ST_T, ST_G, 1e10, path_d, final_d, status_d)
        if timed: e[4].record(stream)
        return e

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    progress("workload resident (%d frames); warm-up" % T)
    for _ in range(args.warmup):
        step(False)
    barrier()
    progress("timed region: %d steps" % args.steps)
    ctx.profile_enable(True)   # HIP event pairs around every kernel launch, on the launch stream
    t0 = time.perf_counter()
    evs = [step(True) for _ in range(args.steps)]
    barrier()
    dt = time.perf_counter() - t0
    for e in evs:
        for k, name in enumerate(("mfcc", "features", "gmm", "viterbi")):
            stage_ms[name] += e[k].elapsed_time(e[k + 1]) / args.steps

    kernel_ms = {}
    for k in ("k_sumsq_scale", mfcc_kernel, "k_features", gmm_prep, gmm_kernel, "k_viterbi"):
        ms, n = ctx.profile_query(k)
        assert n == args.steps, (k, n)
        kernel_ms[k] = ms / n
    ctx.profile_enable(False)

    status = status_d.get()
    assert (status == 0).all(), "alignment failed for some utterance"

    tot = torch.tensor([dt, float(T)], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
    if dist is not None:
        tmax = tot.clone()
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dt_max, frames_all = float(tmax[0]), float(tot[1])
    else:
        dt_max, frames_all = dt, float(T)

    progress("headline done: %.3f ms per step on this rank" % (dt / args.steps * 1e3))
    # ---- the like-for-like precision figure: the f32-input matrix-core kernel on the same batch (not in the headline) ----
    extra = {}
    if args.score_path == "f16x3":
        ctx.profile_enable(True)
        for _ in range(3):
            model.score_dev(feats_d, 3 * NCEPS, T, emis_d, ST_T, ST_G, SRX_MIX_LOGSUMEXP, SRX_SCORE_MFMA)
        ms32, n32 = ctx.profile_query("k_gmm_mfma")
        ctx.profile_enable(False)
        tf32 = 2.0 * (2 * D + 1) * G * M * T / (ms32 / n32 * 1e-3) / 1e12
        extra["gmm_f32_path"] = {"kernel": "k_gmm_mfma<16,4>", "ms_per_launch": ms32 / n32, "achieved_TFLOPs": tf32,
                                 "peak_TFLOPs": PEAK_F32_MFMA_TFLOPS, "frac": tf32 / PEAK_F32_MFMA_TFLOPS,
                                 "note": "same contraction with f32 inputs (exact f32 products); --score-path f32 makes it the headline path"}
    # ---- free the headline buffers, then the strong-scaling legs over BASELINE configs 3-5 ----
    del emis_d, x_d, ceps_d
    progress("f32-path figure done; alignment leg (%d utterances / %d ranks)" % (args.align_utts, world))
    if args.align_utts > 0:
        extra["align_config4"] = alignment_leg(ctx, torch, dist, backend, rank, world, args.align_utts, 5)
    progress("training leg (%d utterances / %d ranks)" % (args.train_utts, world))
    if args.train_utts > 0:
        extra["train_config5"] = training_leg(ctx, torch, dist, backend, rank, world, args.train_utts, args.train_iters, 2)

    if args.decode_words > 0 and rank == 0:
        progress("decode leg (%d words)" % args.decode_words)
        extra["decode_lexicon"] = decode_leg(ctx, args.decode_words, 500)
    progress("legs done")
    if rank == 0:
        ms_step = dt_max / args.steps * 1e3
        flop_per_frame = 2.0 * (2 * D + 1) * G * M          # GMM scoring, all states
        gmm_tflops = flop_per_frame * T / (kernel_ms[gmm_kernel] * 1e-3) / 1e12
        f16 = args.score_path == "f16x3"
        gmm_peak = PEAK_F16_MFMA_TFLOPS if f16 else PEAK_F32_MFMA_TFLOPS
        mfcc_bytes = (HOP * 4 + NCEPS * 4) * T               # 692 B/frame
        mfcc_ms, vit_ms = kernel_ms[mfcc_kernel], kernel_ms["k_viterbi"]
        vit_S = float(np.mean([len(n) for n in ntrs]))
        vit_bytes = (4.25 * vit_S + 5) * T
        out = {
            "metric": "audio frames/sec (node) through MFCC+GMM+Viterbi",
            "value": frames_all * args.steps / dt_max, "unit": "frames/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms_step, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f16x3 split operands, f32 accumulate" if f16 else "f32", "data": "synthetic",
            "config": {"workload": "configs[1] corpus: 1000 synthetic Voxforge-shaped utterances (2-8 s, 16 kHz) per GPU -> "
                                   "MFCC 25ms/10ms, 48 mel filters, 13 ceps -> smoothing+deltas (39 dims) -> dense GMM "
                                   "scoring 2000 tied states x 16 diag Gaussians x 39 dims (log-sum-exp) -> Viterbi forced "
                                   "alignment of every utterance",
                       "utterances_per_gpu": n_utt, "frames_per_gpu": T, "window": WL, "hop": HOP, "G": G, "M": M, "D": D,
                       "mean_states_per_utt": vit_S, "parallelism": f"utterance-sharded x{world}"},
            # achieved = ALGORITHMIC flops (2*(2D+1)*G*M per frame) / kernel time.  The f16x3 kernel executes three
            # f16 products per algorithmic product, so its ceiling in these units is 1/3 of the f16 MFMA peak;
            # against the f32-input MFMA peak (the only other way to hold the 1e-4 bar) see "vs_f32_mfma_peak".
            "roofline": {"bound": "mfma", "achieved": gmm_tflops, "peak": gmm_peak, "unit": "TFLOP/s",
                         "frac": gmm_tflops / gmm_peak, "traffic": pmc_traffic(gmm_kernel, T),
                         "traffic_source": "committed PMC pass (profiles/pmc_traffic.json: separate rocprofv3 --pmc runs of this workload), not measured in this run",
                         "kernel": "k_gmm_f16<16,5>" if f16 else "k_gmm_mfma<16,4>", "flop_per_frame": flop_per_frame,
                         "ms_per_launch": kernel_ms[gmm_kernel], "executed_over_algorithmic_flops": 3 if f16 else 1,
                         "mfma_executed_frac_of_peak": (3 if f16 else 1) * gmm_tflops / gmm_peak,
                         "vs_f32_mfma_peak": gmm_tflops / PEAK_F32_MFMA_TFLOPS},
            "stages_ms": stage_ms, "kernels_ms": kernel_ms, "extra": extra,
            "stage_rooflines": {
                "mfcc_hbm": {"kernel": mfcc_kernel, "achieved_GBs": mfcc_bytes / (mfcc_ms * 1e-3) / 1e9,
                             "peak_GBs": PEAK_HBM_GBS, "frac": mfcc_bytes / (mfcc_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                             "bytes_per_frame": 692, "traffic": pmc_traffic(mfcc_kernel, T)},
                "viterbi_hbm": {"kernel": "k_viterbi", "achieved_GBs": vit_bytes / (vit_ms * 1e-3) / 1e9,
                                "peak_GBs": PEAK_HBM_GBS, "frac": vit_bytes / (vit_ms * 1e-3) / 1e9 / PEAK_HBM_GBS,
                                "bytes_per_frame": 4.25 * vit_S + 5, "traffic": pmc_traffic("k_viterbi", T)},
            },
        }
        if world == 1 and not args.no_cpu:
            out["cpu_baseline"] = cpu_baseline(x, soff, ntrs, trs, tieds, model_np)
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    os.close(json_fd)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
