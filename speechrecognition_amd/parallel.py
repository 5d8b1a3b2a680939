"""Multi-GPU layer: one process per GPU, utterances sharded across ranks, sufficient statistics combined
with all-reduce: RCCL over xGMI through the library's own communicator (SrxExchange over srx_comm_* -- what hmm::train of
the C++ layer uses) or through torch.distributed (Exchange: "nccl" = RCCL, "gloo" for CPU tests).

The reference is single-process; what shards are its sequential `for i < ut.size()` loops (src/hmm.cpp:258,
269, 275, 298, 319; src/main.cpp:70-91).  MFCC, scoring and alignment need no communication.  One EM iteration
exchanges:
    1. MAX  of shift[G*M]  f32   (per-component log-density maximum: makes every rank use the same shift)
    2. SUM  of stats1[G*M*(1+D)] f64   (zeta sums and zeta*x sums)   -> weights and means, identically on every rank
    3. SUM  of stats2[G*M*D] f64       (zeta*(x - new mean)^2 sums)  -> variances, blend
and once after the last iteration the state-model (probsum, count) tables (src/hmm.cpp:350-356).
With the fused statistics (em_path "fused", the default where the shape allows) an iteration exchanges ONE buffer:
    SUM of stats12[G*M*16*ceil((D+1)/8)] f64  ([S1 | S0 | S2 | frame count] per component; srx_em_fused_dev)
("fold": the MAX of the shifts and that one SUM).
"""
import numpy as np


def shard_utterances(costs, world):
    """Greedy longest-processing-time partition: utterances sorted by cost (frames x states) descending, each
    given to the currently lightest rank.  Returns a list of index arrays (ascending within a rank).
    Deterministic; every rank computes the same partition from the same costs."""
    costs = np.asarray(costs, dtype=np.float64)
    order = np.argsort(-costs, kind="stable")
    load = np.zeros(world)
    parts = [[] for _ in range(world)]
    for u in order:
        r = int(np.argmin(load))          # ties -> lowest rank
        parts[r].append(int(u))
        load[r] += costs[u]
    return [np.array(sorted(p), dtype=np.int64) for p in parts]


class Exchange:
    """The collectives of one EM iteration over torch tensors (CUDA tensors under nccl = RCCL, CPU tensors under
    gloo).  world == 1 (or no process group) is a no-op, so the single-GPU path runs the same code."""

    def __init__(self, group=None, force=False):
        """force: run the collectives even in a one-rank group (rehearsal of the RCCL path on one GPU)."""
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or force)
        self.bytes_reduced = 0
        # gloo moves CUDA tensors through its own staging at a few hundred KB/s on this stack (18 s per 10 MB): under
        # gloo (the CPU rehearsal of the multi-GPU path) device tensors are staged through host memory here instead
        self.host_staged = self.active and dist.get_backend(group) == "gloo"

    def _reduce(self, t, op):
        if self.active:
            if self.host_staged and t.is_cuda:
                h = t.cpu()                      # synchronises with the current stream
                self.dist.all_reduce(h, op=op, group=self.group)
                t.copy_(h)
            else:
                self.dist.all_reduce(t, op=op, group=self.group)
            self.bytes_reduced += t.numel() * t.element_size()
        return t

    transport = "torch.distributed"

    def counted_bytes(self):
        return self.bytes_reduced

    def allreduce_max(self, t):
        return self._reduce(t, self.dist.ReduceOp.MAX)

    def allreduce_sum(self, t):
        return self._reduce(t, self.dist.ReduceOp.SUM)


class SrxExchange:
    """The same collectives through the library's communicator (srx_comm_*, csrc/srx_comm.hip): ncclAllReduce in place on the
    context's stream -- the exchange hmm::train and `srx --mode train` of the C++ layer perform.  Created from a launcher's
    RANK / WORLD_SIZE: rank 0 draws the RCCL id (srx_comm_unique_id) and `broadcast_id` hands it to the other ranks
    (torch.distributed's store under torchrun).  transport: "rccl" (one GPU per rank), "host" (rehearsal: ranks sharing a
    GPU, buffers summed on the host through POSIX shared memory), "none" (one rank: every call is a no-op)."""

    def __init__(self, ctx, rank=0, world=1, broadcast_id=None, shm_name=None, force_rccl=False):
        import ctypes as C
        import os
        from ._capi import SRX_COMM_ID_BYTES, check
        self.ctx, self.lib, self.rank, self.world = ctx, ctx.lib, rank, world
        h = C.c_void_p()
        if shm_name:
            check(self.lib.srx_comm_create_host(ctx.h, rank, world, shm_name.encode(), C.byref(h)), ctx.h)
            self.transport = "host" if world > 1 else "none"
        elif world > 1 or force_rccl:
            if force_rccl:
                os.environ["SRX_COMM_FORCE_RCCL"] = "1"
            idb = (C.c_ubyte * SRX_COMM_ID_BYTES)()
            if rank == 0:
                check(self.lib.srx_comm_unique_id(idb), None)
            raw = bytes(idb)
            if world > 1:
                raw = broadcast_id(raw if rank == 0 else None)
            idb = (C.c_ubyte * SRX_COMM_ID_BYTES).from_buffer_copy(raw)
            check(self.lib.srx_comm_create(ctx.h, rank, world, idb, C.byref(h)), ctx.h)
            self.transport = "rccl"
        else:
            check(self.lib.srx_comm_create(ctx.h, 0, 1, None, C.byref(h)), ctx.h)
            self.transport = "none"
        self.h = h
        self.active = self.transport != "none"
        self.host_staged = False

    @property
    def bytes_reduced(self):
        return int(self.lib.srx_comm_bytes_reduced(self.h))

    @bytes_reduced.setter
    def bytes_reduced(self, v):
        self._bytes0 = self.bytes_reduced     # counted from here

    def counted_bytes(self):
        return self.bytes_reduced - getattr(self, "_bytes0", 0)

    def allreduce_max(self, t):
        from ._capi import check
        check(self.lib.srx_allreduce_max_f32_dev(self.h, t.data_ptr(), t.numel()), self.ctx.h)
        return t

    def allreduce_sum(self, t):
        from ._capi import check
        check(self.lib.srx_allreduce_sum_f64_dev(self.h, t.data_ptr(), t.numel()), self.ctx.h)
        return t

    def close(self):
        if self.h:
            self.lib.srx_comm_destroy(self.h)
            self.h = None


class TrainingShard:
    """One rank's share of an embedded-training corpus, resident on its GPU, and the iteration of
    hmm::train (src/hmm.cpp:254-347) over it.  Buffers that are exchanged are torch tensors (so that
    torch.distributed can reduce them in place); libsrx reads and writes them through their data pointers."""

    def __init__(self, ctx, hmm_batch, feats_d, ld, G, M, D, exchange=None, mix_mode=0, score_path=0, emissions="dense",
                 em_path="faithful"):
        """emissions: "dense" = the [G][T] matrix of srx_gmm_score_dev (any score path); "compact" = alignment-mode
        scoring of each utterance's own states (srx_gmm_score_align_dev: split-f16 or exact path; 40x less scoring
        work and memory at G = 2000, the only form that holds a 100k-utterance shard).
        em_path: how the statistics of src/hmm.cpp:274-341 are taken: "faithful" = the five steps that reproduce every f32
        rounding of the reference (three exchanges); "fold" = the same log-densities, the variance pass folded into the first
        on the f64 matrix cores (MAX + one SUM); "fused" = log-densities on the matrix cores too (one SUM; falls back to
        "faithful" for an iteration in which a component was lost to f64 underflow).  See include/srx.h, srx_em_fused_dev."""
        import torch
        from . import host
        self.torch, self.host = torch, host
        self.ctx, self.hb, self.feats_d, self.ld = ctx, hmm_batch, feats_d, ld
        self.G, self.M, self.D = G, M, D
        self.ex = exchange or Exchange()
        self.mix_mode, self.score_path = mix_mode, score_path
        T = hmm_batch.total_frames
        self.T = T
        dev = torch.device("cuda", ctx.device)
        self.emissions = emissions
        if emissions == "compact":
            hmm_batch.align_plan()
            hmm_batch.cache_frames(True)     # the shard's features are fixed for its lifetime: frame operands are built once
            self.emis = ctx.empty((max(hmm_batch.emis_count, 1),), np.float32)   # utterance blocks [T_u][S_u]
        else:
            self.emis = ctx.empty((G, max(T, 1)), np.float32)      # state-major [G][T]
        self.path = ctx.zeros((max(T, 1),), np.int32)
        self.final = ctx.zeros((max(hmm_batch.n_utt, 1),), np.float32)
        self.status = ctx.zeros((max(hmm_batch.n_utt, 1),), np.int32)
        self.gid = ctx.empty((max(T, 1),), np.int32)
        self.logp = ctx.empty((max(T * M, 1),), np.float32)
        self.t_shift = torch.empty(G * M, dtype=torch.float32, device=dev)
        self.t_stats1 = torch.empty(G * M * (1 + D), dtype=torch.float64, device=dev)
        self.t_stats2 = torch.empty(G * M * D, dtype=torch.float64, device=dev)
        self.em_path = em_path
        self.fallbacks = 0                 # iterations the fused path handed back to the five-step form
        if em_path != "faithful":
            assert M <= 16 and D <= 39 and G <= 16383, "shape outside the fused statistics path: use em_path='faithful'"
            n12 = G * M * 16 * ((D + 1 + 7) // 8)        # = srx_em_fused_len(model)
            self.t_stats12 = torch.empty(n12, dtype=torch.float64, device=dev)
            self.stats12 = host.DevArray.wrap(ctx, self.t_stats12.data_ptr(), (n12,), np.float64, self.t_stats12)
            self.under = ctx.zeros((1,), np.int32)
        # the context's stream as a torch stream: collectives are enqueued relative to it (see _collective)
        self.tstream = torch.cuda.ExternalStream(ctx.stream_handle(), device=dev)
        self.time_collectives = False
        self.reuse_scores = True      # skip the scoring pass while the Viterbi model has not been rewritten (see iteration)
        self._scored_key = None
        self.scoring_passes = 0
        self.collective_events = []
        self.shift = host.DevArray.wrap(ctx, self.t_shift.data_ptr(), (G * M,), np.float32, self.t_shift)
        self.stats1 = host.DevArray.wrap(ctx, self.t_stats1.data_ptr(), (G * M * (1 + D),), np.float64, self.t_stats1)
        self.stats2 = host.DevArray.wrap(ctx, self.t_stats2.data_ptr(), (G * M * D,), np.float64, self.t_stats2)

    def _collective(self, fn, t, name):
        """libsrx launches on the context's stream; torch.distributed orders a collective after torch's CURRENT stream and
        makes that stream wait for it.  The exchange therefore runs with the context's stream as torch's current stream
        (an ExternalStream over the same handle): producer kernel -> all-reduce -> consumer kernel are ordered on the
        device, with no host-side wait (round 1 bracketed every collective with two stream synchronisations).  With
        `time_collectives` an event pair on that stream brackets each exchange."""
        if not self.ex.active:
            return
        if not t.is_cuda:
            self.ctx.sync()          # host tensors (CPU rehearsal): the producer kernel has to be done
            fn(t)
            return
        torch = self.torch
        with torch.cuda.stream(self.tstream):
            if self.time_collectives:
                a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                a.record(self.tstream)
                fn(t)
                b.record(self.tstream)
                self.collective_events.append((name, a, b))
            else:
                fn(t)

    def collective_ms(self):
        """Synchronises and returns {name: [ms, ...]} of the timed exchanges since the last call."""
        self.ctx.sync()
        out = {}
        for name, a, b in self.collective_events:
            out.setdefault(name, []).append(a.elapsed_time(b))
        self.collective_events = []
        return out

    def iteration(self, vit_model, A, B, keep=0.2):
        """align under `vit_model`, re-estimate transitions, rebuild B from statistics under A (all ranks end
        with identical B).  The caller swaps (A, B) like src/hmm.cpp:344-346."""
        from ._capi import check
        ctx, hb, lib = self.ctx, self.hb, self.ctx.lib
        T = self.T
        if T > 0:
            # The utterance HMMs score under ONE model object (src/utterance.cpp:20) that hmm::train rewrites only every second
            # iteration (src/hmm.cpp:344-346): while its parameters have not changed, the emission scores this shard holds are
            # the ones a new scoring pass would produce, and only the transition probabilities make the alignment move.
            key = (int(vit_model.h.value), vit_model.version(), self.mix_mode, self.score_path)
            rescore = not (self.reuse_scores and key == self._scored_key)
            if self.emissions == "compact":
                if rescore:
                    hb.score_align(vit_model, self.feats_d, self.ld, self.emis, self.mix_mode, self.score_path)
                hb.viterbi_align(self.emis, 1e10, self.path, self.final, self.status)
            else:
                if rescore:
                    vit_model.score_dev(self.feats_d, self.ld, T, self.emis, 1, T, self.mix_mode, self.score_path)
                hb.viterbi(self.emis, 1, T, 1e10, self.path, self.final, self.status)
            self._scored_key = key
            self.scoring_passes += int(rescore)
            hb.train_transition_probabilities(self.path, self.status)
            check(lib.srx_em_gather_gid_dev(ctx.h, hb.c_ref(), self.path.ptr, self.status.ptr, T, self.gid.ptr), ctx.h)
        if self.em_path != "faithful":
            if self._fused_statistics(A, B, keep):
                return
            self.fallbacks += 1           # a component lost to f64 underflow somewhere: this iteration through the five steps
        check(lib.srx_em_shift_dev(ctx.h, A.h, self.feats_d.ptr, self.ld, self.gid.ptr, T, self.logp.ptr, self.shift.ptr), ctx.h)
        self._collective(self.ex.allreduce_max, self.t_shift, "max_shift_f32")
        check(lib.srx_em_pass1_dev(ctx.h, A.h, self.feats_d.ptr, self.ld, self.gid.ptr, T, self.logp.ptr, self.shift.ptr,
                                   self.stats1.ptr), ctx.h)
        self._collective(self.ex.allreduce_sum, self.t_stats1, "sum_stats1_f64")
        check(lib.srx_em_means_dev(ctx.h, B.h, self.stats1.ptr, self.shift.ptr), ctx.h)
        check(lib.srx_em_pass2_dev(ctx.h, A.h, B.h, self.feats_d.ptr, self.ld, self.gid.ptr, T, self.logp.ptr,
                                   self.shift.ptr, self.stats2.ptr), ctx.h)
        self._collective(self.ex.allreduce_sum, self.t_stats2, "sum_stats2_f64")
        check(lib.srx_em_finish_dev(ctx.h, A.h, B.h, self.stats1.ptr, self.stats2.ptr, keep), ctx.h)

    def _fused_statistics(self, A, B, keep):
        """srx_em_fused_dev / srx_em_fused_finish_dev with their one exchange.  Returns False when the peak-relative form lost
        a component to f64 underflow: the count is formed after the exchange, from the summed statistics, so every rank sees
        the same number and takes the same branch -- the caller then runs the five-step form for this iteration."""
        from ._capi import check
        ctx, lib, T = self.ctx, self.ctx.lib, self.T
        given = self.em_path == "fold"
        if given:
            check(lib.srx_em_shift_dev(ctx.h, A.h, self.feats_d.ptr, self.ld, self.gid.ptr, T, self.logp.ptr, self.shift.ptr), ctx.h)
            self._collective(self.ex.allreduce_max, self.t_shift, "max_shift_f32")
        check(lib.srx_em_fused_dev(ctx.h, A.h, self.feats_d.ptr, self.ld, self.gid.ptr, T, self.logp.ptr if given else None,
                                   self.shift.ptr if given else None, self.stats12.ptr), ctx.h)
        self._collective(self.ex.allreduce_sum, self.t_stats12, "sum_stats12_f64")
        check(lib.srx_em_fused_finish_dev(ctx.h, A.h, B.h, self.stats12.ptr, self.shift.ptr if given else None, keep,
                                          None if given else self.under.ptr), ctx.h)
        if given:
            return True
        return int(self.under.get()[0]) == 0      # one 4-byte read-back per iteration (synchronises the stream)
