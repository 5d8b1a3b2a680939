"""Worker of the world_size-2 tests (tests/test_parallel_cpu.py, tests/test_gpu_parallel.py).  Not collected by
pytest.  Usage: python tests/_dist_worker.py <cpu|gpu|gpu_rccl> <workdir>   with RANK / WORLD_SIZE / MASTER_* in the env.

cpu: every rank holds a shard of the aligned frames, computes its partial shifted statistics in numpy (the same
     algebra as srx_train.hip) and exchanges them with speechrecognition_amd.parallel.Exchange over gloo.
gpu: every rank runs parallel.TrainingShard.iteration on its shard through libsrx on cuda:0 (gloo moves the
     CUDA tensors), two iterations with the acm / temp_acm ping-pong.
gpu_rccl: the same over the "nccl" backend (= RCCL), which needs one GPU per rank: run with one rank, all-reduces forced on.
Rank r writes <workdir>/out<r>.npz."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def logp_f64(x, mean, var, lw):
    """[T, M] log-densities of the frames' own states (mean/var/lw already gathered per frame): f64 statement of
    src/gmm.cpp:41-49."""
    import math
    return -0.5 * (math.log(2 / math.pi) + np.log(var) + (x[:, None, :] - mean) ** 2 / var).sum(-1) + lw


def z_files(work):
    return np.load(os.path.join(work, "in.npz")).files


def main():
    mode, work = sys.argv[1], sys.argv[2]
    import torch
    import torch.distributed as dist
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    rccl = mode == "gpu_rccl"      # the RCCL path itself (device tensors, collectives on the context's stream): one rank per GPU
    if rccl:
        torch.cuda.set_device(0)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", 0))
        mode = "gpu"
    else:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from speechrecognition_amd import parallel
    z = np.load(os.path.join(work, "in.npz"))
    G, M, D = [int(v) for v in z["shape"]]
    mean, var, w = z["mean"], z["var"], z["w"]
    # "exchange": "torch" (default: torch.distributed all_reduce) or "srx" -- the library's own communicator (srx_comm_*: what
    # hmm::train of the C++ layer uses): RCCL with one forced rank under gpu_rccl, the host rehearsal transport for ranks that
    # share the GPU
    use_srx = "exchange" in z_files(work) and str(np.load(os.path.join(work, "in.npz"))["exchange"]) == "srx"
    ex = None
    if not use_srx:
        ex = parallel.Exchange(force=rccl)   # a one-rank RCCL group still runs its all-reduces
        assert ex.active == (world > 1 or rccl)
    if mode == "cpu":
        feats, gid, frame_off = z["feats"], z["gid"], z["frame_off"]
        costs = np.diff(frame_off)
        mine = parallel.shard_utterances(costs, world)[rank]
        sel = np.concatenate([np.arange(frame_off[u], frame_off[u + 1]) for u in mine]) if len(mine) else np.zeros(0, np.int64)
        x, g = feats[sel, :D].astype(np.float64), gid[sel]
        lw = np.log(w.astype(np.float64))
        lp = logp_f64(x, mean[g], var[g], lw[g]) if len(sel) else np.zeros((0, M))
        shift = np.full((G, M), -np.inf, np.float32)
        for m in range(M):
            np.maximum.at(shift[:, m], g, lp[:, m].astype(np.float32))
        t_shift = torch.from_numpy(shift)
        ex.allreduce_max(t_shift)
        sh = t_shift.numpy().astype(np.float64)
        zeta = w[g].astype(np.float64) * np.exp(lp - sh[g]) if len(sel) else np.zeros((0, M))
        s1 = np.zeros((G, M, 1 + D))
        for m in range(M):
            np.add.at(s1[:, m, 0], g, zeta[:, m])
            np.add.at(s1[:, m, 1:], g, zeta[:, m, None] * x)
        t1 = torch.from_numpy(s1)
        ex.allreduce_sum(t1)
        s1 = t1.numpy()
        with np.errstate(invalid="ignore", divide="ignore"):
            newmean = s1[:, :, 1:] / s1[:, :, :1]
        s2 = np.zeros((G, M, D))
        for m in range(M):
            diff = (x - newmean[g, m]).astype(np.float32).astype(np.float64)
            np.add.at(s2[:, m], g, zeta[:, m, None] * diff * diff)
        t2 = torch.from_numpy(s2)
        ex.allreduce_sum(t2)
        with np.errstate(invalid="ignore", divide="ignore"):
            newvar = t2.numpy() / s1[:, :, :1]
            smax = sh.max(axis=1, keepdims=True)
            zz = np.where(np.isneginf(sh), 0.0, s1[:, :, 0] * np.exp(sh - smax))
            neww = (zz / zz.sum(axis=1, keepdims=True)).astype(np.float32)
        np.savez(os.path.join(work, "out%d.npz" % rank), mean=0.2 * newmean + 0.8 * mean, var=0.2 * newvar + 0.8 * var,
                 w=(neww * np.float32(0.2) + w * np.float32(0.8)).astype(np.float32), nframes=len(sel),
                 bytes_reduced=ex.bytes_reduced)
    else:
        from speechrecognition_amd import host
        ctx = host.Context(0)
        if use_srx:
            if rccl:
                ex = parallel.SrxExchange(ctx, 0, 1, force_rccl=True)
                assert ex.transport == "rccl"
            else:
                ex = parallel.SrxExchange(ctx, rank, world, shm_name="/srx_test_%s" % os.environ["MASTER_PORT"])
                assert ex.transport == ("host" if world > 1 else "none")
        feats, frame_off = z["feats"], z["frame_off"]
        n_utt = len(frame_off) - 1
        phones = [z["ph%d" % u] for u in range(n_utt)]
        costs = np.diff(frame_off) * np.array([len(p) for p in phones])
        mine = parallel.shard_utterances(costs, world)[rank]
        fo = np.zeros(len(mine) + 1, np.int64)
        fo[1:] = np.cumsum([frame_off[u + 1] - frame_off[u] for u in mine])
        x = np.concatenate([feats[frame_off[u]:frame_off[u + 1]] for u in mine]) if len(mine) else np.zeros((0, feats.shape[1]), np.float32)
        hm = [host.build_training_hmm(phones[u]) for u in mine]
        hb = host.HmmBatch(ctx, fo, [h[0] for h in hm], [h[1] for h in hm], [phones[u].astype(np.int32) for u in mine])
        O, Cm = host.AcousticModel(ctx, G, M, D), host.AcousticModel(ctx, G, M, D)
        O.set(mean, var, w)
        Cm.copy_from(O)
        shard = parallel.TrainingShard(ctx, hb, ctx.array(x) if len(x) else ctx.zeros((1, feats.shape[1]), np.float32),
                                       feats.shape[1], G, M, D, ex, int(z["mix_mode"]), 1,
                                       emissions=str(z["emissions"]) if "emissions" in z.files else "dense",
                                       em_path=str(z["em_path"]) if "em_path" in z.files else "faithful")
        A, B = O, Cm
        for _ in range(int(z["iters"])):
            shard.iteration(O, A, B, 0.2)     # Viterbi always under O
            A, B = B, A
        ctx.sync()
        m_, v_, w_ = A.get()
        np.savez(os.path.join(work, "out%d.npz" % rank), mean=m_, var=v_, w=w_, trlogp=hb.trlogp(), mine=mine,
                 path=shard.path.get()[:hb.total_frames], bytes_reduced=ex.bytes_reduced, fallbacks=shard.fallbacks)
        if use_srx:
            ex.close()
        ctx.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
