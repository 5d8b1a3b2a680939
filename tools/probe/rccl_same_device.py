"""Does RCCL accept two ranks on ONE device?  (Decides whether the 2-rank GPU tests can use nccl on the 1-GPU box.)
  python tools/probe/rccl_same_device.py   -> spawns 2 ranks on cuda:0, prints the outcome."""
import os
import subprocess
import sys

if len(sys.argv) == 1:
    ps = []
    for r in range(2):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="2", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29517",
                   HSA_ENABLE_IPC_MODE_LEGACY="0", NCCL_DEBUG="WARN")
        ps.append(subprocess.Popen([sys.executable, __file__, "w"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    for r, p in enumerate(ps):
        try:
            o, _ = p.communicate(timeout=90)
        except subprocess.TimeoutExpired:
            p.kill()
            o = "TIMEOUT"
        print("rank", r, "rc", p.returncode, o[-600:])
else:
    import torch
    import torch.distributed as dist
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
    t = torch.ones(4, device="cuda") * (int(os.environ["RANK"]) + 1)
    dist.all_reduce(t)
    torch.cuda.synchronize()
    print("ALLREDUCE OK", t.tolist())
    dist.destroy_process_group()
