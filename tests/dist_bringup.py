"""helper of tests/test_distributed.py: one rank of bench.bring_up_rccl() (no GPU needed: RCCL cannot come up here, which
is the case under test).  argv: rank world port allow_gloo fail_mode
  fail_mode "all"  : every rank tries RCCL and fails (no device)
  fail_mode "some" : rank 0 pretends to hang inside RCCL (sleeps) while the others fail: the watcher must take it out"""
import os
import sys
import time
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rank, world, port, allow, mode = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4] == "1", sys.argv[5]
os.environ.update({"MASTER_ADDR": "127.0.0.1", "MASTER_PORT": port, "RANK": str(rank), "WORLD_SIZE": str(world)})
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402
import bench  # noqa: E402

a = types.SimpleNamespace(allow_gloo=allow)
if mode == "some" and rank == 0:
    real = torch.cuda.set_device

    def hang(_):
        time.sleep(120)
        real(_)
    torch.cuda.set_device = hang
backend, device = bench.bring_up_rccl(a, dist, torch, rank, world, "cuda:0")
t = torch.ones(1)
dist.all_reduce(t)
print("RESULT %s %s %d %s" % (backend, device, int(t.item()), os.environ.get("LSQR_DIST_FALLBACK", "")[:40]))
dist.destroy_process_group()
