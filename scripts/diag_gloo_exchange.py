import os, sys, time, torch, numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch.distributed as dist
from nucliadb_amd import _lib
_lib.lib()
torch.cuda.set_device(0)
dist.init_process_group("gloo")
from nucliadb_amd.shard_merge import exchange_and_merge_vector, all_gather_hits
dev = torch.device("cuda", 0)
B, k = 1024, 10
sc = torch.rand((B, k), device=dev).sort(dim=1, descending=True).values
ids = torch.randint(0, 1 << 30, (B, k), device=dev, dtype=torch.int64)
cnt = torch.full((B,), k, dtype=torch.int32, device=dev)
for _ in range(3):
    exchange_and_merge_vector(sc, ids, cnt, k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    exchange_and_merge_vector(sc, ids, cnt, k)
torch.cuda.synchronize()
print(dist.get_rank(), "exchange ms", (time.perf_counter() - t0) / 20 * 1e3)
t0 = time.perf_counter()
for _ in range(20):
    all_gather_hits(sc, ids, cnt)
torch.cuda.synchronize()
print(dist.get_rank(), "gather only ms", (time.perf_counter() - t0) / 20 * 1e3)
# the same exchange issued the way bench.py --gpus N does: inside a side stream, with kernels queued on another stream
side = torch.cuda.Stream()
busy = torch.cuda.Stream()
x = torch.rand((4096, 4096), device=dev)
ev = torch.cuda.Event()
t0 = time.perf_counter()
for _ in range(20):
    with torch.cuda.stream(busy):
        y = x @ x
        ev.record(busy)
    with torch.cuda.stream(side):
        side.wait_event(ev)
        exchange_and_merge_vector(sc, ids, cnt, k)
torch.cuda.synchronize()
print(dist.get_rank(), "exchange on a side stream behind a busy stream ms", (time.perf_counter() - t0) / 20 * 1e3)
