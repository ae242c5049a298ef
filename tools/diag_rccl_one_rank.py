import os, torch, torch.distributed as dist
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1)
import sys; sys.path.insert(0, "/root/repo")
from d3ga_amd.dist import ViewShardedGrads
s = ViewShardedGrads()
flat = torch.arange(10_000_000, dtype=torch.float32, device="cuda")
factor = torch.randn(500_001, 3, device="cuda")
g = s.exchange(flat.clone(), factor)
torch.cuda.synchronize()
print("world", s.world, "gathered", tuple(g.shape), "ok", bool(torch.equal(g[0], factor)), flush=True)
import time
for _ in range(3): s.exchange(flat, factor)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(20): s.exchange(flat, factor)
torch.cuda.synchronize(); print("1-rank RCCL exchange ms", (time.perf_counter() - t0) / 20 * 1e3)
dist.destroy_process_group()
