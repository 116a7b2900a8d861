"""Child process of tests/test_rccl_gpu.py: world size 1 on the `nccl` backend (= RCCL on ROCm), initialised before anything
touches the GPU.  Runs the library's multi-GPU schedule (feos_torch_amd.dist) with the collectives forced, i.e. the kernels on
torch's stream, the RCCL all-gather behind them on RCCL's stream and the chunk-overlapped form of bench.py, and writes what
it saw as one JSON line."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

os.environ.update(WORLD_SIZE="1", RANK="0", LOCAL_RANK="0", PCS_FORCE_DIST="1")
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", sys.argv[1] if len(sys.argv) > 1 else "29517")

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from feos_torch_amd import dist as pdist  # noqa: E402
from feos_torch_amd import native  # noqa: E402
from feos_torch_amd.synthetic import pure_batch  # noqa: E402

rank, world, device = pdist.init_from_env()
assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1
n = 200_003  # odd: exercises the padded-shard logic's equal-size branch at world 1
P, T = pure_batch(n, seed=11)
Pd, Td = torch.from_numpy(P).to(device), torch.from_numpy(T).to(device)

# 1. the product schedule with the collective forced: shard solve -> all_gather_into_tensor of p (f64) and status (u8)
p, st = pdist.sharded_vapor_pressure(Pd, Td, force_collective=True)
ref = native.pure_vapor_pressure(Pd, Td)
same_p = bool(torch.equal(p, ref["p_sat"]))
same_st = bool(torch.equal(st, ref["status"]))

# 2. all_gather_flat asynchronously behind kernels that write straight into the send buffer (bench.py's all-gather leg):
#    four chunks, the collective of chunk k is enqueued while chunk k+1 is being solved
chunks = 4
bounds = [(n * k) // chunks for k in range(chunks + 1)]
plans = [native.PureVlePlan(bounds[k + 1] - bounds[k], device) for k in range(chunks)]
recv_p = [torch.full((pl.n,), -1.0, dtype=torch.float64, device=device) for pl in plans]
recv_s = [torch.full((pl.n,), 7, dtype=torch.uint8, device=device) for pl in plans]
works = []
for k, pl in enumerate(plans):
    pl.run(Pd[bounds[k]:bounds[k + 1]], Td[bounds[k]:bounds[k + 1]])
    works.append(pdist.all_gather_flat(recv_p[k], pl.p_sat, async_op=True))
    works.append(pdist.all_gather_flat(recv_s[k], pl.status, async_op=True))
for w in works:
    w.wait()
torch.cuda.synchronize()
chunk_p = bool(torch.equal(torch.cat(recv_p), ref["p_sat"]))
chunk_s = bool(torch.equal(torch.cat(recv_s).bool(), ref["status"]))

# 3. which native libraries this process mapped
with open("/proc/self/maps") as f:
    libs = sorted({os.path.basename(line.split()[-1]) for line in f if "rccl" in line.lower() or "libpcsaft_hip" in line})
print(json.dumps({"backend": dist.get_backend(), "world": world, "same_p": same_p, "same_status": same_st, "chunked_p": chunk_p,
                  "chunked_status": chunk_s, "failed_rows": int(ref["status"].sum().item()), "libs": libs}), flush=True)
dist.barrier()
dist.destroy_process_group()
