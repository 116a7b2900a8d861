"""Multi-GPU driver: one process per GPU, rows sharded contiguously, outputs re-assembled.

The reference's only parallelism is rayon over independent rows inside one process
(src/pcsaft.rs:86-92).  Rows never interact, so on a node of MI355X the batch is cut into
contiguous shards [rank*n/W, (rank+1)*n/W), each rank solves its shard on its own GPU with no
data-path communication, and a single all-gather (RCCL over xGMI; `nccl` backend == RCCL on
ROCm) re-assembles `(value fp64, status u8)` on every rank.  Compaction of failed rows happens
after the gather so message sizes are static.

`compute` is injectable so the sharding/gather logic can be exercised on CPU with the `gloo`
backend (tests/test_dist_cpu.py); the product default is the HIP path.
"""
import os

import torch
import torch.distributed as dist


def init_from_env():
    """Initialise torch.distributed from RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* (torchrun).
    Returns (rank, world, device).  Single-process when WORLD_SIZE is unset or 1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_cuda = torch.cuda.is_available()
    # rehearsal on a 1-GPU box: PCS_SINGLE_DEVICE=1 maps every rank to cuda:0 (use with PCS_DIST_BACKEND=gloo,
    # RCCL refuses two ranks on one device)
    if os.environ.get("PCS_SINGLE_DEVICE") == "1":
        local = 0
    device = torch.device("cuda", local) if use_cuda else torch.device("cpu")
    if use_cuda:
        torch.cuda.set_device(device)
    # PCS_FORCE_DIST=1: initialise the process group at world size 1 too, so that the collectives below really run on the
    # backend (RCCL on a one-GPU box: bench.py --force-gather, tests/test_rccl_gpu.py)
    if (world > 1 or os.environ.get("PCS_FORCE_DIST") == "1") and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        # ROCm on these hosts only supports dmabuf IPC handles: with the legacy mode RCCL's (and torch's) cross-process
        # buffer registration fails with `hipIpcGetMemHandle: invalid argument`
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        backend = os.environ.get("PCS_DIST_BACKEND", "nccl" if use_cuda else "gloo")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, device


def shard_bounds(n, rank, world):
    """Contiguous row range of `rank`: [lo, hi).  Sizes differ by at most one row."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def all_gather_flat(recv, send, group=None, async_op=False):
    """all_gather of equal-size 1-D shards into a flat [world * len] tensor.  RCCL: one
    ncclAllGather (all_gather_into_tensor); gloo: list form on views of `recv`."""
    if dist.get_backend(group) == "nccl":
        return dist.all_gather_into_tensor(recv, send, group=group, async_op=async_op)
    world = dist.get_world_size(group)
    return dist.all_gather(list(recv.view(world, -1).unbind(0)), send, group=group, async_op=async_op)


def gather_rows(local, n_total, group=None, force=False):
    """All-gather 1-D (or [rows, k]) shards that were cut with shard_bounds() back into the
    full-length tensor, on every rank.  Shards are padded to the largest shard so the
    collective has a static message size.  At world size 1 the shard is the result and no
    collective runs unless `force` (exercises the backend on a one-GPU box)."""
    if not (dist.is_available() and dist.is_initialized()) or (dist.get_world_size(group) == 1 and not force):
        return local
    world = dist.get_world_size(group)
    sizes = [shard_bounds(n_total, r, world)[1] - shard_bounds(n_total, r, world)[0] for r in range(world)]
    mx = max(sizes)
    tail = tuple(local.shape[1:])
    send = local
    if local.shape[0] != mx:
        send = torch.zeros((mx,) + tail, dtype=local.dtype, device=local.device)
        send[: local.shape[0]] = local
    send = send.contiguous()
    recv = torch.empty((world * mx,) + tail, dtype=local.dtype, device=local.device)
    if dist.get_backend(group) == "nccl":
        dist.all_gather_into_tensor(recv, send, group=group)
    else:  # gloo: list form
        parts = list(recv.view((world, mx) + tail).unbind(0))
        dist.all_gather(parts, send, group=group)
        recv = torch.stack(parts, 0).view((world * mx,) + tail)
    if all(s == mx for s in sizes):
        return recv
    recv = recv.view((world, mx) + tail)
    return torch.cat([recv[r, : sizes[r]] for r in range(world)], dim=0)


def _hip_vapor_pressure(params, temperature):
    from . import native

    r = native.pure_vapor_pressure(params, temperature)  # the kernel behind PcSaftPure.vapor_pressure
    return r["p_sat"], r["status"]


def sharded_vapor_pressure(params, temperature, compute=None, group=None, force_collective=False):
    """Vapour pressures of the FULL batch (same `params` [n,8] / `temperature` [n] on every
    rank): each rank solves its contiguous shard, then one all-gather.  Returns dense
    (p_sat [n], status [n] bool) on every rank.  force_collective: run the all-gather at world size 1 too."""
    compute = compute or _hip_vapor_pressure
    n = temperature.shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n, rank, world)
    p_loc, st_loc = compute(params[lo:hi], temperature[lo:hi])
    p = gather_rows(p_loc, n, group, force_collective)
    st = gather_rows(st_loc.view(torch.uint8) if st_loc.dtype == torch.bool else st_loc, n, group, force_collective).bool()
    return p, st


def sharded_rows(compute, *row_tensors, group=None):
    """Generic form: every tensor in ``row_tensors`` has the batch as its first dimension; each rank applies
    ``compute(*shards)`` (returning a tuple of tensors with the shard's rows first) to its contiguous shard and the
    outputs are re-assembled on every rank.  Bubble / dew points, liquid densities, Jacobians ... shard this way:
    no row interacts with another."""
    n = row_tensors[0].shape[0]
    if dist.is_available() and dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = 0, 1
    lo, hi = shard_bounds(n, rank, world)
    outs = compute(*[t[lo:hi] for t in row_tensors])
    gathered = []
    for o in outs:
        as_u8 = o.dtype == torch.bool
        g = gather_rows(o.to(torch.uint8) if as_u8 else o, n, group)
        gathered.append(g.bool() if as_u8 else g)
    return tuple(gathered)

