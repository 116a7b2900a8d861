#!/usr/bin/env python3
"""Headline benchmark: PC-SAFT pure-component vapour-pressure solves/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path (pcs_pure_vle: fused Helmholtz + Newton/VLE kernel and the
robust pass for rare rows) over one batch of `--rows` synthetic state points per GPU
(SURVEY.md §8d distribution, seed 2026 + rank), inputs resident in HBM.  Weak scaling: every
rank solves its own `--rows` rows and keeps its results (rows are independent: the path has no
exchange step, so there is no data-path collective; the step time is the MAX over ranks between
two barriers).  `--gather` additionally re-assembles the (p_sat fp64, status u8) shards on every
rank with an RCCL all-gather, chunk-overlapped with the solve, for callers that need the whole
result everywhere (90 MB per rank and step: over xGMI that costs more than the 1.7 ms solve).
Rank 0 prints ONE JSON line.

The line also carries
  roofline      for the dominant kernel k_pure_vle: algorithmic bytes (81 B/solve) / its launch
                duration measured with HIP events on the launch stream — reported against the
                HBM roof as the metric demands, next to the fp64-VALU view that actually bounds
                this path (the kernel is compute bound by ~2 orders of magnitude);
  cpu_baseline  the CPU oracle (own port of the same algorithm; the reference's Rust/feos path
                cannot be built here) timed on the host cores on a bounded sample.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

BYTES_PER_SOLVE = 81  # 64 B parameters + 8 B T read, 8 B p_sat + 1 B status written
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec
FP64_FMA_PEAK_GWAVEINSTR = 457.0  # measured on MI355X: independent v_fma_f64 chains, G wave-instructions/s (scratch/valu_peak.hip)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--rows", type=int, default=10_000_000, help="state points per GPU per step")
    ap.add_argument("--chunks", type=int, default=4, help="sub-batches per step (gather/solve overlap, N>1)")
    ap.add_argument("--gather", action="store_true",
                    help="N>1: also re-assemble (p_sat, status) on every rank with an RCCL all-gather, chunk-overlapped with the solve")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample", type=int, default=40_000_000, help="upper bound on the rows of the CPU baseline sample")
    return ap.parse_args()


def relaunch_under_torchrun(args):
    port = 29500 + (os.getpid() % 2000)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.call(cmd))


def load_pmc(rows):
    """Counter-derived figures of k_pure_vle from the committed rocprofv3 PMC passes (profiles/pmc_traffic.json):
    HBM bytes per launch and VALU wave-instructions per launch.  {} if absent or for another launch size."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        with open(path) as f:
            d = json.load(f)
        if int(d.get("rows", -1)) == int(rows):
            return d
    except Exception:
        pass
    return {}


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        relaunch_under_torchrun(args)  # before anything touches the GPU

    import numpy as np
    import torch
    import torch.distributed as dist

    from feos_torch_amd import dist as pdist
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import pure_batch

    rank, world, device = pdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs the GPU (no CPU fallback in the product path)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rows = args.rows
    nchunk = args.chunks if (world > 1 and args.gather) else 1
    assert rows % nchunk == 0

    # ---- inputs resident in HBM --------------------------------------------------------------
    P, T = pure_batch(rows, seed=2026 + rank)
    Pd = torch.from_numpy(P).to(device)
    Td = torch.from_numpy(T).to(device)
    crow = rows // nchunk
    plans = [native.PureVlePlan(crow, device) for _ in range(nchunk)]
    Pc = [Pd[k * crow:(k + 1) * crow] for k in range(nchunk)]
    Tc = [Td[k * crow:(k + 1) * crow] for k in range(nchunk)]
    gather = world > 1 and args.gather
    if gather:
        g_p = [torch.empty(world * crow, dtype=torch.float64, device=device) for _ in range(nchunk)]
        g_s = [torch.empty(world * crow, dtype=torch.uint8, device=device) for _ in range(nchunk)]

    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps * nchunk)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps * nchunk)]

    def step(i=None):
        works = []
        for k in range(nchunk):
            if i is not None:
                ev0[i * nchunk + k].record()
            plans[k].run_fast(Pc[k], Tc[k])
            if i is not None:
                ev1[i * nchunk + k].record()
            plans[k].run_retry(Pc[k], Tc[k])
            if gather:  # NCCL stream waits for the kernels above, the next chunk's solve overlaps it
                works.append(pdist.all_gather_flat(g_p[k], plans[k].p_sat, async_op=True))
                works.append(pdist.all_gather_flat(g_s[k], plans[k].status, async_op=True))
        for w in works:
            w.wait()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # ---- per-kernel duration of k_pure_vle from the HIP events (this rank) --------------------
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))  # per chunk launch
    fails = sum(int(p.status.sum().item()) for p in plans)
    counts = [p.retry_count() for p in plans]
    fallback_rows, retry_rows = sum(c[0] for c in counts), sum(c[1] for c in counts)

    if rank == 0:
        total = world * rows * args.steps
        value = total / dt
        achieved = BYTES_PER_SOLVE * crow / (kern_ms * 1e-3) / 1e9
        pmc = load_pmc(crow)
        traffic = pmc.get("hbm_bytes_per_launch")
        valu = pmc.get("valu_wave_instr_per_launch")
        line = {
            "metric": "pc_saft_pure_vapor_pressure_solves_per_sec",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"PcSaftPure.vapor_pressure batch={rows:.0e} fp64 per GPU (fused Helmholtz+Newton kernel)",
                "rows_per_gpu": rows,
                "global_rows": world * rows,
                "parallelism": f"row-sharded x{world}" + (", all-gather(p_sat,status) overlapped" if gather else ", no data-path collective"),
                "seed": 2026,
                "failed_rows_rank0": fails,
                "fp64_fallback_rows_rank0": fallback_rows,
                "robust_pass_rows_rank0": retry_rows,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_pure_vle<pressure-only> (+ k_pure_vle_fallback on the 0.04 % rows without an fp32 pre-solve)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "bytes_per_solve": BYTES_PER_SOLVE,
                "rows_per_launch": crow,
                "kernel_ms": kern_ms,
                "note": "path is VALU-issue bound (arithmetic intensity >> machine balance); see DESIGN.md",
                # the roofline that actually bounds the kernel: VALU issue.  Counted wave-instructions (PMC
                # SQ_INSTS_VALU, profiles/) / measured kernel time vs the measured v_fma_f64 issue peak
                # (the kernel mixes fp32 and fp64 VALU, so the fraction can exceed what fp64 alone allows)
                "valu": None if not valu else {
                    "wave_instr_per_launch": valu,
                    "instr_per_solve": valu * 64.0 / crow,
                    "achieved_gwaveinstr_s": valu / (kern_ms * 1e-3) / 1e9,
                    "fp64_fma_peak_gwaveinstr_s": FP64_FMA_PEAK_GWAVEINSTR,
                    "valu_busy_pmc": pmc.get("valu_busy"),
                },
            },
        }
        if not args.no_cpu_baseline and world == 1:
            # OpenMP threads = CPUs this process may use (affinity mask capped by the cgroup quota)
            from oracle import pyoracle as _o
            os.environ.setdefault("OMP_NUM_THREADS", str(_o.usable_cpus()))
            from oracle import pyoracle as orc

            orc.pure_vapor_pressure(P[:1000], T[:1000], prec=0)  # load + warm
            # size the sample for ~10 s of wall time on this host (bounded by --cpu-sample and the batch)
            t1 = time.perf_counter()
            orc.pure_vapor_pressure(P[:200_000], T[:200_000], prec=0)
            probe_rate = 200_000 / (time.perf_counter() - t1)
            ns = int(min(rows, args.cpu_sample, max(200_000, probe_rate * 10.0)))
            t1 = time.perf_counter()
            orc.pure_vapor_pressure(P[:ns], T[:ns], prec=0)
            ct = time.perf_counter() - t1
            line["cpu_baseline"] = {
                "value": ns / ct,
                "unit": "solves/s",
                "cores": orc.num_threads(),
                "kind": "port",
                "sample": f"first {ns} rows of the same batch, fp64, OpenMP over rows; own CPU restatement "
                          f"(oracle/), the reference's Rust/feos path is not buildable here",
                "seconds": ct,
            }
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
