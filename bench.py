#!/usr/bin/env python3
"""Headline benchmark: PC-SAFT pure-component vapour-pressure solves/s (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One step = one pass of the hot path at the KERNEL level (the C ABI `pcs_pure_vle_fast` + `pcs_pure_vle_retry` on
pre-allocated outputs: fused Helmholtz + Newton/VLE kernel, all-fp64 fallback kernel and the robust pass for rare rows)
over one batch of `--rows` synthetic state points per GPU (SURVEY.md §8d distribution, seed 2026 + rank), inputs resident
in HBM.  Rank 0 prints ONE JSON line:

  value / ms_per_step   weak scaling, every rank solves its own `--rows` rows and keeps its results: rows are independent,
                        the path has no exchange step (barrier + synchronize on both sides, MAX over ranks);
  allgather             N > 1: the same step with the RCCL all-gather that re-assembles (p_sat fp64, status u8) on every
                        rank (north_star), in `--chunks` sub-batches so that the collective of chunk k overlaps the
                        solve of chunk k+1;
  strong                N > 1: global batch fixed at `--rows` (1e7, the batch the metric is quoted on), rank r solves
                        rows [r n/N, (r+1) n/N), without and with the all-gather;
  variants              N = 1: the same batch through the all-fp64 kernel (densities returned), forward + Jacobian
                        (what a parameter fit runs) and the Python API (PcSaftPure.vapor_pressure, forward and
                        forward + backward, including its allocations and host synchronisation);
  roofline              for the dominant kernel k_pure_vle: algorithmic bytes (81 B/solve) / its launch duration measured
                        with HIP events on the launch stream, against the HBM roof as the metric demands, next to the
                        VALU-issue view that actually bounds this path (profiles/pmc_traffic.json);
  cpu_baseline          N = 1: the CPU oracle (own port; the reference's Rust/feos path cannot be built here) on the host
                        cores, bounded sample.
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


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--rows", type=int, default=10_000_000, help="state points per GPU per step (weak) / global batch (strong)")
    ap.add_argument("--chunks", type=int, default=4, help="sub-batches per step of the all-gather legs (gather/solve overlap)")
    ap.add_argument("--no-extra", action="store_true", help="headline leg only (no all-gather / strong / variants legs)")
    ap.add_argument("--force-gather", action="store_true",
                    help="--gpus 1 only: initialise RCCL at world size 1 and run the chunk-overlapped all-gather leg on it "
                         "(proves the backend, the stream hand-off and the message-size logic; no scaling number)")
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
    HBM bytes per launch, VALU wave-instructions per launch and their issue-cycle weighting.  {} if absent or for another
    launch size."""
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
    if args.force_gather and args.gpus == 1:
        os.environ["PCS_FORCE_DIST"] = "1"  # init_from_env then creates the (RCCL) process group at world size 1
        os.environ.setdefault("MASTER_PORT", str(29500 + (os.getpid() % 2000)))

    # RCCL prints a version banner on stdout when its first communicator comes up: keep stdout for the ONE JSON line (fd 1 points
    # at stderr until the line is printed)
    saved_stdout = None
    if args.gpus > 1 or args.force_gather:
        sys.stdout.flush()
        saved_stdout = os.dup(1)
        os.dup2(2, 1)

    import numpy as np
    import torch
    import torch.distributed as dist

    from feos_torch_amd import dist as pdist
    from feos_torch_amd import native
    from feos_torch_amd.synthetic import pure_batch

    rank, world, device = pdist.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs the GPU (no CPU fallback in the product path)"
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    rows, steps, warmup = args.rows, args.steps, args.warmup

    # ---- inputs resident in HBM --------------------------------------------------------------
    P, T = pure_batch(rows, seed=2026 + rank)
    Pd = torch.from_numpy(P).to(device)
    Td = torch.from_numpy(T).to(device)

    grouped = dist.is_initialized()  # world > 1, or world 1 with --force-gather

    def barrier():
        if grouped:
            dist.barrier()
        torch.cuda.synchronize()

    def timed(step):
        """W untimed + exactly K timed calls of step(i) between barriers; seconds, MAX over ranks."""
        for _ in range(warmup):
            step(None)
        barrier()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i)
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            dt = float(t.item())
        return dt

    # ---- headline leg: weak scaling, no data-path collective -----------------------------------
    plan = native.PureVlePlan(rows, device)
    ev0 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]
    ev1 = [torch.cuda.Event(enable_timing=True) for _ in range(steps)]

    def step_headline(i):
        if i is not None:
            ev0[i].record()
        plan.run_fast(Pd, Td)  # k_pure_vle<pressure-only> + k_pure_vle_fallback
        if i is not None:
            ev1[i].record()
        plan.run_retry(Pd, Td)  # robust pass over the (usually empty) list

    dt = timed(step_headline)
    kern_ms = float(np.mean([a.elapsed_time(b) for a, b in zip(ev0, ev1)]))
    fails = int(plan.status.sum().item())
    fallback_rows, retry_rows = plan.retry_count()

    # ---- sharded legs with the all-gather (N > 1) ---------------------------------------------
    def sharded_leg(Ps, Ts, gather):
        """one step = solve of this rank's rows in chunks (+ all-gather of chunk k overlapped with the solve of k+1)"""
        n_loc = Ts.shape[0]
        nchunk = args.chunks if gather else 1
        bounds = [(n_loc * k) // nchunk for k in range(nchunk + 1)]
        plans = [native.PureVlePlan(bounds[k + 1] - bounds[k], device) for k in range(nchunk)]
        if gather:
            mx = max(p.n for p in plans)  # equal message sizes across ranks: shards differ by at most one row
            mx_t = torch.tensor([mx], dtype=torch.int64, device=device)
            dist.all_reduce(mx_t, op=dist.ReduceOp.MAX)
            mx = int(mx_t.item())
            assert mx >= max(p.n for p in plans)
            send_p = [torch.zeros(mx, dtype=torch.float64, device=device) for _ in range(nchunk)]
            send_s = [torch.zeros(mx, dtype=torch.uint8, device=device) for _ in range(nchunk)]
            g_p = [torch.empty(world * mx, dtype=torch.float64, device=device) for _ in range(nchunk)]
            g_s = [torch.empty(world * mx, dtype=torch.uint8, device=device) for _ in range(nchunk)]
            for k, p in enumerate(plans):  # the kernels write straight into the send buffers
                p.p_sat, p.status = send_p[k][: p.n], send_s[k][: p.n]

        def step(_i):
            works = []
            for k, p in enumerate(plans):
                lo, hi = bounds[k], bounds[k + 1]
                p.run(Ps[lo:hi], Ts[lo:hi])
                if gather:  # the RCCL stream waits for the kernels above only; the next chunk's solve overlaps it
                    works.append(pdist.all_gather_flat(g_p[k], send_p[k], async_op=True))
                    works.append(pdist.all_gather_flat(g_s[k], send_s[k], async_op=True))
            for w in works:
                w.wait()

        return timed(step)

    extra = {}
    if world > 1 and not args.no_extra:
        t_g = sharded_leg(Pd, Td, True)
        extra["allgather"] = {
            "value": world * rows * steps / t_g, "unit": "solves/s", "ms_per_step": t_g / steps * 1e3, "scaling": "weak",
            "chunks": args.chunks, "bytes_gathered_per_rank_per_step": 9 * rows * (world - 1),
            "note": "the headline step + RCCL all-gather of (p_sat fp64, status u8) onto every rank, chunk-overlapped",
        }
        # strong scaling: the global batch of the metric (rank 0's synthetic batch, the same on every rank) cut into
        # contiguous shards
        Pg, Tg = pure_batch(rows, seed=2026)
        lo, hi = pdist.shard_bounds(rows, rank, world)
        Ps, Ts = torch.from_numpy(Pg[lo:hi]).to(device), torch.from_numpy(Tg[lo:hi]).to(device)
        t_s = sharded_leg(Ps, Ts, False)
        t_sg = sharded_leg(Ps, Ts, True)
        extra["strong"] = {
            "global_rows": rows, "rows_per_gpu": hi - lo, "scaling": "strong", "unit": "solves/s",
            "value": rows * steps / t_s, "ms_per_step": t_s / steps * 1e3,
            "value_with_allgather": rows * steps / t_sg, "ms_per_step_with_allgather": t_sg / steps * 1e3,
        }
        del Ps, Ts

    # ---- variants of the same batch on one GPU ---------------------------------------------------
    if world == 1 and grouped:
        t_g = sharded_leg(Pd, Td, True)
        extra["allgather_world1"] = {
            "backend": dist.get_backend(), "value": rows * steps / t_g, "unit": "solves/s", "ms_per_step": t_g / steps * 1e3,
            "chunks": args.chunks,
            "note": "--force-gather: the all-gather leg of the N > 1 runs executed on RCCL at world size 1 (kernels write into "
                    "the send buffers, ncclAllGather per chunk on RCCL's stream overlapped with the next chunk's solve). Proves "
                    "the backend path only; multi-GPU scaling is UNMEASURED on this pool (one GPU per box).",
        }
    if world == 1 and not args.no_extra:
        variants = {}
        full = native.PureVlePlan(rows, device, want_rho_vl=True)  # all-fp64 kernel k_pure_vle<false>, returns the densities
        t_f = timed(lambda _i: full.run(Pd, Td))
        variants["all_fp64_kernel"] = {"value": rows * steps / t_f, "ms_per_step": t_f / steps * 1e3,
                                       "what": "k_pure_vle<false>: fp64 D2 finish, p_sat + (rho_V, rho_L) returned"}

        def fwd_jac(_i):
            full.run(Pd, Td)
            native.pure_jacobian("vapor_pressure", Pd, Td, None, full.rho_vl)

        t_j = timed(fwd_jac)
        variants["forward_plus_jacobian"] = {"value": rows * steps / t_j, "ms_per_step": t_j / steps * 1e3,
                                             "what": "all-fp64 solve + k_pure_jacobian<0>: d p_sat / d(8 parameters, T) per row"}
        del full
        from feos_torch_amd import PcSaftPure

        def api_fwd(_i):
            PcSaftPure(Pd).vapor_pressure(Td)

        t_a = timed(api_fwd)
        Pg = Pd.clone().requires_grad_(True)

        def api_fwd_bwd(_i):
            Pg.grad = None
            _, p = PcSaftPure(Pg).vapor_pressure(Td)
            p.sum().backward()

        t_ab = timed(api_fwd_bwd)
        variants["python_api"] = {"forward_value": rows * steps / t_a, "forward_ms": t_a / steps * 1e3,
                                  "forward_backward_value": rows * steps / t_ab, "forward_backward_ms": t_ab / steps * 1e3,
                                  "what": "PcSaftPure(params).vapor_pressure(T) incl. output allocation, status host sync and row filtering"}
        del Pg
        extra["variants"] = variants

    if rank == 0:
        total = world * rows * steps
        value = total / dt
        achieved = BYTES_PER_SOLVE * rows / (kern_ms * 1e-3) / 1e9
        pmc = load_pmc(rows)
        traffic = pmc.get("hbm_bytes_per_launch")
        valu = pmc.get("valu_wave_instr_per_launch")
        line = {
            "metric": "pc_saft_pure_vapor_pressure_solves_per_sec",
            "value": value,
            "unit": "solves/s",
            "n_gpus": world,
            "steps": steps,
            "warmup": warmup,
            "ms_per_step": dt / steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64 (fp32 pre-solve, fp64 finish)",
            "data": "synthetic",
            "config": {
                "workload": f"pcs_pure_vle (C ABI behind PcSaftPure.vapor_pressure), batch={rows:.0e} fp64 rows per GPU, "
                            "pre-allocated outputs, pressure-only kernel",
                "rows_per_gpu": rows,
                "global_rows": world * rows,
                "parallelism": f"row-sharded x{world}, no data-path collective in `value` (see `allgather` / `strong`)",
                "seed": 2026,
                "failed_rows_rank0": fails,
                "fp64_fallback_rows_rank0": fallback_rows,
                "robust_pass_rows_rank0": retry_rows,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "k_pure_vle<pressure-only> (+ the launch of k_pure_vle_fallback for rows without an fp32 pre-solve: "
                          f"{fallback_rows} on this batch)",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": "profiles/pmc_traffic.json (rocprofv3 --pmc passes of the committed profile, not this run)",
                "bytes_per_solve": BYTES_PER_SOLVE,
                "rows_per_launch": rows,
                "kernel_ms": kern_ms,
                "note": "path is VALU-issue bound (arithmetic intensity >> machine balance); see DESIGN.md",
                # the roofline that actually bounds the kernel: VALU issue.  frac = sum over instruction classes of
                # (counted wave-instructions x issue cycles of the class) / (SIMD-cycles of the launch), a number <= 1
                # (scripts/summarise_profile.py: PMC instruction count split by the static fp32 / fp64 / transcendental mix)
                "valu": None if not valu else {
                    "source": f"profiles/pmc_traffic.json (tag {pmc.get('tag')}: counters and issue costs of the committed "
                              "profile; only kernel_ms / achieved are measured in this run)",
                    "wave_instr_per_launch": valu,
                    "instr_per_solve": valu * 64.0 / rows,
                    "achieved_gwaveinstr_s": valu / (kern_ms * 1e-3) / 1e9,
                    "frac": pmc.get("valu_issue_frac"),
                    "issue_cycles_per_launch": pmc.get("valu_issue_cycles_per_launch"),
                    "simd_cycles_per_launch": pmc.get("simd_cycles_per_launch"),
                    "mix": pmc.get("valu_mix"),
                },
            },
        }
        line.update(extra)
        if not args.no_cpu_baseline and world == 1:
            # OpenMP threads = CPUs this process may use (affinity mask capped by the cgroup quota)
            from oracle import pyoracle as orc

            os.environ.setdefault("OMP_NUM_THREADS", str(orc.usable_cpus()))
            orc.use_fast_build()  # the performance build of the same sources (-O3 -march=native), compiled on this host

            orc.pure_vapor_pressure(P[:1000], T[:1000], prec=0)  # load + warm
            # size the sample for ~10 s of wall time on this host (bounded by --cpu-sample and the batch)
            t1 = time.perf_counter()
            orc.pure_vapor_pressure(P[:200_000], T[:200_000], prec=0)
            probe_rate = 200_000 / (time.perf_counter() - t1)
            ns = int(min(rows, args.cpu_sample, max(200_000, probe_rate * 10.0)))
            # ~10 s of CPU work: the sample is the batch itself (or its head), repeated when the host gets through it faster
            reps = int(min(8, max(1, round(10.0 / max(ns / probe_rate, 1e-9)))))
            t1 = time.perf_counter()
            for _ in range(reps):
                orc.pure_vapor_pressure(P[:ns], T[:ns], prec=0)
            ct = time.perf_counter() - t1
            line["cpu_baseline"] = {
                "value": reps * ns / ct,
                "unit": "solves/s",
                "cores": orc.num_threads(),
                "kind": "port",
                "sample": f"first {ns} rows of the same batch x {reps} passes, fp64 (prec=0), OpenMP over rows; own CPU restatement "
                          f"(oracle/) built with {orc.FAST_FLAGS} on this host; the reference's Rust/feos path is not "
                          "buildable here",
                "seconds": ct,
            }
        if saved_stdout is not None:
            sys.stdout.flush()
            os.dup2(saved_stdout, 1)
        print(json.dumps(line), flush=True)
    if grouped:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
