#!/usr/bin/env python3
"""bench.py — headline benchmark of the MI355X stable-fluids hot path.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = vel_step + dens_step with the resident sources re-injected (sf_bind_sources) over the whole
grid (BASELINE.json metric: Mcells/s per vel_step+dens_step, 20 Jacobi iterations). At N = 1 the
workload is BASELINE.json configs[1]: 256^3 fp32, K = 20. For N > 1 the workload is configs[3]: the
1024^3 fp32 grid, slab-decomposed along k, one slab per GPU, RCCL halo exchange ("scaling": "strong" —
the grid is the same for every N; north_star's target is >= 6x at 8 GPUs on it). After the distributed
phase rank 0 times the SAME grid on its own GPU alone for a few steps, so the line carries an in-run
one-GPU denominator (`single_gpu`) and `speedup`. `--weak` keeps ~256^3 cells per GPU instead
(N_g = 320 / 408 / 512 for 2 / 4 / 8 GPUs, "scaling": "weak"). Inputs are the analytic fields of
docs/SPEC.md §5, resident in HBM before the timed region starts.

Rank 0 prints ONE JSON line. Besides the contract fields it carries
  roofline                : the dominant kernel (four fused Jacobi sweeps of lin_solve per launch) at 512^3 — the working set is far
                            beyond the 256 MiB Infinity Cache, so `achieved` is a statement about HBM — timed with
                            HIP events on the launch stream; `traffic` = counter-measured bytes per launch with its
                            provenance (a committed PMC pass: bench.py cannot run rocprofv3 on itself)
  roofline_cache_resident : the same kernel on the benchmark grid (256^3: x, x0, x' of one field fit the Infinity
                            Cache, so the algorithmic rate can exceed the HBM peak — not an HBM measurement)
  cpu_baseline            : the serial CPU oracle (oracle/, "port") on the same workload, bounded sample
The oracle is only the baseline/checker here, never the thing measured as `value`.
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X spec (MI355X_MICROARCH.md: 8.0 TB/s spec, 6.29 TB/s measured float4 copy)
WEAK_GRID = {1: 256, 2: 320, 4: 408, 8: 512}  # --weak: N_g^3 / gpus ~= 256^3, N_g divisible by gpus and by 4
STRONG_GRID = 1024  # BASELINE.json configs[3]: the grid of every N > 1 run


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--grid", dest="n", type=int, default=0,
                    help="grid size override (default: 256 at N = 1, 1024 for N > 1)")
    ap.add_argument("--weak", action="store_true", help="N > 1: ~256^3 cells per GPU instead of the 1024^3 grid")
    ap.add_argument("--single-steps", type=int, default=3,
                    help="N > 1: steps of the same grid timed on rank 0's GPU alone afterwards (0 = skip)")
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--roofline-n", type=int, default=512,
                    help="grid of the HBM roofline entry (-1: skip the roofline legs, for quick config runs)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-steps", type=int, default=3)
    ap.add_argument("--local-slabs", type=int, default=1,
                    help="rehearsal: cut this rank's part into that many logical slabs on its one GPU")
    return ap.parse_args()


def analytic_planes(N, kb, ke, dt, dtype):
    """SPEC §5 fields for global planes [kb, ke): dict of (ke-kb, N+2, N+2) arrays. Shells are zero
    here and set on the device by sf_set_bnd."""
    S = N + 2
    A = 0.5 / (dt * N)
    i = np.arange(S, dtype=np.float64)
    X = (i - 0.5) / N
    inside = (i >= 1) & (i <= N)
    sx, cx = np.sin(2 * np.pi * X) * inside, np.cos(2 * np.pi * X) * inside
    k = np.arange(kb, ke, dtype=np.float64)
    Z = (k - 0.5) / N
    inz = ((k >= 1) & (k <= N)).astype(np.float64)
    sz = np.sin(2 * np.pi * Z) * inz
    u2 = A * np.outer(cx, sx)  # [j, i] = A sin(2piX) cos(2piY)
    v2 = -A * np.outer(sx, cx)  # [j, i] = -A cos(2piX) sin(2piY)
    out = {
        "u": (inz[:, None, None] * u2[None]).astype(dtype),
        "v": (inz[:, None, None] * v2[None]).astype(dtype),
        "w": np.zeros((ke - kb, S, S), dtype),
        "dens": ((0.5 * inz[:, None, None]) * np.outer(inside, inside)[None]
                 + 0.5 * sz[:, None, None] * np.outer(sx, sx)[None]).astype(dtype),
    }
    c = max(N // 2, 1)
    for name, val in (("su", 0.0), ("sv", A), ("sw", 0.0), ("sd", 100.0)):
        a = np.zeros((ke - kb, S, S), dtype)
        if val and kb <= c < ke:
            a[c - kb, c, c] = val
        out[name] = a
    return out


def upload_inputs(fs, N, dt, batch=32):
    """Analytic inputs for the planes this context stores, generated and uploaded `batch` planes at a time (the
    1024^3 fields are 4.3 GB each: never materialised whole on the host), then the shells."""
    kb, ke = fs.stored_planes()
    for b0 in range(kb, ke, batch):
        b1 = min(b0 + batch, ke)
        f = analytic_planes(N, b0, b1, dt, fs.np_dtype)
        for name, slot in (("u", "u"), ("v", "v"), ("w", "w"), ("dens", "dens"), ("su", "user0"), ("sv", "user1"),
                           ("sw", "user2"), ("sd", "user3")):
            fs.upload_planes(slot, b0, f[name])
    for b, n in ((1, "u"), (2, "v"), (3, "w"), (0, "dens")):
        fs.set_bnd(b, n)
    fs.sync()


def words_per_cell_step(K):
    return 56 + 18 * K  # dens_step 8+3K, vel_step 48+15K


def time_lin_solve(S, N, dtype, K, reps, device):
    """Average duration of ONE Jacobi launch of lin_solve at size N (NF = 1): HIP events on the context's compute stream
    around lin_solve(K), divided by the number of launches it issues (sf_lin_solve_launches). With the k-marching kernel
    a K = 20 solve is five launches of four fused sweeps each (the first reads caller data on the i-shell, the last
    writes the i-shell: same kernel, three template variants — the rocprofv3 kernel-trace averages of the three,
    weighted 1 : 3 : 1, are what this number must agree with). Returns a dict (times in microseconds)."""
    with S.FluidSolver(N, dtype=dtype, iters=K, device=device) as fs:
        rng = np.random.RandomState(1)
        plane = rng.standard_normal((1, N + 2, N + 2)).astype(fs.np_dtype)
        for k in range(0, N + 2):
            fs.upload_planes("dens", k, plane * (1.0 + 0.001 * k))
            fs.upload_planes("dens0", k, plane * (0.5 - 0.001 * k))
        a, c = 0.3, 1 + 6 * 0.3
        launches = fs.lin_solve_launches(K)
        fs.lin_solve(0, "dens", "dens0", a, c, K)  # warm-up
        fs.sync()
        per = []
        for _ in range(reps):
            fs.timer_start()
            fs.lin_solve(0, "dens", "dens0", a, c, K)
            per.append(fs.timer_stop() * 1e3 / launches)
        fs.sync()
        return {"us_per_launch": float(np.mean(per)), "us_per_launch_min": float(np.min(per)),
                "us_per_launch_median": float(np.median(per)), "timed_solves": reps,
                "sweeps_per_launch": K / launches, "launches_per_solve": launches,
                "us_per_sweep_whole_solve": float(np.mean(per)) * launches / K}


def cpu_baseline(N, K, dtype, steps, dt, diff, visc):
    """Serial CPU oracle on this box's host cores (1 thread). Up to 256^3: `steps` full vel_step+dens_step on the
    benchmark inputs. Larger grids: one lin_solve of K Jacobi iterations only, reported in the same
    unit through the step's algorithmic cost ((56+18K) words per cell per step vs 3K per lin_solve)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import oracle_lib as O

    cpu_baseline.fields = None
    npdt = np.float32 if dtype == "f32" else np.float64
    if N > 256:
        rng = np.random.RandomState(3)
        x = rng.standard_normal((N + 2,) * 3).astype(npdt)
        x0 = rng.standard_normal((N + 2,) * 3).astype(npdt)
        t0 = time.perf_counter()
        O.lin_solve(0, x, x0, npdt(0.3), npdt(2.8), K)
        el = time.perf_counter() - t0
        sweeps = N ** 3 * K / el / 1e6
        return {"value": sweeps * 3 * K / words_per_cell_step(K) / K, "unit": "Mcells/s", "cores": 1, "kind": "port",
                "sample": f"one lin_solve of {K} Jacobi iterations at {N}^3 {dtype} ({el:.1f} s, {sweeps:.0f} Mcell-sweeps/s), "
                          f"scaled to a full step by (56+18K)/(3K) words; serial C++ oracle (g++ -O2 -ffp-contract=off)",
                "host_cores_visible": os.cpu_count()}
    f = analytic_planes(N, 0, N + 2, dt, npdt)
    fields = {"u": f["u"], "v": f["v"], "w": f["w"], "dens": f["dens"]}
    for b, n in ((1, "u"), (2, "v"), (3, "w"), (0, "dens")):
        O.set_bnd(b, fields[n])
    t0 = time.perf_counter()
    for _ in range(steps):
        fields.update({"u0": f["su"].copy(), "v0": f["sv"].copy(), "w0": f["sw"].copy(), "dens0": f["sd"].copy()})
        O.step(N, fields, npdt(dt), npdt(diff), npdt(visc), K)
    el = time.perf_counter() - t0
    cpu_baseline.fields = fields  # state after `steps` steps: the expected values of parity_in_run
    return {"value": N ** 3 * steps / el / 1e6, "unit": "Mcells/s", "cores": 1, "kind": "port",
            "sample": f"{steps} x (vel_step+dens_step) at {N}^3 {dtype} K={K}, serial C++ oracle "
                      f"(g++ -O2 -ffp-contract=off), {el:.1f} s", "host_cores_visible": os.cpu_count()}


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        # not launched by torch.distributed.run: start it as a child (nothing has touched the GPU yet)
        port = os.environ.get("MASTER_PORT", "29531")
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
        sys.exit(subprocess.call(cmd))
    if world > 1:
        import torch  # noqa: F401  torch first: its bundled HIP runtime / RCCL are then the ones libsfgpu.so binds to
    from fluidsolvergpu_amd import dist as sfdist

    rank, local_rank, _ = sfdist.env_world()
    if "SF_FORCE_DEVICE" in os.environ:  # rehearsal of the multi-rank path on a one-GPU box
        local_rank = int(os.environ["SF_FORCE_DEVICE"])
    dist = sfdist.init("gloo")  # control plane only; the halo exchange is RCCL inside libsfgpu.so
    if os.environ.get("SF_BENCH_DRYRUN") == "1":
        # CPU-only rehearsal of THIS FILE's multi-rank plumbing (tests/test_dist_gloo.py): a stand-in with the same
        # method names that computes nothing. Never used for a reported number: the output is tagged "dryrun".
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import bench_dryrun_stub as S
    else:
        from fluidsolvergpu_amd import solver as S

    K, dt, diff, visc = args.iters, 0.1, 1e-4, 1e-4
    if args.n > 0:
        N = args.n
    elif world == 1:
        N = 256
    else:
        N = WEAK_GRID.get(world, 256 * world) if args.weak else STRONG_GRID
    assert N % world == 0, f"grid {N} not divisible by {world} ranks"
    # per-GPU work shrinks with N when the grid is fixed (strong), stays put with --weak
    scaling = "weak" if (world == 1 or args.weak) else "strong"
    nccl_id = sfdist.share_nccl_id(dist, S.nccl_unique_id)
    # SF_BENCH_LOOPBACK=1 (with SF_FORCE_DEVICE=0): rehearsal of this file's multi-process path on a one-GPU box, where
    # RCCL refuses two ranks on one device — every rank runs its own slab with SF_FLAG_LOOPBACK_HALO (halo messages
    # become device-local copies, include/sfgpu.h). The line it prints is tagged and is not a result.
    loopback = os.environ.get("SF_BENCH_LOOPBACK") == "1" and world > 1
    kw = {"flags": 1} if loopback else {}
    fs = S.FluidSolver(N, dtype=args.dtype, iters=K, dt=dt, diff=diff, visc=visc, device=local_rank, rank=rank,
                       nranks=world, nccl_id=None if loopback else nccl_id, nslabs_local=args.local_slabs, **kw)
    upload_inputs(fs, N, dt)

    # per-step source re-injection: the sources stay resident in SF_USER0..3 and are bound, which is bit-identical
    # to copying them into u0/v0/w0/dens0 before every step (tests/test_parity_gpu.py::test_bound_sources)
    fs.bind_sources("user0", "user1", "user2", "user3")

    def step():
        fs.vel_step()
        fs.dens_step()

    def barrier():
        fs.sync()
        if dist is not None:
            dist.barrier()
            if "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized():
                sys.modules["torch"].cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = sfdist.max_over_ranks(dist, time.perf_counter() - t0)

    # state sanity: finite, and the halo guard did not fire (sync would have raised)
    kb_o, ke_o = fs.owned_planes()
    probe = fs.download_planes("dens", kb_o, min(kb_o + 2, ke_o))
    assert np.isfinite(probe).all(), "non-finite density after the timed steps"

    # ---- per-step times (HIP events on the compute stream, one host sync per step): OUTSIDE the timed region above,
    # whose K steps run back to back as the contract asks; this second loop shows the spread the mean hides ----
    step_ms = []
    if world == 1:
        for _ in range(args.steps):
            fs.timer_start()
            step()
            step_ms.append(fs.timer_stop())

    # ---- parity in this very run: the state the cpu_baseline leg computes below (cpu_steps steps from the benchmark
    # inputs) computed by the device now, compared bit for bit once the oracle has it ----
    gpu_after = None
    want_parity = world == 1 and not args.no_cpu_baseline and N <= 256 and os.environ.get("SF_BENCH_DRYRUN") != "1"
    if want_parity:
        upload_inputs(fs, N, dt)
        for _ in range(args.cpu_steps):
            step()
        fs.sync()
        gpu_after = {n: fs.download(n) for n in ("u", "v", "w", "dens")}

    out = None
    if rank == 0:
        cells = float(N) ** 3
        ms_per_step = elapsed / args.steps * 1e3
        value = cells * args.steps / elapsed / 1e6
        wsize = 4 if args.dtype == "f32" else 8
        step_bytes = cells * words_per_cell_step(K) * wsize
        copy_gbps = fs.copy_bandwidth_gbps(1 << 30, 5)
        out = {
            "metric": ("Mcells/s per vel_step+dens_step (20 Jacobi iters) at 256^3; achieved HBM GB/s"
                       if (K == 20 and N == 256) else f"Mcells/s per vel_step+dens_step ({K} Jacobi iters) at {N}^3"),
            "value": value,
            "unit": "Mcells/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": scaling,
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": ("DRYRUN - no computation" if os.environ.get("SF_BENCH_DRYRUN") == "1"
                     else "LOOPBACK REHEARSAL on shared devices - not a result" if loopback
                     else "synthetic (analytic fields of docs/SPEC.md §5, resident in HBM)"),
            "config": {"workload": f"{N}^3 {args.dtype}, K={K} Jacobi iters per lin_solve, vel_step+dens_step "
                                   f"with per-step source re-injection", "grid": N, "jacobi_iters": K,
                       "slabs": world * args.local_slabs, "cells_per_gpu": cells / world,
                       "parallelism": f"k-slab x{world}" + (" (RCCL halo exchange)" if world > 1 else ""),
                       "schedule": fs.schedule_info() if hasattr(fs, "schedule_info") else None},
            "achieved_hbm_gbps_step": step_bytes / (elapsed / args.steps) / 1e9 / world,
            "step_algorithmic_bytes_per_cell": words_per_cell_step(K) * wsize,
            "hbm_copy_gbps_same_run": copy_gbps,
            "transport": fs.transport_info() if hasattr(fs, "transport_info") else None,
        }
        if step_ms:
            out["step_times_ms"] = {"n": len(step_ms), "min": float(np.min(step_ms)), "median": float(np.median(step_ms)),
                                    "p90": float(np.percentile(step_ms, 90)), "mean": float(np.mean(step_ms)),
                                    "note": "HIP events around each step, one host sync per step; a second loop after "
                                            "the timed region (`ms_per_step` is the back-to-back wall-clock mean)"}
        # Compulsory traffic of a step: the Jacobi part moves x, x0, x' once per LAUNCH of S fused sweeps (3 words per
        # cell per launch = 18K/S words over the six solves of a step), the rest as in SURVEY.md §8a (56 words)
        launches = fs.lin_solve_launches(K) if hasattr(fs, "lin_solve_launches") and K > 0 else 0
        if launches > 0:
            spl = K / launches
            comp_words = 56 + 18.0 * K / spl
            out["step_compulsory_bytes_per_cell"] = comp_words * wsize
            out["frac_step_compulsory"] = cells * comp_words * wsize / (elapsed / args.steps) / 1e9 / world / HBM_PEAK_GBPS
    fs.close()

    # ---- N > 1: the same grid on ONE GPU (rank 0's), in the same run: the denominator of `speedup` ----
    if dist is not None:
        # every rank has released its slab; the ranks other than 0 have nothing left to do: let them go instead of
        # parking them in a barrier behind rank 0's one-GPU and roofline legs (minutes at 1024^3)
        dist.barrier()
        dist.destroy_process_group()
        dist = None
    if world > 1 and args.single_steps > 0:
        if rank == 0:
            try:
                with S.FluidSolver(N, dtype=args.dtype, iters=K, dt=dt, diff=diff, visc=visc, device=local_rank) as f1:
                    upload_inputs(f1, N, dt)
                    f1.bind_sources("user0", "user1", "user2", "user3")
                    f1.vel_step()
                    f1.dens_step()
                    f1.sync()
                    t0 = time.perf_counter()
                    for _ in range(args.single_steps):
                        f1.vel_step()
                        f1.dens_step()
                    f1.sync()
                    t1 = (time.perf_counter() - t0) / args.single_steps
                out["single_gpu"] = {"grid": N, "steps": args.single_steps, "warmup": 1, "ms_per_step": t1 * 1e3,
                                     "value": float(N) ** 3 / t1 / 1e6, "unit": "Mcells/s",
                                     "note": "same grid, same inputs, rank 0's GPU alone, timed after the distributed phase"}
                out["speedup"] = out["value"] / out["single_gpu"]["value"]
                out["speedup_note"] = (f"PROVISIONAL: {args.steps} timed steps after {args.warmup} warm-up over {world} GPUs "
                                       f"against {args.single_steps} timed steps after 1 warm-up on one GPU; the driver "
                                       "computes scaling efficiency itself from its N = 1 run")
            except Exception as exc:  # e.g. the grid does not fit one GPU: report, do not fail the run
                print(f"bench.py: single-GPU leg failed: {type(exc).__name__}: {exc}", file=sys.stderr, flush=True)
                out["single_gpu"] = {"error": f"{type(exc).__name__}: {exc}"[:300]}
                out["speedup"] = None

    if rank == 0:
        # ---- roofline leg: the dominant kernel, per launch, HIP events on the launch stream -------
        wsize = 4 if args.dtype == "f32" else 8
        tpath = os.path.join(ROOT, "profiles", "traffic_latest.json")
        traffic = json.load(open(tpath)) if os.path.exists(tpath) else {}

        def roofline_entry(n):
            r = time_lin_solve(S, n, args.dtype, K, 20, local_rank)  # >= 20 repetitions, min and median
            spl = r["sweeps_per_launch"]
            alg = float(n) ** 3 * 3 * wsize * spl  # 3 words per cell per sweep x sweeps per launch
            us = max(r["us_per_launch"], 1e-9)
            tr = traffic.get(f"jacobi_{args.dtype}_{n}")
            e = {"bound": "hbm",
                 "kernel": ("jacobi_sk_kernel<T,1,*,S=4>: four fused lin_solve sweeps + set_bnd per launch (k-marching, LDS "
                            "halo exchange)" if spl > 3.5 else
                            "jacobi_sk_kernel<T,1,*,S=3>: three fused lin_solve sweeps + set_bnd per launch (k-marching, LDS "
                            "halo exchange)" if spl > 2.5 else
                            "two fused lin_solve sweeps + set_bnd per launch" if spl > 1.5 else
                            "jacobi_rb_kernel<T,1,*>: one lin_solve sweep + set_bnd per launch"),
                 "achieved": alg / (us * 1e-6) / 1e9, "peak": HBM_PEAK_GBPS, "unit": "GB/s",
                 "frac": alg / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS, "grid": n, "us_per_launch": us,
                 "us_per_launch_min": r["us_per_launch_min"], "us_per_launch_median": r["us_per_launch_median"],
                 "timed_solves": r["timed_solves"], "sweeps_per_launch": spl,
                 "us_per_launch_note": "HIP events around a whole solve / its launches: a per-solve mean over the "
                                       "first-pass, plain and last-pass variants of the kernel (1 : 3 : 1 at K = 20)",
                 "algorithmic_bytes_per_launch": alg, "launches_per_solve": r["launches_per_solve"],
                 "us_per_sweep_whole_solve": r["us_per_sweep_whole_solve"],
                 # what a launch of S fused sweeps MUST move: x and x0 in, x' out, once (3 words per cell); `frac`
                 # (SURVEY.md §8d's per-sweep model) counts that S times and is not a fraction of anything for S > 1
                 "compulsory_bytes_per_launch": alg / spl,
                 "frac_compulsory": alg / spl / (us * 1e-6) / 1e9 / HBM_PEAK_GBPS,
                 "frac_compulsory_best": alg / spl / (r["us_per_launch_min"] * 1e-6) / 1e9 / HBM_PEAK_GBPS}
            if tr:
                e["traffic"] = tr["bytes_per_launch"]
                # counter bytes belong to the PLAIN variant: divide by the kernel-trace time of that same variant from
                # the same profile when it is recorded (one box, one kernel), else by this run's per-solve mean
                t_us = tr.get("kernel_trace_us") or us
                e["frac_traffic"] = tr["bytes_per_launch"] / (t_us * 1e-6) / 1e9 / HBM_PEAK_GBPS
                e["frac_traffic_time_us"] = t_us
                e["traffic_provenance"] = {"file": "profiles/traffic_latest.json", "tag": tr.get("tag"),
                                           "source": tr.get("source"), "kernel": tr.get("kernel"),
                                           "note": "FETCH_SIZE x2 + WRITE_SIZE from separate rocprofv3 --pmc passes of an "
                                                   "earlier run of the same kernel; not measured in this run"}
            else:
                e["traffic"] = None
            return e

        hbm_n = args.roofline_n if args.roofline_n else 512
        if hbm_n < 0:
            print(json.dumps(out), flush=True)
            return
        out["roofline"] = roofline_entry(hbm_n)
        out["roofline"]["note"] = (f"{hbm_n}^3: x, x0 and x' of the solve (3 x {float(hbm_n) ** 3 * wsize / 1e6:.0f} MB) are far "
                                   "beyond the 256 MiB Infinity Cache, so this is an HBM measurement. algorithmic bytes = 3 "
                                   "words/cell/sweep x sweeps per launch: fusing S sweeps into one pass divides the real HBM "
                                   "traffic per sweep by S, which is why `achieved` can exceed the peak while `frac_traffic` "
                                   "(counter bytes / time / peak) cannot")
        n_bench = N if world == 1 else WEAK_GRID[1]
        if n_bench != hbm_n:
            out["roofline_cache_resident"] = roofline_entry(n_bench)
            out["roofline_cache_resident"]["note"] = (
                f"{n_bench}^3: the solve's working set sits largely in the 256 MiB Infinity Cache, so `frac` is NOT a "
                "fraction of HBM bandwidth and may exceed 1")
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(N, K, args.dtype, args.cpu_steps, dt, diff, visc)
            want = cpu_baseline.fields
            if gpu_after is not None and want is not None:
                # the oracle is the checker here, as in tests/: same inputs, same size, same K, same number of steps
                names = ("u", "v", "w", "dens")
                linf = {n: float(np.max(np.abs(gpu_after[n].astype(np.float64) - want[n].astype(np.float64))))
                        for n in names}
                out["parity_in_run"] = {"steps": args.cpu_steps, "fields": list(names),
                                        "bit_exact": bool(all(np.array_equal(gpu_after[n], want[n]) for n in names)),
                                        "linf": max(linf.values()), "linf_per_field": linf,
                                        "checker": "serial CPU oracle (oracle/, the cpu_baseline leg's own result)"}
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    try:
        main()
    except SystemExit:
        raise
    except BaseException as exc:  # report the failure in the contract's shape instead of dying silently
        import traceback

        traceback.print_exc()
        if int(os.environ.get("RANK", "0")) == 0:
            print(json.dumps({"metric": "Mcells/s per vel_step+dens_step", "value": None, "unit": "Mcells/s",
                              "n_gpus": int(os.environ.get("WORLD_SIZE", "1")), "higher_is_better": True,
                              "error": f"{type(exc).__name__}: {exc}"[:500]}), flush=True)
        sys.exit(1)
