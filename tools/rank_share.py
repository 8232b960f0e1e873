#!/usr/bin/env python3
"""One rank's share of an R-rank run, timed on ONE GPU (measurement aid, not the benchmark).

The context is created as rank r of R with SF_FLAG_LOOPBACK_HALO: same slab geometry, ghost planes, streams,
boundary/interior launches and halo-message sizes as in the real multi-process run, but the messages to the
neighbouring ranks are device-local copies (include/sfgpu.h). What it shows: the per-rank compute + launch time
that bounds the weak-scaling efficiency before any xGMI cost. Field values next to the slab faces are meaningless.

  python tools/rank_share.py --ranks 8            # grid from bench.py's WEAK_GRID, middle rank
  python tools/rank_share.py --ranks 2 --grid 320 --rank 0
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (WEAK_GRID, analytic_planes)
from fluidsolvergpu_amd import solver as S  # noqa: E402

SF_FLAG_LOOPBACK_HALO = 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, required=True)
    ap.add_argument("--rank", type=int, default=-1)
    ap.add_argument("--grid", type=int, default=0)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--dtype", default="f32")
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    a = ap.parse_args()
    R = a.ranks
    N = a.grid or bench.WEAK_GRID[R]
    r = a.rank if a.rank >= 0 else R // 2
    fs = S.FluidSolver(N, dtype=a.dtype, iters=a.iters, rank=r, nranks=R, flags=SF_FLAG_LOOPBACK_HALO if R > 1 else 0)
    kb, ke = fs.stored_planes()
    f = bench.analytic_planes(N, kb, ke, 0.1, fs.np_dtype)
    for name, slot in (("u", "u"), ("v", "v"), ("w", "w"), ("dens", "dens"), ("su", "user0"), ("sv", "user1"),
                       ("sw", "user2"), ("sd", "user3")):
        fs.upload_planes(slot, kb, f[name])
    fs.bind_sources("user0", "user1", "user2", "user3")
    fs.sync()
    for _ in range(a.warmup):
        fs.vel_step()
        fs.dens_step()
    fs.sync()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        fs.vel_step()
        fs.dens_step()
    fs.sync()
    ms = (time.perf_counter() - t0) / a.steps * 1e3
    cells_rank = float(N) ** 3 / R
    print(json.dumps({"emulation": "rank-share (loopback halo)", "ranks": R, "rank": r, "grid": N, "ms_per_step": ms,
                      "mcells_per_s_per_rank": cells_rank / ms / 1e3,
                      "implied_aggregate_mcells_per_s": cells_rank * R / ms / 1e3,
                      "schedule": fs.schedule_info()}), flush=True)
    fs.close()


if __name__ == "__main__":
    main()
