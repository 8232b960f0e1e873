"""Debug helper: one lin_solve case against the oracle, with the layout of the differing cells."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import oracle_lib as O
from fluidsolvergpu_amd import solver

N, K, b = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
dtype = np.float64 if sys.argv[4] == "f64" else np.float32
rng = np.random.RandomState(4)
f = {n: (0.2 * rng.standard_normal((N + 2,) * 3)).astype(dtype) for n in ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")}
a, c = 0.37, 1 + 6 * 0.37
with solver.FluidSolver(N, dtype=sys.argv[4], iters=K, dt=0.1, diff=1e-4, visc=1e-4) as fs:
    fs.upload("dens", f["dens"]); fs.upload("dens0", f["dens0"])
    fs.lin_solve(b, "dens", "dens0", a, c, K)
    got = fs.download("dens")
want = f["dens"].copy()
O.lin_solve(b, want, f["dens0"], dtype(a), dtype(c), K)
bad = np.argwhere(got != want)
print(f"N={N} K={K} b={b} {sys.argv[4]} env={ {k: v for k, v in os.environ.items() if k.startswith('SF_')} }: {len(bad)} differ")
if len(bad):
    for ax, nm in enumerate("kji"):
        vals, cnt = np.unique(bad[:, ax], return_counts=True)
        print("  ", nm, dict(zip(vals.tolist(), cnt.tolist())))
if len(bad) and os.environ.get("DBG_STAGES"):
    for kk in range(1, K + 1):
        w = f["dens"].copy()
        O.lin_solve(b, w, f["dens0"], dtype(a), dtype(c), kk)
        for jj in sorted(set(bad[:, 1].tolist())):
            same = np.array_equal(got[:, jj, :], w[:, jj, :])
            same_in = np.array_equal(got[1:-1, jj, 1:-1], w[1:-1, jj, 1:-1])
            print(f"   row j={jj}: equals oracle after {kk} sweeps: all={same} interior(k,i)={same_in}")
if len(bad) and os.environ.get("DBG_MATCH"):
    for jj in sorted(set(bad[:, 1].tolist())):
        row = got[:, jj, :]
        print(f"   row j={jj}: got[1,{jj},1:5]={row[1,1:5]} want={want[1,jj,1:5]}")
        for kk in range(0, K + 1):
            w = f["dens"].copy()
            if kk: O.lin_solve(b, w, f["dens0"], dtype(a), dtype(c), kk)
            for j2 in range(N + 2):
                for dk in (-1, 0, 1):
                    if dk == 0: m = np.array_equal(row[2:-2, 1:-1], w[2:-2, j2, 1:-1])
                    elif dk == 1: m = np.array_equal(row[2:-2, 1:-1], w[3:-1, j2, 1:-1])
                    else: m = np.array_equal(row[2:-2, 1:-1], w[1:-3, j2, 1:-1])
                    if m: print(f"      == oracle after {kk} sweeps, row j={j2}, plane shift {dk}")
