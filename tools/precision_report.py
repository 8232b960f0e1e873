"""tools/precision_report.py — the scale of meaningful error for the parity statement:
L-inf of (fp32 oracle - fp64 oracle) per field after 1, 10 and 100 steps on the benchmark inputs (docs/SPEC.md §5).
CPU only. The GPU path is bit-identical to the oracle of the same precision, so GPU-vs-oracle L-inf is exactly 0."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib as O  # noqa: E402
from bench import analytic_planes  # noqa: E402

N, K, dt, diff, visc = int(sys.argv[1]) if len(sys.argv) > 1 else 64, 20, 0.1, 1e-4, 1e-4
state = {}
for dtype in (np.float32, np.float64):
    f = analytic_planes(N, 0, N + 2, dt, dtype)
    s = {"u": f["u"], "v": f["v"], "w": f["w"], "dens": f["dens"]}
    for b, n in ((1, "u"), (2, "v"), (3, "w"), (0, "dens")):
        O.set_bnd(b, s[n])
    state[dtype] = (s, f)
done = 0
for target in (1, 10, 100):
    for dtype in (np.float32, np.float64):
        s, f = state[dtype]
        for _ in range(target - done):
            s.update({"u0": f["su"].copy(), "v0": f["sv"].copy(), "w0": f["sw"].copy(), "dens0": f["sd"].copy()})
            O.step(N, s, dtype(dt), dtype(diff), dtype(visc), K)
    done = target
    a, b = state[np.float32][0], state[np.float64][0]
    gaps = {n: float(np.max(np.abs(a[n].astype(np.float64) - b[n]))) for n in ("u", "v", "w", "dens")}
    scale = {n: float(np.max(np.abs(b[n]))) for n in ("u", "v", "w", "dens")}
    print(f"N={N} after {target:3d} steps: Linf(f32-f64) " + "  ".join(f"{n}={gaps[n]:.3e} (max|{n}|={scale[n]:.3g})" for n in gaps))
