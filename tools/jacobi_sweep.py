"""tools/jacobi_sweep.py — time the lin_solve sweep (HIP events) for the current build.
usage: python tools/jacobi_sweep.py [N ...]; environment switches are read by libsfgpu.so."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from fluidsolvergpu_amd import solver as S


def run(N, dtype="f32", K=int(os.environ.get("SF_SWEEP_K", 20)), reps=int(os.environ.get("SF_SWEEP_REPS", 4))):
    with S.FluidSolver(N, dtype=dtype, iters=K) as fs:
        rng = np.random.RandomState(1)
        plane = rng.standard_normal((1, N + 2, N + 2)).astype(fs.np_dtype)
        for k in range(N + 2):
            fs.upload_planes("dens", k, plane * (1 + 0.001 * k))
            fs.upload_planes("dens0", k, plane * (0.5 - 0.001 * k))
        fs.lin_solve(0, "dens", "dens0", 0.3, 2.8, 2)
        fs.sync()
        best = 1e30
        for _ in range(reps):
            fs.timer_start()
            fs.lin_solve(0, "dens", "dens0", 0.3, 2.8, K)
            best = min(best, fs.timer_stop() * 1e3 / K)
        w = 4 if dtype == "f32" else 8
        gbs = N ** 3 * 3 * w / (best * 1e-6) / 1e9
        print(f"N={N} {dtype} {os.environ.get('SF_TAG','')}: {best:8.1f} us/sweep  {gbs:7.0f} GB/s algorithmic  "
              f"({gbs / 80:.1f}% of 8 TB/s)", flush=True)


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [512]
    for n in sizes:
        run(n, dtype=os.environ.get("SF_SWEEP_DTYPE", "f32"))
