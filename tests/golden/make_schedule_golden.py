"""Writes the schedule traces tests/test_schedule_check.py checks on the CPU. Needs the MI355X (libsfgpu.so has no CPU
path): run through gpurun, then copy gpurun_out/schedule_golden/*.jsonl to tests/golden/schedule/.

    gpurun -- python tests/golden/make_schedule_golden.py

head_*   : the production schedule (this build), several slab counts / sweep mixes / transports, one full step each
inject_* : the same with the round-2 trapezoid bug re-introduced (SF_TRACE_SCHEDULE=<file>,inject=trap)
The traces are data (launch records of this repo's own library); nothing of the reference is involved."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "gpurun_out", "schedule_golden")
os.makedirs(OUT, exist_ok=True)
os.environ["SF_MARCH_MINCELLS_K"] = "100"  # let the marching kernel (and its four ghost planes) take these small grids

CASES = [  # tag, N, P, K, trap, flags (2 = rccl-self), extra env
    ("n96_p2_k11_trap8", 96, 2, 11, "8", 0, {}),
    ("n128_p2_k7_trap5", 128, 2, 7, "5", 0, {}),
    ("n128_p4_k20_trap3_rccl", 128, 4, 20, "3", 2, {}),
    ("n64_p4_k6_trap5", 64, 4, 6, "5", 0, {}),
    ("n96_p3_k20_trap0_hs2", 96, 3, 20, "0", 0, {"SF_HALO_STREAM": "2"}),
    ("n96_p2_k9_ghost2", 96, 2, 9, "5", 0, {"SF_GHOST": "2"}),
]


def run(tag, N, P, K, trap, flags, env, inject):
    path = os.path.join(OUT, f"{'inject' if inject else 'head'}_{tag}.jsonl")
    if os.path.exists(path):
        os.remove(path)
    os.environ["SF_TRACE_SCHEDULE"] = path + (",inject=trap" if inject else "")
    os.environ["SF_TRAP"] = trap
    for k, v in env.items():
        os.environ[k] = v
    from fluidsolvergpu_amd import solver as S

    rng = np.random.RandomState(5)
    kw = {"flags": flags} if flags else {}
    with S.FluidSolver(N, dtype="f32", iters=K, nslabs_local=P, **kw) as fs:
        for n in ("u", "v", "w", "dens", "user0", "user1", "user2", "user3"):
            scale = 0.2 if n in ("dens", "user3") else 0.5 / N
            fs.upload(n, (scale * rng.standard_normal((N + 2,) * 3)).astype(np.float32))
        fs.bind_sources("user0", "user1", "user2", "user3")
        fs.vel_step()
        fs.dens_step()
        fs.sync()
    for k in env:
        del os.environ[k]
    del os.environ["SF_TRACE_SCHEDULE"]
    print(path, os.path.getsize(path), "bytes", flush=True)


if __name__ == "__main__":
    for c in CASES:
        run(*c, inject=False)
    run(*CASES[0], inject=True)  # K = 11: a three-sweep launch continues the trapezoid block of a four-sweep one
