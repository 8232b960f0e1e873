"""Stand-in for fluidsolvergpu_amd.solver used ONLY by `SF_BENCH_DRYRUN=1 bench.py` in tests/test_dist_gloo.py: same
method names, no computation, no GPU. It lets the CPU suite execute bench.py's multi-rank control flow (rendezvous,
nccl-id hand-over, per-rank plane ranges, barriers, max-over-ranks, rank-0 JSON) which cannot run here otherwise."""
import numpy as np

from fluidsolvergpu_amd import dist as sfdist


def nccl_unique_id():
    return bytes(range(128))


class FluidSolver:
    def __init__(self, N, dtype="f32", iters=20, dt=0.1, diff=1e-4, visc=1e-4, device=0, nslabs_local=1, rank=0,
                 nranks=1, nccl_id=None):
        assert nranks == 1 or (isinstance(nccl_id, bytes) and len(nccl_id) == 128)
        self.N, self.rank, self.nranks, self.iters = N, rank, nranks, iters
        self.np_dtype = np.float32 if dtype == "f32" else np.float64
        self.calls = 0

    def __enter__(self):
        return self

    def __exit__(self, *a):
        pass

    def stored_planes(self):
        kb, ke = sfdist.slab_planes(self.N, self.rank, self.nranks)
        return max(kb - 2, 0), min(ke + 2, self.N + 2)

    def owned_planes(self):
        return sfdist.slab_planes(self.N, self.rank, self.nranks)

    def upload_planes(self, field, k_begin, array):
        assert array.shape[1:] == (self.N + 2, self.N + 2)

    def download_planes(self, field, kb, ke):
        return np.ones((ke - kb, self.N + 2, self.N + 2), self.np_dtype)

    def lin_solve_launches(self, iters):
        return iters // 2 + iters % 2

    def copy_bandwidth_gbps(self, nbytes=0, reps=0):
        return 1.0

    def timer_stop(self):
        return 1.0

    def __getattr__(self, name):  # set_bnd, sync, bind_sources, vel_step, dens_step, lin_solve, timer_start, close ...
        def noop(*a, **k):
            self.calls += 1
        return noop
