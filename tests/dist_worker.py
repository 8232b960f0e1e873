"""Worker of tests/test_dist_gloo.py: run under `python -m torch.distributed.run --nproc-per-node P` on CPU.
Checks the N > 1 host path (fluidsolvergpu_amd/dist.py) and the slab exchange schedule (slab_emulator) over
gloo against the undecomposed CPU oracle. Writes "OK" to the file given as argv[1] on rank 0.

argv: out N K dtype [mode [mincells_k [bound [trace]]]] - mode "g1": one ghost plane, one exchange per sweep (round 1);
"prod": the production schedule (G = 2..4 ghost planes, S sweeps per exchange, folded sources when bound = 1) with the
marching-kernel thresholds of the library (mincells_k = SF_MARCH_MINCELLS_K); trace: a golden SF_TRACE_SCHEDULE file
whose exchange sequence every rank's emulator must reproduce entry by entry."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))

import torch  # noqa: E402
from fluidsolvergpu_amd import dist as sfdist  # noqa: E402
import oracle_lib as O  # noqa: E402
import slab_emulator  # noqa: E402


def main():
    out_path = sys.argv[1]
    N, K = int(sys.argv[2]), int(sys.argv[3])
    dtype = np.float32 if sys.argv[4] == "f32" else np.float64
    rank, _, world = sfdist.env_world()
    dist = sfdist.init("gloo")
    assert dist is not None and dist.get_world_size() == world

    # control plane: the id made on rank 0 reaches every rank unchanged
    secret = bytes(range(128))
    got = sfdist.share_nccl_id(dist, lambda: secret)
    assert got == secret
    assert sfdist.max_over_ranks(dist, 1.0 + rank) == float(world)
    kb, ke = sfdist.slab_planes(N, rank, world)
    assert ke - kb == N // world and sfdist.stored_planes(N, rank, world) == (kb - 1, ke + 1)
    mode = sys.argv[5] if len(sys.argv) > 5 else "g1"
    mincells_k = int(sys.argv[6]) if len(sys.argv) > 6 else 2500
    bound = len(sys.argv) > 7 and sys.argv[7] == "1"
    trace = sys.argv[8] if len(sys.argv) > 8 else None

    # identical global inputs on every rank (seeded), velocities small enough for a one-plane halo
    rng = np.random.RandomState(5)
    names = ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")
    glob = {n: (0.2 * rng.standard_normal((N + 2,) * 3)).astype(dtype) for n in names}
    lim = 0.9 / (0.1 * N)
    for n in names[:6]:
        glob[n] = np.clip(glob[n], -lim / 4, lim / 4).astype(dtype)

    def exchange(send_lo, send_hi):
        """Ghost-plane exchange over gloo (G planes per message): same pairing as the RCCL group of exchange() in
        sf_solver.hpp."""
        reqs, recv_lo, recv_hi = [], None, None
        if send_lo is not None:
            recv_lo = torch.empty(send_lo.shape, dtype=torch.from_numpy(send_lo).dtype)
            reqs.append(dist.isend(torch.from_numpy(send_lo), rank - 1))
            reqs.append(dist.irecv(recv_lo, rank - 1))
        if send_hi is not None:
            recv_hi = torch.empty(send_hi.shape, dtype=torch.from_numpy(send_hi).dtype)
            reqs.append(dist.isend(torch.from_numpy(send_hi), rank + 1))
            reqs.append(dist.irecv(recv_hi, rank + 1))
        for r in reqs:
            r.wait()
        return (None if recv_lo is None else recv_lo.numpy()), (None if recv_hi is None else recv_hi.numpy())

    sched = None
    if mode == "prod":
        sched = slab_emulator.Schedule(N, world, np.dtype(dtype).itemsize, march_mincells_k=mincells_k)
    slab = slab_emulator.Slab(N, rank, world, dtype, exchange, sched)
    local = {n: slab.local(glob[n]) for n in names}
    srcs = {n: slab.local(glob[n]) for n in ("u0", "v0", "w0", "dens0")} if bound else None
    local = slab.step(local, 0.1, 1e-4, 1e-4, K, bound=srcs)

    if trace:  # the device's own record of the same (N, P, K): same exchanges, same order, same depth, same fields
        import schedule_check as SC

        want = SC.exchange_sequence(trace)[0]
        assert slab.xchg_log == want, (len(slab.xchg_log), len(want),
                                       [(a, b) for a, b in zip(slab.xchg_log, want) if a != b][:5])

    ob, oe = sfdist.output_planes(N, rank, world)
    gathered = {}
    for n in ("u", "v", "w", "dens"):
        part = slab.owned(local[n])
        assert part.shape[0] == oe - ob
        gathered[n] = sfdist.gather_field(dist, N, part, dtype)

    if rank == 0:
        assert slab.G == (sched.G if sched else 1)
        O.step(N, glob, dtype(0.1), dtype(1e-4), dtype(1e-4), K)
        for n in gathered:
            assert np.array_equal(gathered[n], glob[n]), f"{n}: slab schedule differs from the undecomposed oracle"
        open(out_path, "w").write("OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
