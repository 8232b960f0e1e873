"""Multi-process plumbing of the slab-decomposed solver: one process per GPU, launched by
`python -m torch.distributed.run` (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in the environment).

torch.distributed is used for the CONTROL plane only (rendezvous, shipping the ncclUniqueId, barriers,
gathering results for output, max-over-ranks timing). The data plane — ghost-plane exchange per Jacobi
sweep — is RCCL send/recv inside libsfgpu.so (csrc/sf_solver.hpp, exchange()). Everything here also runs
on the gloo backend with no GPU, which is how tests/test_dist_gloo.py covers the N > 1 host path.

The reference's counterpart is its two-device scaffolding (solver-unidyn.cu:79-96 split of the cell
range, :187 one-plane ghost `buffer`, :396-470 host-staged exchange); this module is the P-way,
one-process-per-GPU form of the same split.
"""
import os

import numpy as np


def env_world():
    """(rank, local_rank, world_size) from the torch.distributed.run environment (1 process: 0, 0, 1)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def slab_planes(N, rank, world):
    """Global interior planes [k_begin, k_end) owned by `rank` (1-based k), N divisible by world.
    Matches sf_owned_planes(): slab g owns k = g*N/P + 1 .. (g+1)*N/P."""
    if N % world != 0:
        raise ValueError(f"N={N} is not divisible by world={world}")
    nzl = N // world
    return rank * nzl + 1, (rank + 1) * nzl + 1


def stored_planes(N, rank, world):
    """Planes a rank stores: its interior planes plus one ghost (or physical shell) plane each side."""
    kb, ke = slab_planes(N, rank, world)
    return kb - 1, ke + 1


def output_planes(N, rank, world):
    """Planes a rank contributes to a gathered global field: its interior planes, plus the physical
    shell plane k = 0 on the first rank and k = N+1 on the last."""
    kb, ke = slab_planes(N, rank, world)
    return (kb - 1 if rank == 0 else kb), (ke + 1 if rank == world - 1 else ke)


def init(backend="gloo"):
    """Initialise torch.distributed from the environment; returns the module, or None for 1 process."""
    rank, _, world = env_world()
    if world == 1:
        return None
    import torch.distributed as dist

    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29531")
    if not dist.is_initialized():
        # gloo announces its connections on STDOUT ("[Gloo] Rank 0 is connected to ..."), and bench.py owes its caller
        # exactly one JSON line there: while the group is set up (and its connections made, by a first barrier),
        # file descriptor 1 points at stderr
        import sys

        sys.stdout.flush()
        saved = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend, rank=rank, world_size=world)
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved, 1)
            os.close(saved)
    return dist


def share_nccl_id(dist, make_id):
    """Rank 0 calls make_id() (-> 128 bytes from sf_nccl_unique_id); every rank returns those bytes."""
    if dist is None:
        return None
    box = [make_id() if dist.get_rank() == 0 else None]
    dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], (bytes, bytearray)) or len(box[0]) != 128:
        raise RuntimeError("nccl id broadcast failed")
    return bytes(box[0])


def max_over_ranks(dist, value):
    """The slowest rank's time — what bench.py reports."""
    if dist is None:
        return float(value)
    import torch

    t = torch.tensor([float(value)], dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_field(dist, N, local_planes, dtype):
    """Assemble the global (N+2)^3 field on rank 0 from each rank's output_planes() block.
    local_planes: array (k_end-k_begin, N+2, N+2) holding this rank's output_planes(). Other ranks get None."""
    if dist is None:
        return np.ascontiguousarray(local_planes, dtype=dtype)
    import torch

    rank, world = dist.get_rank(), dist.get_world_size()
    kb, ke = output_planes(N, rank, world)
    local = np.ascontiguousarray(local_planes, dtype=dtype)
    if local.shape != (ke - kb, N + 2, N + 2):
        raise ValueError(f"rank {rank}: expected {(ke - kb, N + 2, N + 2)}, got {local.shape}")
    parts = [None] * world if rank == 0 else None
    dist.gather_object(local, parts, dst=0)
    if rank != 0:
        return None
    out = np.empty((N + 2,) * 3, dtype)
    for r, part in enumerate(parts):
        b, e = output_planes(N, r, world)
        out[b:e] = part
    return out
