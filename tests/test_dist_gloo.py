"""N > 1 host path on CPU: world_size 2 (and 4) over gloo, launched exactly like bench.py is
(`python -m torch.distributed.run`, rendezvous on 127.0.0.1)."""
import os
import socket
import subprocess
import sys

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("world,N,K,dtype", [(2, 8, 3, "f32"), (2, 6, 2, "f64"), (4, 8, 2, "f32")])
def test_slab_schedule_over_gloo(tmp_path, world, N, K, dtype):
    out = tmp_path / "result.txt"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "dist_worker.py"), str(out), str(N), str(K), dtype]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out.read_text() == "OK"


# The production exchange schedule over gloo: ghost depth and sweeps per exchange from the library's
# own rules (tests/slab_emulator.Schedule restates them), recomputed ghost-plane levels, folded sources with their
# right-hand side on G - 1 ghost planes; planes the schedule never makes valid hold NaN in the emulator, so a read of
# one fails the comparison with the undecomposed oracle. mincells_k lowers the marching kernel's size threshold the way
# SF_MARCH_MINCELLS_K does, so that small grids take the deep schedules (G = 3, 4; S = 3, 4).
PROD_CASES = [
    # world, N, K, dtype, mincells_k, bound     -> expected ghost depth
    (2, 48, 7, "f32", 10, "0"),    # nzl 24: G = 4 (interior 16 >= 12), passes 4 + 3
    (2, 64, 11, "f32", 10, "1"),   # nzl 32: G = 4, first pass 4 (folded source) + 4 + 3
    (2, 64, 20, "f32", 10, "0"),   # G = 4, 4 + 4 x 4
    (4, 96, 9, "f32", 10, "1"),    # nzl 24: G = 4, bound sources: first pass 4 (folded) + 3 + 2
    (4, 32, 6, "f64", 2500, "1"),  # default thresholds: G = 2, pairs only
    (2, 40, 5, "f64", 10, "0"),    # nzl 20: G = 4 (interior 12), odd K below the first-pass threshold: 2 + 3
    (2, 12, 4, "f32", 2500, "1"),  # thin slabs (nzl 6): G = 2
    (2, 36, 8, "f32", 10, "1"),    # nzl 18: G = 3 (interior 12; four ghost planes would leave 10), pair + 3 + 3
]


@pytest.mark.parametrize("world,N,K,dtype,mincells_k,bound", PROD_CASES)
def test_production_schedule_over_gloo(tmp_path, world, N, K, dtype, mincells_k, bound):
    out = tmp_path / "result.txt"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "dist_worker.py"), str(out), str(N), str(K), dtype, "prod", str(mincells_k), bound]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out.read_text() == "OK"


@pytest.mark.parametrize("world,N,K,trace", [(2, 96, 11, "head_n96_p2_k11_trap8.jsonl"), (2, 128, 7, "head_n128_p2_k7_trap5.jsonl"),
                                             (4, 64, 6, "head_n64_p4_k6_trap5.jsonl")])
def test_emulator_exchange_sequence_equals_the_device_trace(tmp_path, world, N, K, trace):
    """Every rank of the gloo emulation issues the exchanges libsfgpu.so recorded on the MI355X for the same (N, P, K)
    (tests/golden/schedule/, SF_TRACE_SCHEDULE with SF_MARCH_MINCELLS_K=100): same number, order, ghost depth and field
    slots — and the emulated result still equals the undecomposed oracle."""
    out = tmp_path / "result.txt"
    env = dict(os.environ, OMP_NUM_THREADS="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()),
           os.path.join(HERE, "dist_worker.py"), str(out), str(N), str(K), "f32", "prod", "100", "1",
           os.path.join(HERE, "golden", "schedule", trace)]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    assert out.read_text() == "OK"


def test_schedule_rules_restated():
    """tests/slab_emulator.Schedule against hand-derived values of the library's rules."""
    sys.path.insert(0, HERE)
    from slab_emulator import Schedule

    s = Schedule(1024, 8)  # configs[3]: 128-plane slabs
    assert s.G == 4 and s.passes(20) == [4, 4, 4, 4, 4] and s.sk_first_ok(20)
    s = Schedule(512, 8, wsize=8)  # configs[4] over 8 ranks: 64-plane slabs of fp64
    assert s.G == 4 and s.passes(40) == [4] * 10
    s = Schedule(256, 1)
    assert s.G == 1 and s.passes(20) == [4] * 5 and s.passes(6) == [2, 4] and s.passes(9) == [4, 3, 2]
    s = Schedule(128, 1)  # below the marching kernel's size threshold: pairs
    assert s.passes(5) == [2, 2, 1]
    s = Schedule(64, 2, march_mincells_k=10)
    assert s.G == 4 and s.passes(11) == [4, 4, 3] and s.passes(7, continued=True) == [4, 3]
    assert Schedule(64, 2, march_mincells_k=10, ishell=0).passes(11) == [2, 2, 2, 2, 2, 1]
    assert Schedule(30, 2).G == 1  # N not a multiple of the vector width: single sweeps, one ghost plane


def _emulate(N, world, K, bound, defect, mincells_k=10, dtype="f32"):
    import numpy as np

    sys.path.insert(0, HERE)
    import oracle_lib as O
    import slab_emulator as E

    dt_ = np.float32 if dtype == "f32" else np.float64
    rng = np.random.RandomState(5)
    names = ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")
    glob = {n: (0.2 * rng.standard_normal((N + 2,) * 3)).astype(dt_) for n in names}
    lim = 0.9 / (0.1 * N)
    for n in names[:6]:
        glob[n] = np.clip(glob[n], -lim / 4, lim / 4).astype(dt_)
    try:
        got, log = E.run_threads(N, world, dt_, K, glob, {"march_mincells_k": mincells_k}, bound=bound, defect=defect)
    except RuntimeError:  # a NaN velocity trips the emulator's back-trace guard: the defect was noticed
        assert defect is not None
        return False, None
    want = {n: glob[n].copy() for n in names}
    O.step(N, want, dt_(0.1), dt_(1e-4), dt_(1e-4), K)
    return all(np.array_equal(got[n], want[n]) for n in got), log


def test_emulator_notices_schedule_defects():
    """The production-schedule emulator has teeth: the healthy schedule reproduces the oracle bit for bit; leaving the
    folded right-hand side off the ghost planes, or exchanging one plane too few, does not (NaN reaches the result)."""
    ok, log = _emulate(64, 2, 11, True, None)
    assert ok and log[0] == (0, 4, (0,)) and len(log) == 3 * 3 + 2 * (1 + 3 + 1) + 1 + 3 + 1
    assert not _emulate(64, 2, 11, True, "rhs_ghost")[0]
    assert not _emulate(64, 2, 11, True, "shallow_exchange")[0]
    assert not _emulate(64, 2, 11, False, "shallow_exchange")[0]
    assert _emulate(48, 2, 7, False, "shallow_exchange")[0]  # passes 4 + 3 on four ghost planes: the last one is never read


def test_partition_arithmetic():
    from fluidsolvergpu_amd import dist as d

    N, P = 512, 8
    seen = []
    for r in range(P):
        kb, ke = d.slab_planes(N, r, P)
        seen += list(range(kb, ke))
        ob, oe = d.output_planes(N, r, P)
        assert (ob == 0) == (r == 0) and (oe == N + 2) == (r == P - 1)
    assert seen == list(range(1, N + 1))
    with pytest.raises(ValueError):
        d.slab_planes(10, 0, 3)


@pytest.mark.parametrize("world,mode", [(2, "strong"), (4, "strong"), (2, "weak")])
def test_bench_multirank_plumbing_dryrun(world, mode):
    """bench.py --gpus N exactly as the driver launches it, with the solver replaced by a stub (SF_BENCH_DRYRUN=1):
    rank 0 must print ONE JSON line with the contract fields, n_gpus == N, tagged as a dry run. Default for N > 1 is the
    fixed grid ("scaling": "strong") followed by the one-GPU run of the same grid on rank 0 (`single_gpu`, `speedup`);
    `--weak` keeps the per-GPU cell count instead and reports "weak"."""
    import json

    root = os.path.dirname(HERE)
    env = dict(os.environ, OMP_NUM_THREADS="1", SF_BENCH_DRYRUN="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.join(root, "bench.py"),
           "--gpus", str(world), "--steps", "2", "--warmup", "1", "--grid", str(8 * world), "--roofline-n", "8"]
    if mode == "weak":
        cmd.append("--weak")
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    # nothing but that line on stdout (gloo's "[Gloo] Rank .. is connected" chatter is steered to stderr)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                "vs_baseline", "dtype", "data", "config", "roofline", "single_gpu", "speedup"):
        assert key in d, key
    assert d["n_gpus"] == world and d["steps"] == 2 and d["scaling"] == mode and d["vs_baseline"] is None
    assert d["single_gpu"]["grid"] == 8 * world and d["single_gpu"]["steps"] == 3
    assert abs(d["speedup"] - d["value"] / d["single_gpu"]["value"]) < 1e-9
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic", "grid", "us_per_launch", "sweeps_per_launch"):
        assert key in d["roofline"], key
    assert d["data"].startswith("DRYRUN") and "workload" in d["config"]
    assert "cpu_baseline" not in d  # only reported at N = 1
