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
