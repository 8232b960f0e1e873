"""Bit-exact oracle parity at BASELINE.json's own sizes and iteration counts.

The dominant kernel instantiation of configs 2-5 — the plain four-sweep marching pass with non-temporal stores,
`jacobi_sk_kernel<T,1,WL,NT=true,4,TJ,8,false,0>` — only runs when K >= 12 and the working set exceeds the Infinity
Cache (N >~ 320). Every smaller test leaves it out, so these cases run the real sizes against the serial CPU oracle:
the oracle sweeps ~1 G cells/s, i.e. seconds to tens of seconds per case. Comparison: every cell of every field, shells
included, exact equality of the float bits (no tolerance).
"""
import os
import sys

import numpy as np
import pytest

import oracle_lib as O
from test_parity_gpu import DT, DIFF, VISC, assert_same, make, slab_kw, check_transport

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import analytic_planes  # noqa: E402  (the benchmark inputs, docs/SPEC.md §5)

pytestmark = pytest.mark.gpu


def bench_state(N, dtype):
    a = analytic_planes(N, 0, N + 2, DT, dtype)
    f = {"u": a["u"], "v": a["v"], "w": a["w"], "dens": a["dens"]}
    for b, n in ((1, "u"), (2, "v"), (3, "w"), (0, "dens")):
        O.set_bnd(b, f[n])
    src = {"u0": a["su"], "v0": a["sv"], "w0": a["sw"], "dens0": a["sd"]}
    return f, src


def run_bench_steps(N, dtype, K, steps):
    """`steps` x (vel_step + dens_step) of the benchmark inputs with bound sources: (gpu fields, oracle fields)."""
    f, src = bench_state(N, dtype)
    with make(N, dtype, K=K) as fs:
        for n in ("u", "v", "w", "dens"):
            fs.upload(n, f[n])
        for slot, n in (("user0", "u0"), ("user1", "v0"), ("user2", "w0"), ("user3", "dens0")):
            fs.upload(slot, src[n])
        fs.bind_sources("user0", "user1", "user2", "user3")
        for _ in range(steps):
            fs.vel_step()
            fs.dens_step()
        fs.sync()
        got = {n: fs.download(n) for n in ("u", "v", "w", "dens")}
    for _ in range(steps):
        f.update({n: src[n].copy() for n in src})
        O.step(N, f, dtype(DT), dtype(DIFF), dtype(VISC), K)
    return got, f


def test_config2_full_size():
    """BASELINE.json configs[1] / the bench workload itself: 256^3 fp32, K = 20, two steps of the benchmark inputs with
    bound sources (the second step starts from a developed state) — u, v, w, dens bit-identical to the oracle."""
    got, want = run_bench_steps(256, np.float32, 20, 2)
    for n in got:
        assert_same(got[n], want[n], f"config 2 (256^3 f32 K=20, 2 steps): {n}")
    assert np.isfinite(got["dens"]).all() and got["dens"].max() > 1.0


def test_mid_size_default_thresholds():
    """144^3 fp32, K = 20, default switches: since round 3 an undecomposed grid of >= 2.5 M cells (here 3.0 M) runs every
    pass of its solves on the four-sweep marching kernel (SF_MARCH_MINCELLS_K default 2500; 6 M before) — the smallest
    size class that does, with the most chunk ends per plane. One step of the benchmark inputs against the oracle."""
    got, want = run_bench_steps(144, np.float32, 20, 1)
    for n in got:
        assert_same(got[n], want[n], f"144^3 f32 K=20, default thresholds: {n}")
    with make(144, np.float32, K=20) as fs:
        assert fs.lin_solve_launches(20) == 5  # five four-sweep marching launches, no pair launch


def test_fields_in_one_grid_default_thresholds_caller_sources():
    """192^3 fp32, K = 20, default switches, sources uploaded by the caller (no bind_sources): one field's x + x0 + x' take
    89 MB, inside the first window of Solver::batch_march, so u, v, w are diffused as ONE marching grid — add_source as its
    own kernel, one caller-data first pass per field, then plain passes over the three fields. One step against the
    oracle (test_config2_full_size covers the bound-source form at 256^3, in the second window)."""
    N, K, dtype = 192, 20, np.float32
    f, src = bench_state(N, dtype)
    with make(N, dtype, K=K) as fs:
        for n in ("u", "v", "w", "dens"):
            fs.upload(n, f[n])
        for n in src:
            fs.upload(n, src[n])
        fs.vel_step()
        fs.dens_step()
        fs.sync()
        got = {n: fs.download(n) for n in ("u", "v", "w", "dens")}
    f.update({n: src[n].copy() for n in src})
    O.step(N, f, dtype(DT), dtype(DIFF), dtype(VISC), K)
    for n in got:
        assert_same(got[n], f[n], f"192^3 f32 K=20, caller sources, default thresholds: {n}")


def roofline_inputs(N, dtype):
    """The inputs of bench.py's roofline leg (time_lin_solve): one random plane scaled per k."""
    rng = np.random.RandomState(1)
    plane = rng.standard_normal((1, N + 2, N + 2)).astype(dtype)
    k = np.arange(N + 2, dtype=np.float64)[:, None, None]
    x = (plane * (1.0 + 0.001 * k)).astype(dtype)
    x0 = (plane * (0.5 - 0.001 * k)).astype(dtype)
    return x, x0


@pytest.mark.parametrize("N,dtype,K", [(512, np.float32, 20), (512, np.float64, 40)], ids=["512-f32-K20", "512-f64-K40"])
def test_config3_lin_solve_full_size(N, dtype, K):
    """configs[2] / configs[4]: the lin_solve of the HBM roofline entry — 512^3, K = 20 in fp32 on the very inputs
    bench.py times, K = 40 in fp64 — against the oracle. Five (ten) four-sweep marching launches, non-temporal stores."""
    x, x0 = roofline_inputs(N, dtype)
    a, c = 0.3, 1 + 6 * 0.3
    with make(N, dtype, K=K) as fs:
        fs.upload("dens", x)
        fs.upload("dens0", x0)
        assert fs.lin_solve_launches(K) == K // 4  # every pass is a four-sweep marching launch
        fs.lin_solve(0, "dens", "dens0", a, c, K)
        fs.sync()
        got = fs.download("dens")
    O.lin_solve(0, x, x0, dtype(a), dtype(c), K)
    assert_same(got, x, f"{N}^3 lin_solve K={K}")


def test_config3_step_full_size():
    """configs[2]: one full vel_step + dens_step at 512^3 fp32 with K = 40 Jacobi iterations per solve (benchmark
    inputs, bound sources) against the oracle (~40 s of CPU)."""
    got, want = run_bench_steps(512, np.float32, 40, 1)
    for n in got:
        assert_same(got[n], want[n], f"config 3 (512^3 f32 K=40): {n}")


@pytest.mark.parametrize("N,dtype,K", [(1024, np.float32, 20), (512, np.float64, 40)], ids=["config4-1024-f32-K20", "config5-512-f64-K40"])
def test_decomposed_configs_lin_solve_eight_slabs_vs_oracle(N, dtype, K):
    """configs[3] and configs[4] (8 GPUs) of BASELINE.json: the 1024^3 fp32 / 512^3 fp64 grids in eight k-slabs (logical
    slabs on one GPU, ghost planes through RCCL send/recv on a single-rank communicator), lin_solve with the config's own
    K against the ORACLE (not against the one-slab run): 128- / 64-plane slabs, four ghost planes, first pass + plain /
    last four-sweep passes, trapezoid boundary launches."""
    x, x0 = roofline_inputs(N, dtype)
    a, c = 0.3, 1 + 6 * 0.3
    with make(N, dtype, K=K, **slab_kw("rccl-self", 8)) as fs:
        for k0 in range(0, N + 2, 64):  # plane batches: the driver's upload path for grids of this size
            k1 = min(k0 + 64, N + 2)
            fs.upload_planes("dens", k0, x[k0:k1])
            fs.upload_planes("dens0", k0, x0[k0:k1])
        fs.lin_solve(0, "dens", "dens0", a, c, K)
        fs.sync()
        check_transport(fs, "rccl-self", 8)
        got = fs.download("dens")
    O.lin_solve(0, x, x0, dtype(a), dtype(c), K)
    assert_same(got, x, f"{N}^3 lin_solve K={K}, 8 slabs (rccl-self) vs oracle")
