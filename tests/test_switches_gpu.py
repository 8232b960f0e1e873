"""The SF_* switches that change kernels or launch schedules, each run against the CPU oracle (bit for bit) inside
`pytest -m gpu`, so the driver's GPU run covers more than the defaults (tools/switch_matrix.sh repeats a larger
subset of the suite under every documented switch). libsfgpu.so reads the switches at context creation."""
import numpy as np
import pytest

import oracle_lib as O
from test_parity_gpu import DT, DIFF, VISC, NAMES, assert_same, make, rand_fields, small_velocity

pytestmark = pytest.mark.gpu

SWITCHES = [
    {},                             # defaults (four-sweep marching kernel, four ghost planes on slabs)
    {"SF_MARCH": "0"},              # register-blocked pair kernel everywhere
    {"SF_SK_S": "2"}, {"SF_SK_S": "3"},  # marching kernel limited to two / three sweeps per pass
    {"SF_SK_FIRST": "0"},           # first pass of a solve through the register-blocked pair kernel
    {"SF_FUSE2": "0"},              # single sweeps, one ghost plane
    # i-shell read and written by every sweep: no marching launch anywhere (round 2 gated that on one slab only and a
    # decomposed solve mixed marching passes with pair passes: the SF_ERR_HALO_EXCEEDED of gpurun_out/ish.log); the
    # slabs keep their four ghost planes, so every pass is a pair launch of boundary depth four
    {"SF_ISHELL": "0"}, {"SF_ISHELL": "0", "SF_GHOST": "3"}, {"SF_ISHELL": "0", "SF_TRAP": "3"},
    {"SF_ISHELL": "2"},             # as the default, but every solve's last pass writes its i-shell (round 2's rule)
    {"SF_NT": "1"}, {"SF_NT": "0"},  # non-temporal stores everywhere / nowhere (default: beyond the Infinity Cache)
    {"SF_ADVECT_ROW": "0"}, {"SF_ADVECT_ROW": "2"}, {"SF_ADVECT_ROW": "3"},  # advect: one form for every call
    {"SF_OVL": "0"}, {"SF_OVL": "2"},
    {"SF_TRAP": "0"}, {"SF_TRAP": "2"}, {"SF_TRAP": "5"},
    {"SF_HALO_STREAM": "1"}, {"SF_HALO_STREAM": "2"},
    {"SF_SPLIT_FIELDS": "0"}, {"SF_SPLIT_FIELDS": "2"},
    {"SF_GHOST": "1"}, {"SF_GHOST": "2"}, {"SF_GHOST": "3"},
    {"SF_FUSE_SRC": "0"}, {"SF_ZERO_SKIP": "0"}, {"SF_SPLIT": "0"},
    {"SF_MARCH_MINP": "4"},         # thin slabs through the marching kernel
    {"SF_AUTOTUNE": "0"},           # rccl-self contexts keep the default schedule instead of measuring one
    {"SF_MARCH": "0", "SF_TRAP": "3", "SF_HALO_STREAM": "2", "SF_SPLIT_FIELDS": "0"},
]
# (SF_GRAPH: test_parity_gpu.py::test_graph_replay_matches; SF_MARCH_MINCELLS_K: the march_mode fixture there and below;
# SF_TRACE_SCHEDULE: tests/test_schedule_trace.py)


def run_case(N, P, K, steps, transport="copy", dtype=np.float32):
    f = small_velocity(rand_fields(N, dtype, 300 + N + P), N, dtype)
    src = {n: f[n].copy() for n in ("u0", "v0", "w0", "dens0")}
    kw = {"nslabs_local": P}
    if transport == "rccl-self" and P >= 2:
        kw["flags"] = 2
    with make(N, dtype, K=K, **kw) as fs:
        for n in NAMES:
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
        for n in src:
            f[n][...] = src[n]
        O.step(N, f, dtype(DT), dtype(DIFF), dtype(VISC), K)
    return got, f


@pytest.mark.parametrize("env", SWITCHES, ids=lambda e: ",".join(f"{k[3:]}={v}" for k, v in e.items()) or "defaults")
def test_switch_settings_against_the_oracle(env, monkeypatch):
    for k, v in env.items():
        monkeypatch.setenv(k, v)
    monkeypatch.setenv("SF_MARCH_MINCELLS_K", "100")  # let the marching kernel take these small grids too
    # the last two: deep slabs (four ghost planes) whose solves are pair launches only by default (K < 7) - the schedule
    # round 2's SF_ISHELL=0 failure was wrongly blamed on
    for N, P, K, steps, transport in ((64, 1, 7, 2, "copy"), (72, 1, 20, 1, "copy"), (64, 4, 6, 1, "copy"), (96, 2, 11, 1, "rccl-self"),
                                      (96, 3, 20, 1, "copy"), (96, 2, 4, 1, "copy"), (96, 2, 6, 2, "rccl-self")):
        got, want = run_case(N, P, K, steps, transport)
        for n in got:
            assert_same(got[n], want[n], f"{env} N={N} P={P} K={K} {transport}: {n}")


@pytest.mark.parametrize("sweeps", ["2", "3"])
def test_two_and_three_sweep_marching_fp64(sweeps, monkeypatch):
    """The S = 2 / S = 3 instantiations of the marching kernel in fp64 (four-slot rings instead of shift registers: the
    k-wall planes of round 3 sit in different slots there), one slab and three, against the oracle."""
    monkeypatch.setenv("SF_SK_S", sweeps)
    monkeypatch.setenv("SF_MARCH_MINCELLS_K", "100")
    for N, P, K, steps, transport in ((64, 1, 7, 2, "copy"), (96, 3, 20, 1, "copy"), (72, 1, 9, 1, "copy")):
        got, want = run_case(N, P, K, steps, transport, dtype=np.float64)
        for n in got:
            assert_same(got[n], want[n], f"SK_S={sweeps} f64 N={N} P={P} K={K}: {n}")
