"""Checks on the CPU oracle itself (oracle/stable_fluids_oracle.hpp).

The reference holds no golden vectors for this path ("parity unpinned"), so the oracle is
pinned by (a) an independent numpy restatement of docs/SPEC.md written without looking at the C++ loops'
structure (vectorised slicing instead of triple loops) — bit-exact agreement required — and (b) the
known-answer properties of SURVEY.md §8c."""
import numpy as np
import pytest

import oracle_lib as O

DT = 0.1


# ---------------------------------------------------------------------------------- numpy restatement
def np_set_bnd(b, x):
    N = x.shape[0] - 2
    t = x.dtype.type
    sx, sy, sz = (t(-1) if b == 1 else t(1)), (t(-1) if b == 2 else t(1)), (t(-1) if b == 3 else t(1))
    half, third = t(0.5), t(1.0 / 3.0)
    I = slice(1, N + 1)
    x[I, I, 0] = sx * x[I, I, 1]
    x[I, I, N + 1] = sx * x[I, I, N]
    x[I, 0, I] = sy * x[I, 1, I]
    x[I, N + 1, I] = sy * x[I, N, I]
    x[0, I, I] = sz * x[1, I, I]
    x[N + 1, I, I] = sz * x[N, I, I]
    for A, An in ((0, 1), (N + 1, N)):
        for B, Bn in ((0, 1), (N + 1, N)):
            # arrays are [k, j, i]
            x[B, A, I] = half * (x[B, An, I] + x[Bn, A, I])      # x-directed edge (i, J=A, K=B)
            x[B, I, A] = half * (x[B, I, An] + x[Bn, I, A])      # y-directed edge (I=A, j, K=B)
            x[I, B, A] = half * (x[I, B, An] + x[I, Bn, A])      # z-directed edge (I=A, J=B, k)
    for K, Kn in ((0, 1), (N + 1, N)):
        for J, Jn in ((0, 1), (N + 1, N)):
            for Ii, In in ((0, 1), (N + 1, N)):
                x[K, J, Ii] = third * ((x[K, J, In] + x[K, Jn, Ii]) + x[Kn, J, Ii])


def np_lin_solve(b, x, x0, a, c, K):
    t = x.dtype.type
    N = x.shape[0] - 2
    I = slice(1, N + 1)
    inv = t(1) / t(c)
    a = t(a)
    for _ in range(K):
        xn = np.empty_like(x)
        xn[I, I, I] = (x0[I, I, I] + a * (((x[I, I, 0:N] + x[I, I, 2:N + 2]) + (x[I, 0:N, I] + x[I, 2:N + 2, I]))
                                          + (x[0:N, I, I] + x[2:N + 2, I, I]))) * inv
        np_set_bnd(b, xn)
        x[...] = xn


def np_advect(b, d, d0, u, v, w, dt):
    t = d.dtype.type
    N = d.shape[0] - 2
    I = slice(1, N + 1)
    Nf = t(N)
    dt0 = t(dt) * Nf
    kk, jj, ii = np.meshgrid(np.arange(1, N + 1), np.arange(1, N + 1), np.arange(1, N + 1), indexing="ij")
    lo, hi = t(0.5), Nf + t(0.5)
    x = np.clip(ii.astype(d.dtype) - dt0 * u[I, I, I], lo, hi)
    y = np.clip(jj.astype(d.dtype) - dt0 * v[I, I, I], lo, hi)
    z = np.clip(kk.astype(d.dtype) - dt0 * w[I, I, I], lo, hi)
    i0, j0, k0 = x.astype(np.int64), y.astype(np.int64), z.astype(np.int64)
    i1, j1, k1 = i0 + 1, j0 + 1, k0 + 1
    s1 = x - i0.astype(d.dtype)
    s0 = t(1) - s1
    t1 = y - j0.astype(d.dtype)
    t0 = t(1) - t1
    r1 = z - k0.astype(d.dtype)
    r0 = t(1) - r1
    g = lambda i_, j_, k_: d0[k_, j_, i_]
    d[I, I, I] = (s0 * (t0 * (r0 * g(i0, j0, k0) + r1 * g(i0, j0, k1)) + t1 * (r0 * g(i0, j1, k0) + r1 * g(i0, j1, k1)))
                  + s1 * (t0 * (r0 * g(i1, j0, k0) + r1 * g(i1, j0, k1)) + t1 * (r0 * g(i1, j1, k0) + r1 * g(i1, j1, k1))))
    np_set_bnd(b, d)


def np_project(u, v, w, p, div, K):
    t = u.dtype.type
    N = u.shape[0] - 2
    I = slice(1, N + 1)
    Nf = t(N)
    h = t(1) / Nf
    c_div = t(-0.5) * h
    c_grad = t(0.5) * Nf
    p[...] = 0
    div[I, I, I] = c_div * (((u[I, I, 2:N + 2] - u[I, I, 0:N]) + (v[I, 2:N + 2, I] - v[I, 0:N, I]))
                            + (w[2:N + 2, I, I] - w[0:N, I, I]))
    np_set_bnd(0, div)
    np_set_bnd(0, p)
    np_lin_solve(0, p, div, 1, 6, K)
    u[I, I, I] = u[I, I, I] - c_grad * (p[I, I, 2:N + 2] - p[I, I, 0:N])
    v[I, I, I] = v[I, I, I] - c_grad * (p[I, 2:N + 2, I] - p[I, 0:N, I])
    w[I, I, I] = w[I, I, I] - c_grad * (p[2:N + 2, I, I] - p[0:N, I, I])
    np_set_bnd(1, u)
    np_set_bnd(2, v)
    np_set_bnd(3, w)


def np_diffuse(b, x, x0, diff, dt, K):
    t = x.dtype.type
    Nf = t(x.shape[0] - 2)
    a = ((t(dt) * t(diff)) * Nf) * Nf
    np_lin_solve(b, x, x0, a, t(1) + t(6) * a, K)


def np_step(f, dt, diff, visc, K):
    t = f["u"].dtype.type
    for a, s in (("u", "u0"), ("v", "v0"), ("w", "w0")):
        f[a][...] = f[a] + t(dt) * f[s]
    U, V, W, U0, V0, W0 = f["u0"], f["v0"], f["w0"], f["u"], f["v"], f["w"]  # after swap
    np_diffuse(1, U, U0, visc, dt, K)
    np_diffuse(2, V, V0, visc, dt, K)
    np_diffuse(3, W, W0, visc, dt, K)
    np_project(U, V, W, U0, V0, K)
    U, V, W, U0, V0, W0 = U0, V0, W0, U, V, W  # swap back: U is f["u"] again
    np_advect(1, U, U0, U0, V0, W0, dt)
    np_advect(2, V, V0, U0, V0, W0, dt)
    np_advect(3, W, W0, U0, V0, W0, dt)
    np_project(U, V, W, U0, V0, K)
    f["dens"][...] = f["dens"] + t(dt) * f["dens0"]
    X, X0 = f["dens0"], f["dens"]
    np_diffuse(0, X, X0, diff, dt, K)
    X, X0 = X0, X
    np_advect(0, X, X0, f["u"], f["v"], f["w"], dt)


# ---------------------------------------------------------------------------------------- the tests
def rnd(N, dtype, seed, scale=0.3):
    return (scale * np.random.RandomState(seed).standard_normal((N + 2,) * 3)).astype(dtype)


DT_IDS = ["f32", "f64"]


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=DT_IDS)
@pytest.mark.parametrize("b", [0, 1, 2, 3])
@pytest.mark.parametrize("N", [1, 2, 5, 12])
def test_set_bnd_vs_numpy(N, b, dtype):
    a = rnd(N, dtype, 1)
    c = a.copy()
    O.set_bnd(b, a)
    np_set_bnd(b, c)
    assert np.array_equal(a, c)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=DT_IDS)
@pytest.mark.parametrize("N,K", [(1, 2), (4, 3), (9, 5), (16, 4)])
def test_lin_solve_vs_numpy(N, K, dtype):
    x, x0 = rnd(N, dtype, 2), rnd(N, dtype, 3)
    y = x.copy()
    O.lin_solve(2, x, x0, dtype(0.41), dtype(3.46), K)
    np_lin_solve(2, y, x0, 0.41, 3.46, K)
    assert np.array_equal(x, y)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=DT_IDS)
@pytest.mark.parametrize("N", [1, 3, 8, 15])
def test_advect_vs_numpy(N, dtype):
    d0, u, v, w = rnd(N, dtype, 4), rnd(N, dtype, 5, 1.0), rnd(N, dtype, 6, 1.0), rnd(N, dtype, 7, 1.0)
    a, c = rnd(N, dtype, 8), None
    c = a.copy()
    O.advect(1, a, d0, u, v, w, dtype(DT))
    np_advect(1, c, d0, u, v, w, DT)
    assert np.array_equal(a, c)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=DT_IDS)
@pytest.mark.parametrize("N,K", [(2, 2), (6, 4), (12, 6)])
def test_full_step_vs_numpy(N, K, dtype):
    names = ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")
    f = {n: rnd(N, dtype, 10 + q) for q, n in enumerate(names)}
    g = {n: a.copy() for n, a in f.items()}
    O.step(N, f, dtype(DT), dtype(1e-4), dtype(2e-4), K)
    np_step(g, DT, 1e-4, 2e-4, K)
    for n in names:
        assert np.array_equal(f[n], g[n]), n


# ---- known-answer properties, SURVEY.md §8c ----------------------------------------------------------
def test_zero_stays_zero():
    N = 10
    f = {n: np.zeros((N + 2,) * 3, np.float32) for n in ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")}
    O.step(N, f, np.float32(DT), np.float32(1e-4), np.float32(1e-4), 5)
    assert all(not a.any() for a in f.values())


def test_uniform_density_unchanged():
    N = 9
    x = np.full((N + 2,) * 3, 0.625, np.float32)
    z = lambda: np.zeros((N + 2,) * 3, np.float32)
    O.dens_step(x, z(), z(), z(), z(), np.float32(1e-3), np.float32(DT), 8)
    assert np.all(x == np.float32(0.625))


def test_project_reduces_divergence():
    """On a smooth (resolved) velocity field the projection removes most of the divergence and a second
    projection changes little (idempotent to tolerance). White noise is not used: the centred-difference
    projection cannot see checkerboard modes."""
    N, K = 24, 400
    kk, jj, ii = np.meshgrid(*(np.arange(N + 2),) * 3, indexing="ij")
    X, Y, Z = (ii - 0.5) / N, (jj - 0.5) / N, (kk - 0.5) / N
    u = np.sin(np.pi * X) * np.cos(2 * np.pi * Y) * np.cos(np.pi * Z) + 0.3 * np.sin(2 * np.pi * X)
    v = np.cos(np.pi * X) * np.sin(np.pi * Y) * np.sin(2 * np.pi * Z) + 0.2 * np.sin(np.pi * Y)
    w = np.sin(2 * np.pi * X) * np.sin(np.pi * Y) * np.sin(np.pi * Z)
    for b, a in ((1, u), (2, v), (3, w)):
        O.set_bnd(b, a)

    def max_div(u, v, w):
        p, d = np.zeros_like(u), np.zeros_like(u)
        O.project_div(u, v, w, p, d)
        return np.abs(d[1:-1, 1:-1, 1:-1]).max()

    before = max_div(u, v, w)
    O.project(u, v, w, np.zeros_like(u), np.zeros_like(u), K)
    after = max_div(u, v, w)
    assert after < 0.25 * before
    u2, v2, w2 = u.copy(), v.copy(), w.copy()
    O.project(u2, v2, w2, np.zeros_like(u), np.zeros_like(u), K)
    change = max(np.abs(u2 - u).max(), np.abs(v2 - v).max(), np.abs(w2 - w).max())
    assert change < 0.25 * max(np.abs(u).max(), np.abs(v).max(), np.abs(w).max())
    assert max_div(u2, v2, w2) <= after


def test_advect_zero_velocity_is_identity():
    N = 11
    d0 = rnd(N, np.float32, 30)
    d = np.zeros_like(d0)
    z = np.zeros_like(d0)
    O.advect(0, d, d0, z, z, z, np.float32(DT))
    assert np.array_equal(d[1:-1, 1:-1, 1:-1], d0[1:-1, 1:-1, 1:-1])


def test_jacobi_fixed_point_is_noop():
    """If x == (x0 + a*sum_nb(x))*inv holds bitwise, one more sweep reproduces x exactly. Built from a
    converged solve: iterate until the iterate stops changing in float32, then check one more sweep."""
    N = 6
    x, x0 = rnd(N, np.float32, 40), rnd(N, np.float32, 41)
    a, c = np.float32(0.05), np.float32(1.3)
    prev = None
    for _ in range(400):
        prev = x.copy()
        O.lin_solve(0, x, x0, a, c, 1)
        if np.array_equal(prev, x):
            break
    else:
        pytest.skip("did not reach a bitwise fixed point")
    y = x.copy()
    O.lin_solve(0, y, x0, a, c, 1)
    assert np.array_equal(x, y)


def test_mirror_symmetry():
    N = 8
    src = np.zeros((N + 2,) * 3, np.float32)
    src[4:6, 4:6, 4:6] = 10
    x = np.zeros_like(src)
    z = np.zeros_like(src)
    O.dens_step(x, src, z, z.copy(), z.copy(), np.float32(1e-2), np.float32(DT), 12)
    assert x.max() > 0
    assert np.array_equal(x, x[::-1]) and np.array_equal(x, x[:, ::-1]) and np.array_equal(x, x[:, :, ::-1])


def test_diffuse_conserves_mass_in_the_limit():
    N = 8
    x0 = np.abs(rnd(N, np.float64, 50))
    O.set_bnd(0, x0)
    x = x0.copy()
    O.diffuse(0, x, x0, 1e-3, DT, 4000)
    assert abs(x[1:-1, 1:-1, 1:-1].sum() - x0[1:-1, 1:-1, 1:-1].sum()) < 1e-9 * x0.sum()


# ---- tracers (SPEC §6) -------------------------------------------------------------------------------------
def test_tracers_zero_velocity_stay_put_and_are_clamped():
    N = 6
    z = np.zeros((N + 2,) * 3, np.float64)
    pos = np.array([[1.25, 2.5, 3.75], [-3.0, 100.0, 0.5], [N + 0.5, 0.5, 3.0]])
    p = pos.copy()
    O.tracers_advect(p, z, z, z, 0.1)
    assert np.array_equal(p, np.clip(pos, 0.5, N + 0.5))


def test_tracers_uniform_velocity_translates():
    N, dt = 8, 0.1
    u = np.full((N + 2,) * 3, 0.5)
    v = np.full((N + 2,) * 3, -0.25)
    w = np.zeros((N + 2,) * 3)
    p = np.array([[4.0, 4.0, 4.0], [2.5, 6.5, 1.5]])
    q = p.copy()
    O.tracers_advect(q, u, v, w, dt)
    assert np.allclose(q, p + dt * N * np.array([0.5, -0.25, 0.0]), rtol=0, atol=1e-12)


def test_tracer_sample_reproduces_a_trilinear_field_exactly():
    N = 5
    kk, jj, ii = np.meshgrid(*(np.arange(N + 2, dtype=np.float64),) * 3, indexing="ij")
    lin = 0.5 * ii - 0.25 * jj + 2.0 * kk + 1.0  # exactly representable, trilinear interpolation is exact
    z = np.zeros_like(lin)
    pos = np.array([[1.5, 2.25, 3.125], [4.75, 0.5, 5.5], [3.0, 3.0, 3.0]])
    d, s = O.tracers_sample(pos, lin, lin, z, z)
    want = 0.5 * pos[:, 0] - 0.25 * pos[:, 1] + 2.0 * pos[:, 2] + 1.0
    assert np.allclose(d, want, rtol=0, atol=1e-12) and np.allclose(s, np.abs(want), rtol=0, atol=1e-12)
