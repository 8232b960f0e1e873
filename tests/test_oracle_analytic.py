"""Analytic pin of the CPU oracle (and, at the end, of the GPU path): closed forms that do not come from this
repo's own restatements.

The reference holds no grid solver and no fixture for one, so nothing of the reference can pin the
stencil half of the oracle. What can is mathematics: with the mirror shells of set_bnd the discrete cosine modes

    phi_m(i) = cos(pi m (i - 1/2) / N),   i = 0 .. N+1      (phi(0) = phi(1), phi(N+1) = phi(N): b = 0 shells)
    psi_m(i) = sin(pi m (i - 1/2) / N)                        (psi(0) = -psi(1), psi(N+1) = -psi(N): the b = 1/2/3 shells)

are exact eigenvectors of the three-point neighbour sum, INCLUDING the wall cells: phi(i-1) + phi(i+1) = 2 cos(pi m/N)
phi(i) and psi(i+1) - psi(i-1) = 2 sin(pi m/N) phi(i) for every i = 1 .. N. A product mode therefore stays a product
mode under every operator below and only its amplitude changes, by a scalar recursion that is evaluated here in
extended precision (numpy longdouble):

  lin_solve(0, x, x0, a, c, K) on x = A M, x0 = B M:   A <- (B + a lam A) / c,  lam = 2 (cx + cy + cz),  K times
  diffuse(0, x, x0) with x = x0 = A M:                 the same with a = dt diff N^2, c = 1 + 6a, B = A
  project on u = U psi phi phi, v = V phi psi phi, w = W phi phi psi:
      div = D M with D = -(U sx + V sy + W sz) / N;  p = P M with P <- (D + lam P) / 6 from P = 0, K times;
      u' = (U + N P sx) psi phi phi  (v', w' likewise)

Tolerance (stated, not tuned per case): every sweep evaluates ~10 rounded operations on values bounded by
S = |B| + 6 |a| max|A| (the partial sums), so after K sweeps the error is bounded by K * 16 * eps * S / |c| per
sweep-chain; the tests use  tol = 16 * (K + 2) * eps(T) * scale  with scale the largest amplitude that appears. Observed
errors are 10-50x below it. The shells are compared too (they must be the mirror extension of the mode).
"""
import numpy as np
import pytest

import oracle_lib as O

LD = np.longdouble
DTYPES = [np.float32, np.float64]


def modes(N, m):
    i = np.arange(N + 2, dtype=LD)
    th = LD(np.pi) * LD(m) * (i - LD(0.5)) / LD(N)
    return np.cos(th), np.sin(th)


def product(fk, fj, fi):
    return fk[:, None, None] * fj[None, :, None] * fi[None, None, :]


def jacobi_amplitude(A, B, a, c, lam, K):
    A, B, a, c, lam = LD(A), LD(B), LD(a), LD(c), LD(lam)
    for _ in range(K):
        A = (B + a * lam * A) / c
    return A


def tol(dtype, K, scale):
    return 16.0 * (K + 2) * float(np.finfo(dtype).eps) * float(scale)


CASES = [(8, (1, 0, 0), 3), (8, (2, 3, 1), 5), (16, (0, 0, 0), 4), (16, (5, 2, 7), 6), (24, (3, 3, 3), 20), (12, (11, 1, 4), 7)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("N,m,K", CASES)
def test_lin_solve_on_cosine_modes(N, m, K, dtype):
    mx, my, mz = m
    (cx_, _), (cy_, _), (cz_, _) = modes(N, mx), modes(N, my), modes(N, mz)
    M = product(cz_, cy_, cx_)
    lam = 2 * (np.cos(LD(np.pi) * mx / N) + np.cos(LD(np.pi) * my / N) + np.cos(LD(np.pi) * mz / N))
    A0, B0, a = 0.75, -1.25, 0.3
    # the arguments exactly as the oracle receives them (a, c rounded to T; the closed form uses those values)
    a_t, c_t = dtype(a), dtype(1 + 6 * a)
    x = (LD(A0) * M).astype(dtype)
    x0 = (LD(B0) * M).astype(dtype)
    O.lin_solve(0, x, x0, a_t, c_t, K)
    want = jacobi_amplitude(A0, B0, a_t, c_t, lam, K) * M
    scale = abs(B0) + 6 * a * abs(A0) + abs(A0)
    err = float(np.max(np.abs(x.astype(LD) - want)))
    assert err <= tol(dtype, K, scale), (err, tol(dtype, K, scale))


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("N,m,K", [(8, (1, 2, 0), 4), (16, (3, 1, 2), 10), (20, (0, 0, 1), 20)])
def test_diffuse_on_cosine_modes(N, m, K, dtype):
    mx, my, mz = m
    M = product(modes(N, mz)[0], modes(N, my)[0], modes(N, mx)[0])
    lam = 2 * (np.cos(LD(np.pi) * mx / N) + np.cos(LD(np.pi) * my / N) + np.cos(LD(np.pi) * mz / N))
    dt, diff, A0 = 0.1, 2e-3, 1.5
    x0 = (LD(A0) * M).astype(dtype)
    x = x0.copy()
    O.diffuse(0, x, x0, dtype(diff), dtype(dt), K)
    Nf = dtype(N)
    a_t = ((dtype(dt) * dtype(diff)) * Nf) * Nf  # SPEC §2, in T
    c_t = dtype(1) + dtype(6) * a_t
    want = jacobi_amplitude(A0, A0, a_t, c_t, lam, K) * M
    err = float(np.max(np.abs(x.astype(LD) - want)))
    assert err <= tol(dtype, K, (1 + 12 * float(a_t)) * A0), (err, tol(dtype, K, (1 + 12 * float(a_t)) * A0))
    # a diffusion step damps every non-constant mode and leaves the constant one alone
    amp = float(jacobi_amplitude(A0, A0, a_t, c_t, lam, K))
    assert (amp < A0) if any(m) else abs(amp - A0) < 1e-6


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("N,m,K", [(8, (1, 1, 1), 3), (16, (2, 5, 3), 6), (12, (4, 0, 7), 10)])
def test_project_on_sine_cosine_modes(N, m, K, dtype):
    mx, my, mz = m
    (cx_, sx_), (cy_, sy_), (cz_, sz_) = modes(N, mx), modes(N, my), modes(N, mz)
    M = product(cz_, cy_, cx_)
    U, V, W = 0.6, -0.4, 0.9
    u = (LD(U) * product(cz_, cy_, sx_)).astype(dtype)
    v = (LD(V) * product(cz_, sy_, cx_)).astype(dtype)
    w = (LD(W) * product(sz_, cy_, cx_)).astype(dtype)
    p = np.zeros_like(u)
    div = np.zeros_like(u)
    O.project(u, v, w, p, div, K)
    pi = LD(np.pi)
    s = [np.sin(pi * q / N) for q in (mx, my, mz)]
    lam = 2 * (np.cos(pi * mx / N) + np.cos(pi * my / N) + np.cos(pi * mz / N))
    h = LD(dtype(1) / dtype(N))
    D = -LD(dtype(0.5)) * h * 2 * (U * s[0] + V * s[1] + W * s[2])
    P = jacobi_amplitude(0.0, D, dtype(1), dtype(6), lam, K)
    t_div = tol(dtype, 2, 4.0 / N)
    assert float(np.max(np.abs(div.astype(LD) - D * M))) <= t_div
    t_p = tol(dtype, K, abs(float(D)) * 2 + 1e-30)
    assert float(np.max(np.abs(p.astype(LD) - P * M))) <= t_p
    for got, amp, f in ((u, U + N * P * s[0], product(cz_, cy_, sx_)), (v, V + N * P * s[1], product(cz_, sy_, cx_)),
                        (w, W + N * P * s[2], product(sz_, cy_, cx_))):
        # interior and faces: the edges and corners of an antisymmetric (b = 1, 2, 3) field are averages of a mirrored and
        # a negated face, not the product mode's extension (for b = 0 they are: div, p and the tests above include them)
        idx = np.arange(N + 2)
        shell = ((idx == 0) | (idx == N + 1)).astype(int)
        few = (shell[:, None, None] + shell[None, :, None] + shell[None, None, :]) <= 1
        err = float(np.max(np.abs(got.astype(LD) - amp * f)[few]))
        assert err <= tol(dtype, K, 2.0 + N * abs(float(P))), err
    # the projection removes divergence: the amplitude of div(u') is |D| (1 - lam' ...) -> smaller than before
    D2 = -(float((U + N * P * s[0]) * s[0] + (V + N * P * s[1]) * s[1] + (W + N * P * s[2]) * s[2])) / N
    assert abs(D2) < abs(float(D)) or abs(float(D)) < 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,N,K", [(np.float32, 256, 20), (np.float64, 192, 9)], ids=["f32-256", "f64-192"])
def test_gpu_lin_solve_on_a_cosine_mode(dtype, N, K):
    """The same closed form against libsfgpu.so itself (through the C ABI), at the benchmark size: K = 20 is one
    register-blocked pair plus six three-sweep passes of the marching kernel."""
    from fluidsolvergpu_amd import solver as S

    m = (3, 5, 2)
    M = product(modes(N, m[2])[0], modes(N, m[1])[0], modes(N, m[0])[0])
    lam = 2 * sum(np.cos(LD(np.pi) * q / N) for q in m)
    A0, B0, a = 0.75, -1.25, 0.3
    a_t, c_t = dtype(a), dtype(1 + 6 * a)
    x = (LD(A0) * M).astype(dtype)
    x0 = (LD(B0) * M).astype(dtype)
    with S.FluidSolver(N, dtype="f32" if dtype == np.float32 else "f64", iters=K) as fs:
        fs.upload("dens", x)
        fs.upload("dens0", x0)
        fs.lin_solve(0, "dens", "dens0", float(a_t), float(c_t), K)
        fs.sync()
        got = fs.download("dens")
    want = jacobi_amplitude(A0, B0, a_t, c_t, lam, K) * M
    scale = abs(B0) + 6 * a * abs(A0) + abs(A0)
    err = float(np.max(np.abs(got.astype(LD) - want)))
    assert err <= tol(dtype, K, scale), (err, tol(dtype, K, scale))
