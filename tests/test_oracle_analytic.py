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
    """The same closed form against libsfgpu.so itself (through the C ABI), at the benchmark size: K = 20 is five
    four-sweep passes of the marching kernel (first pass over caller data, three plain, one that writes the i-shell);
    tests/test_full_size_gpu.py holds the bit-exact comparison with the oracle at this size."""
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


# ------------------------------------------------------------------------------------------------------------------
# Round 3: the rest of the oracle pinned by closed forms — advect, tracers, and the b = 1, 2, 3 shells (faces, edges,
# corners) that the cosine-mode tests above had to leave out.
#
#  * Trilinear interpolation reproduces affine data exactly, so advect of d0 = alpha + beta . (i, j, k) under a UNIFORM
#    velocity (U, V, W) is   d(i,j,k) = alpha + beta . clamp((i,j,k) - dt N (U,V,W), 0.5, N + 0.5)   on the interior,
#    clamp included (cells whose back-trace leaves the grid read the wall value), followed by set_bnd(b).
#  * A tracer in a uniform flow moves by dt N (U,V,W) per call, clamped to [0.5, N + 0.5]; sampling an affine density
#    returns its value at the (clamped) position; speed = |(U,V,W)|.
#  * set_bnd(b): with s_x = -1 if b == 1 else +1 (s_y, s_z likewise) and v the interior cell nearest to a shell cell:
#    face = s_axis v; edge shelled in axes (p, q) = (s_p + s_q)/2 v — the mean of a mirrored and a negated face is 0 —;
#    corner = (s_x + s_y + s_z)/3 v. And psi_m(i) = sin(pi m (i - 1/2)/N) is the antisymmetric eigenvector
#    (psi(0) = -psi(1), psi(N+1) = -psi(N), psi(i-1) + psi(i+1) = 2 cos(pi m/N) psi(i) for EVERY i = 1..N), so
#    lin_solve(b) on a product mode with psi along axis b follows the same amplitude recursion as the cosine modes.
# Tolerances are stated per test as multiples of eps(T) x the largest magnitude involved.


def shell_rule(x, b):
    """Closed-form shells of set_bnd(b) from the interior of x (longdouble array of shape (N+2,)*3, [k, j, i])."""
    N = x.shape[0] - 2
    sgn = {0: (1, 1, 1), 1: (-1, 1, 1), 2: (1, -1, 1), 3: (1, 1, -1)}[b]  # (s_x, s_y, s_z)
    out = x.copy()
    idx = np.arange(N + 2)
    near = np.clip(idx, 1, N)
    sh = (idx == 0) | (idx == N + 1)
    K3, J3, I3 = np.meshgrid(idx, idx, idx, indexing="ij")
    v = x[near[K3], near[J3], near[I3]]
    shx, shy, shz = sh[I3], sh[J3], sh[K3]
    n = shx.astype(int) + shy.astype(int) + shz.astype(int)
    ssum = shx * LD(sgn[0]) + shy * LD(sgn[1]) + shz * LD(sgn[2])
    fac = np.where(n == 0, LD(1), ssum / np.maximum(n, 1))
    out[...] = fac * v
    return out


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("b", [0, 1, 2, 3])
@pytest.mark.parametrize("N", [1, 2, 5, 12])
def test_set_bnd_faces_edges_corners_closed_form(N, b, dtype):
    rng = np.random.RandomState(100 + N + b)
    x = rng.standard_normal((N + 2,) * 3).astype(dtype)
    want = shell_rule(x.astype(LD), b)
    O.set_bnd(b, x)
    # one rounding per sum / product of at most three terms: 4 eps x max|v|
    err = float(np.max(np.abs(x.astype(LD) - want)))
    assert err <= 4 * float(np.finfo(dtype).eps) * float(np.max(np.abs(want)) + 1), err
    if b == 1 and N >= 2:  # the mean of a mirrored and a negated face is exactly zero
        assert x[3 if N >= 3 else 1, 0, 0] == 0 and x[0, 1, 0] == 0 and x[0, 0, 1] != 0


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("N,m,K,b", [(8, (1, 2, 0), 4, 1), (16, (3, 1, 2), 7, 2), (12, (2, 0, 5), 6, 3), (24, (3, 3, 3), 20, 1)])
def test_lin_solve_on_antisymmetric_modes_with_shells(N, m, K, b, dtype):
    """lin_solve(b = 1, 2, 3) on a product mode that is a sine along axis b: amplitude recursion on the interior AND the
    closed-form faces / edges / corners of the result."""
    mx, my, mz = m
    (cx_, sx_), (cy_, sy_), (cz_, sz_) = modes(N, mx), modes(N, my), modes(N, mz)
    M = product(sz_ if b == 3 else cz_, sy_ if b == 2 else cy_, sx_ if b == 1 else cx_)
    lam = 2 * (np.cos(LD(np.pi) * mx / N) + np.cos(LD(np.pi) * my / N) + np.cos(LD(np.pi) * mz / N))
    A0, B0, a = -0.6, 1.1, 0.27
    a_t, c_t = dtype(a), dtype(1 + 6 * a)
    x = (LD(A0) * M).astype(dtype)
    x0 = (LD(B0) * M).astype(dtype)
    O.set_bnd(b, x)  # the iterate enters with its own set_bnd applied, as every field of a step does
    O.lin_solve(b, x, x0, a_t, c_t, K)
    want = shell_rule(jacobi_amplitude(A0, B0, a_t, c_t, lam, K) * M, b)
    scale = abs(B0) + 6 * a * abs(A0) + abs(A0)
    err = float(np.max(np.abs(x.astype(LD) - want)))
    assert err <= tol(dtype, K, scale), (err, tol(dtype, K, scale))


def affine(N, alpha, beta):
    i = np.arange(N + 2, dtype=LD)
    return LD(alpha) + LD(beta[0]) * i[None, None, :] + LD(beta[1]) * i[None, :, None] + LD(beta[2]) * i[:, None, None]


def advect_affine_closed_form(N, alpha, beta, vel, dt_t, dtype, b):
    """Interior by the closed form, shells by shell_rule. dt0 and the back-traced coordinate are formed in T exactly as
    SPEC §3 writes them (one product, one difference: the clamp then acts on a T value), the affine map in longdouble."""
    dt0 = dtype(dt_t) * dtype(N)
    i = np.arange(N + 2).astype(dtype)
    coords = []
    for q in range(3):
        x = i - dt0 * dtype(vel[q])
        x = np.minimum(np.maximum(x, dtype(0.5)), dtype(N) + dtype(0.5))
        coords.append(x.astype(LD))
    d = (LD(alpha) + LD(beta[0]) * coords[0][None, None, :] + LD(beta[1]) * coords[1][None, :, None]
         + LD(beta[2]) * coords[2][:, None, None])
    return shell_rule(d, b)


ADVECT_CASES = [(8, (0.37, -0.21, 0.05), 0), (16, (1.37, 0.0, -2.6), 0), (12, (-3.6, 3.6, 0.4), 1), (10, (0.0, 0.0, 0.0), 2),
                (20, (0.499, -7.0, 12.0), 3)]


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("N,shift,b", ADVECT_CASES)
def test_advect_of_an_affine_field_under_uniform_velocity(N, shift, b, dtype):
    """`shift` = the back-trace in cells (dt N U): fractions of a cell, several cells, and far beyond the wall (clamp)."""
    dt = 0.1
    vel = tuple(sh / (dt * N) for sh in shift)
    alpha, beta = 0.75, (0.5, -0.25, 0.125)
    d0 = affine(N, alpha, beta).astype(dtype)
    u, v, w = (np.full((N + 2,) * 3, dtype(q), dtype) for q in vel)
    d = np.zeros_like(d0)
    O.advect(b, d, d0, u, v, w, dtype(dt))
    want = advect_affine_closed_form(N, alpha, beta, vel, dt, dtype, b)
    # weights and three nested two-term interpolations: <= 24 roundings on values <= max|d0|
    scale = float(np.max(np.abs(d0.astype(LD))))
    err = float(np.max(np.abs(d.astype(LD) - want)))
    assert err <= 24 * float(np.finfo(dtype).eps) * scale, (err, scale)
    if not any(shift):
        assert np.array_equal(d[1:-1, 1:-1, 1:-1], d0[1:-1, 1:-1, 1:-1])  # zero velocity: the identity, exactly


@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
@pytest.mark.parametrize("N,shift,steps", [(8, (0.3, -0.2, 0.1), 5), (16, (2.5, 0.0, -1.25), 4), (12, (-9.0, 9.0, 0.75), 3)])
def test_tracers_in_a_uniform_flow(N, shift, steps, dtype):
    dt = 0.1
    vel = tuple(sh / (dt * N) for sh in shift)
    u, v, w = (np.full((N + 2,) * 3, dtype(q), dtype) for q in vel)
    alpha, beta = 2.0, (0.25, 0.5, -0.125)
    dens = affine(N, alpha, beta).astype(dtype)
    rng = np.random.RandomState(7)
    pos = rng.uniform(-1.0, N + 2.0, size=(64, 3)).astype(dtype)  # some start outside [0.5, N + 0.5]: clamped first
    p = pos.copy()
    for _ in range(steps):
        O.tracers_advect(p, u, v, w, dtype(dt))
    lo, hi = LD(0.5), LD(N) + LD(0.5)
    want = np.clip(pos.astype(LD), lo, hi)
    dt0 = LD(dtype(dt) * dtype(N))
    for _ in range(steps):
        want = np.clip(want + dt0 * np.array([LD(dtype(q)) for q in vel]), lo, hi)
    # per step: a trilinear sample of a constant (<= 12 roundings on |vel|), one product, one sum
    err = float(np.max(np.abs(p.astype(LD) - want)))
    assert err <= steps * 16 * float(np.finfo(dtype).eps) * (N + 1), err
    dsample, speed = O.tracers_sample(p, dens, u, v, w)
    pc = np.clip(p.astype(LD), lo, hi)
    dwant = LD(alpha) + LD(beta[0]) * pc[:, 0] + LD(beta[1]) * pc[:, 1] + LD(beta[2]) * pc[:, 2]
    assert float(np.max(np.abs(dsample.astype(LD) - dwant))) <= 24 * float(np.finfo(dtype).eps) * float(np.max(np.abs(dens)))
    swant = np.sqrt(sum(LD(dtype(q)) ** 2 for q in vel))
    assert float(np.max(np.abs(speed.astype(LD) - swant))) <= 32 * float(np.finfo(dtype).eps) * float(swant + 1e-30)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", DTYPES, ids=["f32", "f64"])
def test_gpu_advect_tracers_and_antisymmetric_shells_closed_forms(dtype):
    """The three closed forms of round 3 against libsfgpu.so (through the C ABI): advect of an affine field (gather form
    and, through the three velocity components, the one-cell-per-lane form), tracers in a uniform flow, lin_solve(b = 2)
    on an antisymmetric mode with its faces, edges and corners."""
    from fluidsolvergpu_amd import solver as S

    N, dt, K = 96, 0.1, 9
    name = "f32" if dtype == np.float32 else "f64"
    eps = float(np.finfo(dtype).eps)
    shift = (1.37, -0.6, 3.6)
    vel = tuple(sh / (dt * N) for sh in shift)
    alpha, beta = 0.75, (0.5, -0.25, 0.125)
    d0 = affine(N, alpha, beta).astype(dtype)
    u, v, w = (np.full((N + 2,) * 3, dtype(q), dtype) for q in vel)
    with S.FluidSolver(N, dtype=name, iters=K, dt=dt) as fs:
        for n, arr in (("u", u), ("v", v), ("w", w), ("dens0", d0), ("u0", d0), ("v0", d0), ("w0", d0)):
            fs.upload(n, arr)
        fs.advect(0, "dens", "dens0", "u", "v", "w")
        got = fs.download("dens")
        want = advect_affine_closed_form(N, alpha, beta, vel, dt, dtype, 0)
        assert float(np.max(np.abs(got.astype(LD) - want))) <= 24 * eps * float(np.max(np.abs(d0)))
        # tracers
        rng = np.random.RandomState(9)
        pos = rng.uniform(-1.0, N + 2.0, size=(256, 3)).astype(dtype)
        fs.tracers_set(pos)
        for _ in range(3):
            fs.tracers_advect()
        p = fs.tracers_get(sample=False)[0]
        lo, hi = LD(0.5), LD(N) + LD(0.5)
        wantp = np.clip(pos.astype(LD), lo, hi)
        dt0 = LD(dtype(dt) * dtype(N))
        for _ in range(3):
            wantp = np.clip(wantp + dt0 * np.array([LD(dtype(q)) for q in vel]), lo, hi)
        assert float(np.max(np.abs(np.asarray(p).astype(LD) - wantp))) <= 3 * 16 * eps * (N + 1)
        # lin_solve(b = 2) on cos x sin x cos
        m, b = (3, 2, 1), 2
        (cx_, _), (_, sy_), (cz_, _) = modes(N, m[0]), modes(N, m[1]), modes(N, m[2])
        M = product(cz_, sy_, cx_)
        lam = 2 * sum(np.cos(LD(np.pi) * q / N) for q in m)
        A0, B0, a = -0.6, 1.1, 0.27
        a_t, c_t = dtype(a), dtype(1 + 6 * a)
        x = (LD(A0) * M).astype(dtype)
        O.set_bnd(b, x)
        fs.upload("dens", x)
        fs.upload("dens0", (LD(B0) * M).astype(dtype))
        fs.lin_solve(b, "dens", "dens0", float(a_t), float(c_t), K)
        fs.sync()
        got = fs.download("dens")
    want = shell_rule(jacobi_amplitude(A0, B0, a_t, c_t, lam, K) * M, b)
    scale = abs(B0) + 6 * a * abs(A0) + abs(A0)
    assert float(np.max(np.abs(got.astype(LD) - want))) <= tol(dtype, K, scale)
