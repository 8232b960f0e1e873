"""Test helper: ctypes binding of oracle/liboracle.so (the CPU oracle — test infrastructure only).
Arrays are (N+2,)*3 numpy arrays indexed [k, j, i], modified in place."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "liboracle.so")
_lib = C.CDLL(SO)

_SFX = {np.dtype(np.float32): ("f32", C.c_float), np.dtype(np.float64): ("f64", C.c_double)}


def _p(a):
    assert a.flags.c_contiguous
    return a.ctypes.data_as(C.c_void_p)


def _fn(name, dtype):
    sfx, cty = _SFX[np.dtype(dtype)]
    return getattr(_lib, f"oracle_{name}_{sfx}"), cty


def _n(a):
    assert a.ndim == 3 and a.shape[0] == a.shape[1] == a.shape[2]
    return a.shape[0] - 2


def add_source(x, s, dt):
    f, c = _fn("add_source", x.dtype)
    f(C.c_int(_n(x)), _p(x), _p(s), c(dt))


def set_bnd(b, x):
    f, c = _fn("set_bnd", x.dtype)
    f(C.c_int(_n(x)), C.c_int(b), _p(x))


def lin_solve(b, x, x0, a, c_, K):
    f, c = _fn("lin_solve", x.dtype)
    f(C.c_int(_n(x)), C.c_int(b), _p(x), _p(x0), c(a), c(c_), C.c_int(K))


def diffuse(b, x, x0, diff, dt, K):
    f, c = _fn("diffuse", x.dtype)
    f(C.c_int(_n(x)), C.c_int(b), _p(x), _p(x0), c(diff), c(dt), C.c_int(K))


def advect(b, d, d0, u, v, w, dt):
    f, c = _fn("advect", d.dtype)
    f(C.c_int(_n(d)), C.c_int(b), _p(d), _p(d0), _p(u), _p(v), _p(w), c(dt))


def project(u, v, w, p, div, K):
    f, c = _fn("project", u.dtype)
    f(C.c_int(_n(u)), _p(u), _p(v), _p(w), _p(p), _p(div), C.c_int(K))


def project_div(u, v, w, p, div):
    f, c = _fn("project_div", u.dtype)
    f(C.c_int(_n(u)), _p(u), _p(v), _p(w), _p(p), _p(div))


def project_sub(u, v, w, p):
    f, c = _fn("project_sub", u.dtype)
    f(C.c_int(_n(u)), _p(u), _p(v), _p(w), _p(p))


def dens_step(x, x0, u, v, w, diff, dt, K):
    f, c = _fn("dens_step", x.dtype)
    f(C.c_int(_n(x)), _p(x), _p(x0), _p(u), _p(v), _p(w), c(diff), c(dt), C.c_int(K))


def vel_step(u, v, w, u0, v0, w0, visc, dt, K):
    f, c = _fn("vel_step", u.dtype)
    f(C.c_int(_n(u)), _p(u), _p(v), _p(w), _p(u0), _p(v0), _p(w0), c(visc), c(dt), C.c_int(K))


def step(N, fields, dt, diff, visc, K):
    """vel_step then dens_step on a dict of the 8 named fields (modified in place and returned)."""
    f = fields
    vel_step(f["u"], f["v"], f["w"], f["u0"], f["v0"], f["w0"], visc, dt, K)
    dens_step(f["dens"], f["dens0"], f["u"], f["v"], f["w"], diff, dt, K)
    return f


def tracers_advect(pos, u, v, w, dt):
    f, c = _fn("tracers_advect", u.dtype)
    assert pos.dtype == u.dtype and pos.flags.c_contiguous
    f(C.c_int(_n(u)), C.c_int(pos.shape[0]), _p(pos), _p(u), _p(v), _p(w), c(dt))


def tracers_sample(pos, dens, u, v, w):
    f, c = _fn("tracers_sample", u.dtype)
    n = pos.shape[0]
    d, s = np.zeros(n, u.dtype), np.zeros(n, u.dtype)
    f(C.c_int(_n(u)), C.c_int(n), _p(pos), _p(dens), _p(u), _p(v), _p(w), _p(d), _p(s))
    return d, s
