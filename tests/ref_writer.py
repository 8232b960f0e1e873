"""Test helper: the reference's OWN visit_writer, compiled by oracle/Makefile into
oracle/_ref/libvisit_writer_ref.so (from /root/reference/visit_writer.cpp, never copied here).
The reference has C++ linkage, so the symbols are the Itanium-mangled names."""
import ctypes as C
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_SO = os.path.join(ROOT, "oracle", "_ref", "libvisit_writer_ref.so")

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)

MANGLED = {
    "write_point_mesh": "_Z16write_point_meshPKciiPfiPiPKS0_PS1_",
    "write_unstructured_mesh": "_Z23write_unstructured_meshPKciiPfiPiS2_iS2_S2_PKS0_PS1_",
    "write_regular_mesh": "_Z18write_regular_meshPKciPiiS1_S1_PKS0_PPf",
    "write_rectilinear_mesh": "_Z22write_rectilinear_meshPKciPiPfS2_S2_iS1_S1_PKS0_PS2_",
    "write_curvilinear_mesh": "_Z22write_curvilinear_meshPKciPiPfiS1_S1_PKS0_PS2_",
}


def available():
    return os.path.exists(REF_SO)


def load(path=REF_SO):
    return C.CDLL(path)


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32).ravel()


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32).ravel()


def _vars(varnames, vars_):
    arrays = [_f32(v) for v in vars_]
    names = (C.c_char_p * max(len(arrays), 1))(*[s.encode() for s in varnames])
    ptrs = (_fp * max(len(arrays), 1))(*[a.ctypes.data_as(_fp) for a in arrays])
    return arrays, names, ptrs


class Writer:
    """Calls the five C++-linkage entry points of a visit_writer-compatible library."""

    def __init__(self, path=REF_SO):
        self.lib = load(path)
        for fn in MANGLED.values():
            getattr(self.lib, fn).restype = None

    def write_point_mesh(self, filename, ub, npts, pts, nvars, vardim, varnames, vars_):
        keep, names, ptrs = _vars(varnames, vars_)
        p, vd = _f32(pts), _i32(vardim)
        getattr(self.lib, MANGLED["write_point_mesh"])(
            os.fsencode(filename), C.c_int(ub), C.c_int(npts), p.ctypes.data_as(_fp), C.c_int(nvars),
            vd.ctypes.data_as(_ip), names, ptrs)

    def write_unstructured_mesh(self, filename, ub, npts, pts, ncells, celltypes, conn, nvars, vardim, centering,
                                varnames, vars_):
        keep, names, ptrs = _vars(varnames, vars_)
        p, vd, ce, ct, cn = _f32(pts), _i32(vardim), _i32(centering), _i32(celltypes), _i32(conn)
        getattr(self.lib, MANGLED["write_unstructured_mesh"])(
            os.fsencode(filename), C.c_int(ub), C.c_int(npts), p.ctypes.data_as(_fp), C.c_int(ncells),
            ct.ctypes.data_as(_ip), cn.ctypes.data_as(_ip), C.c_int(nvars), vd.ctypes.data_as(_ip),
            ce.ctypes.data_as(_ip), names, ptrs)

    def write_regular_mesh(self, filename, ub, dims, nvars, vardim, centering, varnames, vars_):
        keep, names, ptrs = _vars(varnames, vars_)
        d, vd, ce = _i32(dims), _i32(vardim), _i32(centering)
        getattr(self.lib, MANGLED["write_regular_mesh"])(
            os.fsencode(filename), C.c_int(ub), d.ctypes.data_as(_ip), C.c_int(nvars), vd.ctypes.data_as(_ip),
            ce.ctypes.data_as(_ip), names, ptrs)

    def write_rectilinear_mesh(self, filename, ub, dims, x, y, z, nvars, vardim, centering, varnames, vars_):
        keep, names, ptrs = _vars(varnames, vars_)
        d, vd, ce = _i32(dims), _i32(vardim), _i32(centering)
        xs, ys, zs = _f32(x), _f32(y), _f32(z)
        getattr(self.lib, MANGLED["write_rectilinear_mesh"])(
            os.fsencode(filename), C.c_int(ub), d.ctypes.data_as(_ip), xs.ctypes.data_as(_fp),
            ys.ctypes.data_as(_fp), zs.ctypes.data_as(_fp), C.c_int(nvars), vd.ctypes.data_as(_ip),
            ce.ctypes.data_as(_ip), names, ptrs)

    def write_curvilinear_mesh(self, filename, ub, dims, pts, nvars, vardim, centering, varnames, vars_):
        keep, names, ptrs = _vars(varnames, vars_)
        d, vd, ce, p = _i32(dims), _i32(vardim), _i32(centering), _f32(pts)
        getattr(self.lib, MANGLED["write_curvilinear_mesh"])(
            os.fsencode(filename), C.c_int(ub), d.ctypes.data_as(_ip), p.ctypes.data_as(_fp), C.c_int(nvars),
            vd.ctypes.data_as(_ip), ce.ctypes.data_as(_ip), names, ptrs)
