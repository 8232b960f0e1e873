"""ctypes binding of libsfvtk.so — the legacy-VTK frame writer (include/sf_visit_writer.h).

Mirrors the five entry points of the reference's visit_writer.h (:94-96, :176-179, :216-218,
:262-265, :303-306): same names, same argument order and meaning. Raises OSError if the file
cannot be written; raises ImportError at import time if the library has not been built.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "libsfvtk.so")
if not os.path.exists(_LIB_PATH):
    raise ImportError(f"{_LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
_lib = C.CDLL(_LIB_PATH)

_fp = C.POINTER(C.c_float)
_ip = C.POINTER(C.c_int)
_cpp = C.POINTER(C.c_char_p)
_fpp = C.POINTER(_fp)

_lib.sf_vtk_write_point_mesh.argtypes = [C.c_char_p, C.c_int, C.c_int, _fp, C.c_int, _ip, _cpp, _fpp]
_lib.sf_vtk_write_unstructured_mesh.argtypes = [C.c_char_p, C.c_int, C.c_int, _fp, C.c_int, _ip, _ip,
                                                C.c_int, _ip, _ip, _cpp, _fpp]
_lib.sf_vtk_write_regular_mesh.argtypes = [C.c_char_p, C.c_int, _ip, C.c_int, _ip, _ip, _cpp, _fpp]
_lib.sf_vtk_write_rectilinear_mesh.argtypes = [C.c_char_p, C.c_int, _ip, _fp, _fp, _fp, C.c_int, _ip, _ip,
                                               _cpp, _fpp]
_lib.sf_vtk_write_curvilinear_mesh.argtypes = [C.c_char_p, C.c_int, _ip, _fp, C.c_int, _ip, _ip, _cpp, _fpp]
for _n in ("point", "unstructured", "regular", "rectilinear", "curvilinear"):
    getattr(_lib, f"sf_vtk_write_{_n}_mesh").restype = C.c_int


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _i32(a):
    return np.ascontiguousarray(a, dtype=np.int32)


class _Vars:
    """Keeps the marshalled arrays alive for the duration of one call."""

    def __init__(self, vardim, varnames, vars_, centering=None):
        self.n = len(vars_)
        self.arrays = [_f32(v).ravel() for v in vars_]
        self.vardim = _i32(vardim)
        self.centering = _i32(centering) if centering is not None else None
        self.names = (C.c_char_p * max(self.n, 1))(*[s.encode() for s in varnames])
        self.ptrs = (_fp * max(self.n, 1))(*[a.ctypes.data_as(_fp) for a in self.arrays])

    @property
    def vd(self):
        return self.vardim.ctypes.data_as(_ip)

    @property
    def ce(self):
        return self.centering.ctypes.data_as(_ip)


def _check(rc, filename):
    if rc != 0:
        raise OSError(f"sf_vtk: could not write {filename}")


def write_point_mesh(filename, useBinary, npts, pts, nvars, vardim, varnames, vars):
    v = _Vars(vardim, varnames, vars)
    p = _f32(pts).ravel()
    _check(_lib.sf_vtk_write_point_mesh(os.fsencode(filename), int(useBinary), int(npts), p.ctypes.data_as(_fp),
                                        int(nvars), v.vd, v.names, v.ptrs), filename)


def write_unstructured_mesh(filename, useBinary, npts, pts, ncells, celltypes, conn, nvars, vardim, centering,
                            varnames, vars):
    v = _Vars(vardim, varnames, vars, centering)
    p = _f32(pts).ravel()
    ct, cn = _i32(celltypes), _i32(conn)
    _check(_lib.sf_vtk_write_unstructured_mesh(os.fsencode(filename), int(useBinary), int(npts),
                                               p.ctypes.data_as(_fp), int(ncells), ct.ctypes.data_as(_ip),
                                               cn.ctypes.data_as(_ip), int(nvars), v.vd, v.ce, v.names, v.ptrs),
           filename)


def write_regular_mesh(filename, useBinary, dims, nvars, vardim, centering, varnames, vars):
    v = _Vars(vardim, varnames, vars, centering)
    d = _i32(dims)
    _check(_lib.sf_vtk_write_regular_mesh(os.fsencode(filename), int(useBinary), d.ctypes.data_as(_ip), int(nvars),
                                          v.vd, v.ce, v.names, v.ptrs), filename)


def write_rectilinear_mesh(filename, useBinary, dims, x, y, z, nvars, vardim, centering, varnames, vars):
    v = _Vars(vardim, varnames, vars, centering)
    d = _i32(dims)
    xs, ys, zs = _f32(x), _f32(y), _f32(z)
    _check(_lib.sf_vtk_write_rectilinear_mesh(os.fsencode(filename), int(useBinary), d.ctypes.data_as(_ip),
                                              xs.ctypes.data_as(_fp), ys.ctypes.data_as(_fp),
                                              zs.ctypes.data_as(_fp), int(nvars), v.vd, v.ce, v.names, v.ptrs),
           filename)


def write_curvilinear_mesh(filename, useBinary, dims, pts, nvars, vardim, centering, varnames, vars):
    v = _Vars(vardim, varnames, vars, centering)
    d = _i32(dims)
    p = _f32(pts).ravel()
    _check(_lib.sf_vtk_write_curvilinear_mesh(os.fsencode(filename), int(useBinary), d.ctypes.data_as(_ip),
                                              p.ctypes.data_as(_fp), int(nvars), v.vd, v.ce, v.names, v.ptrs),
           filename)
