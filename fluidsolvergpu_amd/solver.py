"""Host-side mirror of the C ABI in include/sfgpu.h (ctypes; numpy arrays in, numpy arrays out).

`FluidSolver.vel_step()` / `.dens_step()` are the two entry points BASELINE.json's north_star names;
field names follow docs/SPEC.md. All numerics run in libsfgpu.so (hand-written gfx950 HIP); this
module has no fallback: if the library is missing the import fails, and if no MI355X is visible
`FluidSolver(...)` raises SfError(SF_ERR_NO_DEVICE).
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, os.environ.get("SF_LIB", "libsfgpu.so"))  # SF_LIB: experimental builds
if not os.path.exists(LIB_PATH):
    raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
lib = C.CDLL(LIB_PATH)

SF_OK, SF_ERR_INVALID, SF_ERR_HIP, SF_ERR_RCCL, SF_ERR_HALO_EXCEEDED, SF_ERR_NO_DEVICE = range(6)
SF_F32, SF_F64 = 0, 1
FIELD_IDS = {"u": 0, "v": 1, "w": 2, "u0": 3, "v0": 4, "w0": 5, "dens": 6, "dens0": 7,
             "user0": 8, "user1": 9, "user2": 10, "user3": 11}
FIELD_NAMES = ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")
NCCL_ID_BYTES = 128
SF_FLAG_LOOPBACK_HALO, SF_FLAG_RCCL_SELF = 1, 2
TRANSPORTS = ("none", "copy", "rccl", "rccl-self", "loopback")

# every symbol include/sfgpu.h declares (tests check that the library exports all of them)
ABI_SYMBOLS = (
    "sf_version", "sf_status_string", "sf_nccl_unique_id", "sf_create", "sf_destroy", "sf_upload",
    "sf_download", "sf_download_planes", "sf_upload_planes", "sf_owned_planes", "sf_stored_planes", "sf_fill", "sf_copy_field", "vel_step",
    "dens_step", "sf_add_source", "sf_set_bnd", "sf_lin_solve", "sf_diffuse", "sf_advect", "sf_project",
    "sf_set_iters", "sf_set_coefficients", "sf_sync", "sf_last_error", "sf_timer_start", "sf_timer_stop",
    "sf_measure_copy_bandwidth", "sf_layout_info", "sf_schedule_info", "sf_lin_solve_launches", "sf_snapshot", "sf_snapshot_read",
    "sf_tracers_set", "sf_tracers_advect", "sf_tracers_get", "sf_bind_sources", "sf_transport_info", "sf_snapshot_read_planes",
)


class SfParams(C.Structure):
    _fields_ = [("N", C.c_int), ("dtype", C.c_int), ("iters", C.c_int), ("dt", C.c_double),
                ("diff", C.c_double), ("visc", C.c_double), ("device", C.c_int), ("nslabs_local", C.c_int),
                ("rank", C.c_int), ("nranks", C.c_int), ("nccl_id", C.c_void_p), ("flags", C.c_int)]


_ctx = C.c_void_p
lib.sf_version.restype = C.c_char_p
lib.sf_status_string.restype = C.c_char_p
lib.sf_status_string.argtypes = [C.c_int]
lib.sf_last_error.restype = C.c_char_p
lib.sf_last_error.argtypes = [_ctx]
lib.sf_nccl_unique_id.argtypes = [C.c_void_p]
lib.sf_create.argtypes = [C.POINTER(_ctx), C.POINTER(SfParams)]
lib.sf_destroy.argtypes = [_ctx]
lib.sf_destroy.restype = None
lib.sf_upload.argtypes = [_ctx, C.c_int, C.c_void_p]
lib.sf_download.argtypes = [_ctx, C.c_int, C.c_void_p]
lib.sf_download_planes.argtypes = [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.sf_upload_planes.argtypes = [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.sf_owned_planes.argtypes = [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.sf_stored_planes.argtypes = [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.sf_fill.argtypes = [_ctx, C.c_int, C.c_double]
lib.sf_copy_field.argtypes = [_ctx, C.c_int, C.c_int]
lib.vel_step.argtypes = [_ctx]
lib.dens_step.argtypes = [_ctx]
lib.sf_add_source.argtypes = [_ctx, C.c_int, C.c_int]
lib.sf_set_bnd.argtypes = [_ctx, C.c_int, C.c_int]
lib.sf_lin_solve.argtypes = [_ctx, C.c_int, C.c_int, C.c_int, C.c_double, C.c_double, C.c_int]
lib.sf_diffuse.argtypes = [_ctx, C.c_int, C.c_int, C.c_int, C.c_double]
lib.sf_advect.argtypes = [_ctx] + [C.c_int] * 6
lib.sf_project.argtypes = [_ctx] + [C.c_int] * 5
lib.sf_set_iters.argtypes = [_ctx, C.c_int]
lib.sf_set_coefficients.argtypes = [_ctx, C.c_double, C.c_double, C.c_double]
lib.sf_sync.argtypes = [_ctx]
lib.sf_timer_start.argtypes = [_ctx]
lib.sf_timer_stop.argtypes = [_ctx, C.POINTER(C.c_float)]
lib.sf_measure_copy_bandwidth.argtypes = [_ctx, C.c_size_t, C.c_int, C.POINTER(C.c_double)]
lib.sf_lin_solve_launches.argtypes = [_ctx, C.c_int]
lib.sf_bind_sources.argtypes = [_ctx, C.c_int, C.c_int, C.c_int, C.c_int]
lib.sf_snapshot.argtypes = [_ctx, C.POINTER(C.c_int), C.c_int]
lib.sf_snapshot_read.argtypes = [_ctx, C.c_int, C.c_void_p]
lib.sf_snapshot_read_planes.argtypes = [_ctx, C.c_int, C.c_int, C.c_int, C.c_void_p]
lib.sf_tracers_set.argtypes = [_ctx, C.c_int, C.c_void_p]
lib.sf_tracers_advect.argtypes = [_ctx]
lib.sf_tracers_get.argtypes = [_ctx, C.c_void_p, C.c_void_p, C.c_void_p]
lib.sf_layout_info.argtypes = [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_size_t)]
lib.sf_schedule_info.argtypes = [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_int)]
lib.sf_transport_info.argtypes = [_ctx, C.POINTER(C.c_int), C.POINTER(C.c_long)]


class SfError(RuntimeError):
    def __init__(self, status, message):
        super().__init__(f"{lib.sf_status_string(status).decode()}: {message}")
        self.status = status


def version():
    return lib.sf_version().decode()


def nccl_unique_id():
    buf = C.create_string_buffer(NCCL_ID_BYTES)
    rc = lib.sf_nccl_unique_id(buf)
    if rc != SF_OK:
        raise SfError(rc, "ncclGetUniqueId failed")
    return buf.raw


def _fid(f):
    return FIELD_IDS[f] if isinstance(f, str) else int(f)


class FluidSolver:
    """One context of libsfgpu.so. Arrays are (N+2,N+2,N+2) numpy arrays indexed [k, j, i]
    (C order, i fastest), i.e. exactly the dense IX(i,j,k) layout of docs/SPEC.md."""

    def __init__(self, N, dtype="f32", iters=20, dt=0.1, diff=1e-4, visc=1e-4, device=0, nslabs_local=1,
                 rank=0, nranks=1, nccl_id=None, flags=0):
        self.N = int(N)
        self.np_dtype = np.float32 if dtype in ("f32", np.float32, SF_F32) else np.float64
        self._id_buf = C.create_string_buffer(nccl_id, NCCL_ID_BYTES) if nccl_id is not None else None
        p = SfParams(N=self.N, dtype=SF_F32 if self.np_dtype == np.float32 else SF_F64, iters=int(iters),
                     dt=float(dt), diff=float(diff), visc=float(visc), device=int(device),
                     nslabs_local=int(nslabs_local), rank=int(rank), nranks=int(nranks),
                     nccl_id=C.cast(self._id_buf, C.c_void_p) if self._id_buf is not None else None, flags=int(flags))
        self._h = _ctx()
        rc = lib.sf_create(C.byref(self._h), C.byref(p))
        if rc != SF_OK:
            self._h = None
            raise SfError(rc, lib.sf_last_error(None).decode())

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None):
            lib.sf_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        self.close()

    def _ck(self, rc):
        if rc != SF_OK:
            raise SfError(rc, lib.sf_last_error(self._h).decode())

    # -- data ------------------------------------------------------------------------------
    @property
    def shape(self):
        return (self.N + 2,) * 3

    def upload(self, field, array):
        a = np.ascontiguousarray(array, dtype=self.np_dtype)
        if a.shape != self.shape:
            raise ValueError(f"expected shape {self.shape}, got {a.shape}")
        self._ck(lib.sf_upload(self._h, _fid(field), a.ctypes.data_as(C.c_void_p)))

    def download(self, field, out=None):
        if out is None:
            out = np.zeros(self.shape, self.np_dtype)
        assert out.dtype == self.np_dtype and out.shape == self.shape and out.flags.c_contiguous
        self._ck(lib.sf_download(self._h, _fid(field), out.ctypes.data_as(C.c_void_p)))
        return out

    def download_planes(self, field, k_begin, k_end):
        out = np.zeros((k_end - k_begin, self.N + 2, self.N + 2), self.np_dtype)
        self._ck(lib.sf_download_planes(self._h, _fid(field), int(k_begin), int(k_end),
                                        out.ctypes.data_as(C.c_void_p)))
        return out

    def upload_planes(self, field, k_begin, array):
        a = np.ascontiguousarray(array, dtype=self.np_dtype)
        if a.ndim != 3 or a.shape[1:] != (self.N + 2, self.N + 2):
            raise ValueError(f"expected shape (nplanes,{self.N + 2},{self.N + 2}), got {a.shape}")
        self._ck(lib.sf_upload_planes(self._h, _fid(field), int(k_begin), int(k_begin) + a.shape[0],
                                      a.ctypes.data_as(C.c_void_p)))

    def stored_planes(self):
        """Global planes this context stores: its interior planes plus its ghost / shell planes."""
        kb, ke = C.c_int(), C.c_int()
        self._ck(lib.sf_stored_planes(self._h, C.byref(kb), C.byref(ke)))
        return kb.value, ke.value

    def owned_planes(self):
        kb, ke = C.c_int(), C.c_int()
        self._ck(lib.sf_owned_planes(self._h, C.byref(kb), C.byref(ke)))
        return kb.value, ke.value

    def fill(self, field, value):
        self._ck(lib.sf_fill(self._h, _fid(field), float(value)))

    def copy_field(self, dst, src):
        self._ck(lib.sf_copy_field(self._h, _fid(dst), _fid(src)))

    def bind_sources(self, su="user0", sv="user1", sw="user2", sd="user3"):
        ids = [(-1 if f is None else _fid(f)) for f in (su, sv, sw, sd)]
        self._ck(lib.sf_bind_sources(self._h, *ids))

    # -- the path --------------------------------------------------------------------------
    def vel_step(self):
        self._ck(lib.vel_step(self._h))

    def dens_step(self):
        self._ck(lib.dens_step(self._h))

    def add_source(self, x, s):
        self._ck(lib.sf_add_source(self._h, _fid(x), _fid(s)))

    def set_bnd(self, b, x):
        self._ck(lib.sf_set_bnd(self._h, int(b), _fid(x)))

    def lin_solve(self, b, x, x0, a, c, iters):
        self._ck(lib.sf_lin_solve(self._h, int(b), _fid(x), _fid(x0), float(a), float(c), int(iters)))

    def diffuse(self, b, x, x0, diff):
        self._ck(lib.sf_diffuse(self._h, int(b), _fid(x), _fid(x0), float(diff)))

    def advect(self, b, d, d0, u, v, w):
        self._ck(lib.sf_advect(self._h, int(b), _fid(d), _fid(d0), _fid(u), _fid(v), _fid(w)))

    def project(self, u, v, w, p, div):
        self._ck(lib.sf_project(self._h, _fid(u), _fid(v), _fid(w), _fid(p), _fid(div)))

    def set_iters(self, iters):
        self._ck(lib.sf_set_iters(self._h, int(iters)))

    def set_coefficients(self, dt, diff, visc):
        self._ck(lib.sf_set_coefficients(self._h, float(dt), float(diff), float(visc)))

    def sync(self):
        self._ck(lib.sf_sync(self._h))

    # -- measurement -----------------------------------------------------------------------
    def timer_start(self):
        self._ck(lib.sf_timer_start(self._h))

    def timer_stop(self):
        ms = C.c_float()
        self._ck(lib.sf_timer_stop(self._h, C.byref(ms)))
        return ms.value

    def copy_bandwidth_gbps(self, nbytes=1 << 30, reps=5):
        g = C.c_double()
        self._ck(lib.sf_measure_copy_bandwidth(self._h, int(nbytes), int(reps), C.byref(g)))
        return g.value

    # -- asynchronous output / tracers ------------------------------------------------------
    def snapshot(self, fields):
        ids = (C.c_int * len(fields))(*[_fid(f) for f in fields])
        self._ck(lib.sf_snapshot(self._h, ids, len(fields)))

    def snapshot_read(self, index, out=None):
        if out is None:
            out = np.zeros(self.shape, self.np_dtype)
        rc = lib.sf_snapshot_read(self._h, int(index), out.ctypes.data_as(C.c_void_p))
        if rc != SF_OK:
            raise SfError(rc, "sf_snapshot_read failed")
        return out

    def snapshot_read_planes(self, index, k_begin, k_end):
        out = np.zeros((k_end - k_begin, self.N + 2, self.N + 2), self.np_dtype)
        rc = lib.sf_snapshot_read_planes(self._h, int(index), int(k_begin), int(k_end), out.ctypes.data_as(C.c_void_p))
        if rc != SF_OK:
            raise SfError(rc, "sf_snapshot_read_planes failed")
        return out

    def tracers_set(self, xyz):
        a = np.ascontiguousarray(xyz, dtype=self.np_dtype).reshape(-1, 3)
        self._ntr = a.shape[0]
        self._ck(lib.sf_tracers_set(self._h, self._ntr, a.ctypes.data_as(C.c_void_p)))

    def tracers_advect(self):
        self._ck(lib.sf_tracers_advect(self._h))

    def tracers_get(self, sample=True):
        n = getattr(self, "_ntr", 0)
        xyz = np.zeros((n, 3), self.np_dtype)
        dens = np.zeros(n, self.np_dtype)
        speed = np.zeros(n, self.np_dtype)
        self._ck(lib.sf_tracers_get(self._h, xyz.ctypes.data_as(C.c_void_p),
                                    dens.ctypes.data_as(C.c_void_p) if sample else None,
                                    speed.ctypes.data_as(C.c_void_p) if sample else None))
        return xyz, dens, speed

    def lin_solve_launches(self, iters):
        return int(lib.sf_lin_solve_launches(self._h, int(iters)))

    def schedule_info(self):
        trap, measured = C.c_int(), C.c_int()
        self._ck(lib.sf_schedule_info(self._h, C.byref(trap), C.byref(measured)))
        m = measured.value
        return {"trapezoid_pairs": trap.value, "measured": bool(m & 1), "fields_measured": bool(m & 2),
                "fields_per_launch": 3 if (m & 4) else 1}

    def transport_info(self):
        """Halo transport of this context and how many RCCL send/recv groups it has issued so far."""
        t, g = C.c_int(), C.c_long()
        self._ck(lib.sf_transport_info(self._h, C.byref(t), C.byref(g)))
        return {"transport": TRANSPORTS[t.value], "rccl_groups": g.value}

    def layout_info(self):
        pitch, planes, nbytes = C.c_int(), C.c_int(), C.c_size_t()
        self._ck(lib.sf_layout_info(self._h, C.byref(pitch), C.byref(planes), C.byref(nbytes)))
        return {"row_pitch": pitch.value, "planes_per_slab": planes.value, "bytes_per_field": nbytes.value}
