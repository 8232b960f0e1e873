"""Deterministic VTK writer cases shared by the golden generator and the tests.

Each case is (name, function, kwargs) where `function` is one of the five writer entry points and the
file name is filled in by the caller. Cases 'reg_ascii', 'reg_bin', 'pt_ascii' are exactly the probes of
SURVEY.md Appendix B (their SHA-256 is asserted in the tests)."""
import numpy as np


def _rng_vals(n, seed):
    rng = np.random.RandomState(seed)
    v = rng.standard_normal(n).astype(np.float32)
    # sprinkle awkward values: zeros, negative zero, denormal, large, tiny
    specials = np.array([0.0, -0.0, 1e-42, -3.4e38, 1.17549435e-38, 123456.789, -1e-7, 1.0], np.float32)
    v[: min(n, len(specials))] = specials[: min(n, len(specials))]
    return v


def cases():
    out = []
    # --- SURVEY Appendix B probes ---------------------------------------------------------------
    dens = (0.5 * np.arange(8)).astype(np.float32)
    vel = (0.25 * np.arange(24)).astype(np.float32)
    reg = dict(dims=[3, 3, 3], nvars=2, vardim=[1, 3], centering=[0, 0], varnames=["dens", "vel"],
               vars_=[dens, vel])
    out.append(("reg_ascii", "write_regular_mesh", 0, reg, False))  # name without .vtk: suffix gets appended
    out.append(("reg_bin", "write_regular_mesh", 1, reg, True))
    pt = dict(npts=2, pts=[0, 0, 0, 1, 2, 3], nvars=2, vardim=[1, 1], varnames=["dens", "cellnumber"],
              vars_=[np.array([7, 8], np.float32), np.array([9, 10], np.float32)])
    out.append(("pt_ascii", "write_point_mesh", 0, pt, True))
    out.append(("pt_bin", "write_point_mesh", 1, pt, True))

    # --- multiples of 9 (blank-line rule), mixed centering, field-data blocks -----------------------
    dims = [4, 3, 2]  # 24 points, 3*2*1 = 6 cells
    npts, ncells = 24, 6
    many = dict(dims=dims, nvars=7, vardim=[1, 3, 1, 3, 1, 1, 3], centering=[0, 0, 0, 0, 1, 1, 1],
                varnames=["c_s0", "c_v0", "c_s1", "c_v1", "p_s0", "p_s1", "p_v0"],
                vars_=[_rng_vals(ncells, 1), _rng_vals(3 * ncells, 2), _rng_vals(ncells, 3),
                       _rng_vals(3 * ncells, 4), _rng_vals(npts, 5), _rng_vals(npts, 6), _rng_vals(3 * npts, 7)])
    out.append(("reg_many_ascii", "write_regular_mesh", 0, many, True))
    out.append(("reg_many_bin", "write_regular_mesh", 1, many, True))
    nine = dict(dims=[10, 4, 4], nvars=1, vardim=[1], centering=[0], varnames=["nine"],
                vars_=[_rng_vals(81, 8)])  # 9*3*3 = 81 cells -> run length multiple of 9; 10 x-coords
    out.append(("reg_nine_ascii", "write_regular_mesh", 0, nine, True))
    flat = dict(dims=[5, 1, 1], nvars=1, vardim=[3], centering=[1], varnames=["flatvec"],
                vars_=[_rng_vals(15, 9)])  # degenerate axes: cells = max(dims-1,1)
    out.append(("reg_flat_ascii", "write_regular_mesh", 0, flat, True))
    novars = dict(dims=[2, 2, 2], nvars=0, vardim=[], centering=[], varnames=[], vars_=[])
    out.append(("reg_novars_ascii", "write_regular_mesh", 0, novars, True))

    rect = dict(dims=[3, 4, 2], x=[0, 1, 2], y=[1, 1.5, 2, 3], z=[2.5, 3.5], nvars=2, vardim=[1, 1],
                centering=[1, 0], varnames=["pointvar", "cellvar"], vars_=[_rng_vals(24, 10), _rng_vals(6, 11)])
    out.append(("rect_ascii", "write_rectilinear_mesh", 0, rect, True))
    out.append(("rect_bin", "write_rectilinear_mesh", 1, rect, True))

    cdims = [3, 2, 2]
    grid = np.stack(np.meshgrid(np.arange(3), np.arange(2), np.arange(2), indexing="ij"), -1)
    cpts = (grid.transpose(2, 1, 0, 3).reshape(-1, 3) * np.array([1.0, 0.5, 0.25]) + 0.125).astype(np.float32)
    curv = dict(dims=cdims, pts=cpts, nvars=2, vardim=[3, 1], centering=[1, 0], varnames=["pv", "cs"],
                vars_=[_rng_vals(36, 12), _rng_vals(2, 13)])
    out.append(("curv_ascii", "write_curvilinear_mesh", 0, curv, True))
    out.append(("curv_bin", "write_curvilinear_mesh", 1, curv, True))

    # two triangles + a tetra + a vertex, the example of visit_writer.h's comment extended
    upts = np.array([0, 0, 0, 0, 1, 0, 1, 1, 0, 1, 0, 0, 0.5, 0.5, 1], np.float32)
    uns = dict(npts=5, pts=upts, ncells=4, celltypes=[5, 5, 10, 1], conn=[0, 1, 2, 0, 2, 3, 0, 1, 2, 4, 4],
               nvars=3, vardim=[1, 3, 1], centering=[0, 1, 1], varnames=["cell_s", "pt_v", "pt_s"],
               vars_=[_rng_vals(4, 14), _rng_vals(15, 15), _rng_vals(5, 16)])
    out.append(("uns_ascii", "write_unstructured_mesh", 0, uns, True))
    out.append(("uns_bin", "write_unstructured_mesh", 1, uns, True))

    # the one call the reference really makes (solver-unidyn.cu:487): ASCII point mesh, two scalars
    n = 20
    ppts = _rng_vals(3 * n, 17)
    pm = dict(npts=n, pts=ppts, nvars=2, vardim=[1, 1], varnames=["mass", "surface_level"],
              vars_=[_rng_vals(n, 18), _rng_vals(n, 19)])
    out.append(("pt_unidyn_ascii", "write_point_mesh", 0, pm, True))
    return out


def run_case(writer, case, directory):
    """writer: object or module with the five functions. Returns the path of the written file."""
    import os

    name, fn, ub, kw, with_ext = case
    path = os.path.join(directory, name + (".vtk" if with_ext else ""))
    f = getattr(writer, fn)
    if fn == "write_point_mesh":
        f(path, ub, kw["npts"], kw["pts"], kw["nvars"], kw["vardim"], kw["varnames"], kw["vars_"])
    elif fn == "write_unstructured_mesh":
        f(path, ub, kw["npts"], kw["pts"], kw["ncells"], kw["celltypes"], kw["conn"], kw["nvars"], kw["vardim"],
          kw["centering"], kw["varnames"], kw["vars_"])
    elif fn == "write_regular_mesh":
        f(path, ub, kw["dims"], kw["nvars"], kw["vardim"], kw["centering"], kw["varnames"], kw["vars_"])
    elif fn == "write_rectilinear_mesh":
        f(path, ub, kw["dims"], kw["x"], kw["y"], kw["z"], kw["nvars"], kw["vardim"], kw["centering"],
          kw["varnames"], kw["vars_"])
    elif fn == "write_curvilinear_mesh":
        f(path, ub, kw["dims"], kw["pts"], kw["nvars"], kw["vardim"], kw["centering"], kw["varnames"], kw["vars_"])
    else:
        raise ValueError(fn)
    return os.path.join(directory, name + ".vtk")
