"""VTK frame writer parity: fluidsolvergpu_amd/csrc/sf_vtk.cpp vs the reference's visit_writer.

Pinned two ways: (1) byte-equality with tests/golden/vtk/*.vtk, files written by the REFERENCE binary
(tests/golden/make_vtk_golden.py); (2) when oracle/_ref is present, byte-equality on fresh random cases
written by both libraries in this process."""
import hashlib
import json
import os

import numpy as np
import pytest

import ref_writer
import vtk_cases
from fluidsolvergpu_amd import vtk as sfvtk

GOLD = os.path.join(os.path.dirname(__file__), "golden", "vtk")
MANIFEST = json.load(open(os.path.join(GOLD, "MANIFEST.json")))

# hashes of files produced by the compiled reference writer.
SURVEY_SHA = {
    "reg_ascii.vtk": "268453522d9d882bf4c933f1f1693a94f53ae207147aecd560ab782b1d1d8a56",
    "reg_bin.vtk": "95cc91b9b4306210097be2d7960e73963efdeaa6094bf7b8ca8aae7cd7b04e96",
    "pt_ascii.vtk": "e1c2d6bfdde7c319c2aaa9dbe963c8dfc50c7d4b4c65709ad06ec14185ee3634",
}


class _Product:
    """Adapts the package functions to the (…, vars_) calling convention of vtk_cases.run_case."""

    def __getattr__(self, name):
        return getattr(sfvtk, name)


class _ProductMangled(ref_writer.Writer):
    """Calls libsfvtk.so through the reference's C++-linkage names (drop-in link check)."""

    def __init__(self):
        super().__init__(os.path.join(os.path.dirname(sfvtk.__file__), "libsfvtk.so"))


@pytest.mark.parametrize("case", vtk_cases.cases(), ids=lambda c: c[0])
def test_matches_golden(case, tmp_path):
    path = vtk_cases.run_case(_Product(), case, str(tmp_path))
    got = open(path, "rb").read()
    want = open(os.path.join(GOLD, os.path.basename(path)), "rb").read()
    assert got == want
    assert hashlib.sha256(got).hexdigest() == MANIFEST[os.path.basename(path)]["sha256"]


def test_golden_matches_survey_hashes():
    for name, sha in SURVEY_SHA.items():
        assert MANIFEST[name]["sha256"] == sha
        assert hashlib.sha256(open(os.path.join(GOLD, name), "rb").read()).hexdigest() == sha


@pytest.mark.parametrize("case", vtk_cases.cases()[:6], ids=lambda c: c[0])
def test_cxx_linkage_names_are_drop_in(case, tmp_path):
    """The reference's mangled C++ names resolve in libsfvtk.so and write the same bytes."""
    path = vtk_cases.run_case(_ProductMangled(), case, str(tmp_path))
    assert open(path, "rb").read() == open(os.path.join(GOLD, os.path.basename(path)), "rb").read()


@pytest.mark.skipif(not ref_writer.available(), reason="oracle/_ref not built (reference absent)")
@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("ub", [0, 1])
def test_random_regular_mesh_vs_live_reference(seed, ub, tmp_path):
    rng = np.random.RandomState(100 + seed)
    dims = [int(rng.randint(1, 12)) for _ in range(3)]
    npts = dims[0] * dims[1] * dims[2]
    ncells = max(dims[0] - 1, 1) * max(dims[1] - 1, 1) * max(dims[2] - 1, 1)
    nvars = int(rng.randint(0, 6))
    vardim = [int(rng.choice([1, 3])) for _ in range(nvars)]
    centering = [int(rng.randint(0, 2)) for _ in range(nvars)]
    names = [f"var{q}" for q in range(nvars)]
    scale = 10.0 ** rng.randint(-30, 30, size=nvars)
    vars_ = [(rng.standard_normal((npts if c else ncells) * d) * s).astype(np.float32)
             for d, c, s in zip(vardim, centering, scale)]
    a, b = str(tmp_path / "ref.vtk"), str(tmp_path / "mine.vtk")
    ref_writer.Writer().write_regular_mesh(a, ub, dims, nvars, vardim, centering, names, vars_)
    sfvtk.write_regular_mesh(b, ub, dims, nvars, vardim, centering, names, vars_)
    assert open(a, "rb").read() == open(b, "rb").read()


@pytest.mark.skipif(not ref_writer.available(), reason="oracle/_ref not built (reference absent)")
def test_float_formatting_vs_live_reference(tmp_path):
    """%20.12e of 200k random bit patterns (all exponents, denormals, inf/nan) — to_chars == printf."""
    rng = np.random.RandomState(7)
    bits = rng.randint(0, 2 ** 32, size=200_000, dtype=np.uint64).astype(np.uint32)
    vals = bits.view(np.float32)
    n = len(vals)
    a, b = str(tmp_path / "ref.vtk"), str(tmp_path / "mine.vtk")
    args = (0, n, np.zeros(3 * n, np.float32), 1, [1], ["bits"], [vals])
    ref_writer.Writer().write_point_mesh(a, *args)
    sfvtk.write_point_mesh(b, *args)
    assert open(a, "rb").read() == open(b, "rb").read()


def test_unwritable_path_raises(tmp_path):
    with pytest.raises(OSError):
        sfvtk.write_regular_mesh(str(tmp_path / "no_such_dir" / "x.vtk"), 0, [2, 2, 2], 0, [], [], [], [])


def test_extension_rule(tmp_path):
    """'.vtk' is appended only if the name does not already contain it (visit_writer.cpp:136-143)."""
    sfvtk.write_regular_mesh(str(tmp_path / "frame"), 1, [2, 2, 2], 0, [], [], [], [])
    sfvtk.write_regular_mesh(str(tmp_path / "a.vtk.bak"), 1, [2, 2, 2], 0, [], [], [], [])
    assert (tmp_path / "frame.vtk").exists() and (tmp_path / "a.vtk.bak").exists()
