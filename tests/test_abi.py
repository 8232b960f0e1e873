"""The C-ABI libraries load on a CPU-only machine and export every symbol the headers declare.
No compute call is made here (there is no GPU); creating a context must fail loudly instead of
falling back to the CPU."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    return sorted(set(re.findall(r"\b(?:int|void|const char\s*\*)\s+(\w+)\s*\(", text)))


def test_sfgpu_exports_every_declared_symbol():
    from fluidsolvergpu_amd import solver

    names = declared_functions("sfgpu.h")
    assert "vel_step" in names and "dens_step" in names and len(names) >= 25
    assert sorted(names) == sorted(solver.ABI_SYMBOLS)
    for n in names:
        assert hasattr(solver.lib, n), f"libsfgpu.so does not export {n}"


def test_sfvtk_exports_every_declared_symbol():
    from fluidsolvergpu_amd import vtk  # noqa: F401  (import fails loudly if the library is missing)

    lib = C.CDLL(os.path.join(ROOT, "fluidsolvergpu_amd", "libsfvtk.so"))
    names = [n for n in declared_functions("sf_visit_writer.h") if n.startswith("sf_vtk_")]
    assert len(names) == 5
    for n in names:
        assert hasattr(lib, n)
    # the reference's C++-linkage names (visit_writer.h:94-96,176-179,216-218,262-265,303-306)
    out = subprocess.run(["nm", "-DC", os.path.join(ROOT, "fluidsolvergpu_amd", "libsfvtk.so")],
                         capture_output=True, text=True, check=True).stdout
    for n in ("write_point_mesh", "write_unstructured_mesh", "write_regular_mesh", "write_rectilinear_mesh",
              "write_curvilinear_mesh"):
        assert re.search(rf" T {n}\(", out), n


def test_gpu_code_object_is_gfx950():
    out = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes",
                          os.path.join(ROOT, "fluidsolvergpu_amd", "libsfgpu.so")],
                         capture_output=True, text=True).stdout
    raw = open(os.path.join(ROOT, "fluidsolvergpu_amd", "libsfgpu.so"), "rb").read()
    assert b"gfx950" in raw and b"jacobi_sk_kernel" in raw and b"advect_row_kernel" in raw, out[:200]


def _has_gpu():
    return os.path.exists("/dev/kfd")


@pytest.mark.skipif(_has_gpu(), reason="a GPU is present; the no-device path cannot be observed")
def test_no_cpu_fallback():
    from fluidsolvergpu_amd import solver

    with pytest.raises(solver.SfError) as e:
        solver.FluidSolver(8)
    assert e.value.status == solver.SF_ERR_NO_DEVICE


def test_status_strings():
    from fluidsolvergpu_amd import solver

    assert solver.lib.sf_status_string(0) == b"SF_OK"
    assert solver.lib.sf_status_string(4) == b"SF_ERR_HALO_EXCEEDED"
    assert "gfx950" in solver.version()


def test_driver_rejects_rank_override_without_loopback():
    """`sf_driver --rank r --world w` rehearses one rank's share with local copies (--loopback); without it a real
    communicator would wait for an id nobody publishes: refused at once, before anything touches the device."""
    exe = os.path.join(ROOT, "fluidsolvergpu_amd", "sf_driver")
    out = subprocess.run([exe, "--rank", "1", "--world", "2", "--every", "0"], capture_output=True, text=True, timeout=60)
    assert out.returncode == 2 and "--loopback" in out.stderr


def test_switch_table_matches_the_library_and_the_switch_tests():
    """INTEGRATION.md §5 documents exactly the SF_* switches libsfgpu.so reads (at most twenty), and each of them is
    exercised by a GPU test (tests/test_switches_gpu.py, or the test the table / that file names)."""
    src = ""
    for f in ("sf_solver.hpp", "sf_base.hpp", "sf_api.hip", "sf_kernels.hpp"):
        src += open(os.path.join(ROOT, "fluidsolvergpu_amd", "csrc", f)).read()
    read = set(re.findall(r'env_int\("(SF_[A-Z0-9_]+)"', src)) | set(re.findall(r'getenv\("(SF_[A-Z0-9_]+)"\)', src))
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    table = doc[doc.index("## 5. Environment switches"):]
    documented = set(re.findall(r"^\| `(SF_[A-Z0-9_]+)` \|", table, flags=re.M))
    assert read == documented, (sorted(read - documented), sorted(documented - read))
    assert len(documented) <= 20
    tests = ""
    for f in ("test_switches_gpu.py", "test_parity_gpu.py", "test_schedule_check.py", "conftest.py"):
        tests += open(os.path.join(ROOT, "tests", f)).read()
    missing = [s for s in documented if s not in tests]
    assert not missing, missing
