"""BASELINE.json configs[0]: 32^3 grid, one density + velocity source, 10 Jacobi iterations, a single
vel_step + dens_step on the HOST (the CPU oracle) written as a .vtk frame — plumbing only, no GPU.
The frame layout is the one fluidsolvergpu_amd/csrc/sf_driver.cpp writes (density scalar + velocity vector,
cell centred, dims = N+1 points). The fixture tests/golden/config1_frame.json was produced by
tests/golden/make_config1_golden.py (oracle + the REFERENCE writer when oracle/_ref is present)."""
import hashlib
import json
import os

import numpy as np

import oracle_lib as O
import ref_writer
from fluidsolvergpu_amd import vtk as sfvtk

HERE = os.path.dirname(os.path.abspath(__file__))
N, K, DT, DIFF, VISC = 32, 10, 0.1, 1e-4, 1e-4


def run_config1():
    z = lambda: np.zeros((N + 2,) * 3, np.float32)
    f = {n: z() for n in ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")}
    c = N // 2
    f["dens0"][c, c, c] = 100.0
    f["v0"][c, c, c] = 5.0
    O.step(N, f, np.float32(DT), np.float32(DIFF), np.float32(VISC), K)
    dens = np.ascontiguousarray(f["dens"][1:-1, 1:-1, 1:-1]).ravel()
    vel = np.stack([f["u"][1:-1, 1:-1, 1:-1], f["v"][1:-1, 1:-1, 1:-1], f["w"][1:-1, 1:-1, 1:-1]], -1).ravel()
    return f, dens, vel


def frame_args(path, ub, dens, vel):
    return (path, ub, [N + 1] * 3, 2, [1, 3], [0, 0], ["density", "velocity"], [dens, vel])


def test_config1_frame_matches_golden(tmp_path):
    gold = json.load(open(os.path.join(HERE, "golden", "config1_frame.json")))
    f, dens, vel = run_config1()
    assert abs(float(dens.sum()) - gold["dens_sum"]) <= 1e-6 * abs(gold["dens_sum"])
    assert float(dens.max()) == gold["dens_max"] and int(np.argmax(dens)) == gold["dens_argmax"]
    for ub, key in ((1, "binary"), (0, "ascii")):
        p = str(tmp_path / f"anim_s0_{key}.vtk")
        sfvtk.write_regular_mesh(*frame_args(p, ub, dens, vel))
        data = open(p, "rb").read()
        assert len(data) == gold[key]["bytes"]
        assert hashlib.sha256(data).hexdigest() == gold[key]["sha256"]
        if ref_writer.available():
            q = str(tmp_path / f"ref_{key}.vtk")
            ref_writer.Writer().write_regular_mesh(*frame_args(q, ub, dens, vel))
            assert open(q, "rb").read() == data


def test_config1_physics_sanity():
    f, dens, vel = run_config1()
    # the source was dt*100 = 10 units of density in one cell; diffusion + advection spread but keep it positive
    assert 9.0 < float(dens.sum()) < 10.5 and dens.min() > -1e-6
    # velocity field after projection is (numerically) less divergent than the injected impulse
    assert np.isfinite(vel).all() and np.abs(vel).max() > 0
