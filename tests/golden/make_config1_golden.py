"""Generates tests/golden/config1_frame.json: BASELINE.json configs[0] run through the CPU oracle and written
with the REFERENCE's visit_writer (oracle/_ref; falls back to this repo's writer if the reference is absent —
the two are byte-identical, tests/test_vtk_writer.py). Run: python tests/golden/make_config1_golden.py"""
import hashlib
import json
import os
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import numpy as np  # noqa: E402
import ref_writer  # noqa: E402
import test_config1_plumbing as T  # noqa: E402

f, dens, vel = T.run_config1()
out = {"dens_sum": float(dens.sum()), "dens_max": float(dens.max()), "dens_argmax": int(np.argmax(dens)),
       "writer": "reference (oracle/_ref)" if ref_writer.available() else "sfvtk"}
w = ref_writer.Writer() if ref_writer.available() else T.sfvtk
d = tempfile.mkdtemp()
for ub, key in ((1, "binary"), (0, "ascii")):
    p = os.path.join(d, key + ".vtk")
    w.write_regular_mesh(*T.frame_args(p, ub, dens, vel))
    data = open(p, "rb").read()
    out[key] = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
json.dump(out, open(os.path.join(HERE, "config1_frame.json"), "w"), indent=1, sort_keys=True)
print(out)
