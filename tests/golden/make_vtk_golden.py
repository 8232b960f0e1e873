"""Generates tests/golden/vtk/*.vtk by running the REFERENCE's own visit_writer (compiled from
/root/reference/visit_writer.cpp into oracle/_ref by `make -C oracle ref`) on the deterministic cases
of tests/vtk_cases.py. Run in the build container (the reference is not on the GPU box):

    make -C oracle ref && python tests/golden/make_vtk_golden.py

The outputs are data (files written by the reference binary), committed as fixtures."""
import hashlib
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import ref_writer  # noqa: E402
import vtk_cases  # noqa: E402


def main():
    out_dir = os.path.join(HERE, "vtk")
    os.makedirs(out_dir, exist_ok=True)
    w = ref_writer.Writer()
    manifest = {}
    for case in vtk_cases.cases():
        path = vtk_cases.run_case(w, case, out_dir)
        data = open(path, "rb").read()
        manifest[os.path.basename(path)] = {"bytes": len(data), "sha256": hashlib.sha256(data).hexdigest()}
        print(f"{os.path.basename(path):24s} {len(data):7d} {manifest[os.path.basename(path)]['sha256']}")
    json.dump(manifest, open(os.path.join(out_dir, "MANIFEST.json"), "w"), indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
