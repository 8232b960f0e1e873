"""fluidsolvergpu_amd — MI355X-native 3-D stable-fluids solver (host-side Python mirror).

The product is the C-ABI library built from fluidsolvergpu_amd/csrc (include/sfgpu.h): hand-written
gfx950 HIP kernels behind `vel_step` / `dens_step`, plus the visit_writer-compatible VTK frame
writer (include/sf_visit_writer.h). This package only marshals numpy arrays into those C entry
points; it contains no numerics and no CPU fallback.
"""
__all__ = ["vtk", "solver"]
