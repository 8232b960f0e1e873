/* include/sf_visit_writer.h — legacy-VTK (v2.0) frame writer of the MI355X stable-fluids solver.
 *
 * Drop-in for the five entry points the reference declares in /root/reference/visit_writer.h
 * (write_point_mesh :94-96, write_unstructured_mesh :176-179, write_regular_mesh :216-218,
 * write_rectilinear_mesh :262-265, write_curvilinear_mesh :303-306). Same names, same C++ linkage
 * (the reference has no extern "C"), same argument meaning, byte-identical files — so a driver written
 * against the reference header links against libsfvtk.so unchanged. This is an independent
 * implementation (buffered output, bulk byte swap, <charconv> formatting); see
 * fluidsolvergpu_amd/csrc/sf_vtk.cpp for the behaviours it reproduces and the defects it does not.
 *
 * The sf_vtk_* functions are the same calls behind a C ABI for FFI users (ctypes, cgo, JNI): they
 * return 0 on success, non-zero if the file could not be opened or written.
 */
#ifndef SF_VISIT_WRITER_H
#define SF_VISIT_WRITER_H

/* VTK cell type ids accepted by write_unstructured_mesh (values as in visit_writer.h:167-174). */
#define VISIT_VERTEX 1
#define VISIT_LINE 3
#define VISIT_TRIANGLE 5
#define VISIT_QUAD 9
#define VISIT_TETRA 10
#define VISIT_HEXAHEDRON 12
#define VISIT_WEDGE 13
#define VISIT_PYRAMID 14

#ifdef __cplusplus

/* filename: ".vtk" is appended unless the name already contains it. useBinary: 0 = ASCII,
 * otherwise big-endian binary. vardim[i] is 1 (scalar) or 3 (vector, xyz interleaved);
 * centering[i] == 0 means cell data, otherwise point data. */

/* npts vertices, pts = x0 y0 z0 x1 y1 z1 ...; every variable is point centred. */
void write_point_mesh(const char* filename, int useBinary, int npts, float* pts, int nvars,
                      int* vardim, const char* const* varnames, float** vars);

/* conn holds, cell after cell, the point indices of each cell (count implied by celltypes[i]). */
void write_unstructured_mesh(const char* filename, int useBinary, int npts, float* pts, int ncells,
                             int* celltypes, int* conn, int nvars, int* vardim, int* centering,
                             const char* const* varnames, float** vars);

/* dims = number of POINTS per axis; coordinates are 0..dims-1. Cells per axis = max(dims-1, 1). */
void write_regular_mesh(const char* filename, int useBinary, int* dims, int nvars, int* vardim,
                        int* centering, const char* const* varnames, float** vars);

void write_rectilinear_mesh(const char* filename, int useBinary, int* dims, float* x, float* y,
                            float* z, int nvars, int* vardim, int* centering,
                            const char* const* varnames, float** vars);

/* pts = 3 * dims[0]*dims[1]*dims[2] floats, i fastest. */
void write_curvilinear_mesh(const char* filename, int useBinary, int* dims, float* pts, int nvars,
                            int* vardim, int* centering, const char* const* varnames, float** vars);

extern "C" {
#endif

int sf_vtk_write_point_mesh(const char* filename, int useBinary, int npts, const float* pts,
                            int nvars, const int* vardim, const char* const* varnames,
                            const float* const* vars);
int sf_vtk_write_unstructured_mesh(const char* filename, int useBinary, int npts, const float* pts,
                                   int ncells, const int* celltypes, const int* conn, int nvars,
                                   const int* vardim, const int* centering,
                                   const char* const* varnames, const float* const* vars);
int sf_vtk_write_regular_mesh(const char* filename, int useBinary, const int* dims, int nvars,
                              const int* vardim, const int* centering, const char* const* varnames,
                              const float* const* vars);
int sf_vtk_write_rectilinear_mesh(const char* filename, int useBinary, const int* dims,
                                  const float* x, const float* y, const float* z, int nvars,
                                  const int* vardim, const int* centering,
                                  const char* const* varnames, const float* const* vars);
int sf_vtk_write_curvilinear_mesh(const char* filename, int useBinary, const int* dims,
                                  const float* pts, int nvars, const int* vardim,
                                  const int* centering, const char* const* varnames,
                                  const float* const* vars);

#ifdef __cplusplus
}
#endif
#endif /* SF_VISIT_WRITER_H */
