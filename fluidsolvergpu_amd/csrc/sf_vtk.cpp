// sf_vtk.cpp — legacy-VTK frame writer, byte-compatible with the reference's visit_writer.
//
// Independent implementation of the six entry points of /root/reference/visit_writer.cpp
// (write_variables :358-644, write_point_mesh :673-719, write_unstructured_mesh :801-853,
// write_rectilinear_mesh :894-932, write_regular_mesh :968-991, write_curvilinear_mesh :1032-1061).
// Output is byte-identical to the reference for every input the reference handles without
// undefined behaviour; tests/test_vtk_writer.py checks that against the compiled reference
// (oracle/_ref) and against tests/golden/*.vtk.
//
// Byte rules reproduced (reference helpers :110-335):
//   * header: "# vtk DataFile Version 2.0\nWritten using VisIt writer\n" then "ASCII\n" | "BINARY\n";
//   * ASCII numbers: "%20.12e " / "%d ", a newline after every 9th number of a run; a run is closed
//     by a newline only if it is not already at column 0 ("new section") — except the data runs of
//     write_variables and the end of the file, which always add one (so a run whose length is a
//     multiple of 9 is followed by an empty line, and every ASCII file ends with an extra "\n");
//   * binary numbers: 4-byte big-endian, no separators at all between a blob and the next keyword;
//   * ".vtk" is appended unless the file name already contains ".vtk" anywhere;
//   * variables: CELL_DATA first, then POINT_DATA; per centering the first scalar is written as
//     SCALARS + LOOKUP_TABLE, the first vector as VECTORS, further scalars in one
//     "FIELD FieldData k" block and further vectors in another.
// Reference defects NOT reproduced: names are written literally (the reference passes them to
// fprintf as the format, :115,220,267,306); an unopenable file is reported instead of crashing
// (:145 leaves the NULL unchecked); files are opened "wb" rather than "w+".
//
// Unlike the reference (three file-scope globals, :92-94) the state lives in a local object, so
// concurrent writers of different files are safe, and output is buffered: binary blobs are swapped
// and written in bulk, ASCII numbers are formatted with std::to_chars (exactly printf's digits).
#include "../../include/sf_visit_writer.h"

#include <charconv>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

class VtkSink {
public:
    VtkSink(const char* filename, int use_binary) : binary_(use_binary != 0) {
        std::string name(filename);
        if (name.find(".vtk") == std::string::npos) name += ".vtk";
        fp_ = std::fopen(name.c_str(), "wb");
        if (!fp_) {
            std::fprintf(stderr, "sf_vtk: cannot open %s for writing\n", name.c_str());
            failed_ = true;
        }
        buf_.reserve(kFlush + 4096);
    }
    ~VtkSink() { close(); }

    bool ok() const { return !failed_; }

    void text(const char* s) { append(s, std::strlen(s)); }
    void text(const std::string& s) { append(s.data(), s.size()); }

    void header() {
        text("# vtk DataFile Version 2.0\n");
        text("Written using VisIt writer\n");
        text(binary_ ? "BINARY\n" : "ASCII\n");
    }

    // Unconditional line end (ASCII only) — the reference's end_line().
    void end_line() {
        if (!binary_) {
            put('\n');
            col_ = 0;
        }
    }
    // Line end only if a number run is open — the reference's new_section().
    void new_section() {
        if (col_ != 0) end_line();
        col_ = 0;
    }

    void put_int(int v) {
        if (binary_) {
            uint32_t u;
            std::memcpy(&u, &v, 4);
            u = __builtin_bswap32(u);
            append(reinterpret_cast<const char*>(&u), 4);
            return;
        }
        char tmp[32];
        int n = std::snprintf(tmp, sizeof tmp, "%d ", v);
        append(tmp, (size_t)n);
        if ((col_++ % 9) == 8) {
            put('\n');
            col_ = 0;
        }
    }

    void put_floats(const float* v, size_t n) {
        if (binary_) {
            size_t done = 0;
            while (done < n) {
                size_t chunk = n - done;
                if (chunk > kFlush / 4) chunk = kFlush / 4;
                reserve(chunk * 4);
                const size_t at = buf_.size();
                buf_.resize(at + chunk * 4);
                uint32_t* dst = reinterpret_cast<uint32_t*>(buf_.data() + at);
                for (size_t q = 0; q < chunk; ++q) {
                    uint32_t u;
                    std::memcpy(&u, v + done + q, 4);
                    u = __builtin_bswap32(u);
                    std::memcpy(dst + q, &u, 4);
                }
                done += chunk;
                maybe_flush();
            }
            return;
        }
        for (size_t q = 0; q < n; ++q) {
            reserve(48);
            format_e12(v[q]);
            if ((col_++ % 9) == 8) end_line();
            maybe_flush();
        }
    }
    void put_float(float v) { put_floats(&v, 1); }

    int close() {
        if (fp_) {
            end_line();
            flush();
            if (std::fclose(fp_) != 0) failed_ = true;
            fp_ = nullptr;
        }
        return failed_ ? 1 : 0;
    }

private:
    static constexpr size_t kFlush = 1u << 20;

    // "%20.12e " of (double)v, appended to buf_.
    void format_e12(float v) {
        char tmp[48];
        const double d = (double)v;
        if (std::isfinite(d)) {
            char* p = tmp;
            auto r = std::to_chars(p, p + sizeof tmp, d, std::chars_format::scientific, 12);
            const size_t len = (size_t)(r.ptr - p);
            for (size_t pad = len; pad < 20; ++pad) buf_.push_back(' ');
            buf_.insert(buf_.end(), p, p + len);
            buf_.push_back(' ');
        } else {
            int n = std::snprintf(tmp, sizeof tmp, "%20.12e ", d);
            buf_.insert(buf_.end(), tmp, tmp + n);
        }
    }

    void put(char c) {
        buf_.push_back(c);
        maybe_flush();
    }
    void append(const char* s, size_t n) {
        buf_.insert(buf_.end(), s, s + n);
        maybe_flush();
    }
    void reserve(size_t extra) {
        if (buf_.capacity() < buf_.size() + extra) buf_.reserve(buf_.size() + extra + kFlush);
    }
    void maybe_flush() {
        if (buf_.size() >= kFlush) flush();
    }
    void flush() {
        if (fp_ && !buf_.empty()) {
            if (std::fwrite(buf_.data(), 1, buf_.size(), fp_) != buf_.size()) failed_ = true;
        }
        buf_.clear();
    }

    FILE* fp_ = nullptr;
    bool binary_;
    bool failed_ = false;
    int col_ = 0;
    std::vector<char> buf_;
};

// One centering class (cell or point) of the variable section.
void emit_centering(VtkSink& out, const char* keyword, int count, bool want_point, int nvars,
                    const int* vardim, const int* centering, const char* const* names,
                    const float* const* vars) {
    out.new_section();
    out.text(std::string(keyword) + " " + std::to_string(count) + "\n");

    auto mine = [&](int i) { return (centering[i] != 0) == want_point; };
    auto data = [&](int i) {
        out.put_floats(vars[i], (size_t)count * (size_t)vardim[i]);
        out.end_line();
    };

    bool have_scalar = false, have_vector = false;
    int extra_scalars = 0, extra_vectors = 0;
    for (int i = 0; i < nvars; ++i) {
        if (!mine(i)) continue;
        if (vardim[i] == 1) {
            if (!have_scalar) {
                have_scalar = true;
                out.text(std::string("SCALARS ") + names[i] + " float\n");
                out.text("LOOKUP_TABLE default\n");
                data(i);
            } else {
                ++extra_scalars;
            }
        } else if (vardim[i] == 3) {
            if (!have_vector) {
                have_vector = true;
                out.text(std::string("VECTORS ") + names[i] + " float\n");
                data(i);
            } else {
                ++extra_vectors;
            }
        } else {
            std::printf("Only supported variable dimensions are 1 and 3.\n");
            std::printf("Ignoring variable %s.\n", names[i]);
        }
    }
    for (int dim : {1, 3}) {
        const int extra = (dim == 1) ? extra_scalars : extra_vectors;
        if (extra <= 0) continue;
        out.text("FIELD FieldData " + std::to_string(extra) + "\n");
        bool skipped_primary = false;
        for (int i = 0; i < nvars; ++i) {
            if (!mine(i) || vardim[i] != dim) continue;
            if (!skipped_primary) {
                skipped_primary = true;
                continue;
            }
            out.text(std::string(names[i]) + " " + std::to_string(dim) + " " +
                     std::to_string(count) + " float\n");
            data(i);
        }
    }
}

void emit_variables(VtkSink& out, int nvars, const int* vardim, const int* centering,
                    const char* const* names, const float* const* vars, int npts, int ncells) {
    emit_centering(out, "CELL_DATA", ncells, false, nvars, vardim, centering, names, vars);
    emit_centering(out, "POINT_DATA", npts, true, nvars, vardim, centering, names, vars);
}

int points_per_cell(int celltype) {
    switch (celltype) {
        case VISIT_VERTEX: return 1;
        case VISIT_LINE: return 2;
        case VISIT_TRIANGLE: return 3;
        case VISIT_QUAD: return 4;
        case VISIT_TETRA: return 4;
        case VISIT_HEXAHEDRON: return 8;
        case VISIT_WEDGE: return 6;
        case VISIT_PYRAMID: return 5;
        default: return 0;
    }
}

inline int cells_along(int npoints) { return npoints - 1 < 1 ? 1 : npoints - 1; }

}  // namespace

extern "C" {

int sf_vtk_write_point_mesh(const char* filename, int useBinary, int npts, const float* pts,
                            int nvars, const int* vardim, const char* const* varnames,
                            const float* const* vars) {
    VtkSink out(filename, useBinary);
    if (!out.ok()) return 1;
    out.header();
    out.text("DATASET UNSTRUCTURED_GRID\n");
    out.text("POINTS " + std::to_string(npts) + " float\n");
    out.put_floats(pts, (size_t)3 * (size_t)npts);
    out.new_section();
    out.text("CELLS " + std::to_string(npts) + " " + std::to_string(2 * npts) + "\n");
    for (int i = 0; i < npts; ++i) {
        out.put_int(1);
        out.put_int(i);
        out.end_line();
    }
    out.new_section();
    out.text("CELL_TYPES " + std::to_string(npts) + "\n");
    for (int i = 0; i < npts; ++i) {
        out.put_int(VISIT_VERTEX);
        out.end_line();
    }
    std::vector<int> centering((size_t)(nvars > 0 ? nvars : 0), 1);
    emit_variables(out, nvars, vardim, centering.data(), varnames, vars, npts, npts);
    return out.close();
}

int sf_vtk_write_unstructured_mesh(const char* filename, int useBinary, int npts, const float* pts,
                                   int ncells, const int* celltypes, const int* conn, int nvars,
                                   const int* vardim, const int* centering,
                                   const char* const* varnames, const float* const* vars) {
    VtkSink out(filename, useBinary);
    if (!out.ok()) return 1;
    out.header();
    out.text("DATASET UNSTRUCTURED_GRID\n");
    out.text("POINTS " + std::to_string(npts) + " float\n");
    out.put_floats(pts, (size_t)3 * (size_t)npts);
    out.new_section();
    long conn_size = 0;
    for (int i = 0; i < ncells; ++i) conn_size += points_per_cell(celltypes[i]) + 1;
    out.text("CELLS " + std::to_string(ncells) + " " + std::to_string(conn_size) + "\n");
    const int* next = conn;
    for (int i = 0; i < ncells; ++i) {
        const int n = points_per_cell(celltypes[i]);
        out.put_int(n);
        for (int j = 0; j < n; ++j) out.put_int(*next++);
        out.end_line();
    }
    out.new_section();
    out.text("CELL_TYPES " + std::to_string(ncells) + "\n");
    for (int i = 0; i < ncells; ++i) {
        out.put_int(celltypes[i]);
        out.end_line();
    }
    emit_variables(out, nvars, vardim, centering, varnames, vars, npts, ncells);
    return out.close();
}

int sf_vtk_write_rectilinear_mesh(const char* filename, int useBinary, const int* dims,
                                  const float* x, const float* y, const float* z, int nvars,
                                  const int* vardim, const int* centering,
                                  const char* const* varnames, const float* const* vars) {
    const int npts = dims[0] * dims[1] * dims[2];
    const int ncells = cells_along(dims[0]) * cells_along(dims[1]) * cells_along(dims[2]);
    VtkSink out(filename, useBinary);
    if (!out.ok()) return 1;
    out.header();
    out.text("DATASET RECTILINEAR_GRID\n");
    out.text("DIMENSIONS " + std::to_string(dims[0]) + " " + std::to_string(dims[1]) + " " +
             std::to_string(dims[2]) + "\n");
    const float* coords[3] = {x, y, z};
    const char* axis[3] = {"X", "Y", "Z"};
    for (int a = 0; a < 3; ++a) {
        if (a > 0) out.new_section();
        out.text(std::string(axis[a]) + "_COORDINATES " + std::to_string(dims[a]) + " float\n");
        out.put_floats(coords[a], (size_t)dims[a]);
    }
    emit_variables(out, nvars, vardim, centering, varnames, vars, npts, ncells);
    return out.close();
}

int sf_vtk_write_regular_mesh(const char* filename, int useBinary, const int* dims, int nvars,
                              const int* vardim, const int* centering, const char* const* varnames,
                              const float* const* vars) {
    std::vector<float> c[3];
    for (int a = 0; a < 3; ++a) {
        c[a].resize((size_t)(dims[a] > 0 ? dims[a] : 0));
        for (int i = 0; i < dims[a]; ++i) c[a][(size_t)i] = (float)i;
    }
    return sf_vtk_write_rectilinear_mesh(filename, useBinary, dims, c[0].data(), c[1].data(),
                                         c[2].data(), nvars, vardim, centering, varnames, vars);
}

int sf_vtk_write_curvilinear_mesh(const char* filename, int useBinary, const int* dims,
                                  const float* pts, int nvars, const int* vardim,
                                  const int* centering, const char* const* varnames,
                                  const float* const* vars) {
    const int npts = dims[0] * dims[1] * dims[2];
    const int ncells = cells_along(dims[0]) * cells_along(dims[1]) * cells_along(dims[2]);
    VtkSink out(filename, useBinary);
    if (!out.ok()) return 1;
    out.header();
    out.text("DATASET STRUCTURED_GRID\n");
    out.text("DIMENSIONS " + std::to_string(dims[0]) + " " + std::to_string(dims[1]) + " " +
             std::to_string(dims[2]) + "\n");
    out.text("POINTS " + std::to_string(npts) + " float\n");
    out.put_floats(pts, (size_t)3 * (size_t)npts);
    emit_variables(out, nvars, vardim, centering, varnames, vars, npts, ncells);
    return out.close();
}

}  // extern "C"

// ---- the reference's C++-linkage names (visit_writer.h) ---------------------------------------

void write_point_mesh(const char* filename, int useBinary, int npts, float* pts, int nvars,
                      int* vardim, const char* const* varnames, float** vars) {
    sf_vtk_write_point_mesh(filename, useBinary, npts, pts, nvars, vardim, varnames, vars);
}

void write_unstructured_mesh(const char* filename, int useBinary, int npts, float* pts, int ncells,
                             int* celltypes, int* conn, int nvars, int* vardim, int* centering,
                             const char* const* varnames, float** vars) {
    sf_vtk_write_unstructured_mesh(filename, useBinary, npts, pts, ncells, celltypes, conn, nvars,
                                   vardim, centering, varnames, vars);
}

void write_regular_mesh(const char* filename, int useBinary, int* dims, int nvars, int* vardim,
                        int* centering, const char* const* varnames, float** vars) {
    sf_vtk_write_regular_mesh(filename, useBinary, dims, nvars, vardim, centering, varnames, vars);
}

void write_rectilinear_mesh(const char* filename, int useBinary, int* dims, float* x, float* y,
                            float* z, int nvars, int* vardim, int* centering,
                            const char* const* varnames, float** vars) {
    sf_vtk_write_rectilinear_mesh(filename, useBinary, dims, x, y, z, nvars, vardim, centering,
                                  varnames, vars);
}

void write_curvilinear_mesh(const char* filename, int useBinary, int* dims, float* pts, int nvars,
                            int* vardim, int* centering, const char* const* varnames,
                            float** vars) {
    sf_vtk_write_curvilinear_mesh(filename, useBinary, dims, pts, nvars, vardim, centering,
                                  varnames, vars);
}
