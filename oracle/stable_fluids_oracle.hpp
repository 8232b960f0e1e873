// oracle/stable_fluids_oracle.hpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE.
//
// Serial CPU implementation of docs/SPEC.md (the frozen 3-D stable-fluids step).
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.
//
// PARITY UNPINNED for the stencil numerics: the reference (robbergen/FluidSolverGPU) is an SPH
// particle solver and holds no dens_step / vel_step / lin_solve / advect / project / set_bnd
// (the only hits for those names under /root/reference are the XML word "Project"
// in Bleh.vcxproj:2 and FluidSolver.sln:6; solver.cu:171-216 and solver-unidyn.cu:313-573 are
// particle loops). There is nothing to transcribe and no golden vector to pin against, so each
// function below cites docs/SPEC.md instead of a reference file:line. What the reference does pin
// (the .vtk byte format, visit_writer.cpp) is checked against the compiled reference in oracle/_ref.
//
// Build with: g++ -O2 -ffp-contract=off  (no -ffast-math) so every expression rounds exactly as
// written; the HIP kernels are built with -ffp-contract=off too and must match bit for bit.
#pragma once
#include <cmath>
#include <cstddef>
#include <cstring>
#include <utility>
#include <vector>

namespace sf_oracle {

template <class T>
struct Grid {
    int N;
    int S;
    explicit Grid(int n) : N(n), S(n + 2) {}
    inline size_t IX(int i, int j, int k) const {
        return (size_t)i + (size_t)S * ((size_t)j + (size_t)S * (size_t)k);
    }
    inline size_t size() const { return (size_t)S * S * S; }
};

// SPEC §3 add_source: all S^3 entries.
template <class T>
void add_source(int N, T* x, const T* s, T dt) {
    Grid<T> g(N);
    const size_t n = g.size();
    for (size_t q = 0; q < n; ++q) x[q] = x[q] + dt * s[q];
}

// SPEC §3 set_bnd: faces, then 12 edges, then 8 corners.
template <class T>
void set_bnd(int N, int b, T* x) {
    Grid<T> g(N);
    const T sx = (b == 1) ? T(-1) : T(1);
    const T sy = (b == 2) ? T(-1) : T(1);
    const T sz = (b == 3) ? T(-1) : T(1);
    const T half = T(0.5);
    const T third = (T)(1.0 / 3.0);
    const int E = N + 1;
    for (int k = 1; k <= N; ++k)
        for (int j = 1; j <= N; ++j) {
            x[g.IX(0, j, k)] = sx * x[g.IX(1, j, k)];
            x[g.IX(E, j, k)] = sx * x[g.IX(N, j, k)];
        }
    for (int k = 1; k <= N; ++k)
        for (int i = 1; i <= N; ++i) {
            x[g.IX(i, 0, k)] = sy * x[g.IX(i, 1, k)];
            x[g.IX(i, E, k)] = sy * x[g.IX(i, N, k)];
        }
    for (int j = 1; j <= N; ++j)
        for (int i = 1; i <= N; ++i) {
            x[g.IX(i, j, 0)] = sz * x[g.IX(i, j, 1)];
            x[g.IX(i, j, E)] = sz * x[g.IX(i, j, N)];
        }
    const int lo_hi[2] = {0, E};
    const int inner[2] = {1, N};
    // x-directed edges: (i, J, K)
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) {
            const int J = lo_hi[a], Jn = inner[a], K = lo_hi[c], Kn = inner[c];
            for (int i = 1; i <= N; ++i)
                x[g.IX(i, J, K)] = half * (x[g.IX(i, Jn, K)] + x[g.IX(i, J, Kn)]);
        }
    // y-directed edges: (I, j, K)
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) {
            const int I = lo_hi[a], In = inner[a], K = lo_hi[c], Kn = inner[c];
            for (int j = 1; j <= N; ++j)
                x[g.IX(I, j, K)] = half * (x[g.IX(In, j, K)] + x[g.IX(I, j, Kn)]);
        }
    // z-directed edges: (I, J, k)
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c) {
            const int I = lo_hi[a], In = inner[a], J = lo_hi[c], Jn = inner[c];
            for (int k = 1; k <= N; ++k)
                x[g.IX(I, J, k)] = half * (x[g.IX(In, J, k)] + x[g.IX(I, Jn, k)]);
        }
    // corners
    for (int a = 0; a < 2; ++a)
        for (int c = 0; c < 2; ++c)
            for (int e = 0; e < 2; ++e) {
                const int I = lo_hi[a], In = inner[a];
                const int J = lo_hi[c], Jn = inner[c];
                const int K = lo_hi[e], Kn = inner[e];
                x[g.IX(I, J, K)] =
                    third * ((x[g.IX(In, J, K)] + x[g.IX(I, Jn, K)]) + x[g.IX(I, J, Kn)]);
            }
}

// One Jacobi sweep (SPEC §3 lin_solve body): xn <- f(x, x0), interior only.
template <class T>
void jacobi_sweep(int N, T* xn, const T* x, const T* x0, T a, T inv) {
    Grid<T> g(N);
    const size_t S = (size_t)g.S, S2 = S * S;
    for (int k = 1; k <= N; ++k)
        for (int j = 1; j <= N; ++j) {
            const size_t row = g.IX(0, j, k);
            for (int i = 1; i <= N; ++i) {
                const size_t q = row + i;
                xn[q] = (x0[q] + a * (((x[q - 1] + x[q + 1]) + (x[q - S] + x[q + S])) +
                                      (x[q - S2] + x[q + S2]))) * inv;
            }
        }
}

// SPEC §3 lin_solve. Result ends in x. `scratch` must hold S^3 entries.
template <class T>
void lin_solve(int N, int b, T* x, const T* x0, T a, T c, int K, T* scratch) {
    Grid<T> g(N);
    const T inv = T(1) / c;
    T* cur = x;
    T* nxt = scratch;
    for (int it = 0; it < K; ++it) {
        jacobi_sweep(N, nxt, cur, x0, a, inv);
        set_bnd(N, b, nxt);
        std::swap(cur, nxt);
    }
    if (cur != x) std::memcpy(x, cur, g.size() * sizeof(T));
}

template <class T>
void diffuse(int N, int b, T* x, const T* x0, T diff, T dt, int K, T* scratch) {
    const T Nf = (T)N;
    const T a = ((dt * diff) * Nf) * Nf;
    // sweep + set_bnd together write all S^3 entries of the target, so scratch needs no initialisation.
    lin_solve(N, b, x, x0, a, T(1) + T(6) * a, K, scratch);
}

// SPEC §3 advect.
template <class T>
void advect(int N, int b, T* d, const T* d0, const T* u, const T* v, const T* w, T dt) {
    Grid<T> g(N);
    const T Nf = (T)N;
    const T dt0 = dt * Nf;
    const T lo = T(0.5), hi = Nf + T(0.5);
    for (int k = 1; k <= N; ++k)
        for (int j = 1; j <= N; ++j)
            for (int i = 1; i <= N; ++i) {
                const size_t q = g.IX(i, j, k);
                T x = (T)i - dt0 * u[q];
                T y = (T)j - dt0 * v[q];
                T z = (T)k - dt0 * w[q];
                if (x < lo) x = lo;
                if (x > hi) x = hi;
                if (y < lo) y = lo;
                if (y > hi) y = hi;
                if (z < lo) z = lo;
                if (z > hi) z = hi;
                int i0 = (x == x) ? (int)x : 0;
                int j0 = (y == y) ? (int)y : 0;
                int k0 = (z == z) ? (int)z : 0;
                i0 = i0 < 0 ? 0 : (i0 > N ? N : i0);
                j0 = j0 < 0 ? 0 : (j0 > N ? N : j0);
                k0 = k0 < 0 ? 0 : (k0 > N ? N : k0);
                const int i1 = i0 + 1, j1 = j0 + 1, k1 = k0 + 1;
                const T s1 = x - (T)i0, s0 = T(1) - s1;
                const T t1 = y - (T)j0, t0 = T(1) - t1;
                const T r1 = z - (T)k0, r0 = T(1) - r1;
                d[q] = s0 * (t0 * (r0 * d0[g.IX(i0, j0, k0)] + r1 * d0[g.IX(i0, j0, k1)]) +
                             t1 * (r0 * d0[g.IX(i0, j1, k0)] + r1 * d0[g.IX(i0, j1, k1)])) +
                       s1 * (t0 * (r0 * d0[g.IX(i1, j0, k0)] + r1 * d0[g.IX(i1, j0, k1)]) +
                             t1 * (r0 * d0[g.IX(i1, j1, k0)] + r1 * d0[g.IX(i1, j1, k1)]));
            }
    set_bnd(N, b, d);
}

// SPEC §3 project, split so tests can check the halves.
template <class T>
void project_div(int N, const T* u, const T* v, const T* w, T* p, T* div) {
    Grid<T> g(N);
    const size_t S = (size_t)g.S, S2 = S * S;
    const T Nf = (T)N;
    const T h = T(1) / Nf;
    const T c_div = T(-0.5) * h;
    const size_t n = g.size();
    for (size_t q = 0; q < n; ++q) p[q] = T(0);
    for (int k = 1; k <= N; ++k)
        for (int j = 1; j <= N; ++j)
            for (int i = 1; i <= N; ++i) {
                const size_t q = g.IX(i, j, k);
                div[q] = c_div * (((u[q + 1] - u[q - 1]) + (v[q + S] - v[q - S])) +
                                  (w[q + S2] - w[q - S2]));
            }
    set_bnd(N, 0, div);
    set_bnd(N, 0, p);
}

template <class T>
void project_sub(int N, T* u, T* v, T* w, const T* p) {
    Grid<T> g(N);
    const size_t S = (size_t)g.S, S2 = S * S;
    const T c_grad = T(0.5) * (T)N;
    for (int k = 1; k <= N; ++k)
        for (int j = 1; j <= N; ++j)
            for (int i = 1; i <= N; ++i) {
                const size_t q = g.IX(i, j, k);
                u[q] = u[q] - c_grad * (p[q + 1] - p[q - 1]);
                v[q] = v[q] - c_grad * (p[q + S] - p[q - S]);
                w[q] = w[q] - c_grad * (p[q + S2] - p[q - S2]);
            }
    set_bnd(N, 1, u);
    set_bnd(N, 2, v);
    set_bnd(N, 3, w);
}

template <class T>
void project(int N, T* u, T* v, T* w, T* p, T* div, int K, T* scratch) {
    project_div(N, u, v, w, p, div);
    lin_solve(N, 0, p, div, T(1), T(6), K, scratch);
    project_sub(N, u, v, w, p);
}

// "swap" exchanges which array a name refers to (SPEC §3). Each step swaps twice, so on return the
// caller's arrays hold: x / u,v,w = new state; x0 / u0,v0,w0 = scratch (diffused field / p, div, pre-advect w).

// SPEC §3 dens_step. Same name / argument meaning as the C-ABI entry point (include/sfgpu.h).
template <class T>
void dens_step(int N, T* x, T* x0, T* u, T* v, T* w, T diff, T dt, int K) {
    std::vector<T> scratch(Grid<T>(N).size());
    add_source(N, x, x0, dt);
    std::swap(x0, x);
    diffuse(N, 0, x, x0, diff, dt, K, scratch.data());
    std::swap(x0, x);
    advect(N, 0, x, x0, u, v, w, dt);
}

// SPEC §3 vel_step.
template <class T>
void vel_step(int N, T* u, T* v, T* w, T* u0, T* v0, T* w0, T visc, T dt, int K) {
    std::vector<T> scratch(Grid<T>(N).size());
    add_source(N, u, u0, dt);
    add_source(N, v, v0, dt);
    add_source(N, w, w0, dt);
    std::swap(u0, u);
    std::swap(v0, v);
    std::swap(w0, w);
    diffuse(N, 1, u, u0, visc, dt, K, scratch.data());
    diffuse(N, 2, v, v0, visc, dt, K, scratch.data());
    diffuse(N, 3, w, w0, visc, dt, K, scratch.data());
    project(N, u, v, w, u0, v0, K, scratch.data());
    std::swap(u0, u);
    std::swap(v0, v);
    std::swap(w0, w);
    advect(N, 1, u, u0, u0, v0, w0, dt);
    advect(N, 2, v, v0, u0, v0, w0, dt);
    advect(N, 3, w, w0, u0, v0, w0, dt);
    project(N, u, v, w, u0, v0, K, scratch.data());
}

// SPEC §6 tracers. pos = x0 y0 z0 x1 y1 z1 ... in grid-index coordinates.
template <class T>
struct TriSample {
    size_t q000;
    size_t dj, dk;
    T s0, s1, t0, t1, r0, r1;
    TriSample(int N, T x, T y, T z) {
        Grid<T> g(N);
        const T Nf = (T)N, lo = T(0.5), hi = Nf + T(0.5);
        if (x < lo) x = lo;
        if (x > hi) x = hi;
        if (y < lo) y = lo;
        if (y > hi) y = hi;
        if (z < lo) z = lo;
        if (z > hi) z = hi;
        int i0 = (x == x) ? (int)x : 0, j0 = (y == y) ? (int)y : 0, k0 = (z == z) ? (int)z : 0;
        i0 = i0 < 0 ? 0 : (i0 > N ? N : i0);
        j0 = j0 < 0 ? 0 : (j0 > N ? N : j0);
        k0 = k0 < 0 ? 0 : (k0 > N ? N : k0);
        s1 = x - (T)i0;
        s0 = T(1) - s1;
        t1 = y - (T)j0;
        t0 = T(1) - t1;
        r1 = z - (T)k0;
        r0 = T(1) - r1;
        q000 = g.IX(i0, j0, k0);
        dj = (size_t)g.S;
        dk = (size_t)g.S * g.S;
    }
    T operator()(const T* d0) const {
        const size_t q = q000;
        return s0 * (t0 * (r0 * d0[q] + r1 * d0[q + dk]) + t1 * (r0 * d0[q + dj] + r1 * d0[q + dj + dk])) +
               s1 * (t0 * (r0 * d0[q + 1] + r1 * d0[q + 1 + dk]) +
                     t1 * (r0 * d0[q + 1 + dj] + r1 * d0[q + 1 + dj + dk]));
    }
};

template <class T>
inline T clamp_coord(int N, T x) {
    const T lo = T(0.5), hi = (T)N + T(0.5);
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

template <class T>
void tracers_sample(int N, int n, const T* pos, const T* dens, const T* u, const T* v, const T* w, T* dens_out,
                    T* speed_out) {
    for (int t = 0; t < n; ++t) {
        const TriSample<T> S(N, pos[3 * t], pos[3 * t + 1], pos[3 * t + 2]);
        const T a = S(u), b = S(v), c = S(w);
        dens_out[t] = S(dens);
        speed_out[t] = std::sqrt((a * a + b * b) + c * c);
    }
}

template <class T>
void tracers_advect(int N, int n, T* pos, const T* u, const T* v, const T* w, T dt) {
    const T dt0 = dt * (T)N;
    for (int t = 0; t < n; ++t) {
        const T x = clamp_coord(N, pos[3 * t]), y = clamp_coord(N, pos[3 * t + 1]), z = clamp_coord(N, pos[3 * t + 2]);
        const TriSample<T> S(N, x, y, z);
        const T vx = S(u), vy = S(v), vz = S(w);
        pos[3 * t] = clamp_coord(N, x + dt0 * vx);
        pos[3 * t + 1] = clamp_coord(N, y + dt0 * vy);
        pos[3 * t + 2] = clamp_coord(N, z + dt0 * vz);
    }
}

}  // namespace sf_oracle
