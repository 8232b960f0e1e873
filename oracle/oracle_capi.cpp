// oracle/oracle_capi.cpp — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see stable_fluids_oracle.hpp).
// Plain-C entry points over the templated CPU oracle so tests/ and bench.py's cpu_baseline leg can
// call it through ctypes. Suffix _f32 / _f64 selects T. Arrays are dense (N+2)^3, x fastest.
#include "stable_fluids_oracle.hpp"

#include <vector>

using namespace sf_oracle;

#define SF_ORACLE_API(T, SFX)                                                                       \
    extern "C" void oracle_add_source_##SFX(int N, T* x, const T* s, T dt) {                        \
        add_source<T>(N, x, s, dt);                                                                 \
    }                                                                                               \
    extern "C" void oracle_set_bnd_##SFX(int N, int b, T* x) { set_bnd<T>(N, b, x); }               \
    extern "C" void oracle_lin_solve_##SFX(int N, int b, T* x, const T* x0, T a, T c, int K) {      \
        std::vector<T> scratch(Grid<T>(N).size());                                                  \
        lin_solve<T>(N, b, x, x0, a, c, K, scratch.data());                                         \
    }                                                                                               \
    extern "C" void oracle_diffuse_##SFX(int N, int b, T* x, const T* x0, T diff, T dt, int K) {    \
        std::vector<T> scratch(Grid<T>(N).size());                                                  \
        diffuse<T>(N, b, x, x0, diff, dt, K, scratch.data());                                       \
    }                                                                                               \
    extern "C" void oracle_advect_##SFX(int N, int b, T* d, const T* d0, const T* u, const T* v,    \
                                        const T* w, T dt) {                                         \
        advect<T>(N, b, d, d0, u, v, w, dt);                                                        \
    }                                                                                               \
    extern "C" void oracle_project_##SFX(int N, T* u, T* v, T* w, T* p, T* div, int K) {            \
        std::vector<T> scratch(Grid<T>(N).size());                                                  \
        project<T>(N, u, v, w, p, div, K, scratch.data());                                          \
    }                                                                                               \
    extern "C" void oracle_project_div_##SFX(int N, const T* u, const T* v, const T* w, T* p,       \
                                             T* div) {                                              \
        project_div<T>(N, u, v, w, p, div);                                                         \
    }                                                                                               \
    extern "C" void oracle_project_sub_##SFX(int N, T* u, T* v, T* w, const T* p) {                 \
        project_sub<T>(N, u, v, w, p);                                                              \
    }                                                                                               \
    extern "C" void oracle_dens_step_##SFX(int N, T* x, T* x0, T* u, T* v, T* w, T diff, T dt,      \
                                           int K) {                                                 \
        dens_step<T>(N, x, x0, u, v, w, diff, dt, K);                                               \
    }                                                                                               \
    extern "C" void oracle_tracers_advect_##SFX(int N, int n, T* pos, const T* u, const T* v, const T* w, T dt) { \
        tracers_advect<T>(N, n, pos, u, v, w, dt);                                                  \
    }                                                                                               \
    extern "C" void oracle_tracers_sample_##SFX(int N, int n, const T* pos, const T* dens, const T* u,  \
                                                const T* v, const T* w, T* dout, T* sout) {         \
        tracers_sample<T>(N, n, pos, dens, u, v, w, dout, sout);                                    \
    }                                                                                               \
    extern "C" void oracle_vel_step_##SFX(int N, T* u, T* v, T* w, T* u0, T* v0, T* w0, T visc,     \
                                          T dt, int K) {                                            \
        vel_step<T>(N, u, v, w, u0, v0, w0, visc, dt, K);                                           \
    }

SF_ORACLE_API(float, f32)
SF_ORACLE_API(double, f64)
