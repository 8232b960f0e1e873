// sf_base.hpp — what the translation units of libsfgpu.so share: the error type and macros, the type-erased solver
// interface behind the C ABI (include/sfgpu.h) and the per-precision factories (sf_solver_f32.hip / sf_solver_f64.hip
// instantiate sfi::Solver<float> / <double> separately, so the two halves of the library compile in parallel).
#pragma once
#include "../../include/sfgpu.h"
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <algorithm>
#include <chrono>
#include <climits>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace sfi {


struct Failure {
    int code;
    std::string msg;
};

#define SF_HIP(expr)                                                                               \
    do {                                                                                           \
        hipError_t _e = (expr);                                                                    \
        if (_e != hipSuccess)                                                                      \
            throw Failure{SF_ERR_HIP, std::string("Error ") + hipGetErrorString(_e) + " at line " + \
                                          std::to_string(__LINE__) + " in file " + __FILE__ +      \
                                          " (" #expr ")"};                                         \
    } while (0)

#define SF_NCCL(expr)                                                                              \
    do {                                                                                           \
        ncclResult_t _r = (expr);                                                                  \
        if (_r != ncclSuccess)                                                                     \
            throw Failure{SF_ERR_RCCL, std::string("RCCL error ") + ncclGetErrorString(_r) +       \
                                           " at line " + std::to_string(__LINE__) + " (" #expr ")"}; \
    } while (0)

#define SF_REQUIRE(cond, text)                                          \
    do {                                                                \
        if (!(cond)) throw Failure{SF_ERR_INVALID, std::string(text)};  \
    } while (0)

inline int env_int(const char* name, int dflt) {
    const char* s = std::getenv(name);
    return (s && *s) ? std::atoi(s) : dflt;
}

inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
inline long ceil_div(long a, long b) { return (a + b - 1) / b; }

class SolverBase {
public:
    virtual ~SolverBase() {}
    virtual void upload(int field, const void* host) = 0;
    virtual void download(int field, void* host) = 0;
    virtual void download_planes(int field, int kb, int ke, void* host) = 0;
    virtual void upload_planes(int field, int kb, int ke, const void* host) = 0;
    virtual void owned_planes(int* kb, int* ke) const = 0;
    virtual void stored_planes(int* kb, int* ke) const = 0;
    virtual void fill(int field, double value) = 0;
    virtual void copy_field(int dst, int src) = 0;
    virtual void bind_sources(int su, int sv, int sw, int sd) = 0;
    virtual void vel_step() = 0;
    virtual void dens_step() = 0;
    virtual void add_source(int x, int s) = 0;
    virtual void set_bnd(int b, int x) = 0;
    virtual void lin_solve(int b, int x, int x0, double a, double c, int iters) = 0;
    virtual void diffuse(int b, int x, int x0, double diff) = 0;
    virtual void advect(int b, int d, int d0, int u, int v, int w) = 0;
    virtual void project(int u, int v, int w, int p, int div) = 0;
    virtual void set_iters(int iters) = 0;
    virtual void set_coefficients(double dt, double diff, double visc) = 0;
    virtual void sync() = 0;
    virtual void timer_start() = 0;
    virtual float timer_stop() = 0;
    virtual double copy_bandwidth(size_t bytes, int reps) = 0;
    virtual void layout_info(int* pitch, int* planes, size_t* bytes) const = 0;
    virtual void schedule_info(int* trap, int* measured) const = 0;
    virtual void transport_info(int* transport, long* groups) const = 0;
    virtual int lin_solve_launches(int iters) const = 0;
    virtual void snapshot(const int* fields, int nfields) = 0;
    virtual void snapshot_read(int index, void* host) = 0;
    virtual void snapshot_read_planes(int index, int kb, int ke, void* host) = 0;
    virtual void tracers_set(int n, const void* xyz) = 0;
    virtual void tracers_advect() = 0;
    virtual void tracers_get(void* xyz, void* dens, void* speed) = 0;
};

SolverBase* make_solver_f32(const sf_params& p);
SolverBase* make_solver_f64(const sf_params& p);

}  // namespace sfi
