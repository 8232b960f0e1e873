// sf_api.hip — the C ABI of include/sfgpu.h over the type-erased solver (sf_base.hpp). No exception crosses it: every
// entry point returns a status code and leaves the message in sf_last_error (the reference's convention is print +
// exit(1), FluidGPU.cuh:34-41; the driver does that with the code it gets back).
#include "sf_base.hpp"

using namespace sfi;

namespace {
thread_local std::string g_create_error;
}


struct sf_ctx {
    std::unique_ptr<SolverBase> impl;
    std::string err;
    std::string snap_err;  // sf_snapshot_read may run on another thread: it gets its own message buffer
};

namespace {

template <class F>
int guarded(sf_ctx* ctx, F&& body) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    try {
        body(*ctx->impl);
        return SF_OK;
    } catch (const Failure& f) {
        ctx->err = f.msg;
        return f.code;
    } catch (const std::exception& e) {
        ctx->err = e.what();
        return SF_ERR_INVALID;
    }
}

}  // namespace

extern "C" {

const char* sf_version(void) { return "sfgpu 0.1 gfx950 hip"; }

const char* sf_status_string(int status) {
    switch (status) {
        case SF_OK: return "SF_OK";
        case SF_ERR_INVALID: return "SF_ERR_INVALID";
        case SF_ERR_HIP: return "SF_ERR_HIP";
        case SF_ERR_RCCL: return "SF_ERR_RCCL";
        case SF_ERR_HALO_EXCEEDED: return "SF_ERR_HALO_EXCEEDED";
        case SF_ERR_NO_DEVICE: return "SF_ERR_NO_DEVICE";
        default: return "SF_ERR_UNKNOWN";
    }
}

int sf_nccl_unique_id(void* out) {
    if (!out) return SF_ERR_INVALID;
    ncclUniqueId id;
    if (ncclGetUniqueId(&id) != ncclSuccess) return SF_ERR_RCCL;
    std::memset(out, 0, SF_NCCL_ID_BYTES);
    std::memcpy(out, &id, sizeof id);
    return SF_OK;
}

int sf_create(sf_ctx** out, const sf_params* p) {
    if (!out) return SF_ERR_INVALID;
    *out = nullptr;
    if (!p) {
        g_create_error = "null params";
        return SF_ERR_INVALID;
    }
    try {
        std::unique_ptr<sf_ctx> ctx(new sf_ctx);
        if (p->dtype == SF_F32)
            ctx->impl.reset(make_solver_f32(*p));
        else if (p->dtype == SF_F64)
            ctx->impl.reset(make_solver_f64(*p));
        else
            throw Failure{SF_ERR_INVALID, "dtype must be SF_F32 or SF_F64"};
        *out = ctx.release();
        return SF_OK;
    } catch (const Failure& f) {
        g_create_error = f.msg;
        return f.code;
    } catch (const std::exception& e) {
        g_create_error = e.what();
        return SF_ERR_INVALID;
    }
}

void sf_destroy(sf_ctx* ctx) { delete ctx; }

int sf_upload(sf_ctx* ctx, int field, const void* host) {
    return guarded(ctx, [&](SolverBase& s) { s.upload(field, host); });
}
int sf_download(sf_ctx* ctx, int field, void* host) {
    return guarded(ctx, [&](SolverBase& s) { s.download(field, host); });
}
int sf_download_planes(sf_ctx* ctx, int field, int k_begin, int k_end, void* host) {
    return guarded(ctx, [&](SolverBase& s) { s.download_planes(field, k_begin, k_end, host); });
}
int sf_upload_planes(sf_ctx* ctx, int field, int k_begin, int k_end, const void* host) {
    return guarded(ctx, [&](SolverBase& s) { s.upload_planes(field, k_begin, k_end, host); });
}
int sf_stored_planes(const sf_ctx* ctx, int* k_begin, int* k_end) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    ctx->impl->stored_planes(k_begin, k_end);
    return SF_OK;
}
int sf_owned_planes(const sf_ctx* ctx, int* k_begin, int* k_end) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    ctx->impl->owned_planes(k_begin, k_end);
    return SF_OK;
}
int sf_fill(sf_ctx* ctx, int field, double value) {
    return guarded(ctx, [&](SolverBase& s) { s.fill(field, value); });
}
int sf_copy_field(sf_ctx* ctx, int dst, int src) {
    return guarded(ctx, [&](SolverBase& s) { s.copy_field(dst, src); });
}
int sf_bind_sources(sf_ctx* ctx, int su, int sv, int sw, int sd) {
    return guarded(ctx, [&](SolverBase& s) { s.bind_sources(su, sv, sw, sd); });
}
int vel_step(sf_ctx* ctx) {
    return guarded(ctx, [&](SolverBase& s) { s.vel_step(); });
}
int dens_step(sf_ctx* ctx) {
    return guarded(ctx, [&](SolverBase& s) { s.dens_step(); });
}
int sf_add_source(sf_ctx* ctx, int x, int src) {
    return guarded(ctx, [&](SolverBase& s) { s.add_source(x, src); });
}
int sf_set_bnd(sf_ctx* ctx, int b, int x) {
    return guarded(ctx, [&](SolverBase& s) { s.set_bnd(b, x); });
}
int sf_lin_solve(sf_ctx* ctx, int b, int x, int x0, double a, double c, int iters) {
    return guarded(ctx, [&](SolverBase& s) { s.lin_solve(b, x, x0, a, c, iters); });
}
int sf_diffuse(sf_ctx* ctx, int b, int x, int x0, double diff) {
    return guarded(ctx, [&](SolverBase& s) { s.diffuse(b, x, x0, diff); });
}
int sf_advect(sf_ctx* ctx, int b, int d, int d0, int u, int v, int w) {
    return guarded(ctx, [&](SolverBase& s) { s.advect(b, d, d0, u, v, w); });
}
int sf_project(sf_ctx* ctx, int u, int v, int w, int p, int div) {
    return guarded(ctx, [&](SolverBase& s) { s.project(u, v, w, p, div); });
}
int sf_snapshot(sf_ctx* ctx, const int* fields, int nfields) {
    return guarded(ctx, [&](SolverBase& s) { s.snapshot(fields, nfields); });
}
int sf_snapshot_read(sf_ctx* ctx, int index, void* host) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    try {
        ctx->impl->snapshot_read(index, host);
        return SF_OK;
    } catch (const Failure& f) {
        ctx->snap_err = f.msg;
        return f.code;
    } catch (const std::exception& e) {  // e.g. std::bad_alloc: nothing may cross the C ABI
        ctx->snap_err = e.what();
        return SF_ERR_INVALID;
    }
}
int sf_snapshot_read_planes(sf_ctx* ctx, int index, int k_begin, int k_end, void* host) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    try {
        ctx->impl->snapshot_read_planes(index, k_begin, k_end, host);
        return SF_OK;
    } catch (const Failure& f) {
        ctx->snap_err = f.msg;
        return f.code;
    } catch (const std::exception& e) {
        ctx->snap_err = e.what();
        return SF_ERR_INVALID;
    }
}
int sf_tracers_set(sf_ctx* ctx, int n, const void* xyz) {
    return guarded(ctx, [&](SolverBase& s) { s.tracers_set(n, xyz); });
}
int sf_tracers_advect(sf_ctx* ctx) {
    return guarded(ctx, [&](SolverBase& s) { s.tracers_advect(); });
}
int sf_tracers_get(sf_ctx* ctx, void* xyz, void* dens_sample, void* speed_sample) {
    return guarded(ctx, [&](SolverBase& s) { s.tracers_get(xyz, dens_sample, speed_sample); });
}
int sf_set_iters(sf_ctx* ctx, int iters) {
    return guarded(ctx, [&](SolverBase& s) { s.set_iters(iters); });
}
int sf_set_coefficients(sf_ctx* ctx, double dt, double diff, double visc) {
    return guarded(ctx, [&](SolverBase& s) { s.set_coefficients(dt, diff, visc); });
}
int sf_sync(sf_ctx* ctx) {
    return guarded(ctx, [&](SolverBase& s) { s.sync(); });
}
const char* sf_last_error(const sf_ctx* ctx) {
    if (!ctx) return g_create_error.c_str();
    return ctx->err.c_str();
}
int sf_timer_start(sf_ctx* ctx) {
    return guarded(ctx, [&](SolverBase& s) { s.timer_start(); });
}
int sf_timer_stop(sf_ctx* ctx, float* ms) {
    return guarded(ctx, [&](SolverBase& s) {
        const float t = s.timer_stop();
        if (ms) *ms = t;
    });
}
int sf_measure_copy_bandwidth(sf_ctx* ctx, size_t bytes, int reps, double* gbps) {
    return guarded(ctx, [&](SolverBase& s) {
        const double r = s.copy_bandwidth(bytes, reps);
        if (gbps) *gbps = r;
    });
}
int sf_lin_solve_launches(const sf_ctx* ctx, int iters) {
    if (!ctx || !ctx->impl || iters < 0) return -1;
    return ctx->impl->lin_solve_launches(iters);
}
int sf_layout_info(const sf_ctx* ctx, int* row_pitch, int* planes_per_slab, size_t* bytes_per_field) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    ctx->impl->layout_info(row_pitch, planes_per_slab, bytes_per_field);
    return SF_OK;
}
int sf_schedule_info(const sf_ctx* ctx, int* trapezoid_pairs, int* measured) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    ctx->impl->schedule_info(trapezoid_pairs, measured);
    return SF_OK;
}
int sf_transport_info(const sf_ctx* ctx, int* transport, long* rccl_groups) {
    if (!ctx || !ctx->impl) return SF_ERR_INVALID;
    ctx->impl->transport_info(transport, rccl_groups);
    return SF_OK;
}

}  // extern "C"
