/* include/sfgpu.h — C ABI of libsfgpu.so: the MI355X (gfx950) stable-fluids hot path.
 *
 * What this replaces in the reference (robbergen/FluidSolverGPU): the reference has NO plugin /
 * FFI boundary — its host loop launches kernels with <<<>>> on raw device pointers inside one
 * program (kernel prototypes FluidGPU.cuh:417-419, FluidGPU-unidyn.cuh:537-544; host loop
 * solver.cu:171-216, solver-unidyn.cu:313-573) and owns every buffer in main() (solver.cu:74-106).
 * BASELINE.json's north_star asks for "dens_step / vel_step entry points ... host code in C++
 * calling HIP through a thin C-ABI"; this header is that boundary. Per entry point
 * the comment names the reference site whose role it takes.
 *
 * Conventions
 *   - Every function returns an sf_status (0 = SF_OK). No exception crosses the boundary.
 *     sf_last_error() gives the message; a driver prints it and exits, mirroring the reference's
 *     CUDA_CHECK_RETURN (FluidGPU.cuh:34-41).
 *   - Host arrays are dense (N+2)^3, x fastest: IX(i,j,k) = i + (N+2)*(j + (N+2)*k), element type
 *     float (SF_F32) or double (SF_F64). The caller owns host memory; the context owns all device
 *     memory, streams, events and the RCCL communicator. Device layout (pitched rows, ghost planes)
 *     is internal.
 *   - Step functions are asynchronous on the context's streams; sf_sync() waits and reports
 *     deferred errors. One context per host thread; contexts are independent.
 *   - Numerics are those of docs/SPEC.md, bit-for-bit.
 */
#ifndef SFGPU_H
#define SFGPU_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sf_ctx sf_ctx;

typedef enum sf_status {
    SF_OK = 0,
    SF_ERR_INVALID = 1,       /* bad argument */
    SF_ERR_HIP = 2,           /* a HIP runtime call failed (message has hipGetErrorString) */
    SF_ERR_RCCL = 3,          /* an RCCL call failed */
    SF_ERR_HALO_EXCEEDED = 4, /* advect back-traced beyond the slab's ghost plane (SPEC §4) */
    SF_ERR_NO_DEVICE = 5      /* no usable gfx950 device: there is NO CPU fallback */
} sf_status;

typedef enum sf_dtype { SF_F32 = 0, SF_F64 = 1 } sf_dtype;

/* Field slots. p and div of project() alias SF_U0 and SF_V0 (SPEC §1). SF_USER0..3 are extra
 * resident slots (e.g. per-step sources kept in HBM), allocated on first use. */
typedef enum sf_field {
    SF_U = 0, SF_V = 1, SF_W = 2, SF_U0 = 3, SF_V0 = 4, SF_W0 = 5, SF_DENS = 6, SF_DENS0 = 7,
    SF_USER0 = 8, SF_USER1 = 9, SF_USER2 = 10, SF_USER3 = 11, SF_NUM_FIELDS = 12
} sf_field;

#define SF_NCCL_ID_BYTES 128

typedef struct sf_params {
    int N;            /* interior cells per axis (>= 1)                                         */
    int dtype;        /* sf_dtype                                                               */
    int iters;        /* Jacobi iterations K per lin_solve                                      */
    double dt, diff, visc;
    int device;       /* HIP device ordinal for this process                                    */
    int nslabs_local; /* logical k-slabs held by this context (>= 1); total = nranks * this     */
    int rank, nranks; /* this process / number of processes (one per GPU). nranks == 1: no RCCL */
    const void* nccl_id; /* SF_NCCL_ID_BYTES from sf_nccl_unique_id() of rank 0 when nranks > 1 */
    int flags;        /* 0, or SF_FLAG_* bits                                                   */
} sf_params;

/* Measurement aid: the context takes the geometry, streams and launch schedule of rank `rank` of `nranks`, but the
 * halo messages to ranks in other processes become device-local copies of its own planes of the same size and no
 * communicator is created (nccl_id may be NULL). Timing of one rank's share on a single GPU; the field values
 * next to the slab boundaries are then meaningless. */
#define SF_FLAG_LOOPBACK_HALO 1

/* Verification aid for the inter-process data plane on a ONE-GPU box (RCCL refuses two ranks on one device):
 * nranks == 1, nslabs_local >= 2, and the ghost planes of the logical slabs travel through a real single-rank RCCL
 * communicator as grouped ncclSend / ncclRecv to self — the same calls, counts, datatypes, buffer offsets, streams
 * and system-scope halo event as the multi-process exchange (whose role is that of the reference's 2-GPU buffer
 * exchange, solver-unidyn.cu:396-470) — instead of the device-local copy kernel. Results are bit-identical to the
 * copy transport and to the undecomposed solve; sf_create also runs the schedule measurement and its all-reduce
 * vote over that communicator, as a multi-process context does. */
#define SF_FLAG_RCCL_SELF 2

/* Library / build identification ("sfgpu <ver> gfx950 hip"). */
const char* sf_version(void);
const char* sf_status_string(int status);

/* Fills `out` (SF_NCCL_ID_BYTES) with a fresh ncclUniqueId; rank 0 calls it and ships the bytes to
 * the other ranks by any means (bench.py uses torch.distributed). */
int sf_nccl_unique_id(void* out);

/* Takes the role of the allocation block of solver.cu:74-106 / solver-unidyn.cu:90-114:
 * all fields, streams and (nranks > 1) the communicator. *out is NULL on failure; the message is
 * then available from sf_last_error(NULL). N must be divisible by nranks*nslabs_local. */
int sf_create(sf_ctx** out, const sf_params* p);
void sf_destroy(sf_ctx* ctx);

/* Host <-> device copies (cudaMemcpy sites solver.cu:131,151,168-169; solver-unidyn.cu:475).
 * `host` is the dense GLOBAL (N+2)^3 array; each context reads / writes only the planes of its own
 * slabs (download: planes it owns, plus the physical shell planes on the end slabs). Synchronous. */
int sf_upload(sf_ctx* ctx, int field, const void* host);
int sf_download(sf_ctx* ctx, int field, void* host);
/* Same for a run of whole planes: global k in [k_begin, k_end) must lie inside this context's
 * stored range; host points at plane k_begin ((N+2)^2 elements per plane). */
int sf_download_planes(sf_ctx* ctx, int field, int k_begin, int k_end, void* host);
int sf_upload_planes(sf_ctx* ctx, int field, int k_begin, int k_end, const void* host);
/* First / one-past-last global interior plane owned by this context (1-based k). sf_upload_planes
 * accepts any [k_begin, k_end) and copies the planes of it that this context STORES (its interior
 * planes and the ghost / shell plane either side), so a rank can upload just its own part. */
int sf_owned_planes(const sf_ctx* ctx, int* k_begin, int* k_end);
/* Global planes [k_begin, k_end) this context stores (owned planes plus its ghost / shell planes, clipped
 * to 0 .. N+2): the range a rank should fill with sf_upload_planes before the first step. */
int sf_stored_planes(const sf_ctx* ctx, int* k_begin, int* k_end);

/* Device-side helpers, asynchronous. */
int sf_fill(sf_ctx* ctx, int field, double value);        /* every stored entry = value           */
int sf_copy_field(sf_ctx* ctx, int dst, int src);         /* dst <- src (device to device)        */

/* The two entry points north_star names. They take the place of the per-step kernel sequence of the
 * reference loop (solver.cu:181-198; solver-unidyn.cu:324-393). Sources are whatever is currently in
 * SF_U0/V0/W0 (vel_step) and SF_DENS0 (dens_step); on return those slots hold scratch (SPEC §3). */
int vel_step(sf_ctx* ctx);
int dens_step(sf_ctx* ctx);

/* Resident sources. After sf_bind_sources(ctx, su, sv, sw, sd) every vel_step / dens_step behaves exactly as if
 * SF_U0, SF_V0, SF_W0 (vel_step) and SF_DENS0 (dens_step) had first been overwritten with copies of the bound
 * slots (normally SF_USER0..3) — same bits, one pass less over memory than sf_copy_field + the step. Pass -1 for
 * a source that stays "whatever is in the x0 slot"; sf_bind_sources(ctx, -1, -1, -1, -1) restores the default. */
int sf_bind_sources(sf_ctx* ctx, int su, int sv, int sw, int sd);

/* The operators of the path, exposed singly for parity tests and for timing the Jacobi sweep in
 * isolation. Field arguments are sf_field slots; `b` is the boundary mode 0..3. */
int sf_add_source(sf_ctx* ctx, int x, int s);
int sf_set_bnd(sf_ctx* ctx, int b, int x);
int sf_lin_solve(sf_ctx* ctx, int b, int x, int x0, double a, double c, int iters);
int sf_diffuse(sf_ctx* ctx, int b, int x, int x0, double diff);
int sf_advect(sf_ctx* ctx, int b, int d, int d0, int u, int v, int w);
int sf_project(sf_ctx* ctx, int u, int v, int w, int p, int div);

/* Asynchronous frame output (the reference blocks on cudaDeviceSynchronize + cudaMemcpy +
 * per-value sprintf every output step, solver-unidyn.cu:475-487). sf_snapshot copies up to 4 fields into
 * context-owned snapshot buffers on the compute stream (device to device, ordered after everything issued
 * so far) and returns at once. sf_snapshot_read(index, host) waits for that copy only, then downloads snapshot
 * `index` as a dense global (N+2)^3 array on a separate copy stream. It may be called from ANOTHER host
 * thread while the owner keeps stepping; do not call sf_snapshot again before all reads have returned. */
int sf_snapshot(sf_ctx* ctx, const int* fields, int nfields);
int sf_snapshot_read(sf_ctx* ctx, int index, void* host);
/* The same for global planes [k_begin, k_end) only; `host` holds exactly those planes, dense (N+2)^2 each (what a
 * rank of a decomposed run needs for its own frame file: solver-unidyn.cu:484-490 writes one file per device). */
int sf_snapshot_read_planes(sf_ctx* ctx, int index, int k_begin, int k_end, void* host);

/* Tracer particles (docs/SPEC.md §6; feeds the write_point_mesh call of solver-unidyn.cu:487). Positions are
 * x y z triples in grid-index coordinates, element type = the context's dtype. Single-slab contexts only.
 * sf_tracers_advect moves them through SF_U/V/W by one dt; sf_tracers_get returns positions and, if the
 * pointers are non-NULL, the density and speed sampled at each tracer. */
int sf_tracers_set(sf_ctx* ctx, int n, const void* xyz);
int sf_tracers_advect(sf_ctx* ctx);
int sf_tracers_get(sf_ctx* ctx, void* xyz, void* dens_sample, void* speed_sample);

/* Run-time parameters (the reference only has compile-time #defines, FluidGPU.cuh:1-31). */
int sf_set_iters(sf_ctx* ctx, int iters);
int sf_set_coefficients(sf_ctx* ctx, double dt, double diff, double visc);

/* Waits for all streams of the context; returns deferred errors (SF_ERR_HALO_EXCEEDED, HIP faults).
 * Role of the cudaDeviceSynchronize calls of solver-unidyn.cu:369,380,403. */
int sf_sync(sf_ctx* ctx);
/* Message of the last failing call on ctx (ctx == NULL: of the last failing sf_create). */
const char* sf_last_error(const sf_ctx* ctx);

/* Device timers on the context's compute stream (the cudaEvent pair of solver.cu:175-197).
 * sf_timer_stop synchronises on the stop event and returns milliseconds between the two records. */
int sf_timer_start(sf_ctx* ctx);
int sf_timer_stop(sf_ctx* ctx, float* ms);

/* Measures a plain 16-byte-per-lane device copy of `bytes` bytes (read + write = 2*bytes of traffic)
 * on the context's device, `reps` times after one warm-up; returns the best rate in GB/s of
 * traffic. Used by bench.py to quote the achievable-HBM figure in the same run. */
int sf_measure_copy_bandwidth(sf_ctx* ctx, size_t bytes, int reps, double* gbps);

/* Number of kernel launches sf_lin_solve(..., iters) issues per field group on this context: sweeps are
 * fused in pairs where the layout allows (docs: DESIGN.md §4), so this is iters/2 (+1 if odd) or iters.
 * Lets a benchmark convert a lin_solve time into a per-launch time comparable with rocprof. */
int sf_lin_solve_launches(const sf_ctx* ctx, int iters);

/* Geometry of the internal layout, for reports: row pitch (elements), planes stored per slab,
 * bytes per field per slab. Any pointer may be NULL. */
int sf_layout_info(const sf_ctx* ctx, int* row_pitch, int* planes_per_slab, size_t* bytes_per_field);

/* Launch schedule of a decomposed lin_solve, for reports: pairs per trapezoid block (0 = boundary launch of fixed
 * size); `measured` bit 0: sf_create measured the depth on this machine (else default / SF_TRAP), bit 1: it also
 * measured "u,v,w one field at a time" against "three fields per launch", bit 2: the three-field form is in use. */
int sf_schedule_info(const sf_ctx* ctx, int* trapezoid_pairs, int* measured);

/* Which halo transport this context uses and how often it ran: *transport = 0 none (one slab), 1 device-local copy
 * kernel between logical slabs, 2 RCCL send/recv between processes, 3 RCCL send/recv to self (SF_FLAG_RCCL_SELF),
 * 4 loopback copies (SF_FLAG_LOOPBACK_HALO); *rccl_groups = ncclGroupEnd calls of the halo exchange issued so far. */
int sf_transport_info(const sf_ctx* ctx, int* transport, long* rccl_groups);

#ifdef __cplusplus
}
#endif
#endif /* SFGPU_H */
