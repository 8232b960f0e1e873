// sf_driver.cpp — C++ host driver over the C ABI (include/sfgpu.h) and the VTK frame writer.
//
// Keeps the SHAPE of the reference's host loop (solver.cu:171-216, solver-unidyn.cu:313-573):
// per step print "t= <t>", bracket the device work with an event pair, print
// "done.\nElapsed kernel time: <ms> ms", and every <every> steps copy the fields back and write
// "anim_s<frame>.vtk" (naming of solver.cu:210 / solver-unidyn.cu:484). Errors follow the
// reference's CUDA_CHECK_RETURN convention (FluidGPU.cuh:34-41): print and exit(1).
// Unlike the reference (compile-time #defines, argv ignored — solver.cu:17-19,64) everything is a
// run-time option, and the grid solver behind vel_step/dens_step is the stable-fluids path of
// docs/SPEC.md, not the reference's SPH kernels.
//
// Output is asynchronous: at an output step the fields are snapshotted on the device and a
// writer thread downloads and encodes the frame while the main loop keeps stepping; the reference blocks on
// cudaDeviceSynchronize + cudaMemcpy + per-value sprintf instead (solver-unidyn.cu:475-487). With --tracers N a
// cloud of passive tracers is advected through the velocity field and written with write_point_mesh, the one
// writer call the reference really makes (solver-unidyn.cu:487: ASCII, two point scalars).
//
// Several GPUs: start one process per GPU with RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT in the environment
// (e.g. `python -m torch.distributed.run --no-python --nproc-per-node 8 ./sf_driver ...`). Rank 0 creates the
// ncclUniqueId and hands it to the others through a file under /tmp keyed by MASTER_PORT (single node); every
// rank then owns one k-slab and writes its own frames as anim_s_GPU<rank>_<frame>.vtk — the per-device naming
// of solver-unidyn.cu:484-490 — as a rectilinear mesh carrying its z range.
//
//   sf_driver [--n 64] [--steps 20] [--iters 20] [--dtype f32|f64] [--every 10] [--out DIR]
//             [--binary] [--device 0] [--slabs 1] [--plumbing] [--quiet] [--sync-output] [--tracers 0]
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <filesystem>
#include <iostream>
#include <sstream>
#include <string>
#include <thread>
#include <ctime>
#include <sys/stat.h>
#include <unistd.h>
#include <vector>

#include "../../include/sf_visit_writer.h"
#include "../../include/sfgpu.h"

static sf_ctx* g_ctx = nullptr;

#define SF_CHECK_RETURN(value)                                                                  \
    {                                                                                           \
        int _m_stat = (value);                                                                  \
        if (_m_stat != SF_OK) {                                                                 \
            fprintf(stderr, "Error %s (%s) at line %d in file %s\n", sf_status_string(_m_stat), \
                    sf_last_error(g_ctx), __LINE__, __FILE__);                                  \
            exit(1);                                                                            \
        }                                                                                       \
    }

struct Options {
    int n = 64, steps = 20, iters = 20, every = 10, device = 0, slabs = 1, tracers = 0;
    bool f64 = false, binary = false, plumbing = false, quiet = false, sync_output = false, loopback = false;
    std::string out = ".";
    int rank = 0, world = 1, local_rank = 0;
};

static int env_int(const char* name, int dflt) {
    const char* v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

// Single-node exchange of the 128-byte ncclUniqueId: rank 0 writes it (atomically, via rename), the others poll.
// The file name is unique per launch — it carries the pid of the launching agent (torch.distributed.run starts every
// local rank as its child) besides the rendezvous port — so a file left by an earlier run can never be taken for
// this run's; rank 0 removes a stale one before it publishes, and removes its own once sf_create has returned (the
// communicator exists then, so every rank has read the id). SF_NCCL_ID_FILE names the file explicitly for launchers
// whose ranks do not share a parent.
static std::string nccl_id_path() {
    if (const char* f = getenv("SF_NCCL_ID_FILE")) return f;
    char path[256];
    snprintf(path, sizeof path, "/tmp/sf_ncclid_%d_%ld_%s.bin", env_int("MASTER_PORT", 29500), (long)getppid(),
             getenv("TORCHELASTIC_RUN_ID") ? getenv("TORCHELASTIC_RUN_ID") : "run");
    return path;
}

// Start of this process (seconds since the epoch), from /proc: files older than it cannot belong to this launch.
static time_t process_start_time() {
    struct stat st;
    return stat("/proc/self", &st) == 0 ? st.st_ctime : time(nullptr);
}

static void share_nccl_id(const Options& o, unsigned char* id) {
    const std::string path = nccl_id_path();
    if (o.rank == 0) {
        unlink(path.c_str());  // whatever is there is not ours
        if (sf_nccl_unique_id(id) != SF_OK) {
            fprintf(stderr, "Error: sf_nccl_unique_id failed\n");
            exit(1);
        }
        const std::string tmp = path + ".tmp" + std::to_string((long)getpid());
        FILE* f = fopen(tmp.c_str(), "wb");
        if (!f || fwrite(id, 1, SF_NCCL_ID_BYTES, f) != SF_NCCL_ID_BYTES || fclose(f) != 0 ||
            rename(tmp.c_str(), path.c_str()) != 0) {
            fprintf(stderr, "Error: cannot publish the nccl id at %s\n", path.c_str());
            exit(1);
        }
    } else {
        // With SF_NCCL_ID_FILE the name is NOT unique per launch: a file older than this process was left by an
        // earlier (crashed) run — rank 0 of this run has not published yet — and is not taken.
        const time_t born = process_start_time();
        for (int tries = 0; tries < 1200; ++tries) {  // up to 120 s
            struct stat st;
            if (stat(path.c_str(), &st) == 0 && st.st_mtime + 1 >= born) {
                FILE* f = fopen(path.c_str(), "rb");
                if (f) {
                    const size_t n = fread(id, 1, SF_NCCL_ID_BYTES, f);
                    fclose(f);
                    if (n == SF_NCCL_ID_BYTES) return;
                }
            }
            usleep(100000);
        }
        fprintf(stderr, "Error: rank %d timed out waiting for %s\n", o.rank, path.c_str());
        exit(1);
    }
}

static Options parse(int argc, char** argv) {
    Options o;
    o.rank = env_int("RANK", 0);
    o.world = env_int("WORLD_SIZE", 1);
    o.local_rank = env_int("LOCAL_RANK", 0);
    bool rank_or_world_given = false;
    for (int a = 1; a < argc; ++a) {
        const std::string s = argv[a];
        auto next = [&]() -> const char* {
            if (a + 1 >= argc) {
                fprintf(stderr, "missing value after %s\n", s.c_str());
                exit(2);
            }
            return argv[++a];
        };
        if (s == "--n") o.n = atoi(next());
        else if (s == "--steps") o.steps = atoi(next());
        else if (s == "--iters") o.iters = atoi(next());
        else if (s == "--every") o.every = atoi(next());
        else if (s == "--device") o.device = atoi(next());
        else if (s == "--slabs") o.slabs = atoi(next());
        else if (s == "--out") o.out = next();
        else if (s == "--dtype") o.f64 = (std::string(next()) == "f64");
        else if (s == "--binary") o.binary = true;
        else if (s == "--plumbing") o.plumbing = true;
        else if (s == "--quiet") o.quiet = true;
        else if (s == "--sync-output") o.sync_output = true;
        else if (s == "--tracers") o.tracers = atoi(next());
        // rehearsal of ONE rank's share on a one-GPU box: the geometry, buffers, launches and frame file of rank
        // --rank of --world, halo messages replaced by device-local copies (SF_FLAG_LOOPBACK_HALO), no communicator
        else if (s == "--loopback") o.loopback = true;
        else if (s == "--rank") { o.rank = atoi(next()); rank_or_world_given = true; }
        else if (s == "--world") { o.world = atoi(next()); rank_or_world_given = true; }
        else {
            fprintf(stderr, "unknown option %s\n", s.c_str());
            exit(2);
        }
    }
    // --rank / --world override the launcher's environment only to rehearse one rank's share with local copies: without
    // --loopback a real communicator would wait 120 s for an id nobody publishes
    if (rank_or_world_given && !o.loopback) {
        fprintf(stderr, "--rank / --world need --loopback (ranks of a real run come from RANK / WORLD_SIZE)\n");
        exit(2);
    }
    // frames go to <out>/: create it (the reference writes into the working directory, which always exists)
    if (o.every > 0) {
        std::error_code ec;
        std::filesystem::create_directories(o.out, ec);
    }
    return o;
}

// Analytic inputs of docs/SPEC.md §5, evaluated in double and rounded to T: ONE field on the global planes [kb, ke)
// this process stores (never the whole (N+2)^3 array: at 1024^3 that is 4.3 GB per field and rank).
// which: 0 u, 1 v, 2 w, 3 dens, 4 su, 5 sv, 6 sw, 7 sd.
template <class T>
static std::vector<T> make_input(int which, int N, int kb, int ke, double dt, bool plumbing) {
    const size_t S = (size_t)N + 2;
    std::vector<T> f(S * S * (size_t)(ke - kb), T(0));
    const double PI2 = 6.283185307179586476925286766559;
    const double A = 0.5 / (dt * N);
    auto at = [&](int i, int j, int k) -> T& { return f[(size_t)i + S * ((size_t)j + S * (size_t)(k - kb))]; };
    if (!plumbing && (which == 0 || which == 1 || which == 3)) {
        std::vector<double> sn(S), cs(S);
        for (int i = 1; i <= N; ++i) {
            sn[i] = std::sin(PI2 * (i - 0.5) / N);
            cs[i] = std::cos(PI2 * (i - 0.5) / N);
        }
        for (int k = std::max(kb, 1); k < std::min(ke, N + 1); ++k)
            for (int j = 1; j <= N; ++j)
                for (int i = 1; i <= N; ++i) {
                    if (which == 0) at(i, j, k) = (T)(A * sn[i] * cs[j]);
                    if (which == 1) at(i, j, k) = (T)(-A * cs[i] * sn[j]);
                    if (which == 3) at(i, j, k) = (T)(0.5 + 0.5 * sn[i] * sn[j] * sn[k]);
                }
    }
    const int c = N / 2 > 0 ? N / 2 : 1;
    if (c >= kb && c < ke) {
        if (which == 7) at(c, c, c) = T(100);
        if (which == 5) at(c, c, c) = plumbing ? T(5) : (T)A;
    }
    return f;
}

// One frame = the interior cells of global planes [kb, ke) (1-based k; the host arrays hold exactly those planes),
// density scalar + velocity vector, cell centred. Single process: the whole cube as a regular mesh "anim_s<frame>.vtk". Several processes: each rank
// writes its slab "anim_s_GPU<rank>_<frame>.vtk" as a rectilinear mesh whose z coordinates are its plane range.
template <class T>
static void write_frame(const Options& o, int frame, int kb, int ke, const std::vector<T>& dens,
                        const std::vector<T>& u, const std::vector<T>& v, const std::vector<T>& w) {
    const int N = o.n;
    const size_t S = (size_t)N + 2;
    const size_t ncell = (size_t)N * N * (size_t)(ke - kb);
    std::vector<float> d(ncell), vel(3 * ncell);
    size_t q = 0;
    for (int k = kb; k < ke; ++k)
        for (int j = 1; j <= N; ++j)
            for (int i = 1; i <= N; ++i, ++q) {
                const size_t s = (size_t)i + S * ((size_t)j + S * (size_t)(k - kb));  // buffers start at plane kb
                d[q] = (float)dens[s];
                vel[3 * q + 0] = (float)u[s];
                vel[3 * q + 1] = (float)v[s];
                vel[3 * q + 2] = (float)w[s];
            }
    std::ostringstream oss;
    int vardim[2] = {1, 3}, centering[2] = {0, 0};
    const char* names[2] = {"density", "velocity"};
    float* vars[2] = {d.data(), vel.data()};
    if (o.world == 1) {
        oss << o.out << "/anim_s" << frame << ".vtk";
        int dims[3] = {N + 1, N + 1, N + 1};  // point counts; cells = N^3 (visit_writer.cpp:901-905)
        write_regular_mesh(oss.str().c_str(), o.binary ? 1 : 0, dims, 2, vardim, centering, names, vars);
    } else {
        oss << o.out << "/anim_s_GPU" << o.rank << "_" << frame << ".vtk";  // solver-unidyn.cu:484
        int dims[3] = {N + 1, N + 1, ke - kb + 1};
        std::vector<float> xs(N + 1), zs(ke - kb + 1);
        for (int a = 0; a <= N; ++a) xs[a] = (float)a;
        for (int a = 0; a <= ke - kb; ++a) zs[a] = (float)(kb - 1 + a);
        write_rectilinear_mesh(oss.str().c_str(), o.binary ? 1 : 0, dims, xs.data(), xs.data(), zs.data(), 2, vardim,
                               centering, names, vars);
    }
}

template <class T>
static int run(const Options& o) {
    const double dt = 0.1, diff = 1e-4, visc = 1e-4;
    sf_params p;
    std::memset(&p, 0, sizeof p);
    p.N = o.n;
    p.dtype = sizeof(T) == 4 ? SF_F32 : SF_F64;
    p.iters = o.iters;
    p.dt = dt;
    p.diff = diff;
    p.visc = visc;
    p.device = (o.world > 1 && !o.loopback) ? o.local_rank : o.device;
    p.flags = o.loopback ? SF_FLAG_LOOPBACK_HALO : 0;
    p.nslabs_local = o.slabs;
    p.rank = o.rank;
    p.nranks = o.world;
    unsigned char nccl_id[SF_NCCL_ID_BYTES];
    if (o.world > 1 && !o.loopback) {
        share_nccl_id(o, nccl_id);
        p.nccl_id = nccl_id;
    }
    int rc = sf_create(&g_ctx, &p);
    if (rc != SF_OK) {
        fprintf(stderr, "Error %s (%s) at line %d in file %s\n", sf_status_string(rc), sf_last_error(nullptr),
                __LINE__, __FILE__);
        exit(1);
    }
    if (o.world > 1 && !o.loopback && o.rank == 0) unlink(nccl_id_path().c_str());  // every rank has joined the communicator
    const bool talk = (o.rank == 0);
    if (talk)
        std::cout << sf_version() << "  N=" << o.n << " K=" << o.iters << " dtype=" << (sizeof(T) == 4 ? "f32" : "f64")
                  << " slabs=" << o.slabs * o.world << " ranks=" << o.world << "\n";
    int own_kb = 1, own_ke = o.n + 1;
    SF_CHECK_RETURN(sf_owned_planes(g_ctx, &own_kb, &own_ke));

    // inputs: only the planes this process stores, one field at a time (host memory per rank stays O(N^3 / ranks))
    int st_kb = 0, st_ke = o.n + 2;
    SF_CHECK_RETURN(sf_stored_planes(g_ctx, &st_kb, &st_ke));
    {
        const int slot[8] = {SF_U, SF_V, SF_W, SF_DENS, SF_USER0, SF_USER1, SF_USER2, SF_USER3};
        for (int q = 0; q < 8; ++q) {
            const std::vector<T> f = make_input<T>(q, o.n, st_kb, st_ke, dt, o.plumbing);
            SF_CHECK_RETURN(sf_upload_planes(g_ctx, slot[q], st_kb, st_ke, f.data()));
        }
    }
    for (int f : {SF_U, SF_V, SF_W}) SF_CHECK_RETURN(sf_set_bnd(g_ctx, f + 1, f));
    SF_CHECK_RETURN(sf_set_bnd(g_ctx, 0, SF_DENS));
    // sources stay resident in HBM (SF_USER0..3) and are re-injected every step
    SF_CHECK_RETURN(sf_bind_sources(g_ctx, SF_USER0, SF_USER1, SF_USER2, SF_USER3));

    // frame buffers: the planes this process owns, nothing else
    const size_t n = ((size_t)o.n + 2) * ((size_t)o.n + 2) * (size_t)(own_ke - own_kb);
    std::vector<T> hd, hu, hv, hw;
    if (o.every > 0) {
        hd.resize(n);
        hu.resize(n);
        hv.resize(n);
        hw.resize(n);
    }

    // tracers: a small lattice in the middle of the box, in grid-index coordinates (SPEC §6)
    std::vector<T> tpos, tdens, tspeed;
    if (o.tracers > 0 && o.slabs == 1 && o.world == 1) {
        int side = 1;
        while (side * side * side < o.tracers) ++side;
        for (int c = 0; c < side && (int)tpos.size() / 3 < o.tracers; ++c)
            for (int b = 0; b < side && (int)tpos.size() / 3 < o.tracers; ++b)
                for (int a = 0; a < side && (int)tpos.size() / 3 < o.tracers; ++a) {
                    tpos.push_back((T)(0.25 * o.n + 0.5 * o.n * (a + 0.5) / side));
                    tpos.push_back((T)(0.25 * o.n + 0.5 * o.n * (b + 0.5) / side));
                    tpos.push_back((T)(0.25 * o.n + 0.5 * o.n * (c + 0.5) / side));
                }
        tdens.resize(tpos.size() / 3);
        tspeed.resize(tpos.size() / 3);
        SF_CHECK_RETURN(sf_tracers_set(g_ctx, (int)(tpos.size() / 3), tpos.data()));
    }
    const int ntr = (int)(tpos.size() / 3);

    std::thread writer;  // at most one frame in flight
    auto write_tracers = [&](int frame) {
        std::vector<float> pts(3 * (size_t)ntr), m(ntr), sp(ntr);
        for (int q = 0; q < 3 * ntr; ++q) pts[q] = (float)tpos[q];
        for (int q = 0; q < ntr; ++q) {
            m[q] = (float)tdens[q];
            sp[q] = (float)tspeed[q];
        }
        std::ostringstream oss;
        oss << o.out << "/tracers_s" << frame << ".vtk";
        int vardims[2] = {1, 1};
        const char* names[2] = {"density", "speed"};
        float* arrays[2] = {m.data(), sp.data()};
        write_point_mesh(oss.str().c_str(), 0, ntr, pts.data(), 2, vardims, names, arrays);  // solver-unidyn.cu:487
    };

    double total_ms = 0;
    const auto wall0 = std::chrono::steady_clock::now();
    for (int t = 0; t < o.steps; t++) {
        if (!o.quiet && talk) std::cout << "t= " << t << "\n";
        float elapsedTime = 0.f;
        SF_CHECK_RETURN(sf_timer_start(g_ctx));
        SF_CHECK_RETURN(vel_step(g_ctx));  // sources are bound (sf_bind_sources above)
        SF_CHECK_RETURN(dens_step(g_ctx));
        if (ntr > 0) SF_CHECK_RETURN(sf_tracers_advect(g_ctx));
        SF_CHECK_RETURN(sf_timer_stop(g_ctx, &elapsedTime));
        total_ms += elapsedTime;
        if (!o.quiet && talk) std::cout << "done.\nElapsed kernel time: " << elapsedTime << " ms\n";

        if (o.every > 0 && t % o.every == 0) {
            const int frame = t / o.every;
            if (writer.joinable()) writer.join();  // the previous frame must be out before its buffers are reused
            if (ntr > 0) SF_CHECK_RETURN(sf_tracers_get(g_ctx, tpos.data(), tdens.data(), tspeed.data()));
            if (o.sync_output) {
                SF_CHECK_RETURN(sf_sync(g_ctx));
                SF_CHECK_RETURN(sf_download_planes(g_ctx, SF_DENS, own_kb, own_ke, hd.data()));
                SF_CHECK_RETURN(sf_download_planes(g_ctx, SF_U, own_kb, own_ke, hu.data()));
                SF_CHECK_RETURN(sf_download_planes(g_ctx, SF_V, own_kb, own_ke, hv.data()));
                SF_CHECK_RETURN(sf_download_planes(g_ctx, SF_W, own_kb, own_ke, hw.data()));
                write_frame<T>(o, frame, own_kb, own_ke, hd, hu, hv, hw);
                if (ntr > 0) write_tracers(frame);
            } else {
                const int fields[4] = {SF_DENS, SF_U, SF_V, SF_W};
                SF_CHECK_RETURN(sf_snapshot(g_ctx, fields, 4));
                writer = std::thread([&, frame]() {
                    T* dst[4] = {hd.data(), hu.data(), hv.data(), hw.data()};
                    for (int q = 0; q < 4; ++q)
                        if (sf_snapshot_read_planes(g_ctx, q, own_kb, own_ke, dst[q]) != SF_OK) {
                            fprintf(stderr, "Error: snapshot read failed for frame %d\n", frame);
                            exit(1);
                        }
                    write_frame<T>(o, frame, own_kb, own_ke, hd, hu, hv, hw);
                    if (ntr > 0) write_tracers(frame);
                });
            }
        }
    }
    if (writer.joinable()) writer.join();
    SF_CHECK_RETURN(sf_sync(g_ctx));
    const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
    if (talk) std::cout << "wall time of the loop incl. output: " << wall_s << " s ("
              << (o.sync_output ? "synchronous" : "asynchronous") << " output)\n";
    const double cells = (double)o.n * o.n * o.n;
    if (o.steps > 0 && talk)
        std::cout << "mean step " << total_ms / o.steps << " ms, " << cells * o.steps / (total_ms * 1e-3) / 1e6
                  << " Mcells/s\n";
    sf_destroy(g_ctx);
    g_ctx = nullptr;
    return 0;
}

int main(int argc, char** argv) {
    const Options o = parse(argc, argv);
    return o.f64 ? run<double>(o) : run<float>(o);
}
