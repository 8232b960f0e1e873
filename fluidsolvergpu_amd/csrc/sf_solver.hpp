// sf_solver.hpp — host side of libsfgpu.so: contexts, slab decomposition, halo exchange, the step
// sequences of docs/SPEC.md §3 and the C ABI of include/sfgpu.h.
//
// Structure (MI355X-first, not a translation of the reference's host loop, solver.cu:171-216):
//   * one context = L logical k-slabs of one process on one GPU; P = nranks*L slabs in total.
//     Each slab has a compute stream and a halo stream. An operator is launched first on the two
//     slab-boundary planes, then on the interior planes; the halo stream ships the boundary planes
//     (device-to-device copy between slabs of the same process, RCCL send/recv grouped over xGMI
//     between processes) while the interior sweep runs. No collective reduction exists anywhere in
//     the step, only neighbour exchange.
//   * fields are named slots holding device pointers, so SPEC's "swap" is a pointer swap.
//   * there is NO CPU fallback: without a gfx950 device sf_create fails with SF_ERR_NO_DEVICE.
#pragma once
#include "sf_base.hpp"
#include "sf_kernels.hpp"

namespace sfi {

template <class T>
class Solver final : public SolverBase {
    static constexpr int W = sfk::VecT<T>::W;
    static constexpr int NSCRATCH = 3;

    struct Slab {
        int gid = 0;  // global slab index 0..P-1
        sfk::Geom geom{};
        T* field[SF_NUM_FIELDS] = {};
        T* scratch[NSCRATCH] = {};
        T* snap[4] = {};               // snapshot buffers for asynchronous output
        hipStream_t os = nullptr;      // output (copy) stream
        hipEvent_t snap_done = nullptr;
        hipStream_t cs = nullptr;   // compute (interior planes, whole-field operators)
        hipStream_t bs = nullptr;   // boundary planes of a decomposed grid: runs beside the interior launch
        hipStream_t hs = nullptr;   // halo
        hipStream_t cur = nullptr;  // where the launch being issued goes (cs or bs)
        hipEvent_t cs_mark = nullptr;
        hipEvent_t boundary_done = nullptr;
        hipEvent_t halo_done = nullptr;
        int* d_flag = nullptr;
    };

public:
    explicit Solver(const sf_params& p) : N_(p.N), K_(p.iters), device_(p.device) {
        SF_REQUIRE(p.N >= 1, "N must be >= 1");
        SF_REQUIRE(p.iters >= 0, "iters must be >= 0");
        L_ = p.nslabs_local > 0 ? p.nslabs_local : 1;
        nranks_ = p.nranks > 0 ? p.nranks : 1;
        rank_ = p.rank;
        SF_REQUIRE(rank_ >= 0 && rank_ < nranks_, "rank out of range");
        P_ = nranks_ * L_;
        SF_REQUIRE(N_ % P_ == 0, "N must be divisible by nranks*nslabs_local");
        loopback_ = (p.flags & SF_FLAG_LOOPBACK_HALO) != 0;
        rccl_self_ = (p.flags & SF_FLAG_RCCL_SELF) != 0;
        SF_REQUIRE(!rccl_self_ || (nranks_ == 1 && L_ >= 2 && !loopback_),
                   "SF_FLAG_RCCL_SELF needs nranks == 1, nslabs_local >= 2 and no loopback flag");
        SF_REQUIRE(nranks_ == 1 || loopback_ || p.nccl_id != nullptr, "nccl_id required when nranks > 1");
        set_coefficients(p.dt, p.diff, p.visc);

        int ndev = 0;
        if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
            throw Failure{SF_ERR_NO_DEVICE, "no HIP device visible: libsfgpu has no CPU fallback"};
        SF_REQUIRE(device_ >= 0 && device_ < ndev, "device ordinal out of range");
        SF_HIP(hipSetDevice(device_));
        hipDeviceProp_t prop;
        SF_HIP(hipGetDeviceProperties(&prop, device_));
        if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0)
            throw Failure{SF_ERR_NO_DEVICE, std::string("device is ") + prop.gcnArchName +
                                                ", this library is built for gfx950 only"};
        num_cu_ = prop.multiProcessorCount;

        // layout
        nzl_ = N_ / P_;
        lead_ = 128 / (int)sizeof(T);
        const int line = 128 / (int)sizeof(T);
        px_ = ceil_div(lead_ + N_ + 1 + W, line) * line;
        plane_ = (long)px_ * (N_ + 2);
        // two ghost planes per side let sweep pairs be fused across slab boundaries (one exchange per pair); grids
        // the fused kernel does not take (rows wider than 512 vectors, N not a multiple of W) keep one ghost plane
        // and exchange one plane per sweep
        const bool fusable = env_int("SF_FUSE2", 1) != 0 && N_ % W == 0 && N_ / W <= fuse_maxvec_;
        G_ = (P_ > 1 && nzl_ >= 2 && fusable && env_int("SF_GHOST", 4) >= 2) ? 2 : 1;
        // three / four ghost planes where the marching kernel will run three / four sweeps per pass on the slab
        // interiors (one exchange per pass): the interior launch [2G, nzl) must be long and large enough for it
        {
            const long min_cells = (long)env_int("SF_MARCH_MINCELLS_K", 2500) * 1000L;  // (march_min_cells_, below)
            const int ghost_max = env_int("SF_GHOST", 4), smax = env_int("SF_SK_S", 4);
            for (int gs = 3; gs <= 4; ++gs) {  // S = gs sweeps per exchange need gs ghost planes
                const int interior = nzl_ - 2 * gs;
                // (SF_ISHELL=0 — every sweep reads and writes the i-shell in memory — rules the marching kernel out in
                // sweeps_in_launch(); the deeper ghost zone is kept all the same: the solve then runs pair launches of
                // boundary depth pair_depth() on it, the schedule that K < 7 and remainders take by default)
                if (G_ == gs - 1 && ghost_max >= gs && env_int("SF_MARCH", 1) != 0 &&
                    smax >= gs && env_int("SF_SPLIT", 1) != 0 && interior >= env_int("SF_MARCH_MINP", 12) &&
                    (long)N_ * N_ * interior >= min_cells)
                    G_ = gs;
            }
        }
        nplanes_ = nzl_ + 2 * G_;
        field_elems_ = plane_ * nplanes_ + 256;  // slack so whole-vector accesses never leave the buffer
        field_elems_ = (field_elems_ + W - 1) / W * W;
        // Rows of padding before and after every field: the marching kernel addresses the rows of a workgroup's tile
        // without clamping them into the plane (sfk::jsk_step), so the tile of the first j-block reaches up to three
        // rows below the first plane and the tile of the last one up to NW x TJ rows beyond the last plane. Never
        // stored to, and what is loaded there only feeds rows that are not stored.
        pad_front_ = (sfk::SK_PAD_ROWS_FRONT * (long)px_ + 63) / 64 * 64;
        pad_back_ = sfk::SK_PAD_ROWS_BACK * (long)px_;

        slabs_.resize(L_);
        for (int s = 0; s < L_; ++s) {
            Slab& sl = slabs_[s];
            sl.gid = rank_ * L_ + s;
            sl.geom.N = N_;
            sl.geom.nzl = nzl_;
            sl.geom.G = G_;
            sl.geom.np = nplanes_;
            sl.geom.kg0 = sl.gid * nzl_ + 1 - G_;  // first interior k = gid*nzl + 1 is local plane G
            sl.geom.px = px_;
            sl.geom.lead = lead_;
            sl.geom.plane = plane_;
            sl.geom.wall_lo = (sl.gid == 0);
            sl.geom.wall_hi = (sl.gid == P_ - 1);
            SF_HIP(hipStreamCreateWithFlags(&sl.cs, hipStreamNonBlocking));
            SF_HIP(hipStreamCreateWithFlags(&sl.bs, hipStreamNonBlocking));
            // Events that only order kernels of THIS device against each other skip the system-scope fence of the
            // default event (a cache write-back + invalidate per record: ~12 us between consecutive sweeps of a
            // decomposed grid). halo_done keeps it when the ghost planes are written by another GPU through RCCL.
            const unsigned ev_local = hipEventDisableTiming | (unsigned)hipEventDisableSystemFence;
            const unsigned ev_halo = ((nranks_ > 1 && !loopback_) || rccl_self_) ? (unsigned)hipEventDisableTiming : ev_local;
            SF_HIP(hipEventCreateWithFlags(&sl.cs_mark, ev_local));
            sl.cur = sl.cs;
            {
                // One slab per process (production): every halo is an RCCL message, and the chain
                // boundary launch -> message -> next boundary launch is what limits a pair once the messages take as
                // long as the interior work. Issued on ONE stream that chain needs no cross-stream hand-over (each
                // costs ~10 us, tools/evgap.hip): the halo "stream" is then the boundary stream itself
                // (SF_HALO_STREAM=1 keeps a separate one). With several slabs per process the copies pull from the
                // neighbours' buffers and stay on their own stream (SF_HALO_STREAM=2 shares there too: used by the
                // parity tests to run the shared-stream ordering against the oracle). (A high-priority halo stream
                // was measured in round 1: with logical slabs on one GPU the copy kernel pre-empts the sweeps, 2x slower.)
                const int hmode = env_int("SF_HALO_STREAM", 0);  // 0 as described, 1 always separate, 2 always shared
                if ((L_ == 1 && nranks_ > 1 && hmode == 0) || hmode == 2)
                    sl.hs = sl.bs;
                else
                    SF_HIP(hipStreamCreateWithFlags(&sl.hs, hipStreamNonBlocking));
            }
            SF_HIP(hipEventCreateWithFlags(&sl.boundary_done, ev_local));
            SF_HIP(hipEventCreateWithFlags(&sl.halo_done, ev_halo));
            for (int f = 0; f < SF_USER0; ++f) sl.field[f] = alloc_field();
            for (int f = 0; f < NSCRATCH; ++f) sl.scratch[f] = alloc_field();
            SF_HIP(hipMalloc(&sl.d_flag, sizeof(int)));
            SF_HIP(hipMemset(sl.d_flag, 0, sizeof(int)));
            SF_HIP(hipDeviceSynchronize());
        }
        SF_HIP(hipEventCreate(&t0_));
        SF_HIP(hipEventCreate(&t1_));
        if (nranks_ > 1 && !loopback_) {
            ncclUniqueId id;
            static_assert(sizeof(ncclUniqueId) <= SF_NCCL_ID_BYTES, "ncclUniqueId larger than ABI slot");
            std::memcpy(&id, p.nccl_id, sizeof id);
            SF_NCCL(ncclCommInitRank(&comm_, nranks_, id, rank_));
        } else if (rccl_self_) {
            // a real communicator of one rank: the logical slabs' ghost planes travel as grouped ncclSend / ncclRecv
            // to self (see exchange()), so the RCCL data plane runs on a one-GPU box
            ncclUniqueId id;
            SF_NCCL(ncclGetUniqueId(&id));
            SF_NCCL(ncclCommInitRank(&comm_, 1, id, 0));
        }
        nt_mode_ = env_int("SF_NT", 2);  // non-temporal stores: 0 never, 1 always, 2 beyond the Infinity Cache
        ishell_skip_ = env_int("SF_ISHELL", 1) != 0;
        dead_ishell_opt_ = env_int("SF_ISHELL", 1) == 1;
        fuse2_ = env_int("SF_FUSE2", 1) != 0;  // 0 single sweeps, 1 fused sweeps (default)
        advect_row_ = env_int("SF_ADVECT_ROW", 1);
        zero_skip_ = env_int("SF_ZERO_SKIP", 1) != 0;
        split_enabled_ = env_int("SF_SPLIT", 1) != 0;
        ovl_mode_ = env_int("SF_OVL", 1);
        split_fields_ = env_int("SF_SPLIT_FIELDS", 1);  // 0 never, 1 when one field fits the Infinity Cache, 2 always
        fuse_src_ = env_int("SF_FUSE_SRC", 1) != 0;  // fold add_source (bound sources) into diffuse's first pass
        // Trapezoid blocks shorten the interior chain (no cross-stream wait) but lengthen the boundary chain
        // B(j) -> halo(j) -> B(j+1), because B grows by two planes per side and pair. With halos that are copies on
        // this GPU the interior chain is the critical one (default 5 pairs per block); with RCCL messages over xGMI
        // (2.4 MB per direction and pair at 512^2: tens of microseconds) the boundary chain is, so the default there
        // keeps B at its minimum size (0 = off) unless the measurement at the end of this constructor
        // (tune_schedule) says otherwise. SF_TRAP overrides and switches the measurement off.
        trap_m_ = env_int("SF_TRAP", (nranks_ > 1 || rccl_self_) ? 0 : 5);  // pairs per trapezoid block of a decomposed lin_solve (<= 1: off)
        graphs_ = env_int("SF_GRAPH", 0) != 0 && P_ == 1;
        march_k_ = env_int("SF_MARCH", 1);  // 0: the register-blocked pair kernel everywhere
        march_min_planes_ = env_int("SF_MARCH_MINP", 12);
        // Smallest launch the marching kernel takes: 2.5 M cells (~136^3; 6 M until round 3) — with 16 thin waves per
        // workgroup it overtakes the pair kernel there (us per sweep of a 20-sweep solve, pair / marching: 128^3 4.4 /
        // 5.8, 144^3 9.0 / 6.0, 160^3 10.6 / 6.3, 176^3 13.4 / 7.1), slab interiors included (one rank's share of the
        // full step: 256^3 over 4 ranks 1.335 -> 1.147 ms, 384^3 over 8 ranks 1.806 -> 1.492).
        march_min_cells_ = (long)env_int("SF_MARCH_MINCELLS_K", 2500) * 1000L;
        sk2_min_cells_ = std::max(march_min_cells_ == 0 ? 0L : 60000000L, march_min_cells_);  // ~390^3
        sk_s_ = env_int("SF_SK_S", 4);
        sk_first_ = env_int("SF_SK_FIRST", 1) != 0;  // first pass of a solve through the marching kernel
        SF_HIP(hipDeviceSynchronize());
        if ((nranks_ > 1 || rccl_self_) && std::getenv("SF_TRAP") == nullptr && env_int("SF_AUTOTUNE", 1)) tune_schedule();
        trace_open();
    }

    // Which trapezoid depth suits THIS machine's halo latency (see the comment at trap_m_)? Times a 20-sweep
    // lin_solve on the (still zero) density slots for 0, 2 and 5 pairs per block and keeps the fastest, preferring
    // the shallower one unless the deeper is 3 % faster. Every rank runs the same sequence of exchanges whatever it
    // picks (the depth only moves planes between this rank's own two launches).
    void tune_schedule() {
        if (!(G_ >= 2 && can_fuse2()) || nzl_ <= 2 * (G_ + 2) + 2) return;
        const int x[1] = {SF_DENS}, x0[1] = {SF_DENS0}, b0[1] = {0};
        const T a = T(0.25), c = T(1) + T(6) * a;
        auto drain = [&] {
            join();
            for (Slab& sl : slabs_) {
                SF_HIP(hipStreamSynchronize(sl.cs));
                SF_HIP(hipStreamSynchronize(sl.bs));
                SF_HIP(hipStreamSynchronize(sl.hs));
            }
        };
        // The ranks measure TOGETHER: a one-word all-reduce lines them up before the clock starts (a rank that starts
        // its solves early would time its neighbours' set-up), and the time that counts is the slowest rank's — the
        // step runs at that pace — so every rank sees the same numbers and takes the same decision. (Round 2 timed each
        // rank by itself.) d_sync[0]: barrier word, d_sync[1]: the time in microseconds.
        long long* d_sync = nullptr;
        if (comm_) {
            SF_HIP(hipMalloc(&d_sync, 2 * sizeof(long long)));
            SF_HIP(hipMemset(d_sync, 0, 2 * sizeof(long long)));
            SF_HIP(hipDeviceSynchronize());
        }
        auto line_up = [&] {
            if (!comm_) return;
            SF_NCCL(ncclAllReduce(d_sync, d_sync, 1, ncclInt64, ncclSum, comm_, slabs_[0].cs));
            SF_HIP(hipStreamSynchronize(slabs_[0].cs));
        };
        auto slowest = [&](double t) -> double {
            if (!comm_) return t;
            long long us = (long long)(t * 1e6);
            SF_HIP(hipMemcpy(d_sync + 1, &us, sizeof us, hipMemcpyHostToDevice));
            SF_NCCL(ncclAllReduce(d_sync + 1, d_sync + 1, 1, ncclInt64, ncclMax, comm_, slabs_[0].cs));
            SF_HIP(hipStreamSynchronize(slabs_[0].cs));
            SF_HIP(hipMemcpy(&us, d_sync + 1, sizeof us, hipMemcpyDeviceToHost));
            return (double)us * 1e-6;
        };
        const int cand[3] = {0, 2, 5};
        double best = 0;
        int best_m = 0;
        for (int q = 0; q < 3; ++q) {
            trap_m_ = cand[q];
            op_lin_solve<1>(x, x0, b0, a, c, 10);  // warm-up (first use of the communicator, caches)
            drain();
            line_up();
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 3; ++r) op_lin_solve<1>(x, x0, b0, a, c, 20);
            drain();
            const double t = slowest(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            if (q == 0 || t < 0.97 * best) {
                best = t;
                best_m = cand[q];
            }
        }
        trap_m_ = best_m;
        tuned_trap_ = best_m;
        // Same question for "u, v, w one field at a time" (Infinity-Cache resident, three small messages per pair)
        // against "three fields per launch" (one message of three times the size: the message latency is paid once):
        // the first wins when a pair is compute-bound, the second when the messages are the critical chain.
        const double one = 3.0 * (double)(N_ + 2) * (N_ + 2) * nplanes_ * sizeof(T);
        if (std::getenv("SF_SPLIT_FIELDS") == nullptr && one <= 0.9 * 256.0 * 1048576.0) {
            const int vel[3] = {SF_U, SF_V, SF_W}, vel0[3] = {SF_U0, SF_V0, SF_W0}, b123[3] = {1, 2, 3};
            double t_split = 0;
            for (int q = 0; q < 2; ++q) {
                split_fields_ = q == 0 ? 1 : 0;
                op_lin_solve<3>(vel, vel0, b123, a, c, 4);
                drain();
                line_up();
                const auto t0 = std::chrono::steady_clock::now();
                for (int r = 0; r < 2; ++r) op_lin_solve<3>(vel, vel0, b123, a, c, 20);
                drain();
                const double t = slowest(std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
                if (q == 0)
                    t_split = t;
                else if (!(t < 0.97 * t_split))
                    split_fields_ = 1;
            }
            // unlike the trapezoid depth this changes the sequence of exchanges, so the ranks must agree: they now compare
            // the same (slowest-rank) times, and the vote below stays as the guard that they really did
            if (comm_) {
                int* d_vote = nullptr;
                int vote = split_fields_ == 0 ? 1 : 0, sum = 0;
                SF_HIP(hipMalloc(&d_vote, sizeof(int)));
                SF_HIP(hipMemcpy(d_vote, &vote, sizeof(int), hipMemcpyHostToDevice));
                SF_NCCL(ncclAllReduce(d_vote, d_vote, 1, ncclInt, ncclSum, comm_, slabs_[0].cs));
                SF_HIP(hipStreamSynchronize(slabs_[0].cs));
                SF_HIP(hipMemcpy(&sum, d_vote, sizeof(int), hipMemcpyDeviceToHost));
                SF_HIP(hipFree(d_vote));
                split_fields_ = (sum == nranks_) ? 0 : 1;
            }
            tuned_split_ = split_fields_;
        }
        if (d_sync) SF_HIP(hipFree(d_sync));
    }

    ~Solver() override {
        (void)hipSetDevice(device_);
        (void)hipDeviceSynchronize();
        for (GraphEntry& e : graph_cache_) (void)hipGraphExecDestroy(e.exec);
        if (trace_) std::fclose(trace_);
        if (comm_) ncclCommDestroy(comm_);
        for (Slab& sl : slabs_) {
            for (T*& f : sl.field) free_field(f);
            for (T*& f : sl.scratch) free_field(f);
            if (sl.d_flag) (void)hipFree(sl.d_flag);
            for (T*& f : sl.snap) free_field(f);
            if (sl.os) (void)hipStreamDestroy(sl.os);
            if (sl.snap_done) (void)hipEventDestroy(sl.snap_done);
            if (sl.cs) (void)hipStreamDestroy(sl.cs);
            if (sl.bs) (void)hipStreamDestroy(sl.bs);
            if (sl.cs_mark) (void)hipEventDestroy(sl.cs_mark);
            if (sl.hs && sl.hs != sl.bs) (void)hipStreamDestroy(sl.hs);
            if (sl.boundary_done) (void)hipEventDestroy(sl.boundary_done);
            if (sl.halo_done) (void)hipEventDestroy(sl.halo_done);
        }
        if (t0_) (void)hipEventDestroy(t0_);
        if (t1_) (void)hipEventDestroy(t1_);
        if (tr_pos_) (void)hipFree(tr_pos_);
        if (tr_dens_) (void)hipFree(tr_dens_);
        if (tr_speed_) (void)hipFree(tr_speed_);
        if (copy_src_) (void)hipFree(copy_src_);
        if (copy_dst_) (void)hipFree(copy_dst_);
    }

    // ---- host <-> device ------------------------------------------------------------------
    void upload(int field, const void* host) override {
        join();
        check_field(field);
        SF_REQUIRE(host != nullptr, "null host pointer");
        SF_HIP(hipSetDevice(device_));
        const size_t S = (size_t)N_ + 2;
        for (Slab& sl : slabs_) {
            T* dev = ensure(sl, field);
            // every stored plane that exists globally (ghosts included) comes from the global array
            const int gb = std::max(sl.geom.kg0, 0), ge = std::min(sl.geom.kg0 + nplanes_, N_ + 2);
            const T* src = static_cast<const T*>(host) + (size_t)gb * S * S;
            SF_HIP(hipMemcpy2DAsync(dev + (size_t)(gb - sl.geom.kg0) * plane_ + (lead_ - 1), (size_t)px_ * sizeof(T), src,
                                    S * sizeof(T), S * sizeof(T), S * (size_t)(ge - gb), hipMemcpyHostToDevice, sl.cs));
        }
        for (Slab& sl : slabs_) SF_HIP(hipStreamSynchronize(sl.cs));
    }

    void download(int field, void* host) override {
        join();
        check_field(field);
        SF_REQUIRE(host != nullptr, "null host pointer");
        for (Slab& sl : slabs_) {
            const int kb = sl.geom.kg0 + G_ - (sl.geom.wall_lo ? 1 : 0);
            const int ke = sl.geom.kg0 + G_ + nzl_ + (sl.geom.wall_hi ? 1 : 0);
            const size_t S = (size_t)N_ + 2;
            copy_planes_out(sl, field, kb, ke, static_cast<T*>(host) + (size_t)kb * S * S);
        }
        for (Slab& sl : slabs_) SF_HIP(hipStreamSynchronize(sl.cs));
    }

    void download_planes(int field, int kb, int ke, void* host) override {
        join();
        check_field(field);
        SF_REQUIRE(host != nullptr, "null host pointer");
        SF_REQUIRE(kb < ke, "empty plane range");
        const size_t S = (size_t)N_ + 2;
        bool any = false;
        for (Slab& sl : slabs_) {
            // planes of [kb,ke) this slab is the owner of (interior; shell planes on wall slabs)
            const int ob = sl.geom.kg0 + G_ - (sl.geom.wall_lo ? 1 : 0);
            const int oe = sl.geom.kg0 + G_ + nzl_ + (sl.geom.wall_hi ? 1 : 0);
            const int b = std::max(kb, ob), e = std::min(ke, oe);
            if (b >= e) continue;
            any = true;
            copy_planes_out(sl, field, b, e, static_cast<T*>(host) + (size_t)(b - kb) * S * S);
        }
        SF_REQUIRE(any, "plane range not stored by this context");
        for (Slab& sl : slabs_) SF_HIP(hipStreamSynchronize(sl.cs));
    }

    void upload_planes(int field, int kb, int ke, const void* host) override {
        join();
        check_field(field);
        SF_REQUIRE(host != nullptr, "null host pointer");
        SF_REQUIRE(kb < ke, "empty plane range");
        SF_HIP(hipSetDevice(device_));
        const size_t S = (size_t)N_ + 2;
        bool any = false;
        for (Slab& sl : slabs_) {
            T* dev = ensure(sl, field);
            const int b = std::max(std::max(kb, sl.geom.kg0), 0);
            const int e = std::min(std::min(ke, sl.geom.kg0 + nplanes_), N_ + 2);
            if (b >= e) continue;
            any = true;
            const T* src = static_cast<const T*>(host) + (size_t)(b - kb) * S * S;
            SF_HIP(hipMemcpy2DAsync(dev + (size_t)(b - sl.geom.kg0) * plane_ + (lead_ - 1), (size_t)px_ * sizeof(T),
                                    src, S * sizeof(T), S * sizeof(T), S * (size_t)(e - b), hipMemcpyHostToDevice,
                                    sl.cs));
        }
        SF_REQUIRE(any, "plane range not stored by this context");
        for (Slab& sl : slabs_) SF_HIP(hipStreamSynchronize(sl.cs));
    }

    void stored_planes(int* kb, int* ke) const override {
        if (kb) *kb = std::max(slabs_.front().geom.kg0, 0);
        if (ke) *ke = std::min(slabs_.back().geom.kg0 + nplanes_, N_ + 2);
    }

    void owned_planes(int* kb, int* ke) const override {
        if (kb) *kb = slabs_.front().geom.kg0 + G_;
        if (ke) *ke = slabs_.back().geom.kg0 + G_ + nzl_;
    }

    void fill(int field, double value) override {
        join();
        check_field(field);
        SF_HIP(hipSetDevice(device_));
        for (Slab& sl : slabs_) {
            T* dev = ensure(sl, field);
            const long nvec = field_elems_ / W;
            hipLaunchKernelGGL((sfk::fill_kernel<T>), dim3(stream_grid(nvec)), dim3(256), 0, sl.cs, dev,
                               (T)value, nvec);
        }
        SF_HIP(hipGetLastError());
    }

    void copy_field(int dst, int src) override {
        join();
        check_field(dst);
        check_field(src);
        SF_REQUIRE(dst != src, "copy_field: dst == src");
        SF_HIP(hipSetDevice(device_));
        for (Slab& sl : slabs_) {
            T* d = ensure(sl, dst);
            T* s = ensure(sl, src);
            SF_HIP(hipMemcpyAsync(d, s, (size_t)field_elems_ * sizeof(T), hipMemcpyDeviceToDevice, sl.cs));
        }
    }

    // ---- operators ------------------------------------------------------------------------
    void add_source(int x, int s) override {
        check_field(x);
        check_field(s);
        SF_HIP(hipSetDevice(device_));
        const int xs[1] = {x}, ss[1] = {s};
        op_add_source<1>(xs, ss);
    }

    void set_bnd(int b, int x) override {
        join();
        check_field(x);
        check_b(b);
        SF_HIP(hipSetDevice(device_));
        for (Slab& sl : slabs_) {
            T* dev = ensure(sl, x);
            const long n0 = std::max((long)N_ * nzl_, (long)N_ * N_);
            const long n1 = std::max(N_, nzl_);
            hipLaunchKernelGGL((sfk::set_bnd_kernel<T>), dim3((unsigned)((n0 + 255) / 256)), dim3(256), 0,
                               sl.cs, sl.geom, dev, b, 0);
            hipLaunchKernelGGL((sfk::set_bnd_kernel<T>), dim3((unsigned)((n1 + 255) / 256)), dim3(256), 0,
                               sl.cs, sl.geom, dev, b, 1);
            hipLaunchKernelGGL((sfk::set_bnd_kernel<T>), dim3(1), dim3(64), 0, sl.cs, sl.geom, dev, b, 2);
        }
        SF_HIP(hipGetLastError());
        if (P_ > 1) {
            for (Slab& sl : slabs_) ev_record(sl, &Slab::boundary_done, sl.cs);
            const int fs[1] = {x};
            exchange<1>(fs);
        }
    }

    void lin_solve(int b, int x, int x0, double a, double c, int iters) override {
        check_field(x);
        check_field(x0);
        check_b(b);
        SF_REQUIRE(x != x0, "lin_solve: x and x0 must be different fields");
        SF_REQUIRE(iters >= 0, "iters must be >= 0");
        SF_HIP(hipSetDevice(device_));
        const int xs[1] = {x}, x0s[1] = {x0}, bs[1] = {b};
        op_lin_solve<1>(xs, x0s, bs, (T)a, (T)c, iters);
    }

    void diffuse(int b, int x, int x0, double diff) override {
        check_field(x);
        check_field(x0);
        check_b(b);
        SF_REQUIRE(x != x0, "diffuse: x and x0 must be different fields");
        SF_HIP(hipSetDevice(device_));
        const int xs[1] = {x}, x0s[1] = {x0}, bs[1] = {b};
        const T a = diffusion_a((T)diff);
        op_lin_solve<1>(xs, x0s, bs, a, T(1) + T(6) * a, K_);
    }

    void advect(int b, int d, int d0, int u, int v, int w) override {
        check_field(d);
        check_field(d0);
        check_field(u);
        check_field(v);
        check_field(w);
        check_b(b);
        SF_REQUIRE(d != d0 && d != u && d != v && d != w, "advect: output must not alias an input");
        SF_HIP(hipSetDevice(device_));
        const int ds[1] = {d}, d0s[1] = {d0}, bs[1] = {b};
        op_advect<1>(ds, d0s, bs, u, v, w);
    }

    void project(int u, int v, int w, int p, int div) override {
        const int all[5] = {u, v, w, p, div};
        for (int a = 0; a < 5; ++a) {
            check_field(all[a]);
            for (int c = a + 1; c < 5; ++c) SF_REQUIRE(all[a] != all[c], "project: fields must be distinct");
        }
        SF_HIP(hipSetDevice(device_));
        op_project(u, v, w, p, div);
    }

    void bind_sources(int su, int sv, int sw, int sd) override {
        const int b[4] = {su, sv, sw, sd};
        const int own[4] = {SF_U0, SF_V0, SF_W0, SF_DENS0};
        for (int q = 0; q < 4; ++q) {
            SF_REQUIRE(b[q] >= -1 && b[q] < SF_NUM_FIELDS, "bind_sources: slot out of range");
            SF_REQUIRE(b[q] < 0 || b[q] >= SF_USER0, "bind_sources: sources must live in SF_USER0..3");
            (void)own;
        }
        for (int q = 0; q < 4; ++q) bound_[q] = b[q];
    }

    // SF_GRAPH=1 (opt-in): the ~55 launches of a step are captured once into a hipGraph per distinct buffer arrangement
    // and replayed, so the host issues one graph launch instead of one launch per kernel. Single-slab contexts only.
    // Off by default: measured on MI355X it changes nothing (32^3: 0.31 ms/step, 64^3: 0.35, 128^3: 0.63 either
    // way) — small grids are bound by the ~5 us dependent-kernel boundary on the device, not by host launches.
    struct GraphEntry {
        int op;
        std::vector<T*> before, after;
        int K;
        T dt, diff, visc;
        int bound[4];
        hipGraphExec_t exec;
    };
    std::vector<T*> pointer_state() const {
        std::vector<T*> st;
        const Slab& sl = slabs_[0];
        for (T* f : sl.field) st.push_back(f);
        for (T* f : sl.scratch) st.push_back(f);
        return st;
    }
    void apply_state(const std::vector<T*>& st) {
        Slab& sl = slabs_[0];
        size_t q = 0;
        for (T*& f : sl.field) f = st[q++];
        for (T*& f : sl.scratch) f = st[q++];
    }
    template <class Body>
    void run_maybe_graphed(int op, Body body) {
        if (!graphs_ || P_ != 1) {
            body();
            return;
        }
        Slab& sl = slabs_[0];
        for (int q = 0; q < 4; ++q)
            if (bound_[q] >= 0) ensure(sl, bound_[q]);  // no allocation may happen inside a capture
        const std::vector<T*> before = pointer_state();
        for (GraphEntry& e : graph_cache_)
            if (e.op == op && e.K == K_ && e.dt == dt_ && e.diff == diff_ && e.visc == visc_ &&
                std::equal(e.bound, e.bound + 4, bound_) && e.before == before) {
                SF_HIP(hipGraphLaunch(e.exec, sl.cs));
                apply_state(e.after);
                return;
            }
        hipGraph_t graph = nullptr;
        SF_HIP(hipStreamBeginCapture(sl.cs, hipStreamCaptureModeThreadLocal));
        try {
            body();
        } catch (...) {
            (void)hipStreamEndCapture(sl.cs, &graph);
            if (graph) (void)hipGraphDestroy(graph);
            apply_state(before);
            throw;
        }
        SF_HIP(hipStreamEndCapture(sl.cs, &graph));
        GraphEntry e;
        e.op = op;
        e.before = before;
        e.after = pointer_state();
        e.K = K_;
        e.dt = dt_;
        e.diff = diff_;
        e.visc = visc_;
        std::copy(bound_, bound_ + 4, e.bound);
        SF_HIP(hipGraphInstantiate(&e.exec, graph, nullptr, nullptr, 0));
        SF_HIP(hipGraphDestroy(graph));
        graph_cache_.push_back(e);
        SF_HIP(hipGraphLaunch(e.exec, sl.cs));
    }

    void vel_step() override {
        SF_HIP(hipSetDevice(device_));
        run_maybe_graphed(0, [&] { vel_step_body(); });
    }
    void dens_step() override {
        SF_HIP(hipSetDevice(device_));
        run_maybe_graphed(1, [&] { dens_step_body(); });
    }

    // SPEC §3 vel_step.
    void vel_step_body() {
        const int vel[3] = {SF_U, SF_V, SF_W}, vel0[3] = {SF_U0, SF_V0, SF_W0}, b123[3] = {1, 2, 3};
        const T a = diffusion_a(visc_);
        // Nobody reads the i-shell of the diffused velocity: project_div reads v and w at interior i only and mirrors u's
        // (b = 1), project_sub then rewrites all three shells from the corrected interior. Likewise the pressure of this
        // FIRST projection: its slot is overwritten by advect before anything else looks at it. Their solves therefore
        // leave the i-shell unwritten (no partial writes to HBM: a 512^3 last pass takes 441 instead of 520 us). K_ = 0
        // runs no sweep: the fields keep their caller-written shells and are read from memory as before.
        const bool dead = ishell_skip_ && K_ >= 1 && dead_ishell_opt_;
        dead_ishell_ = dead;
        if (bound_[0] >= 0 && bound_[1] >= 0 && bound_[2] >= 0) {
            const int src[3] = {bound_[0], bound_[1], bound_[2]};
            op_diffuse_src<3>(vel, vel0, b123, src, a, T(1) + T(6) * a, K_);
        } else {
            for (int q = 0; q < 3; ++q)
                if (bound_[q] >= 0) copy_field(vel0[q], bound_[q]);
            op_add_source<3>(vel, vel0);
            swap_slots(SF_U0, SF_U);
            swap_slots(SF_V0, SF_V);
            swap_slots(SF_W0, SF_W);
            op_lin_solve<3>(vel, vel0, b123, a, T(1) + T(6) * a, K_);
        }
        dead_ishell_ = false;
        op_project(SF_U, SF_V, SF_W, SF_U0, SF_V0, dead, dead);
        swap_slots(SF_U0, SF_U);
        swap_slots(SF_V0, SF_V);
        swap_slots(SF_W0, SF_W);
        // the advected velocity goes straight into the second projection: same argument as for the diffused one
        dead_ishell_ = dead;
        op_advect<3>(vel, vel0, b123, SF_U0, SF_V0, SF_W0);
        dead_ishell_ = false;
        op_project(SF_U, SF_V, SF_W, SF_U0, SF_V0, dead, false);
    }

    // SPEC §3 dens_step.
    void dens_step_body() {
        const int x[1] = {SF_DENS}, x0[1] = {SF_DENS0}, b0[1] = {0};
        const T a = diffusion_a(diff_);
        if (bound_[3] >= 0) {
            const int src[1] = {bound_[3]};
            op_diffuse_src<1>(x, x0, b0, src, a, T(1) + T(6) * a, K_);
        } else {
            op_add_source<1>(x, x0);
            swap_slots(SF_DENS0, SF_DENS);
            op_lin_solve<1>(x, x0, b0, a, T(1) + T(6) * a, K_);
        }
        swap_slots(SF_DENS0, SF_DENS);
        op_advect<1>(x, x0, b0, SF_U, SF_V, SF_W);
    }

    void set_iters(int iters) override {
        SF_REQUIRE(iters >= 0, "iters must be >= 0");
        K_ = iters;
    }
    void set_coefficients(double dt, double diff, double visc) override {
        dt_ = (T)dt;
        diff_ = (T)diff;
        visc_ = (T)visc;
    }

    void sync() override {
        join();
        SF_HIP(hipSetDevice(device_));
        for (Slab& sl : slabs_) {
            SF_HIP(hipStreamSynchronize(sl.cs));
            SF_HIP(hipStreamSynchronize(sl.bs));
            SF_HIP(hipStreamSynchronize(sl.hs));
        }
        if (P_ > 1) {
            bool exceeded = false;
            for (Slab& sl : slabs_) {
                int flag = 0;
                SF_HIP(hipMemcpy(&flag, sl.d_flag, sizeof(int), hipMemcpyDeviceToHost));
                if (flag) {
                    exceeded = true;
                    SF_HIP(hipMemset(sl.d_flag, 0, sizeof(int)));
                    SF_HIP(hipDeviceSynchronize());
                }
            }
            if (exceeded)
                throw Failure{SF_ERR_HALO_EXCEEDED,
                              "advect back-traced more than one plane across a slab boundary "
                              "(|dt*N*w| >= 1): results differ from the undecomposed solve"};
        }
    }

    void timer_start() override {
        join();
        SF_HIP(hipSetDevice(device_));
        SF_HIP(hipEventRecord(t0_, slabs_[0].cs));
    }
    float timer_stop() override {
        join();
        SF_HIP(hipEventRecord(t1_, slabs_[0].cs));
        SF_HIP(hipEventSynchronize(t1_));
        float ms = 0.f;
        SF_HIP(hipEventElapsedTime(&ms, t0_, t1_));
        return ms;
    }

    double copy_bandwidth(size_t bytes, int reps) override {
        join();
        SF_HIP(hipSetDevice(device_));
        bytes = (bytes + 4095) / 4096 * 4096;
        if (copy_bytes_ != bytes) {
            if (copy_src_) (void)hipFree(copy_src_);
            if (copy_dst_) (void)hipFree(copy_dst_);
            copy_src_ = copy_dst_ = nullptr;
            copy_bytes_ = 0;
            SF_HIP(hipMalloc(&copy_src_, bytes));
            SF_HIP(hipMalloc(&copy_dst_, bytes));
            SF_HIP(hipMemset(copy_src_, 1, bytes));
            SF_HIP(hipMemset(copy_dst_, 0, bytes));
            SF_HIP(hipDeviceSynchronize());
            copy_bytes_ = bytes;
        }
        const long n = (long)(bytes / 16);
        hipStream_t st = slabs_[0].cs;
        double best_ms = 1e30;
        for (int r = 0; r <= reps; ++r) {
            SF_HIP(hipEventRecord(t0_, st));
            // one thread per 16 bytes, the grid in memory order: the shape that reaches 6.2-6.3 TB/s here
            // (tools/membench.hip); a grid-stride loop over a few thousand blocks stays at 5.0-5.9
            hipLaunchKernelGGL(sfk::copy16_kernel, dim3((unsigned)ceil_div(n, 256L)), dim3(256), 0, st,
                               (const float4*)copy_src_, (float4*)copy_dst_, n);
            SF_HIP(hipEventRecord(t1_, st));
            SF_HIP(hipEventSynchronize(t1_));
            float ms = 0.f;
            SF_HIP(hipEventElapsedTime(&ms, t0_, t1_));
            if (r > 0 && ms < best_ms) best_ms = ms;
        }
        return 2.0 * (double)bytes / (best_ms * 1e-3) / 1e9;
    }

    // ---- asynchronous output -----------------------------------------------------------------
    void snapshot(const int* fields, int nfields) override {
        join();
        SF_REQUIRE(fields != nullptr && nfields >= 1 && nfields <= 4, "snapshot takes 1..4 fields");
        SF_HIP(hipSetDevice(device_));
        for (int q = 0; q < nfields; ++q) check_field(fields[q]);
        for (Slab& sl : slabs_) {
            if (!sl.os) SF_HIP(hipStreamCreateWithFlags(&sl.os, hipStreamNonBlocking));
            if (!sl.snap_done) SF_HIP(hipEventCreateWithFlags(&sl.snap_done, hipEventDisableTiming));
            for (int q = 0; q < nfields; ++q) {
                if (!sl.snap[q]) sl.snap[q] = alloc_field();
                SF_HIP(hipMemcpyAsync(sl.snap[q], ensure(sl, fields[q]), (size_t)field_elems_ * sizeof(T),
                                      hipMemcpyDeviceToDevice, sl.cs));
            }
            SF_HIP(hipEventRecord(sl.snap_done, sl.cs));
        }
        snap_count_ = nfields;
    }

    // May run on another host thread: touches only the snapshot buffers, the output stream and snap_done.
    void snapshot_read(int index, void* host) override {
        SF_REQUIRE(host != nullptr, "null host pointer");
        // the planes this context is the owner of (shell planes on the end slabs), at their place in the GLOBAL array
        const int kb = slabs_.front().geom.kg0 + G_ - (slabs_.front().geom.wall_lo ? 1 : 0);
        const int ke = slabs_.back().geom.kg0 + G_ + nzl_ + (slabs_.back().geom.wall_hi ? 1 : 0);
        const size_t S = (size_t)N_ + 2;
        snapshot_read_planes(index, kb, ke, static_cast<T*>(host) + (size_t)kb * S * S);
    }

    // Planes [kb, ke) of snapshot `index` into a host array that holds exactly those planes (dense (N+2)^2 each).
    void snapshot_read_planes(int index, int kb, int ke, void* host) override {
        SF_REQUIRE(index >= 0 && index < snap_count_, "snapshot index out of range");
        SF_REQUIRE(host != nullptr, "null host pointer");
        SF_REQUIRE(kb < ke, "empty plane range");
        SF_HIP(hipSetDevice(device_));
        const size_t S = (size_t)N_ + 2;
        bool any = false;
        for (Slab& sl : slabs_) {
            SF_HIP(hipStreamWaitEvent(sl.os, sl.snap_done, 0));
            const int ob = sl.geom.kg0 + G_ - (sl.geom.wall_lo ? 1 : 0);
            const int oe = sl.geom.kg0 + G_ + nzl_ + (sl.geom.wall_hi ? 1 : 0);
            const int b = std::max(kb, ob), e = std::min(ke, oe);
            if (b >= e) continue;
            any = true;
            SF_HIP(hipMemcpy2DAsync(static_cast<T*>(host) + (size_t)(b - kb) * S * S, S * sizeof(T),
                                    sl.snap[index] + (size_t)(b - sl.geom.kg0) * plane_ + (lead_ - 1),
                                    (size_t)px_ * sizeof(T), S * sizeof(T), S * (size_t)(e - b), hipMemcpyDeviceToHost,
                                    sl.os));
        }
        SF_REQUIRE(any, "plane range not stored by this context");
        for (Slab& sl : slabs_) SF_HIP(hipStreamSynchronize(sl.os));
    }

    // ---- tracers (SPEC §6) ----------------------------------------------------------------------
    void tracers_set(int n, const void* xyz) override {
        join();
        SF_REQUIRE(P_ == 1, "tracers need a single-slab context");
        SF_REQUIRE(n >= 0 && (n == 0 || xyz != nullptr), "bad tracer array");
        SF_HIP(hipSetDevice(device_));
        if (tr_pos_) (void)hipFree(tr_pos_);
        if (tr_dens_) (void)hipFree(tr_dens_);
        if (tr_speed_) (void)hipFree(tr_speed_);
        tr_pos_ = tr_dens_ = tr_speed_ = nullptr;
        tr_n_ = n;
        if (n == 0) return;
        SF_HIP(hipMalloc(&tr_pos_, (size_t)3 * n * sizeof(T)));
        SF_HIP(hipMalloc(&tr_dens_, (size_t)n * sizeof(T)));
        SF_HIP(hipMalloc(&tr_speed_, (size_t)n * sizeof(T)));
        SF_HIP(hipMemcpyAsync(tr_pos_, xyz, (size_t)3 * n * sizeof(T), hipMemcpyHostToDevice, slabs_[0].cs));
        SF_HIP(hipStreamSynchronize(slabs_[0].cs));
    }
    void tracers_advect() override {
        join();
        SF_REQUIRE(P_ == 1, "tracers need a single-slab context");
        if (tr_n_ == 0) return;
        SF_HIP(hipSetDevice(device_));
        Slab& sl = slabs_[0];
        hipLaunchKernelGGL((sfk::tracers_advect_kernel<T>), dim3((unsigned)ceil_div(tr_n_, 256)), dim3(256), 0, sl.cs,
                           sl.geom, tr_n_, tr_pos_, sl.field[SF_U], sl.field[SF_V], sl.field[SF_W], dt_ * (T)N_);
        SF_HIP(hipGetLastError());
    }
    void tracers_get(void* xyz, void* dens, void* speed) override {
        join();
        SF_REQUIRE(P_ == 1, "tracers need a single-slab context");
        if (tr_n_ == 0) return;
        SF_HIP(hipSetDevice(device_));
        Slab& sl = slabs_[0];
        if (dens || speed) {
            hipLaunchKernelGGL((sfk::tracers_sample_kernel<T>), dim3((unsigned)ceil_div(tr_n_, 256)), dim3(256), 0,
                               sl.cs, sl.geom, tr_n_, tr_pos_, sl.field[SF_DENS], sl.field[SF_U], sl.field[SF_V],
                               sl.field[SF_W], tr_dens_, tr_speed_);
            SF_HIP(hipGetLastError());
        }
        if (xyz) SF_HIP(hipMemcpyAsync(xyz, tr_pos_, (size_t)3 * tr_n_ * sizeof(T), hipMemcpyDeviceToHost, sl.cs));
        if (dens) SF_HIP(hipMemcpyAsync(dens, tr_dens_, (size_t)tr_n_ * sizeof(T), hipMemcpyDeviceToHost, sl.cs));
        if (speed) SF_HIP(hipMemcpyAsync(speed, tr_speed_, (size_t)tr_n_ * sizeof(T), hipMemcpyDeviceToHost, sl.cs));
        SF_HIP(hipStreamSynchronize(sl.cs));
    }

    int lin_solve_launches(int iters) const override {
        int n = 0;
        for (int it = 0; it < iters; ++n) it += sweeps_in_launch(it, iters, false);
        return n;
    }

    void schedule_info(int* trap, int* measured) const override {
        if (trap) *trap = trap_m_ > 1 ? trap_m_ : 0;
        if (measured) *measured = (tuned_trap_ >= 0 ? 1 : 0) | (tuned_split_ >= 0 ? 2 : 0) | (split_fields_ == 0 ? 4 : 0);
    }
    void transport_info(int* transport, long* groups) const override {
        if (transport)
            *transport = P_ == 1 ? 0 : (rccl_self_ ? 3 : (loopback_ ? 4 : (nranks_ > 1 ? 2 : 1)));
        if (groups) *groups = rccl_groups_;
    }
    void layout_info(int* pitch, int* planes, size_t* bytes) const override {
        if (pitch) *pitch = px_;
        if (planes) *planes = nplanes_;
        if (bytes) *bytes = (size_t)field_elems_ * sizeof(T);
    }

private:
    // ---- helpers --------------------------------------------------------------------------
    static void check_field(int f) { SF_REQUIRE(f >= 0 && f < SF_NUM_FIELDS, "field id out of range"); }
    static void check_b(int b) { SF_REQUIRE(b >= 0 && b <= 3, "boundary mode b must be 0..3"); }

    T* alloc_field() {
        T* p = nullptr;
        const size_t total = (size_t)(pad_front_ + field_elems_ + pad_back_) * sizeof(T);
        SF_HIP(hipMalloc(&p, total));
        SF_HIP(hipMemset(p, 0, total));
        // hipMemset on device memory may return before the fill has run, and the context's streams are
        // non-blocking (they do not order against the null stream): wait here.
        SF_HIP(hipDeviceSynchronize());
        return p + pad_front_;
    }
    void free_field(T* f) const {
        if (f) (void)hipFree(f - pad_front_);
    }
    T* ensure(Slab& sl, int f) {
        if (!sl.field[f]) {
            // allocation is synchronous with respect to the device; fine for the lazily created user slots
            sl.field[f] = alloc_field();
        }
        return sl.field[f];
    }
    void swap_slots(int a, int b) {
        for (Slab& sl : slabs_) std::swap(sl.field[a], sl.field[b]);
    }
    T diffusion_a(T coeff) const {
        const T Nf = (T)N_;
        return ((dt_ * coeff) * Nf) * Nf;
    }
    unsigned stream_grid(long nvec) const {
        long blocks = (nvec + 255) / 256;
        const long cap = (long)num_cu_ * 8;
        return (unsigned)std::max(1L, std::min(blocks, cap));
    }

    void copy_planes_out(Slab& sl, int field, int kb, int ke, T* dst) {
        SF_HIP(hipSetDevice(device_));
        T* dev = ensure(sl, field);
        const size_t S = (size_t)N_ + 2;
        const int klb = kb - sl.geom.kg0;
        SF_REQUIRE(klb >= 0 && ke - sl.geom.kg0 <= nplanes_, "plane range outside slab");
        SF_HIP(hipMemcpy2DAsync(dst, S * sizeof(T), dev + (size_t)klb * plane_ + (lead_ - 1),
                                (size_t)px_ * sizeof(T), S * sizeof(T), S * (size_t)(ke - kb),
                                hipMemcpyDeviceToHost, sl.cs));
    }

    // 1-D banded grid for the one-vector-per-thread kernels: fills block, returns the map and block count.
    sfk::TileMap flat_map(int nplanes, dim3& block, unsigned& nblocks) const {
        // these kernels take their i+-1 values by loads, so a row tile may have any width: one tile per row up
        // to 256 vectors (no idle lanes for N = 324, 408, ...), 64-lane tiles beyond
        const int nvec = ceil_div(N_, W);
        const int tx = nvec <= 256 ? nvec : 64;
        const int ty = std::max(1, 256 / tx);
        block = dim3(tx, ty, 1);
        sfk::TileMap m{};
        m.rows = 0;
        m.gx = ceil_div(nvec, tx);
        m.gy = ceil_div(N_, ty);
        m.nxcd = 8;
        m.band = (m.gy >= 16) ? ceil_div(m.gy, 8) : 0;
        m.ishell_mem = 1;
        m.ishell_write = 1;
        m.split = split_;
        m.gap = gap_;
        const long per_plane = m.band > 0 ? (long)m.nxcd * m.gx * m.band : (long)m.gx * m.gy;
        nblocks = (unsigned)(per_plane * nplanes);
        return m;
    }

    // Runs `launch(slab, kb, ke)` over the interior planes of every slab. With P > 1 the two
    // slab-boundary planes go first, their completion is recorded, and the rest follows so that the
    // halo exchange issued by the caller overlaps the interior work.
    // Streams of a decomposed grid (P > 1). Per operator and slab:
    //   bs: boundary launch B (first / last G interior planes; needs the previous halo and everything issued so far)
    //   cs: interior launch I (needs the previous B and I, never a halo — its stencil stays inside the slab)
    //   hs: halo exchange of B's planes, concurrent with I
    // so a pair costs max(I, B + exchange) instead of B + max(I, exchange). Whole-field operators run on cs after
    // join(), which makes cs wait for the last B and the last halo.
    // ---- schedule trace (SF_TRACE_SCHEDULE=<file>) -------------------------------------------------------------
    // Every launch, exchange and stream-ordering call of the decomposed step is appended to <file> as one JSON line:
    //   {"t":"ctx", ...}                                   context geometry (first line of a context)
    //   {"t":"op","name":..,"slab":g,"stream":"cs|bs|hs","acc":[["r"|"w",buffer,lo,hi],...]}   plane ranges [lo,hi)
    //   {"t":"rec","slab":g,"stream":..,"ev":..}           hipEventRecord
    //   {"t":"wait","slab":g,"stream":..,"ev":..,"evslab":h}   hipStreamWaitEvent (on the event's latest record)
    //   {"t":"xchg","seq":n,"fields":[..],"G":G}           one halo exchange (sequence number: the same on every rank)
    // Reads carry the stencil reach of the kernel (an S-sweep launch reads x on S planes either side). tests/
    // schedule_check.py rebuilds the happens-before relation (stream order + event edges) and asserts that no two
    // accesses to overlapping planes of one buffer, one of them a write, are unordered — the write-after-read race of
    // round 2 (DESIGN §4 log) is such a pair. ",inject=trap" after the file name re-introduces that bug (growth S_j
    // instead of max(S_j, S_{j-1})) so the checker can be shown to catch it; results may then be wrong by design.
    struct Acc {
        const void* buf;
        bool write;
        int lo, hi;
    };
    using AccFn = std::function<void(Slab&, int, int, std::vector<Acc>&)>;
    void trace_open() {
        const char* t = std::getenv("SF_TRACE_SCHEDULE");
        if (!t || !*t) return;
        std::string path(t);
        const size_t c = path.find(',');
        if (c != std::string::npos) {
            inject_trap_bug_ = path.substr(c + 1) == "inject=trap";
            path.resize(c);
        }
        trace_ = std::fopen(path.c_str(), "a");
        if (!trace_) throw Failure{SF_ERR_INVALID, "SF_TRACE_SCHEDULE: cannot open " + path};
        std::fprintf(trace_, "{\"t\":\"ctx\",\"N\":%d,\"P\":%d,\"L\":%d,\"rank\":%d,\"G\":%d,\"nzl\":%d,\"np\":%d,\"trap\":%d,"
                             "\"hs_is_bs\":%d,\"inject\":%d}\n",
                     N_, P_, L_, rank_, G_, nzl_, nplanes_, trap_m_, slabs_[0].hs == slabs_[0].bs ? 1 : 0,
                     inject_trap_bug_ ? 1 : 0);
    }
    const char* sname(const Slab& sl, hipStream_t st) const { return st == sl.cs ? "cs" : (st == sl.bs ? "bs" : "hs"); }
    static const char* ename(hipEvent_t Slab::*e) {
        return e == &Slab::cs_mark ? "cs_mark" : (e == &Slab::boundary_done ? "boundary" : "halo");
    }
    int buf_id(const void* p) {
        auto it = buf_ids_.find(p);
        if (it == buf_ids_.end()) it = buf_ids_.emplace(p, (int)buf_ids_.size()).first;
        return it->second;
    }
    void ev_record(Slab& sl, hipEvent_t Slab::*e, hipStream_t st) {
        SF_HIP(hipEventRecord(sl.*e, st));
        if (trace_)
            std::fprintf(trace_, "{\"t\":\"rec\",\"slab\":%d,\"stream\":\"%s\",\"ev\":\"%s\"}\n", sl.gid, sname(sl, st), ename(e));
    }
    void st_wait(Slab& wsl, hipStream_t st, Slab& esl, hipEvent_t Slab::*e) {
        SF_HIP(hipStreamWaitEvent(st, esl.*e, 0));
        if (trace_)
            std::fprintf(trace_, "{\"t\":\"wait\",\"slab\":%d,\"stream\":\"%s\",\"ev\":\"%s\",\"evslab\":%d}\n", wsl.gid,
                         sname(wsl, st), ename(e), esl.gid);
    }
    void tr_op(const char* name, const Slab& sl, hipStream_t st, const std::vector<Acc>& acc) {
        if (!trace_) return;
        std::fprintf(trace_, "{\"t\":\"op\",\"name\":\"%s\",\"slab\":%d,\"stream\":\"%s\",\"acc\":[", name, sl.gid, sname(sl, st));
        bool first = true;
        for (const Acc& a : acc) {
            const int lo = std::max(a.lo, 0), hi = std::min(a.hi, nplanes_);
            if (lo >= hi) continue;
            std::fprintf(trace_, "%s[\"%s\",%d,%d,%d]", first ? "" : ",", a.write ? "w" : "r", buf_id(a.buf), lo, hi);
            first = false;
        }
        std::fprintf(trace_, "]}\n");
        std::fflush(trace_);
    }
    // whole-field operator on the compute stream
    void tr_whole(const char* name, const Slab& sl, std::initializer_list<const void*> reads,
                  std::initializer_list<const void*> writes) {
        if (!trace_) return;
        std::vector<Acc> acc;
        for (const void* r : reads) acc.push_back({r, false, 0, nplanes_});
        for (const void* w : writes) acc.push_back({w, true, 0, nplanes_});
        tr_op(name, sl, sl.cs, acc);
    }
    // planes [kb, ke) written by a launch, widened by the physical shell plane a wall slab's launch also writes
    void wr_range(const Slab& sl, int kb, int ke, int& lo, int& hi) const {
        lo = (sl.geom.wall_lo && kb == G_) ? kb - 1 : kb;
        hi = (sl.geom.wall_hi && ke == G_ + nzl_) ? ke + 1 : ke;
    }

    void join() {
        if (P_ == 1 || !pending_join_) return;
        for (int s = 0; s < L_; ++s) {
            Slab& sl = slabs_[s];
            st_wait(sl, sl.cs, sl, &Slab::boundary_done);
            st_wait(sl, sl.cs, sl, &Slab::halo_done);
            if (s > 0) st_wait(sl, sl.cs, slabs_[s - 1], &Slab::halo_done);
            if (s < L_ - 1) st_wait(sl, sl.cs, slabs_[s + 1], &Slab::halo_done);
        }
        pending_join_ = false;
    }

    template <class F>
    void for_planes(F launch, int depth = 1, bool can_split = true, bool interior_reads_ghosts = false) {
        // the exchange that follows ships G_ planes per side, so at least G_ planes per side must come out of
        // the boundary launch (whose completion the halo stream waits for), not out of the interior launch
        depth = std::max(depth, G_);
        const int kb = G_, ke = G_ + nzl_;
        // trace: the accesses of this operator over the plane ranges one launch covers
        auto emit = [&](Slab& sl, hipStream_t st, int a0, int a1, int b0 = 0, int b1 = 0) {
            if (!trace_ || !acc_fn_) return;
            std::vector<Acc> acc;
            acc_fn_(sl, a0, a1, acc);
            if (b1 > b0) acc_fn_(sl, b0, b1, acc);
            tr_op(acc_name_, sl, st, acc);
        };
        if (P_ == 1) {
            slabs_[0].cur = slabs_[0].cs;
            launch(slabs_[0], kb, ke);
            emit(slabs_[0], slabs_[0].cs, kb, ke);
            SF_HIP(hipGetLastError());
            return;
        }
        // trap_extra_ > 0 (lin_solve only): the boundary launch takes that many more planes per side than the last
        // one did, so this interior launch reads nothing a boundary launch wrote since the last resync and the
        // compute stream needs no cross-stream wait (see op_lin_solve)
        const int extra = (can_split && nzl_ > 2 * (depth + trap_extra_) && split_enabled_ && !interior_reads_ghosts)
                              ? trap_extra_ : 0;
        const bool two_streams = can_split && nzl_ > 2 * depth && split_enabled_ && !interior_reads_ghosts;
        if (!two_streams) join();
        depth += extra;
        for (Slab& sl : slabs_) {
            if (!two_streams) {
                sl.cur = sl.cs;
                if (nzl_ <= 2 * depth) {
                    launch(sl, kb, ke);
                    emit(sl, sl.cs, kb, ke);
                    ev_record(sl, &Slab::boundary_done, sl.cs);
                } else {
                    launch(sl, kb, kb + depth);
                    emit(sl, sl.cs, kb, kb + depth);
                    launch(sl, ke - depth, ke);
                    emit(sl, sl.cs, ke - depth, ke);
                    ev_record(sl, &Slab::boundary_done, sl.cs);
                    launch(sl, kb + depth, ke - depth);
                    emit(sl, sl.cs, kb + depth, ke - depth);
                }
                continue;
            }
            // I of this operator reads what the previous B wrote (unless B has grown, see above); B reads everything
            // issued on cs so far
            if (extra == 0) st_wait(sl, sl.cs, sl, &Slab::boundary_done);
            ev_record(sl, &Slab::cs_mark, sl.cs);
            st_wait(sl, sl.bs, sl, &Slab::cs_mark);
            // ONE launch over the first and the last `depth` interior planes (split plane range)
            sl.cur = sl.bs;
            split_ = depth;
            gap_ = nzl_ - 2 * depth;
            launch(sl, kb, kb + 2 * depth);
            emit(sl, sl.bs, kb, kb + depth, ke - depth, ke);
            split_ = INT_MAX;
            gap_ = 0;
            ev_record(sl, &Slab::boundary_done, sl.bs);
            sl.cur = sl.cs;
            launch(sl, kb + depth, ke - depth);
            emit(sl, sl.cs, kb + depth, ke - depth);
        }
        SF_HIP(hipGetLastError());
    }

    // Kernel launch on sl.cur.
    template <class F, class... Args>
    void launch_k(Slab& sl, F kernel, dim3 nblocks, unsigned nthreads, Args... args) {
        hipLaunchKernelGGL(kernel, nblocks, dim3(nthreads), 0, sl.cur, args...);
    }

    // trace of one slab's share of a halo exchange on stream `st`: its low / high ghost planes are written from the last
    // / first interior planes of `lo` / `hi` (the neighbouring slab of this process; with a neighbour in another
    // process — or the loopback stand-in — the planes READ are this slab's own outgoing ones)
    template <int NF>
    void tr_halo(const char* name, Slab& sl, hipStream_t st, const int (&fields)[NF], Slab* lo, Slab* hi, bool lo_remote,
                 bool hi_remote) {
        if (!trace_) return;
        std::vector<Acc> acc;
        for (int f = 0; f < NF; ++f) {
            const T* mine = sl.field[fields[f]];
            if (lo) acc.push_back({lo->field[fields[f]], false, nzl_, nzl_ + G_});
            if (lo_remote) acc.push_back({mine, false, G_, 2 * G_});
            if (lo || lo_remote) acc.push_back({mine, true, 0, G_});
            if (hi) acc.push_back({hi->field[fields[f]], false, G_, 2 * G_});
            if (hi_remote) acc.push_back({mine, false, nzl_, nzl_ + G_});
            if (hi || hi_remote) acc.push_back({mine, true, G_ + nzl_, 2 * G_ + nzl_});
        }
        tr_op(name, sl, st, acc);
    }
    template <int NF>
    void tr_xchg(const int (&fields)[NF]) {
        if (!trace_) return;
        std::fprintf(trace_, "{\"t\":\"xchg\",\"seq\":%ld,\"G\":%d,\"fields\":[", xchg_seq_, G_);
        for (int f = 0; f < NF; ++f) std::fprintf(trace_, "%s%d", f ? "," : "", fields[f]);
        std::fprintf(trace_, "]}\n");
    }

    // Halo exchange of NF fields: first / last interior plane -> neighbour's ghost plane.
    // Must follow for_planes (uses boundary_done). Compute streams wait on the result.
    template <int NF>
    void exchange(const int (&fields)[NF]) {
        if (P_ == 1) return;
        // G_ planes per direction: the first / last G_ interior planes go to the neighbour's ghost planes
        const size_t gcount = (size_t)G_ * plane_;
        const size_t bytes = gcount * sizeof(T);
        const size_t send_lo = (size_t)G_ * plane_, send_hi = (size_t)nzl_ * plane_;
        const size_t recv_lo = 0, recv_hi = (size_t)(G_ + nzl_) * plane_;
        tr_xchg<NF>(fields);
        ++xchg_seq_;
        if (rccl_self_) {
            exchange_rccl_self<NF>(fields, gcount, send_lo, send_hi, recv_lo, recv_hi);
            return;
        }
        for (int s = 0; s < L_; ++s) {
            Slab& sl = slabs_[s];
            st_wait(sl, sl.hs, sl, &Slab::boundary_done);
            const bool has_lo = sl.gid > 0, has_hi = sl.gid < P_ - 1;
            const bool lo_local = has_lo && s > 0, hi_local = has_hi && s < L_ - 1;
            if (lo_local) st_wait(sl, sl.hs, slabs_[s - 1], &Slab::boundary_done);
            if (hi_local) st_wait(sl, sl.hs, slabs_[s + 1], &Slab::boundary_done);
            // pull from neighbours that live in this process: one copy kernel for all fields and both sides
            if (lo_local || hi_local) {
                sfk::HaloCopyArgs H;
                H.nseg = 0;
                H.n16 = (long)(bytes / 16);
                for (int f = 0; f < NF; ++f) {
                    T* mine = sl.field[fields[f]];
                    if (lo_local) {
                        H.src[H.nseg] = reinterpret_cast<const float4*>(slabs_[s - 1].field[fields[f]] + send_hi);
                        H.dst[H.nseg++] = reinterpret_cast<float4*>(mine + recv_lo);
                    }
                    if (hi_local) {
                        H.src[H.nseg] = reinterpret_cast<const float4*>(slabs_[s + 1].field[fields[f]] + send_lo);
                        H.dst[H.nseg++] = reinterpret_cast<float4*>(mine + recv_hi);
                    }
                }
                const unsigned gx = (unsigned)std::max(1L, std::min((H.n16 + 255) / 256, 512L));
                hipLaunchKernelGGL(sfk::halo_copy_kernel, dim3(gx, H.nseg), dim3(256), 0, sl.hs, H);
                SF_HIP(hipGetLastError());
                tr_halo<NF>("halo_pull", sl, sl.hs, fields, lo_local ? &slabs_[s - 1] : nullptr,
                            hi_local ? &slabs_[s + 1] : nullptr, false, false);
            }
            // neighbours in other processes: grouped send/recv over RCCL (xGMI point-to-point)
            const bool lo_remote = has_lo && !lo_local, hi_remote = has_hi && !hi_local;
            if ((lo_remote || hi_remote) && loopback_) {
                // SF_FLAG_LOOPBACK_HALO: same bytes, same stream, same dependencies, but from this slab's own planes
                sfk::HaloCopyArgs H;
                H.nseg = 0;
                H.n16 = (long)(bytes / 16);
                for (int f = 0; f < NF; ++f) {
                    T* mine = sl.field[fields[f]];
                    if (lo_remote) {
                        H.src[H.nseg] = reinterpret_cast<const float4*>(mine + send_lo);
                        H.dst[H.nseg++] = reinterpret_cast<float4*>(mine + recv_lo);
                    }
                    if (hi_remote) {
                        H.src[H.nseg] = reinterpret_cast<const float4*>(mine + send_hi);
                        H.dst[H.nseg++] = reinterpret_cast<float4*>(mine + recv_hi);
                    }
                }
                const unsigned gx = (unsigned)std::max(1L, std::min((H.n16 + 255) / 256, 512L));
                hipLaunchKernelGGL(sfk::halo_copy_kernel, dim3(gx, H.nseg), dim3(256), 0, sl.hs, H);
                SF_HIP(hipGetLastError());
                tr_halo<NF>("halo_loopback", sl, sl.hs, fields, nullptr, nullptr, lo_remote, hi_remote);
            } else if (lo_remote || hi_remote) {
                const ncclDataType_t dt = sizeof(T) == 4 ? ncclFloat : ncclDouble;
                SF_NCCL(ncclGroupStart());
                for (int f = 0; f < NF; ++f) {
                    T* mine = sl.field[fields[f]];
                    if (lo_remote) {
                        SF_NCCL(ncclSend(mine + send_lo, gcount, dt, rank_ - 1, comm_, sl.hs));
                        SF_NCCL(ncclRecv(mine + recv_lo, gcount, dt, rank_ - 1, comm_, sl.hs));
                    }
                    if (hi_remote) {
                        SF_NCCL(ncclSend(mine + send_hi, gcount, dt, rank_ + 1, comm_, sl.hs));
                        SF_NCCL(ncclRecv(mine + recv_hi, gcount, dt, rank_ + 1, comm_, sl.hs));
                    }
                }
                SF_NCCL(ncclGroupEnd());
                ++rccl_groups_;
                tr_halo<NF>("halo_rccl", sl, sl.hs, fields, nullptr, nullptr, lo_remote, hi_remote);
            }
            ev_record(sl, &Slab::halo_done, sl.hs);
        }
        // consumers: the next boundary launch reads my ghosts, and neighbours that pulled from my planes must be
        // done before I overwrite them two sweeps later. The compute stream only waits when it runs a
        // whole-field operator (join()).
        for (int s = 0; s < L_; ++s) {
            Slab& sl = slabs_[s];
            st_wait(sl, sl.bs, sl, &Slab::halo_done);
            if (s > 0) st_wait(sl, sl.bs, slabs_[s - 1], &Slab::halo_done);
            if (s < L_ - 1) st_wait(sl, sl.bs, slabs_[s + 1], &Slab::halo_done);
        }
        pending_join_ = true;
    }

    // SF_FLAG_RCCL_SELF: the ghost planes of the L logical slabs travel through a real RCCL communicator (one rank,
    // this GPU) as grouped ncclSend / ncclRecv to self — the calls, datatype, counts, buffer offsets, stream choice
    // (each slab's halo stream, which is its boundary stream under SF_HALO_STREAM=2 as in production) and the
    // system-scope halo_done event of the multi-process branch of exchange(). RCCL matches the sends and the receives
    // of one peer in issue order, so every transfer is issued as the pair (send from the owner's planes, receive into
    // the neighbour's ghost planes); one group spans all slabs because a send to self needs its receive in the same
    // group.
    template <int NF>
    void exchange_rccl_self(const int (&fields)[NF], size_t gcount, size_t send_lo, size_t send_hi, size_t recv_lo,
                            size_t recv_hi) {
        const ncclDataType_t dt = sizeof(T) == 4 ? ncclFloat : ncclDouble;
        for (int s = 0; s < L_; ++s) {
            Slab& sl = slabs_[s];
            st_wait(sl, sl.hs, sl, &Slab::boundary_done);
            if (s > 0) st_wait(sl, sl.hs, slabs_[s - 1], &Slab::boundary_done);
            if (s < L_ - 1) st_wait(sl, sl.hs, slabs_[s + 1], &Slab::boundary_done);
        }
        SF_NCCL(ncclGroupStart());
        for (int s = 0; s + 1 < L_; ++s) {
            Slab& lo = slabs_[s];
            Slab& hi = slabs_[s + 1];
            for (int f = 0; f < NF; ++f) {
                T* a = lo.field[fields[f]];
                T* b = hi.field[fields[f]];
                // upward: last interior planes of slab s -> low ghost planes of slab s+1
                SF_NCCL(ncclSend(a + send_hi, gcount, dt, 0, comm_, lo.hs));
                SF_NCCL(ncclRecv(b + recv_lo, gcount, dt, 0, comm_, hi.hs));
                // downward: first interior planes of slab s+1 -> high ghost planes of slab s
                SF_NCCL(ncclSend(b + send_lo, gcount, dt, 0, comm_, hi.hs));
                SF_NCCL(ncclRecv(a + recv_hi, gcount, dt, 0, comm_, lo.hs));
            }
        }
        SF_NCCL(ncclGroupEnd());
        ++rccl_groups_;
        for (int s = 0; s < L_; ++s)
            tr_halo<NF>("halo_rccl_self", slabs_[s], slabs_[s].hs, fields, s > 0 ? &slabs_[s - 1] : nullptr,
                        s < L_ - 1 ? &slabs_[s + 1] : nullptr, false, false);
        for (int s = 0; s < L_; ++s) ev_record(slabs_[s], &Slab::halo_done, slabs_[s].hs);
        for (int s = 0; s < L_; ++s) {
            Slab& sl = slabs_[s];
            st_wait(sl, sl.bs, sl, &Slab::halo_done);
            if (s > 0) st_wait(sl, sl.bs, slabs_[s - 1], &Slab::halo_done);
            if (s < L_ - 1) st_wait(sl, sl.bs, slabs_[s + 1], &Slab::halo_done);
        }
        pending_join_ = true;
    }

    template <int NF>
    void op_add_source(const int (&x)[NF], const int (&s)[NF]) {
        join();
        const long nvec = field_elems_ / W;
        for (Slab& sl : slabs_) {
            sfk::AddSourceArgs<T, NF> A;
            for (int f = 0; f < NF; ++f) {
                A.x[f] = ensure(sl, x[f]);
                A.s[f] = ensure(sl, s[f]);
            }
            A.dt = dt_;
            A.nvec = nvec;
            hipLaunchKernelGGL((sfk::add_source_kernel<T, NF>), dim3((unsigned)ceil_div(nvec, 256L)), dim3(256), 0, sl.cs, A);
            for (int f = 0; f < NF; ++f) tr_whole("add_source", sl, {A.x[f], A.s[f]}, {A.x[f]});
        }
        SF_HIP(hipGetLastError());
        // ghosts of x and s were current, so the ghosts of the result are current: no exchange
    }

    template <int NF>
    void op_add_source_bound(const int (&x)[NF], const int (&s_copy)[NF], const int (&src)[NF]) {
        join();
        const long nvec = field_elems_ / W;
        for (Slab& sl : slabs_) {
            sfk::AddSourceBoundArgs<T, NF> A;
            for (int f = 0; f < NF; ++f) {
                A.x[f] = ensure(sl, x[f]);
                A.s_copy[f] = ensure(sl, s_copy[f]);
                A.src[f] = ensure(sl, src[f]);
            }
            A.dt = dt_;
            A.nvec = nvec;
            hipLaunchKernelGGL((sfk::add_source_bound_kernel<T, NF>), dim3((unsigned)ceil_div(nvec, 256L)), dim3(256), 0, sl.cs, A);
            for (int f = 0; f < NF; ++f) tr_whole("add_source_bound", sl, {A.x[f], A.src[f]}, {A.x[f], A.s_copy[f]});
        }
        SF_HIP(hipGetLastError());
    }

    // Single-sweep launcher (flat register-blocked kernel with XCD bands): odd last sweeps, grids the fused kernels do
    // not take. SF_NT: 0 never / 1 always / 2 auto non-temporal stores. SF_ISHELL: 0 = always read+write the i-shell,
    // 1 = recompute it in intermediate sweeps (default).
    template <int NF, bool NT, int RJ, int RK>
    void launch_rb(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool first, bool last) {
        const int nvec = ceil_div(N_, W);
        int tx = 1;
        while (tx < nvec && tx < 64) tx <<= 1;
        const int ty = 256 / tx;
        sfk::TileMap m{};
        m.gx = ceil_div(nvec, tx);
        m.gy = ceil_div(N_, ty * RJ);
        m.nxcd = 8;
        m.band = (m.gy >= 16) ? ceil_div(m.gy, 8) : 0;
        m.ishell_mem = (!ishell_skip_ || first) ? 1 : 0;
        m.ishell_write = (!ishell_skip_ || last) ? 1 : 0;
        m.split = split_;
        m.gap = gap_;
        const long per_plane = m.band > 0 ? (long)m.nxcd * m.gx * m.band : (long)m.gx * m.gy;
        const long nblocks = per_plane * ceil_div(ke - kb, RK) * NF;
        hipLaunchKernelGGL((sfk::jacobi_rb_kernel<T, NF, NT, RJ, RK>), dim3((unsigned)nblocks), dim3(tx, ty), 0,
                           sl.cur, sl.geom, A, kb, ke, m);
    }

    template <int NF, bool NT>
    void launch_rb_shape(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool first, bool last) {
        // measured (512^3 / 256^3 fp32): 2x2 blocks win once the sweep streams from HBM (286 vs 309 us),
        // 1x1 wins while x, x0, x' sit in the Infinity Cache (30.8 vs 34.0 us)
        const bool blocks = NT && !(split_ != INT_MAX && split_ % 2 != 0);  // a plane block must not straddle the split
        if (blocks)
            launch_rb<NF, NT, 2, 2>(sl, A, kb, ke, first, last);
        else
            launch_rb<NF, NT, 1, 1>(sl, A, kb, ke, first, last);
    }

    template <int NF>
    void launch_jacobi(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool first, bool last) {
        // non-temporal stores pay once x, x0 and x' of all NF fields no longer fit the 256 MiB Infinity Cache
        const bool nt = nt_mode_ == 1 ||
                        (nt_mode_ == 2 && (size_t)field_elems_ * sizeof(T) * 3 * NF > ((size_t)384 << 20));
        if (nt)
            launch_rb_shape<NF, true>(sl, A, kb, ke, first, last);
        else
            launch_rb_shape<NF, false>(sl, A, kb, ke, first, last);
    }

    // Two fused sweeps (temporal blocking). Usable when a row fits one workgroup, N is a multiple of the
    // vector width and the grid is not decomposed (a second ghost plane would be needed).
    // (measured against single sweeps: +20 % at 512^3, +35 % at 256^3, +11 % at 1024^3 fp32, +15 % at 512^3 fp64)
    bool can_fuse2() const {
        return fuse2_ && (P_ == 1 || G_ >= 2) && N_ % W == 0 && N_ / W <= fuse_maxvec_;
    }

    template <int NF>
    static sfk::JacobiArgs<T, 1> first_field(const sfk::JacobiArgs<T, NF>& A) {
        sfk::JacobiArgs<T, 1> B;
        B.x[0] = A.x[0];
        B.x0[0] = A.x0[0];
        B.xn[0] = A.xn[0];
        B.b[0] = A.b[0];
        B.a = A.a;
        B.inv = A.inv;
        return B;
    }

    template <int NF, bool NT, int RJ, int RK, bool SRC = false>
    void launch_fused2(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool first, bool last) {
        const int nvec = N_ / W;
        sfk::TileMap m{};
        // row strips per 256-thread workgroup: packed densely (strip = nvec lanes; rows then start anywhere inside a
        // wave and both ends of most waves are seams -> LDS hand-over of x) or aligned to wave boundaries (strip
        // = multiple of 64; idle lanes, but one seam per wave at most). Aligned wins unless it idles >10 % more lanes.
        const int aligned = ceil_div(nvec, 64) * 64;
        const double eff_dense = (double)((256 / nvec) * nvec), eff_aligned = (double)((256 / aligned) * nvec);
        m.strip = (nvec % 64 == 0 || 64 % nvec == 0 || eff_aligned >= 0.9 * eff_dense) ? std::max(aligned, nvec) : nvec;
        if (64 % nvec == 0) m.strip = nvec;  // narrow rows: several whole rows per wave, never a seam
        m.rows = std::max(1, 256 / m.strip);
        m.gx = 1;
        m.gy = ceil_div(N_, m.rows * RJ);
        m.nxcd = 8;
        m.band = (m.gy >= 16) ? ceil_div(m.gy, 8) : 0;
        m.ishell_mem = (!ishell_skip_ || first) ? 1 : 0;
        m.ishell_write = (!ishell_skip_ || last) ? 1 : 0;
        m.split = split_;
        m.gap = gap_;
        // rows that neither fill whole waves nor divide one: the seam-free overlapped mapping (SF_OVL: 0 never,
        // 1 for such rows (default), 2 for every width)
        // ... and rows wider than two waves, where it also beats one row strip per workgroup (1024^3 fp32: 2026 vs
        // 1986 us/sweep; the single-sweep kernel: 2207)
        // (a row strip must fit the 256 threads of a workgroup: beyond that only the overlapped mapping exists)
        const bool ovl = ovl_mode_ == 2 || (ovl_mode_ == 1 && ((nvec % 64 != 0 && 64 % nvec != 0) || nvec > 128)) ||
                         nvec > 256;
        if (ovl) {
            const int items = ceil_div(N_, RJ) * nvec;
            m.gy = ceil_div(ceil_div(items, sfk::SF_OVL_OUT), 4);
            m.band = (m.gy >= 16) ? ceil_div(m.gy, 8) : 0;
        }
        // warm-up loads (see TileMap::pf_dz): about 32 workgroups ahead on the same XCD (measured best at 256^3 and
        // 512^3), expressed in plane blocks at the same j position
        // Automatic mode: only where the data comes from HBM (x, x0, x' of the launch's fields exceed the Infinity
        // Cache: a resident working set gains nothing, 256^3) and only where a plane block is a fine enough unit of
        // distance (<= 48 workgroups per XCD and plane block) and rows are at most 128 vectors wide: with rows of 256
        // vectors the early lines evict the j / k reuse from the 4 MB L2 (1024^3 fp32 ran 8 % slower, 512^3 fp64 5-15 %).
        {
            const int per_xcd_round = m.band > 0 ? m.band : m.gy;
            const double ws = 3.0 * NF * (double)(N_ + 2) * (N_ + 2) * nplanes_ * sizeof(T);
            if (per_xcd_round <= 48 && nvec <= 128 && ws > 0.9 * 256.0 * 1048576.0)
                m.pf_dz = std::max(1, (32 + per_xcd_round / 2) / std::max(1, per_xcd_round));
            else
                m.pf_dz = 0;
        }
        m.strip_shift = (m.strip & (m.strip - 1)) == 0 ? __builtin_ctz((unsigned)m.strip) : -1;
        m.nvec_magic = nvec > 1 ? 0xFFFFFFFFu / (unsigned)nvec + 1u : 0u;
        const dim3 nb(m.band > 0 ? (unsigned)m.nxcd : (unsigned)m.gy, m.band > 0 ? (unsigned)m.band : 1u,
                      (unsigned)(ceil_div(ke - kb, RK) * NF));
        const bool xlds = m.strip % 64 != 0 && 64 % m.strip != 0;
        if constexpr (!SRC) {
            if (NF == 1 && x_is_zero_) {  // implicit-zero first pair of project's lin_solve
                if (ovl)
                    launch_k(sl, sfk::jacobi2_kernel<T, 1, NT, RJ, RK, false, true, true>, nb, 256u, sl.geom, first_field(A),
                             kb, ke, m);
                else if (xlds)
                    launch_k(sl, sfk::jacobi2_kernel<T, 1, NT, RJ, RK, true, true>, nb, 256u, sl.geom, first_field(A), kb, ke,
                             m);
                else
                    launch_k(sl, sfk::jacobi2_kernel<T, 1, NT, RJ, RK, false, true>, nb, 256u, sl.geom, first_field(A), kb,
                             ke, m);
                return;
            }
        }
        if (ovl)
            launch_k(sl, sfk::jacobi2_kernel<T, NF, NT, RJ, RK, false, false, true, SRC>, nb, 256u, sl.geom, A, kb, ke, m);
        else if (xlds)
            launch_k(sl, sfk::jacobi2_kernel<T, NF, NT, RJ, RK, true, false, false, SRC>, nb, 256u, sl.geom, A, kb, ke, m);
        else
            launch_k(sl, sfk::jacobi2_kernel<T, NF, NT, RJ, RK, false, false, false, SRC>, nb, 256u, sl.geom, A, kb, ke, m);
    }

    // k-marching S-sweep kernel (sfk::jacobi_sk_kernel): the plain passes of a solve (iterate already swept once, so its
    // i-shell is recomputed in registers) on plane ranges long enough to march. SF_MARCH=0 switches it off.
    bool can_march_k(int nplanes, bool first) const {
        // small grids do not fill the chip with workgroups of 32-48 rows (128^3: 9.0 vs 4.3 us/sweep)
        return march_k_ != 0 && !first && ishell_skip_ && split_ == INT_MAX && nplanes >= march_min_planes_ &&
               !x_is_zero_ && (long)N_ * N_ * nplanes >= march_min_cells_;
    }

    // S fused sweeps with LDS halo exchange (sfk::jacobi_sk_kernel). Same eligibility as the two-sweep marching kernel;
    // three sweeps only on an undecomposed grid (a slab boundary would need three ghost planes).
    // (two-sweep launches — slab interiors with two ghost planes, remainders — only pay on large grids: at 256^3 the
    // register-blocked pair kernel takes 49 us, the marching kernel 67; at 512^3 437 against 365)
    bool can_sk(int nplanes, bool first, int sweeps = 3) const {
        return can_march_k(nplanes, first) && (sweeps >= 3 || (long)N_ * N_ * nplanes >= sk2_min_cells_);
    }

    int sk_chunks(int ncb, int np, int S) const {
        const int max_chunks = std::max(1, np / 8);
        double best = -1;
        int nchunk = 1;
        for (int c = 1; c <= max_chunks; ++c) {
            const int kc = ceil_div(np, c);
            const long total = (long)ncb * ceil_div(np, kc);
            // (workgroups are dealt to the eight XCDs in turn; an XCD runs one per CU at a time)
            const double tm = (double)ceil_div(ceil_div(total, 8L), (long)std::max(1, num_cu_ / 8)) * (kc + 2 * S - 2 + 2);
            if (best < 0 || tm < best * 0.999) {
                best = tm;
                nchunk = c;
            }
        }
        return nchunk;
    }
    // Tile of the four-sweep launches (rows per wave x waves stacked in j; the lane vector is 8 bytes in both types).
    // fp32: 2 x 16 — four waves per SIMD at <= 128 registers: a march step is bound by each wave's serial instruction
    // stream (tools/sk_probe.hip -DSF_SK_DIAG=4: 88 % of its time with every load and store removed), so 16 thin waves
    // beat 8 waves of four rows (256^3: 62 -> 59 us per pass, 512^3: 441 -> 392, same 32-row tile and bytes).
    // fp64: 5 x 8 — half the cells per lane, twice the bytes per cell: there the 40-row tile (32 rows stored instead
    // of 24 of 32) is worth more than the waves (512^3 K = 40 step: 50.5 against 56.4 ms).
    // First passes over caller data (FIRST = 1, 2) hold a ring of shell cells and the right-hand side on top and
    // spill at 128 registers: 4 x 8 in both types.
    static constexpr int SK4_TJ = sizeof(T) == 4 ? 2 : 5, SK4_NW = sizeof(T) == 4 ? 16 : 8;
    static constexpr int SK4F_TJ = 4, SK4F_NW = 8;
    template <bool NT, int S, int TJ, int NW, int FIRST = 0, int NF = 1>
    void launch_sk_cfg(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool last) {
        constexpr int WL = W / 2;  // 8 bytes per lane
        constexpr int V = NW * TJ - 2 * S, P = 64 - 2 * ((S + WL - 1) / WL);
        const int nvec = N_ / WL;
        sfk::SkMap m{};
        m.njb = ceil_div(N_, V);
        const long items = (long)m.njb * nvec;
        m.ncb = (int)ceil_div(items, (long)P);
        m.band = ceil_div(m.ncb, 8);
        m.nvec_magic = nvec > 1 ? 0xFFFFFFFFu / (unsigned)nvec + 1u : 0u;
        const int np = ke - kb;
        int nchunk;
        if (split_ != INT_MAX) {
            // boundary launch of a decomposed grid: the first and the last `split_` interior planes as two chunks
            nchunk = 2;
            m.gap = gap_;
        } else {
            // One workgroup per CU at a time (LDS, registers): time ~ (workgroups per CU, rounded up) x (steps per
            // chunk: kc + 2S-2, plus start-up).
            nchunk = sk_chunks(m.ncb * NF, np, S);  // (NF fields in one grid: NF times the column blocks)
        }
        m.kc = split_ != INT_MAX ? split_ : ceil_div(np, nchunk);
        nchunk = ceil_div(np, m.kc);
        dim3 nb(8u, (unsigned)m.band, (unsigned)(nchunk * NF));
        if constexpr (FIRST != 0) {  // a first pass is never the last one (sk_first_ok)
            launch_k(sl, sfk::jacobi_sk_kernel<T, NF, WL, NT, S, TJ, NW, false, FIRST>, nb, 64u * NW, sl.geom, A, kb, ke, m);
        } else if constexpr (NF > 1) {  // (fields in one grid: plain passes only, launch_sk)
            launch_k(sl, sfk::jacobi_sk_kernel<T, NF, WL, NT, S, TJ, NW, false>, nb, 64u * NW, sl.geom, A, kb, ke, m);
        } else {
            if (last)
                launch_k(sl, sfk::jacobi_sk_kernel<T, 1, WL, NT, S, TJ, NW, true>, nb, 64u * NW, sl.geom, A, kb, ke, m);
            else
                launch_k(sl, sfk::jacobi_sk_kernel<T, 1, WL, NT, S, TJ, NW, false>, nb, 64u * NW, sl.geom, A, kb, ke, m);
        }
    }

    // First pass of a solve through the marching kernel (four sweeps; sfk::SkFirst): mode 1 caller data, 2 folded
    // add_source, 3 zero iterate. One launch per field.
    template <int NF>
    void launch_sk_first(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, int mode) {
        const bool nt = nt_mode_ == 1 ||
                        (nt_mode_ == 2 && (size_t)field_elems_ * sizeof(T) * 3 * NF > ((size_t)384 << 20));
        // (tiles: SK4F_* for the passes that read the caller's i-shell, SK4_* for the zero-iterate pass, see above)
        constexpr int TJ0 = SK4F_TJ, NWF = SK4F_NW, TJ3 = SK4_TJ, NW4 = SK4_NW;
        if constexpr (NF > 1) {
            if (mode == 2 && fields_in_one_grid()) {
                if (nt) launch_sk_cfg<true, 4, TJ0, NWF, 2, NF>(sl, A, kb, ke, false);
                else launch_sk_cfg<false, 4, TJ0, NWF, 2, NF>(sl, A, kb, ke, false);
                return;
            }
        }
        for (int f = 0; f < NF; ++f) {
            sfk::JacobiArgs<T, 1> B;
            B.x[0] = A.x[f];
            B.x0[0] = A.x0[f];
            B.xn[0] = A.xn[f];
            B.x0out[0] = A.x0out[f];
            B.b[0] = A.b[f];
            B.a = A.a;
            B.inv = A.inv;
            B.dt = A.dt;
            if (mode == 1) {
                if (nt) launch_sk_cfg<true, 4, TJ0, NWF, 1>(sl, B, kb, ke, false);
                else launch_sk_cfg<false, 4, TJ0, NWF, 1>(sl, B, kb, ke, false);
            } else if (mode == 2) {
                if (nt) launch_sk_cfg<true, 4, TJ0, NWF, 2>(sl, B, kb, ke, false);
                else launch_sk_cfg<false, 4, TJ0, NWF, 2>(sl, B, kb, ke, false);
            } else {
                if (nt) launch_sk_cfg<true, 4, TJ3, NW4, 3>(sl, B, kb, ke, false);
                else launch_sk_cfg<false, 4, TJ3, NW4, 3>(sl, B, kb, ke, false);
            }
        }
    }
    // May the FIRST pass of a K-sweep solve go through the marching kernel? Undecomposed grid, four-sweep launches
    // enabled, a grid the kernel takes, and sweeps left over afterwards (no i-shell-writing variant of a first pass).
    bool sk_first_ok(int K) const {
        if (!(sk_first_ && march_k_ != 0 && sk_s_ >= 4 && can_fuse2() && ishell_skip_ && K >= 7)) return false;
        if (P_ == 1) return nzl_ >= march_min_planes_ && (long)N_ * N_ * nzl_ >= march_min_cells_;
        const int interior = nzl_ - 2 * std::max(4, G_);  // slabs: four ghost planes and an interior launch the kernel takes
        return G_ >= 4 && split_enabled_ && interior >= march_min_planes_ && (long)N_ * N_ * interior >= march_min_cells_;
    }

    // The NF fields of a batched solve (u, v, w of a diffusion) as ONE marching grid on an undecomposed slab: NF times
    // the column blocks let the launch fill the chip with NF times fewer, longer chunks — a chunk pays 2(S-1) warm-up
    // planes whatever its length (256^3: 3 chunks of 86 planes instead of 10 of 26 per field: 92 instead of 96 steps per
    // workgroup; plain pass 51.3 against 53.3 us per field, folded-source first pass 74 against 86).
    struct BatchScope {  // sets batch_now_ for the launches of one solve
        bool& flag;
        bool saved;
        BatchScope(bool& f, bool v) : flag(f), saved(f) { flag = v; }
        ~BatchScope() { flag = saved; }
    };
    bool fields_in_one_grid() const { return P_ == 1 && split_ == INT_MAX && split_fields_ != 2 && batch_now_; }
    // ... which the solve only asks for when every pass is a four-sweep marching launch (K a multiple of four on a
    // grid the kernel takes): a pair-kernel pass over three fields at once would leave the Infinity Cache. And only
    // where it was measured to pay (same-box A/B of the full fp32 K = 20 step, ms, fields apart / in one grid; `one` =
    // x + x0 + x' of one field): 160^3 (52 MB) 0.846 / 0.881; 192^3 (89 MB) 1.666 / 1.506; 208^3 (113 MB) 1.399 / 1.281;
    // 224^3 (141 MB) 1.960 / 2.069 — one field still fits the 256 MiB Infinity Cache there, three do not —; 256^3 (201 MB)
    // 2.17 / 2.09; 320^3 3.95 / 3.92; 384^3 6.59 / 6.54; 512^3 K = 40 25.64 / 25.35; 1024^3 121.8 / 123.5 and 512^3 fp64
    // 49.6 / 49.8 (chunks are long anyway); fp64: 144^3 (76 MB) 1.579 / 1.409; 176^3 (137 MB) 2.048 / 2.162; 192^3
    // (178 MB) 2.662 / 2.646; 256^3 (403 MB) 4.60 / 4.32; 320^3 7.89 / 7.75. Hence two windows in bytes, whatever the
    // precision: 75..125 MB and 170 MB..2 GB.
    bool batch_march(int K, bool continued) const {
        if (!(P_ == 1 && split_fields_ != 2 && K % 4 == 0 && K >= 4 && sk_first_ && march_k_ != 0 && sk_s_ >= 4 &&
              can_fuse2() && ishell_skip_ && nzl_ >= march_min_planes_ && (long)N_ * N_ * nzl_ >= march_min_cells_))
            return false;
        const double one = 3.0 * (double)(N_ + 2) * (N_ + 2) * nplanes_ * sizeof(T);  // x, x0, x' of one field
        const double mb = one / 1048576.0;
        if (split_fields_ == 1 && !((mb >= 75.0 && mb <= 125.0) || (mb >= 170.0 && mb <= 2048.0))) return false;
        return continued || K >= 8;
    }

    template <int NF, int S>
    void launch_sk(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool last) {
        SF_REQUIRE(ishell_skip_ && march_k_ != 0, "internal: marching launch while SF_ISHELL=0 / SF_MARCH=0");
        const bool nt = nt_mode_ == 1 ||
                        (nt_mode_ == 2 && (size_t)field_elems_ * sizeof(T) * 3 * NF > ((size_t)384 << 20));
        if constexpr (NF > 1 && S == 4) {
            if (!last && fields_in_one_grid()) {
                if (nt) launch_sk_cfg<true, S, SK4_TJ, SK4_NW, 0, NF>(sl, A, kb, ke, false);
                else launch_sk_cfg<false, S, SK4_TJ, SK4_NW, 0, NF>(sl, A, kb, ke, false);
                return;
            }
        }
        for (int f = 0; f < NF; ++f) {  // one launch per field (fields are independent)
            sfk::JacobiArgs<T, 1> B;
            B.x[0] = A.x[f];
            B.x0[0] = A.x0[f];
            B.xn[0] = A.xn[f];
            B.b[0] = A.b[f];
            B.a = A.a;
            B.inv = A.inv;
            // Tile: SK4_* at S = 4 (above); six rows x eight waves at S <= 3. Four levels hold 17 planes of rows per
            // lane (x 3, x0 5, three intermediate levels x 3). (A spill in the wall workgroups alone doubles a launch
            // at 256^3, where every workgroup runs at once and the slowest one is the launch.)
            constexpr int TJ0 = S == 4 ? SK4_TJ : 6, NW0 = S == 4 ? SK4_NW : 8;
            if (nt)
                launch_sk_cfg<true, S, TJ0, NW0>(sl, B, kb, ke, last);
            else
                launch_sk_cfg<false, S, TJ0, NW0>(sl, B, kb, ke, last);
        }
    }

    template <int NF>
    void launch_jacobi_s(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool last, int sweeps) {
        if (sweeps == 4)
            launch_sk<NF, 4>(sl, A, kb, ke, last);
        else
            launch_sk<NF, 3>(sl, A, kb, ke, last);
    }

    template <int NF, bool SRC = false>
    void launch_jacobi2(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool first, bool last) {
        const bool nt = nt_mode_ == 1 ||
                        (nt_mode_ == 2 && (size_t)field_elems_ * sizeof(T) * 3 * NF > ((size_t)384 << 20));
        if constexpr (!SRC) {
            if (can_sk(ke - kb, first, 2)) {
                launch_sk<NF, 2>(sl, A, kb, ke, last);
                return;
            }
        }
        if (nt)
            launch_fused2_shape<NF, true, SRC>(sl, A, kb, ke, first, last);
        else
            launch_fused2_shape<NF, false, SRC>(sl, A, kb, ke, first, last);
    }

    template <int NF, bool NT, bool SRC = false>
    void launch_fused2_shape(Slab& sl, const sfk::JacobiArgs<T, NF>& A, int kb, int ke, bool first, bool last) {
        // 2x2 output vectors per thread: measured best of 1x1, 2x1, 1x2, 2x2, 4x2 (4x2 spills)
        launch_fused2<NF, NT, 2, 2, SRC>(sl, A, kb, ke, first, last);
    }

    // Sweeps fused into the launch that starts at iteration `it` of a K-sweep solve: 3 where the S-sweep kernel is in
    // use (never the first pass of a solve, whose iterate is caller data / zero / a source; a remainder of four goes
    // as 2 + 2), else 2 where pairs can be fused, else 1.
    int sweeps_in_launch(int it, int K, bool continued, int extra = 0) const {
        const bool pair = can_fuse2() && it + 2 <= K;
        const int left = K - it;
        if (it == 0 && !continued && sk_first_ok(K)) return 4;
        // (the marching kernel leaves the i-shell implicit between passes: with SF_ISHELL=0 — every pass reads it from
        // memory — it must not run at all, on one slab or many. Round 2 checked that for P_ == 1 only, and a
        // decomposed solve mixed marching passes with pair passes that read a stale i-shell.)
        bool marching = pair && (it > 0 || continued) && march_k_ != 0 && ishell_skip_ && !x_is_zero_ && sk_s_ >= 3 &&
                        left >= 3;
        if (marching && P_ == 1) marching = can_sk(nzl_, false);
        // S sweeps per pass need S ghost planes on a decomposed grid, the two-stream schedule, and an interior launch
        // [G+S+extra, ...) the marching kernel takes (the boundary launch always goes through it: there is no other
        // kernel of that depth)
        auto slab_ok = [&](int S) {
            const int interior = nzl_ - 2 * (std::max(S, G_) + extra);  // for_planes: depth = max(S, G) + extra
            return P_ == 1 || (G_ >= S && split_enabled_ && interior >= march_min_planes_ &&
                               (long)N_ * N_ * interior >= march_min_cells_);
        };
        // four sweeps per pass; remainders of 5 and 6 go as 3 + 2 and 3 + 3; without four-sweep launches a remainder of
        // 4 goes as 2 + 2
        if (marching && sk_s_ >= 4 && left >= 4 && left != 5 && left != 6 && slab_ok(4)) return 4;
        if (marching && left != 4 && slab_ok(3)) return 3;
        return pair ? 2 : 1;
    }
    // boundary depth of a two-sweep launch: the register-blocked pair kernel works on plane pairs, and a plane block
    // must not straddle the split of a boundary launch, so with three ghost planes it takes four planes per side
    int pair_depth() const { return G_ >= 3 ? 4 : 2; }

    // K Jacobi sweeps on NF fields at once; scratch buffers are swapped into the slots.
    template <int NF>
    void op_lin_solve(const int (&x)[NF], const int (&x0)[NF], const int (&b)[NF], T a, T c, int K,
                      bool x_zero = false, bool continued = false) {
        static_assert(NF <= NSCRATCH, "not enough scratch buffers");
        if constexpr (NF > 1) {
            // x, x0 and x' of ONE field fit the 256 MiB Infinity Cache where those of NF fields together do not:
            // solving the fields one after the other then keeps every pair after the first out of HBM (256^3 fp32:
            // 3 x 50.8 us against 175.9 us per pair of three fields). Independent fields: same results.
            const double one = 3.0 * (double)(N_ + 2) * (N_ + 2) * nplanes_ * sizeof(T);
            const bool fits = one <= 0.9 * 256.0 * 1048576.0;
            if (split_fields_ == 2 || (split_fields_ == 1 && fits && !batch_march(K, continued))) {
                for (int f = 0; f < NF; ++f) {
                    const int xf[1] = {x[f]}, x0f[1] = {x0[f]}, bf[1] = {b[f]};
                    op_lin_solve<1>(xf, x0f, bf, a, c, K, x_zero, continued);
                }
                return;
            }
        }
        BatchScope batch_scope(batch_now_, NF > 1 && batch_march(K, continued));
        const T inv = T(1) / c;
        for (Slab& sl : slabs_)
            for (int f = 0; f < NF; ++f) {
                ensure(sl, x[f]);
                ensure(sl, x0[f]);
            }
        // Decomposed grid, fused pairs: a cross-stream wait in front of every interior launch costs ~10 us of idle
        // GPU per pair (measured: tools/evgap.hip, profiles of tools/rank_share.py). So the boundary launch grows by
        // two planes per side and pair ("trapezoid") for trap_m_ pairs: interior launch j then covers planes
        // [G+2+2j, ...) and reads only what interior launch j-1 wrote (planes [G+2j, ...)), back to back on the
        // compute stream, while boundary launch j (planes [G, G+2+2j), on its own stream, after interior j-1 and halo
        // j-1) feeds the halo exchange. Every trap_m_ pairs the interior snaps back and waits for the boundary once.
        // Same arithmetic on every plane whichever launch computes it: results do not change.
        // With S sweeps per launch the growth is S planes per side: if interior launch j-1 started D planes into the
        // slab, launch j starts D + max(S_j, S_{j-1}) planes in (S_j = its sweeps). S_j: it reads only what interior
        // launch j-1 wrote. S_{j-1}: it WRITES the buffer that was the input of launch j-1, which boundary launch j-1
        // (another stream, possibly still running) reads up to D + S_{j-1} planes in — a four-sweep launch followed
        // by a three-sweep one raced there until this was the maximum. Its boundary launch takes those planes.
        int tj = 0, dprev = 0, sprev = 0;
        int it = 0;
        while (it < K) {
            const bool pair = can_fuse2() && it + 2 <= K;
            // three sweeps per pass where the S-sweep kernel is in use (never the first pass of a solve, whose iterate
            // is caller data; a remainder of four goes as 2 + 2)
            x_is_zero_ = x_zero && it == 0 && pair;  // the first fused pair then loads no x at all
            const int step = sweeps_in_launch(it, K, continued, 0);
            const int depth0 = std::max(step == 2 ? pair_depth() : step, G_);  // boundary depth without growth
            int extra = 0;
            {
                // does this launch continue the trapezoid block?
                // (inject_trap_bug_: SF_TRACE_SCHEDULE's ",inject=trap" — the round-2 race, for the checker's own test)
                const int d = dprev + (inject_trap_bug_ ? step : std::max(step, sprev));  // where its interior launch would start
                bool cont = pair && P_ > 1 && G_ >= 2 && trap_m_ > 1 && tj > 0 && tj < trap_m_ && d >= depth0 &&
                            nzl_ > 2 * d + 2;
                if (cont && step >= 3 && sweeps_in_launch(it, K, continued, d - depth0) != step) cont = false;
                if (cont && step == 2 && (d & 1)) cont = false;  // plane pairs: even boundary depth
                if (!cont) tj = 0;
                extra = cont ? d - depth0 : 0;
                dprev = depth0 + extra;
                sprev = step;
            }
            const bool triple = step >= 3;  // three or four sweeps: the marching kernel
            // the pass that writes the i-shell: the last one, unless nothing will read that shell (dead_ishell_)
            const bool last = it + step == K && !dead_ishell_;
            trap_extra_ = extra;
            ++tj;
            if (trace_) {
                acc_name_ = step == 4 ? "jacobi4" : (step == 3 ? "jacobi3" : (step == 2 ? "jacobi2" : "jacobi1"));
                acc_fn_ = [&, step](Slab& sl, int a, int b, std::vector<Acc>& acc) {
                    int lo, hi;
                    wr_range(sl, a, b, lo, hi);
                    for (int f = 0; f < NF; ++f) {
                        if (!x_is_zero_) acc.push_back({sl.field[x[f]], false, a - step, b + step});
                        acc.push_back({sl.field[x0[f]], false, a - (step - 1), b + (step - 1)});
                        acc.push_back({sl.scratch[f], true, lo, hi});
                    }
                };
            }
            for_planes([&](Slab& sl, int kb, int ke) {
                sfk::JacobiArgs<T, NF> A;
                for (int f = 0; f < NF; ++f) {
                    A.x[f] = sl.field[x[f]];
                    A.x0[f] = sl.field[x0[f]];
                    A.xn[f] = sl.scratch[f];
                    A.b[f] = b[f];
                }
                A.a = a;
                A.inv = inv;
                if (it == 0 && !continued && step == 4)
                    launch_sk_first<NF>(sl, A, kb, ke, x_zero ? 3 : 1);
                else if (triple)
                    launch_jacobi_s<NF>(sl, A, kb, ke, last, step);
                else if (pair)
                    launch_jacobi2<NF>(sl, A, kb, ke, it == 0 && !continued, last);
                else
                    launch_jacobi<NF>(sl, A, kb, ke, it == 0 && !continued, last);
            }, step == 2 ? pair_depth() : step, true);
            acc_fn_ = nullptr;
            // the new iterate becomes the field; the old buffer becomes scratch
            for (Slab& sl : slabs_)
                for (int f = 0; f < NF; ++f) std::swap(sl.field[x[f]], sl.scratch[f]);
            exchange<NF>(x);
            it += step;
        }
        trap_extra_ = 0;
        x_is_zero_ = false;
    }

    // Right-hand side x + dt*src of a folded add_source on the G-1 ghost planes next to the slab on either side (an
    // S-sweep launch evaluates its first S-1 levels there and needs x0 for them; the source pass itself stores it on the
    // planes it computes). It reads ghost planes of x, so it must follow the last halo: on the boundary stream when
    // for_planes ran its two-stream schedule with that boundary depth (bs waits for every halo and the next boundary
    // launch follows in stream order), on the compute stream otherwise (for_planes has just joined it).
    template <int NF>
    void rhs_on_ghost_planes(const int (&x)[NF], const int (&x0)[NF], const int (&src)[NF], int depth) {
        if (P_ == 1) return;
        const bool two = split_enabled_ && nzl_ > 2 * std::max(depth, G_);
        for (Slab& sl : slabs_) {
            sfk::RhsPlanesArgs<T, NF> R;
            for (int f = 0; f < NF; ++f) {
                R.out[f] = sl.field[x0[f]];
                R.a[f] = sl.field[x[f]];
                R.s[f] = sl.field[src[f]];
            }
            R.dt = dt_;
            R.off[0] = (long)1 * plane_;            // planes 1 .. G-1
            R.off[1] = (long)(G_ + nzl_) * plane_;  // planes G+nzl .. G+nzl+G-2
            R.nvec = (long)(G_ - 1) * plane_ / W;
            hipLaunchKernelGGL((sfk::rhs_planes_kernel<T, NF>), dim3((unsigned)ceil_div(R.nvec, 256L), 2), dim3(256), 0,
                               two ? sl.bs : sl.cs, R);
            if (trace_) {
                std::vector<Acc> acc;
                for (int f = 0; f < NF; ++f)
                    for (int side = 0; side < 2; ++side) {
                        const int lo = side ? G_ + nzl_ : 1, hi = lo + G_ - 1;
                        acc.push_back({sl.field[x[f]], false, lo, hi});
                        acc.push_back({sl.field[src[f]], false, lo, hi});
                        acc.push_back({sl.field[x0[f]], true, lo, hi});
                    }
                tr_op("rhs_ghost", sl, two ? sl.bs : sl.cs, acc);
            }
        }
        SF_HIP(hipGetLastError());
    }

    // diffuse with add_source folded in (sources bound to resident slots): replaces
    //     add_source_bound(x, x0 <- src); swap(x0, x); lin_solve(x, x0)
    // The first sweep pair reads the source as its iterate and the field before add_source, forms x + dt*src in
    // registers and stores it to the x0 slot's buffer (whose old content is dead) for the later pairs. One pass
    // over the arrays less per field. The pair stores the right-hand side on the planes it computes; on a
    // decomposed grid the later pairs also read it on the first ghost plane of either side, which a small launch
    // fills from the (current) ghost planes of x and src.
    template <int NF>
    void op_diffuse_src(const int (&x)[NF], const int (&x0)[NF], const int (&b)[NF], const int (&src)[NF], T a, T c,
                        int K) {
        if (!(fuse_src_ && can_fuse2() && K >= 2)) {
            op_add_source_bound<NF>(x, x0, src);
            for (int f = 0; f < NF; ++f) swap_slots(x0[f], x[f]);
            op_lin_solve<NF>(x, x0, b, a, c, K);
            return;
        }
        if constexpr (NF > 1) {
            const double one = 3.0 * (double)(N_ + 2) * (N_ + 2) * nplanes_ * sizeof(T);
            if (split_fields_ == 2 || (split_fields_ == 1 && one <= 0.9 * 256.0 * 1048576.0 && !batch_march(K, false))) {
                for (int f = 0; f < NF; ++f) {
                    const int xf[1] = {x[f]}, x0f[1] = {x0[f]}, bf[1] = {b[f]}, sf[1] = {src[f]};
                    op_diffuse_src<1>(xf, x0f, bf, sf, a, c, K);
                }
                return;
            }
        }
        BatchScope batch_scope(batch_now_, NF > 1 && batch_march(K, false));
        const T inv = T(1) / c;
        for (Slab& sl : slabs_)
            for (int f = 0; f < NF; ++f) {
                ensure(sl, x[f]);
                ensure(sl, x0[f]);
                ensure(sl, src[f]);
            }
        const int sreach = sk_first_ok(K) ? 4 : 2;  // sweeps of the first pass = its reach in planes
        if (trace_) {
            acc_name_ = "jacobi_src";
            acc_fn_ = [&, sreach](Slab& sl, int a, int b, std::vector<Acc>& acc) {
                int lo, hi;
                wr_range(sl, a, b, lo, hi);
                for (int f = 0; f < NF; ++f) {
                    acc.push_back({sl.field[src[f]], false, a - sreach, b + sreach});
                    acc.push_back({sl.field[x[f]], false, a - (sreach - 1), b + (sreach - 1)});
                    acc.push_back({sl.scratch[f], true, lo, hi});
                    acc.push_back({sl.field[x0[f]], true, a, b});
                }
            };
        }
        if (sk_first_ok(K)) {
            // the same pass as four sweeps of the marching kernel (undecomposed grid): rhs formed per plane as it
            // arrives, stored for the later launches
            for_planes([&](Slab& sl, int kb, int ke) {
                sfk::JacobiArgs<T, NF> A;
                for (int f = 0; f < NF; ++f) {
                    A.x[f] = sl.field[src[f]];
                    A.x0[f] = sl.field[x[f]];
                    A.xn[f] = sl.scratch[f];
                    A.x0out[f] = sl.field[x0[f]];
                    A.b[f] = b[f];
                }
                A.a = a;
                A.inv = inv;
                A.dt = dt_;
                launch_sk_first<NF>(sl, A, kb, ke, 2);
            }, 4, true);
            acc_fn_ = nullptr;
            rhs_on_ghost_planes<NF>(x, x0, src, 4);
            for (Slab& sl : slabs_)
                for (int f = 0; f < NF; ++f) std::swap(sl.field[x[f]], sl.scratch[f]);
            exchange<NF>(x);
            op_lin_solve<NF>(x, x0, b, a, c, K - 4, false, true);
            return;
        }
        for_planes([&](Slab& sl, int kb, int ke) {
            sfk::JacobiArgs<T, NF> A;
            for (int f = 0; f < NF; ++f) {
                A.x[f] = sl.field[src[f]];    // iterate = the source (Stam's initial guess)
                A.x0[f] = sl.field[x[f]];     // the field before add_source
                A.xn[f] = sl.scratch[f];
                A.x0out[f] = sl.field[x0[f]];  // right-hand side x + dt*src for the later pairs
                A.b[f] = b[f];
            }
            A.a = a;
            A.inv = inv;
            A.dt = dt_;
            launch_jacobi2<NF, true>(sl, A, kb, ke, true, K == 2 && !dead_ishell_);
        }, pair_depth(), true);
        acc_fn_ = nullptr;
        rhs_on_ghost_planes<NF>(x, x0, src, pair_depth());
        for (Slab& sl : slabs_)
            for (int f = 0; f < NF; ++f) std::swap(sl.field[x[f]], sl.scratch[f]);
        exchange<NF>(x);
        op_lin_solve<NF>(x, x0, b, a, c, K - 2, false, true);
    }

    template <int NF>
    void op_advect(const int (&d)[NF], const int (&d0)[NF], const int (&b)[NF], int u, int v, int w) {
        const T dt0 = dt_ * (T)N_;
        for (Slab& sl : slabs_) {
            for (int f = 0; f < NF; ++f) {
                ensure(sl, d[f]);
                ensure(sl, d0[f]);
            }
            ensure(sl, u);
            ensure(sl, v);
            ensure(sl, w);
        }
        if (trace_) {
            acc_name_ = "advect";
            acc_fn_ = [&](Slab& sl, int a, int b_, std::vector<Acc>& acc) {
                int lo, hi;
                wr_range(sl, a, b_, lo, hi);
                for (int f = 0; f < NF; ++f) {
                    acc.push_back({sl.field[d0[f]], false, a - 1, b_ + 1});
                    acc.push_back({sl.field[d[f]], true, lo, hi});
                }
                acc.push_back({sl.field[u], false, a, b_});
                acc.push_back({sl.field[v], false, a, b_});
                acc.push_back({sl.field[w], false, a, b_});
            };
        }
        for_planes([&](Slab& sl, int kb, int ke) {
            sfk::AdvectArgs<T, NF> A;
            for (int f = 0; f < NF; ++f) {
                A.d[f] = sl.field[d[f]];
                A.d0[f] = sl.field[d0[f]];
                A.b[f] = b[f];
            }
            A.u = sl.field[u];
            A.v = sl.field[v];
            A.w = sl.field[w];
            A.dt0 = dt0;
            A.flag = sl.d_flag;
            A.skip_ishell = dead_ishell_ ? 1 : 0;
            dim3 block;
            unsigned nblocks;
            const sfk::TileMap m = flat_map(ke - kb, block, nblocks);
            if (advect_row_ >= 2 || (advect_row_ == 1 && NF >= 2)) {
                // one cell per lane for the three velocity components. fp32: the i0+1 samples from the neighbour lane
                // (256^3 245 -> 171 us, 512^3 1628 -> 1217; with own (i0, i0+1) pair loads 215 / 1537). fp64: own pair
                // loads (256^3 376 -> 307 us; with neighbour-lane sharing 415). One field: the gather form stays
                // (fp32 79 vs 88 / 113, fp64 130 vs 150 / 129). SF_ADVECT_ROW = 0 never, 2 / 3 always the sharing /
                // the pair form.
                const int wpr = ceil_div(N_, 64);
                const long waves = (long)wpr * N_ * (ke - kb);
                if (advect_row_ == 3 || (advect_row_ == 1 && sizeof(T) == 8))
                    hipLaunchKernelGGL((sfk::advect_row_kernel<T, NF, true>), dim3((unsigned)ceil_div(waves, 4L)),
                                       dim3(256), 0, sl.cur, sl.geom, A, kb, ke, wpr);
                else
                    hipLaunchKernelGGL((sfk::advect_row_kernel<T, NF>), dim3((unsigned)ceil_div(waves, 4L)), dim3(256), 0,
                                       sl.cur, sl.geom, A, kb, ke, wpr);
            } else
                hipLaunchKernelGGL((sfk::advect_kernel<T, NF>), dim3(nblocks), block, 0, sl.cur, sl.geom, A, kb, ke, m);
        }, 1, true, /*interior_reads_ghosts=*/true);  // a long back-trace may reach a ghost plane from any plane
        acc_fn_ = nullptr;
        exchange<NF>(d);
    }

    // mirror_u: u's i-shell was left unwritten by the solve before (b = 1: mirrored in project_div); dead_p: nothing reads
    // p after this projection (its slot is overwritten before anyone looks), so the solve leaves p's i-shell unwritten
    // and project_sub mirrors it. Both false for the public sf_project().
    void op_project(int u, int v, int w, int p, int div, bool mirror_u = false, bool dead_p = false) {
        mirror_u = mirror_u && ishell_skip_;
        dead_p = dead_p && ishell_skip_ && K_ >= 1;
        // (a projection whose pressure is dead is the first one of vel_step: its div slot is overwritten as well)
        const bool dead_div = dead_p;
        const T Nf = (T)N_;
        const T h = T(1) / Nf;
        auto args = [&](Slab& sl) {
            sfk::ProjectArgs<T> A;
            A.u = ensure(sl, u);
            A.v = ensure(sl, v);
            A.w = ensure(sl, w);
            A.p = ensure(sl, p);
            A.div = ensure(sl, div);
            A.c_div = T(-0.5) * h;
            A.c_grad = T(0.5) * Nf;
            A.mirror_u = mirror_u ? 1 : 0;
            A.mirror_p = dead_p ? 1 : 0;
            A.skip_div_ishell = dead_div ? 1 : 0;
            return A;
        };
        // p = 0: when the first two sweeps are fused the kernel treats x as literal zeros and p is never read,
        // so the fill (one word per cell) is skipped; otherwise zero the whole field (ghosts and shells included)
        const bool implicit_zero = can_fuse2() && K_ >= 2 && zero_skip_;
        if (!implicit_zero) join();
        for (Slab& sl : slabs_) {
            ensure(sl, p);
            if (!implicit_zero) {
                SF_HIP(hipMemsetAsync(sl.field[p], 0, (size_t)field_elems_ * sizeof(T), sl.cs));
                tr_whole("zero_p", sl, {}, {sl.field[p]});
            }
        }
        if (trace_) {
            acc_name_ = "project_div";
            acc_fn_ = [&](Slab& sl, int a, int b_, std::vector<Acc>& acc) {
                int lo, hi;
                wr_range(sl, a, b_, lo, hi);
                acc.push_back({sl.field[u], false, a, b_});
                acc.push_back({sl.field[v], false, a, b_});
                acc.push_back({sl.field[w], false, a - 1, b_ + 1});
                acc.push_back({sl.field[div], true, lo, hi});
            };
        }
        for_planes([&](Slab& sl, int kb, int ke) {
            dim3 block;
            unsigned nblocks;
            const sfk::TileMap m = flat_map(ke - kb, block, nblocks);
            hipLaunchKernelGGL((sfk::project_div_kernel<T>), dim3(nblocks), block, 0, sl.cur, sl.geom, args(sl), kb, ke, m);
        });
        acc_fn_ = nullptr;
        // div's ghost planes are exchanged although a single sweep reads div at cell centres only: the fused
        // sweep pair evaluates its first sweep on the first ghost plane and needs x0 = div there, and div is left
        // in the v0 slot, where the caller may use it as the next step's source / initial guess (all G planes).
        // p is zero, ghosts included.
        const int dv[1] = {div};
        exchange<1>(dv);
        const int ps[1] = {p}, b0[1] = {0};
        dead_ishell_ = dead_p;
        op_lin_solve<1>(ps, dv, b0, T(1), T(6), K_, implicit_zero);
        dead_ishell_ = false;
        if (trace_) {
            acc_name_ = "project_sub";
            acc_fn_ = [&](Slab& sl, int a, int b_, std::vector<Acc>& acc) {
                int lo, hi;
                wr_range(sl, a, b_, lo, hi);
                acc.push_back({sl.field[p], false, a - 1, b_ + 1});
                for (int q : {u, v, w}) {
                    acc.push_back({sl.field[q], false, a, b_});
                    acc.push_back({sl.field[q], true, lo, hi});
                }
            };
        }
        for_planes([&](Slab& sl, int kb, int ke) {
            dim3 block;
            unsigned nblocks;
            const sfk::TileMap m = flat_map(ke - kb, block, nblocks);
            hipLaunchKernelGGL((sfk::project_sub_kernel<T>), dim3(nblocks), block, 0, sl.cur, sl.geom, args(sl), kb, ke, m);
        });
        acc_fn_ = nullptr;
        const int uvw[3] = {u, v, w};
        exchange<3>(uvw);
    }

    int N_, K_, device_;
    int L_ = 1, nranks_ = 1, rank_ = 0, P_ = 1, G_ = 1;
    int fuse_maxvec_ = 512, ovl_mode_ = 1;  // fuse_maxvec_: widest row (vectors) the fused kernels take
    int trap_m_ = 4, trap_extra_ = 0, split_fields_ = 1, tuned_trap_ = -1, tuned_split_ = -1;
    int bound_[4] = {-1, -1, -1, -1};  // resident source slots (sf_bind_sources)
    bool pending_join_ = false, split_enabled_ = true, graphs_ = false;
    std::vector<GraphEntry> graph_cache_;
    int split_ = INT_MAX, gap_ = 0;  // plane-range split of the launch being issued (for_planes)
    T dt_{}, diff_{}, visc_{};
    int num_cu_ = 256;
    int nzl_ = 0, lead_ = 0, px_ = 0, nplanes_ = 0, nt_mode_ = 2;
    int advect_row_ = 1;  // 0 gather form always, 1 one cell per lane for the three velocity components, 2 / 3 always
    bool dead_ishell_ = false;      // the solve being issued may leave its result's i-shell unwritten
    bool dead_ishell_opt_ = true;   // SF_ISHELL=2 switches the dead-shell elision off (1: on, 0: every sweep writes it)
    bool ishell_skip_ = true, zero_skip_ = true, x_is_zero_ = false, fuse_src_ = true;
    bool fuse2_ = true;
    int march_k_ = 1, march_min_planes_ = 12;
    long march_min_cells_ = 2500000, sk2_min_cells_ = 60000000;
    int sk_s_ = 4;
    bool sk_first_ = true;
    bool batch_now_ = false;  // the running solve launches its NF fields as one marching grid (batch_march)
    long plane_ = 0, field_elems_ = 0, pad_front_ = 0, pad_back_ = 0;
    std::vector<Slab> slabs_;
    FILE* trace_ = nullptr;  // SF_TRACE_SCHEDULE
    bool inject_trap_bug_ = false;
    std::map<const void*, int> buf_ids_;
    AccFn acc_fn_;           // accesses of the operator being issued through for_planes (trace only)
    const char* acc_name_ = "op";
    long xchg_seq_ = 0;
    ncclComm_t comm_ = nullptr;
    bool loopback_ = false, rccl_self_ = false;
    long rccl_groups_ = 0;  // RCCL send/recv groups issued so far (sf_schedule_info: proof the transport ran)
    hipEvent_t t0_ = nullptr, t1_ = nullptr;
    T* tr_pos_ = nullptr;
    T* tr_dens_ = nullptr;
    T* tr_speed_ = nullptr;
    int tr_n_ = 0, snap_count_ = 0;
    void* copy_src_ = nullptr;
    void* copy_dst_ = nullptr;
    size_t copy_bytes_ = 0;
};

}  // namespace sfi
