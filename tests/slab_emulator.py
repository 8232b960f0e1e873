"""Test helper: numpy model of the slab-decomposed step (docs/SPEC.md §4) with a pluggable halo transport, following
the PRODUCTION exchange schedule of libsfgpu.so (fluidsolvergpu_amd/csrc/sf_solver.hpp):

  * G = 1 ... 4 ghost planes per side, chosen by the same rule as the Solver constructor (`Schedule.ghost_planes`);
  * lin_solve in passes of S = 1 ... 4 fused sweeps (`Schedule.sweeps_in_launch`, the rule of the same name), each pass
    = S sweep levels of which the first S-1 are RE-COMPUTED on the ghost planes next to the slab (what the marching /
    pair kernels do in registers), then ONE exchange of G planes per side;
  * the folded `add_source` of bound sources (`op_diffuse_src`): the right-hand side x + dt*src exists on the planes the
    first pass computes and on G-1 ghost planes per side (`rhs_on_ghost_planes`), nowhere else;
  * exchanges after project_div (div), after every pass (the iterate), after project_sub (u, v, w together) and after
    advect — each one recorded as (sequence number, G, field slots): the list `SF_TRACE_SCHEDULE` writes for the same
    (N, P, K) on the device (tests/schedule_check.exchange_sequence), so the two can be compared entry by entry.

Planes that the production schedule never makes valid (the outermost ghost plane of a folded right-hand side, the
ghost planes of a fresh iterate before its exchange, ghost planes beyond the physical shell of a wall slab) hold NaN
here: a schedule that read one of them would carry the NaN into the result and fail the comparison with the oracle.

Local arrays have shape (nzl + 2G, N+2, N+2); local plane q is global k = kg0 + q, kg0 = rank*nzl + 1 - G.
Test infrastructure only (the numerics are docs/SPEC.md's, restated with numpy as in tests/test_oracle.py)."""
import numpy as np

SLOT = {"u": 0, "v": 1, "w": 2, "u0": 3, "v0": 4, "w0": 5, "dens": 6, "dens0": 7}  # include/sfgpu.h


class Schedule:
    """Host-side launch-schedule rules of sf_solver.hpp, restated (constructor: ghost planes; sweeps_in_launch;
    sk_first_ok; can_fuse2). Defaults = the library's defaults; keyword arguments = its SF_* switches."""

    def __init__(self, N, P, wsize=4, march=1, sk_s=4, sk_first=1, ishell=1, ghost=4, split=1, march_minp=12,
                 march_mincells_k=None, fuse2=1, fuse_src=1, zero_skip=1, split_fields=1):
        self.N, self.P, self.W = N, P, 16 // wsize
        self.wsize = wsize
        self.nzl = N // P
        self.march, self.sk_s, self.sk_first, self.ishell = march, sk_s, sk_first, ishell
        if march_mincells_k is None:  # the library's default: 2.5 M cells
            march_mincells_k = 2500
        self.split, self.minp, self.mincells = split, march_minp, march_mincells_k * 1000
        self.sk2_mincells = max(60000000 if self.mincells else 0, self.mincells)
        self.fuse2, self.fuse_src, self.zero_skip, self.split_fields = fuse2, fuse_src, zero_skip, split_fields
        fusable = bool(fuse2) and N % self.W == 0 and N // self.W <= 512
        G = 2 if (P > 1 and self.nzl >= 2 and fusable and ghost >= 2) else 1
        for gs in (3, 4):
            interior = self.nzl - 2 * gs
            if (G == gs - 1 and ghost >= gs and march and sk_s >= gs and split and interior >= self.minp
                    and N * N * interior >= self.mincells):
                G = gs
        self.G = G
        self.nplanes = self.nzl + 2 * G

    def can_fuse2(self):
        return bool(self.fuse2) and (self.P == 1 or self.G >= 2) and self.N % self.W == 0 and self.N // self.W <= 512

    def _interior_ok(self, S):
        interior = self.nzl - 2 * max(S, self.G)
        return (self.P == 1 or (self.G >= S and self.split and interior >= self.minp
                                and self.N * self.N * interior >= self.mincells))

    def sk_first_ok(self, K):
        if not (self.sk_first and self.march and self.sk_s >= 4 and self.can_fuse2() and self.ishell and K >= 7):
            return False
        if self.P == 1:
            return self.nzl >= self.minp and self.N * self.N * self.nzl >= self.mincells
        interior = self.nzl - 2 * max(4, self.G)
        return self.G >= 4 and bool(self.split) and interior >= self.minp and self.N * self.N * interior >= self.mincells

    def sweeps_in_launch(self, it, K, continued):
        pair = self.can_fuse2() and it + 2 <= K
        left = K - it
        if it == 0 and not continued and self.sk_first_ok(K):
            return 4
        marching = bool(pair and (it > 0 or continued) and self.march and self.ishell and self.sk_s >= 3 and left >= 3)
        if marching and self.P == 1:
            cells = self.N * self.N * self.nzl
            marching = self.nzl >= self.minp and cells >= self.mincells  # can_sk(.., sweeps = 3)
        if marching and self.sk_s >= 4 and left >= 4 and left not in (5, 6) and self._interior_ok(4):
            return 4
        if marching and left != 4 and self._interior_ok(3):
            return 3
        return 2 if pair else 1

    def passes(self, K, continued=False):
        out, it = [], 0
        while it < K:
            s = self.sweeps_in_launch(it, K, continued)
            out.append(s)
            it += s
        return out

    def fields_split(self):
        """u, v, w solved one field at a time (x, x0, x' of ONE field fit the Infinity Cache) or three per launch."""
        one = 3.0 * (self.N + 2) * (self.N + 2) * self.nplanes * self.wsize
        return self.split_fields == 2 or (self.split_fields == 1 and one <= 0.9 * 256.0 * 1048576.0)


class Slab:
    def __init__(self, N, rank, world, dtype, exchange, schedule=None):
        """exchange(send_lo, send_hi) -> (recv_lo, recv_hi): arrays of G planes each, None at the walls.
        schedule = None: the one-ghost-plane, one-exchange-per-sweep schedule (SF_GHOST=1 / grids the fused kernels do
        not take), as in round 1."""
        self.N, self.rank, self.world = N, rank, world
        self.sch = schedule if schedule is not None else Schedule(N, world, np.dtype(dtype).itemsize, ghost=1, fuse2=0)
        assert self.sch.N == N and self.sch.P == world
        self.G = self.sch.G
        self.nzl = N // world
        self.np_ = self.nzl + 2 * self.G
        self.kg0 = rank * self.nzl + 1 - self.G  # global k of local plane 0
        self.wall_lo, self.wall_hi = rank == 0, rank == world - 1
        self.t = np.dtype(dtype).type
        self.exchange_planes = exchange
        self.xchg_log = []  # (seq, G, field slots) of every exchange, as SF_TRACE_SCHEDULE records them
        # deliberate schedule defects, for the emulator's own test (does it notice?): "rhs_ghost" leaves the folded
        # right-hand side off the ghost planes, "shallow_exchange" ships G - 1 planes per side instead of G
        self.defect = None

    # ---- helpers -----------------------------------------------------------------------------
    def local(self, global_field):
        """The planes this slab stores, cut from a global array; planes outside [0, N+1] (ghosts beyond a wall) are NaN."""
        out = np.full((self.np_, self.N + 2, self.N + 2), np.nan, global_field.dtype)
        for q in range(self.np_):
            k = self.kg0 + q
            if 0 <= k <= self.N + 1:
                out[q] = global_field[k]
        return out

    def owned(self, x):
        """Output planes: interior + the physical shell plane on a wall slab (fluidsolvergpu_amd/dist.output_planes)."""
        G, nzl = self.G, self.nzl
        return x[G - (1 if self.wall_lo else 0):G + nzl + (1 if self.wall_hi else 0)]

    def exchange(self, names, *fields):
        G, nzl = self.G, self.nzl
        self.xchg_log.append((len(self.xchg_log), G, tuple(SLOT[n] for n in names)))
        for x in fields:
            lo, hi = self.exchange_planes(None if self.wall_lo else x[G:2 * G].copy(),
                                          None if self.wall_hi else x[nzl:nzl + G].copy())
            if self.defect == "shallow_exchange":
                lo = None if lo is None else np.concatenate([np.full_like(lo[:1], np.nan), lo[1:]])
                hi = None if hi is None else np.concatenate([hi[:-1], np.full_like(hi[:1], np.nan)])
            if lo is not None:
                x[0:G] = lo
            if hi is not None:
                x[G + nzl:2 * G + nzl] = hi

    def krange(self, ext):
        """Local plane range [lo, hi) of the interior planes widened by `ext` planes per side, cut at the physical
        interior (global k = 1 .. N)."""
        G, nzl = self.G, self.nzl
        lo, hi = G - ext, G + nzl + ext
        lo = max(lo, 1 - self.kg0)
        hi = min(hi, self.N + 1 - self.kg0)
        return lo, hi

    def set_bnd(self, b, x, lo, hi):
        """set_bnd restricted to what depends on the interior planes [lo, hi) (local): their i / j faces and
        z-directed edges; the k face, its edges and corners when the range touches a wall plane."""
        N, t = self.N, self.t
        sx, sy, sz = (t(-1) if b == 1 else t(1)), (t(-1) if b == 2 else t(1)), (t(-1) if b == 3 else t(1))
        half, third = t(0.5), t(1.0 / 3.0)
        I, K = slice(1, N + 1), slice(lo, hi)
        x[K, I, 0] = sx * x[K, I, 1]
        x[K, I, N + 1] = sx * x[K, I, N]
        x[K, 0, I] = sy * x[K, 1, I]
        x[K, N + 1, I] = sy * x[K, N, I]
        walls = []
        if self.kg0 + lo == 1:
            x[lo - 1, I, I] = sz * x[lo, I, I]
            walls.append((lo - 1, lo))
        if self.kg0 + hi - 1 == N:
            x[hi, I, I] = sz * x[hi - 1, I, I]
            walls.append((hi, hi - 1))
        for A, An in ((0, 1), (N + 1, N)):
            for B, Bn in ((0, 1), (N + 1, N)):
                x[K, B, A] = half * (x[K, B, An] + x[K, Bn, A])          # z-directed edge (I=A, J=B)
            for Kw, Kn in walls:
                x[Kw, A, I] = half * (x[Kw, An, I] + x[Kn, A, I])        # x-directed edge (J=A, K=Kw)
                x[Kw, I, A] = half * (x[Kw, I, An] + x[Kn, I, A])        # y-directed edge (I=A, K=Kw)
        for Kw, Kn in walls:
            for J, Jn in ((0, 1), (N + 1, N)):
                for Ii, In in ((0, 1), (N + 1, N)):
                    x[Kw, J, Ii] = third * ((x[Kw, J, In] + x[Kw, Jn, Ii]) + x[Kn, J, Ii])

    def fresh(self, like):
        return np.full_like(like, np.nan)

    # ---- operators (product schedule) ----------------------------------------------------------
    def add_source(self, x, s, dt):
        x[...] = x + self.t(dt) * s  # all stored planes (ghosts of x and s are current); no exchange

    def sweep_levels(self, b, x, x0, a, inv, S, zero=False):
        """One pass of S fused sweeps: level l = 1..S on the interior planes widened by S - l planes per side. Returns
        the new iterate: valid on the interior planes (+ wall shells), NaN on the ghost planes (until exchanged)."""
        N = self.N
        I = slice(1, N + 1)
        prev = np.zeros_like(x) if zero else x
        for l in range(1, S + 1):
            lo, hi = self.krange(S - l)
            cur = self.fresh(x)
            Kc = slice(lo, hi)
            cur[Kc, I, I] = (x0[Kc, I, I] + a * (((prev[Kc, I, 0:N] + prev[Kc, I, 2:N + 2])
                                                  + (prev[Kc, 0:N, I] + prev[Kc, 2:N + 2, I]))
                                                 + (prev[lo - 1:hi - 1, I, I] + prev[lo + 1:hi + 1, I, I]))) * inv
            self.set_bnd(b, cur, lo, hi)
            prev = cur
        return prev

    def lin_solve(self, names, bs, xs, x0s, a, c, K, zero=False, continued=False):
        """names: slot names of xs (for the exchange record). Fields are solved one after the other or together exactly
        as op_lin_solve does (Schedule.fields_split). Returns the new xs."""
        t = self.t
        if len(xs) > 1 and self.sch.fields_split():
            return [self.lin_solve([n], [b], [x], [x0], a, c, K, zero, continued)[0]
                    for n, b, x, x0 in zip(names, bs, xs, x0s)]
        inv, a = t(1) / t(c), t(a)
        it = 0
        while it < K:
            S = self.sch.sweeps_in_launch(it, K, continued)
            assert self.G >= S or self.world == 1, (self.G, S)
            xs = [self.sweep_levels(b, x, x0, a, inv, S, zero and it == 0) for b, x, x0 in zip(bs, xs, x0s)]
            self.exchange(names, *xs)
            it += S
        return xs

    def diffuse_bound(self, names, names0, bs, xs, x0s, srcs, coeff, dt, K):
        """op_diffuse_src: diffuse with add_source folded in (sources bound to resident slots). Returns (xs, x0s)."""
        t, N, G, nzl = self.t, self.N, self.G, self.nzl
        Nf = t(N)
        a = ((t(dt) * t(coeff)) * Nf) * Nf
        c = t(1) + t(6) * a
        if not (self.sch.fuse_src and self.sch.can_fuse2() and K >= 2):
            new0 = []
            for x, src in zip(xs, srcs):  # add_source_bound: x += dt*src, x0 slot = copy of src; then swap
                x[...] = x + t(dt) * src
                new0.append(src.copy())
            xs, x0s = new0, xs
            return self.lin_solve(names, bs, xs, x0s, a, c, K), x0s
        if len(xs) > 1 and self.sch.fields_split():
            outs = [self.diffuse_bound([n], [n0], [b], [x], [x0], [s], coeff, dt, K)
                    for n, n0, b, x, x0, s in zip(names, names0, bs, xs, x0s, srcs)]
            return [o[0][0] for o in outs], [o[1][0] for o in outs]
        S = 4 if self.sch.sk_first_ok(K) else 2
        inv = t(1) / c
        new, rhs_all = [], []
        for b, x, src in zip(bs, xs, srcs):
            # right-hand side: stored on the planes the pass computes and (rhs_on_ghost_planes) on G-1 ghost planes per
            # side; the outermost ghost plane of the x0 slot is never written: NaN
            rhs = self.fresh(x)
            full = x + t(dt) * src
            rhs[G:G + nzl] = full[G:G + nzl]
            if self.world == 1:
                rhs[...] = full  # one slab: the ghosts are the physical shell planes, never read as x0
            elif self.defect != "rhs_ghost":
                rhs[1:G] = full[1:G]
                rhs[G + nzl:G + nzl + G - 1] = full[G + nzl:G + nzl + G - 1]
            new.append(self.sweep_levels(b, src, rhs, a, inv, S))  # iterate 0 = the source (Stam's initial guess)
            rhs_all.append(rhs)
        self.exchange(names, *new)
        return self.lin_solve(names, bs, new, rhs_all, a, c, K - S, continued=True), rhs_all

    def advect(self, names, bs, ds, d0s, u, v, w, dt):
        N, t, nzl, G = self.N, self.t, self.nzl, self.G
        I, Kc = slice(1, N + 1), slice(G, G + nzl)
        Nf = t(N)
        dt0 = t(dt) * Nf
        lo, hi = t(0.5), Nf + t(0.5)
        kk, jj, ii = np.meshgrid(np.arange(G, G + nzl) + self.kg0, np.arange(1, N + 1), np.arange(1, N + 1),
                                 indexing="ij")
        dtp = ds[0].dtype
        x = np.clip(ii.astype(dtp) - dt0 * u[Kc, I, I], lo, hi)
        y = np.clip(jj.astype(dtp) - dt0 * v[Kc, I, I], lo, hi)
        z = np.clip(kk.astype(dtp) - dt0 * w[Kc, I, I], lo, hi)
        i0, j0, k0 = x.astype(np.int64), y.astype(np.int64), z.astype(np.int64)
        s1 = x - i0.astype(dtp)
        s0 = t(1) - s1
        t1 = y - j0.astype(dtp)
        t0 = t(1) - t1
        r1 = z - k0.astype(dtp)
        r0 = t(1) - r1
        kl0 = k0 - self.kg0
        if kl0.min() < 0 or kl0.max() + 1 > self.np_ - 1:
            raise RuntimeError("SF_ERR_HALO_EXCEEDED (emulator)")
        for b, d, d0 in zip(bs, ds, d0s):
            g = lambda i_, j_, k_: d0[k_, j_, i_]
            d[...] = np.nan
            d[Kc, I, I] = (s0 * (t0 * (r0 * g(i0, j0, kl0) + r1 * g(i0, j0, kl0 + 1))
                                 + t1 * (r0 * g(i0, j0 + 1, kl0) + r1 * g(i0, j0 + 1, kl0 + 1)))
                           + s1 * (t0 * (r0 * g(i0 + 1, j0, kl0) + r1 * g(i0 + 1, j0, kl0 + 1))
                                   + t1 * (r0 * g(i0 + 1, j0 + 1, kl0) + r1 * g(i0 + 1, j0 + 1, kl0 + 1))))
            self.set_bnd(b, d, G, G + nzl)
        self.exchange(names, *ds)

    def project(self, u, v, w, K):
        """project(u, v, w; p = slot u0, div = slot v0) in place on u, v, w; returns (p, div)."""
        N, t, nzl, G = self.N, self.t, self.nzl, self.G
        I, Kc = slice(1, N + 1), slice(G, G + nzl)
        Nf = t(N)
        c_div, c_grad = t(-0.5) * (t(1) / Nf), t(0.5) * Nf
        div = self.fresh(u)
        div[Kc, I, I] = c_div * (((u[Kc, I, 2:N + 2] - u[Kc, I, 0:N]) + (v[Kc, 2:N + 2, I] - v[Kc, 0:N, I]))
                                 + (w[G + 1:G + nzl + 1, I, I] - w[G - 1:G + nzl - 1, I, I]))
        self.set_bnd(0, div, G, G + nzl)
        self.exchange(["v0"], div)
        implicit_zero = self.sch.can_fuse2() and K >= 2 and self.sch.zero_skip
        p = np.zeros_like(u)  # (an explicit fill and the implicit zero of the first pass are the same numbers)
        (p,) = self.lin_solve(["u0"], [0], [p], [div], 1, 6, K, zero=bool(implicit_zero))
        u[Kc, I, I] = u[Kc, I, I] - c_grad * (p[Kc, I, 2:N + 2] - p[Kc, I, 0:N])
        v[Kc, I, I] = v[Kc, I, I] - c_grad * (p[Kc, 2:N + 2, I] - p[Kc, 0:N, I])
        w[Kc, I, I] = w[Kc, I, I] - c_grad * (p[G + 1:G + nzl + 1, I, I] - p[G - 1:G + nzl - 1, I, I])
        for b, q in ((1, u), (2, v), (3, w)):
            # ghost planes of u, v, w are stale from here until the exchange
            q[0:G] = np.nan
            q[G + nzl:] = np.nan
            self.set_bnd(b, q, G, G + nzl)
        self.exchange(["u", "v", "w"], u, v, w)
        return p, div

    def step(self, f, dt, diff, visc, K, bound=None):
        """f: dict of the 8 local fields; returns the dict after vel_step + dens_step (names re-bound as in SPEC).
        bound: None, or dict {"u0": src_u, "v0": .., "w0": .., "dens0": ..} of local source arrays (sf_bind_sources)."""
        t, N = self.t, self.N
        Nf = t(N)
        vel, vel0, b123 = ["u", "v", "w"], ["u0", "v0", "w0"], [1, 2, 3]
        if bound is not None:
            xs, x0s = self.diffuse_bound(vel, vel0, b123, [f[n] for n in vel], [f[n] for n in vel0],
                                         [bound[n] for n in vel0], visc, dt, K)
        else:
            for a, s in zip(vel, vel0):
                self.add_source(f[a], f[s], dt)
            a = ((t(dt) * t(visc)) * Nf) * Nf
            x0s = [f[n] for n in vel]  # swap: the summed field becomes the right-hand side ...
            xs = self.lin_solve(vel, b123, [f[n] for n in vel0], x0s, a, t(1) + t(6) * a, K)  # ... the source the iterate
        for n, x, x0 in zip(vel, xs, x0s):
            f[n], f[n + "0"] = x, x0
        f["u0"], f["v0"] = self.project(f["u"], f["v"], f["w"], K)
        for n in vel:
            f[n], f[n + "0"] = f[n + "0"], f[n]
        self.advect(vel, b123, [f[n] for n in vel], [f[n] for n in vel0], f["u0"], f["v0"], f["w0"], dt)
        f["u0"], f["v0"] = self.project(f["u"], f["v"], f["w"], K)
        if bound is not None:
            (x,), (x0,) = self.diffuse_bound(["dens"], ["dens0"], [0], [f["dens"]], [f["dens0"]], [bound["dens0"]], diff,
                                             dt, K)
        else:
            self.add_source(f["dens"], f["dens0"], dt)
            a = ((t(dt) * t(diff)) * Nf) * Nf
            x0 = f["dens"]
            (x,) = self.lin_solve(["dens"], [0], [f["dens0"]], [x0], a, t(1) + t(6) * a, K)
        f["dens"], f["dens0"] = x, x0
        f["dens"], f["dens0"] = f["dens0"], f["dens"]
        self.advect(["dens"], [0], [f["dens"]], [f["dens0"]], f["u"], f["v"], f["w"], dt)
        return f


def run_threads(N, world, dtype, K, glob, schedule_kw=None, bound=False, defect=None, steps=1):
    """All `world` slabs of one decomposed step in this process, one thread per slab, ghost planes handed over through
    queues (the in-process stand-in for the gloo transport of tests/dist_worker.py). Returns (global fields assembled
    from the slabs' output planes, exchange log of slab 0)."""
    import queue
    import threading

    names = ("u", "v", "w", "u0", "v0", "w0", "dens", "dens0")
    q = {(a, b): queue.Queue() for a in range(world) for b in range(world) if abs(a - b) == 1}
    out = {n: np.zeros((N + 2,) * 3, dtype) for n in ("u", "v", "w", "dens")}
    logs, errors = {}, []

    def worker(rank):
        try:
            def exchange(send_lo, send_hi):
                if send_lo is not None:
                    q[(rank, rank - 1)].put(send_lo)
                if send_hi is not None:
                    q[(rank, rank + 1)].put(send_hi)
                lo = q[(rank - 1, rank)].get(timeout=120) if send_lo is not None else None
                hi = q[(rank + 1, rank)].get(timeout=120) if send_hi is not None else None
                return lo, hi

            sch = Schedule(N, world, np.dtype(dtype).itemsize, **(schedule_kw or {}))
            slab = Slab(N, rank, world, dtype, exchange, sch)
            slab.defect = defect
            local = {n: slab.local(glob[n]) for n in names}
            srcs = {n: slab.local(glob[n]) for n in ("u0", "v0", "w0", "dens0")} if bound else None
            for s_ in range(steps):
                if s_ > 0 and not bound:
                    for n in ("u0", "v0", "w0", "dens0"):
                        local[n] = slab.local(glob[n])
                local = slab.step(local, 0.1, 1e-4, 1e-4, K, bound=srcs)
            nzl, G = slab.nzl, slab.G
            kb = rank * nzl + 1 - (1 if slab.wall_lo else 0)
            for n in out:
                part = slab.owned(local[n])
                out[n][kb:kb + part.shape[0]] = part
            logs[rank] = slab.xchg_log
        except Exception as e:  # noqa: BLE001
            errors.append((rank, repr(e)))

    th = [threading.Thread(target=worker, args=(r,)) for r in range(world)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    if errors:
        raise RuntimeError(f"emulator worker failed: {errors}")
    return out, logs[0]
