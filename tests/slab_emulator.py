"""Test helper: numpy model of the slab-decomposed step (docs/SPEC.md §4) with a pluggable halo transport.

It follows the EXCHANGE SCHEDULE of the product (fluidsolvergpu_amd/csrc/sf_api.hip: op_add_source has
no exchange, op_lin_solve exchanges after every sweep, project_div / project_sub / op_advect exchange their
outputs) on local arrays of shape (nzl+2, S, S) — interior planes 1..nzl, ghosts 0 and
nzl+1 — so a world_size-2 gloo run on CPU can check that this schedule reproduces the undecomposed
oracle bit for bit. Test infrastructure only."""
import numpy as np


class Slab:
    def __init__(self, N, rank, world, dtype, exchange):
        self.N, self.rank, self.world = N, rank, world
        self.nzl = N // world
        self.kg0 = rank * self.nzl  # global k of local plane 0
        self.wall_lo, self.wall_hi = rank == 0, rank == world - 1
        self.t = np.dtype(dtype).type
        self.exchange_planes = exchange  # f(send_lo, send_hi) -> (recv_lo, recv_hi); None entries at walls

    # ---- helpers -----------------------------------------------------------------------------
    def local(self, global_field):
        return np.ascontiguousarray(global_field[self.kg0:self.kg0 + self.nzl + 2]).copy()

    def exchange(self, *fields):
        for x in fields:
            lo, hi = self.exchange_planes(None if self.wall_lo else x[1].copy(),
                                          None if self.wall_hi else x[self.nzl].copy())
            if lo is not None:
                x[0] = lo
            if hi is not None:
                x[self.nzl + 1] = hi

    def set_bnd(self, b, x):
        """Shells owned by this slab: i/j faces and z-directed edges of its interior planes; k faces, the other
        edges and the corners only on the wall slabs."""
        N, t, nzl = self.N, self.t, self.nzl
        sx, sy, sz = (t(-1) if b == 1 else t(1)), (t(-1) if b == 2 else t(1)), (t(-1) if b == 3 else t(1))
        half, third = t(0.5), t(1.0 / 3.0)
        I, K = slice(1, N + 1), slice(1, nzl + 1)
        x[K, I, 0] = sx * x[K, I, 1]
        x[K, I, N + 1] = sx * x[K, I, N]
        x[K, 0, I] = sy * x[K, 1, I]
        x[K, N + 1, I] = sy * x[K, N, I]
        walls = []
        if self.wall_lo:
            x[0, I, I] = sz * x[1, I, I]
            walls.append((0, 1))
        if self.wall_hi:
            x[nzl + 1, I, I] = sz * x[nzl, I, I]
            walls.append((nzl + 1, nzl))
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

    # ---- operators (product schedule) ----------------------------------------------------------
    def add_source(self, x, s, dt):
        x[...] = x + self.t(dt) * s  # all stored planes; no exchange

    def lin_solve(self, b, xs, x0s, a, c, K):
        """xs, x0s: lists of fields solved together (NF = len). Returns the new xs."""
        N, t, nzl = self.N, self.t, self.nzl
        I, Kc = slice(1, N + 1), slice(1, nzl + 1)
        inv, a = t(1) / t(c), t(a)
        bs = b if isinstance(b, (list, tuple)) else [b] * len(xs)
        for _ in range(K):
            new = []
            for x, x0, bb in zip(xs, x0s, bs):
                xn = np.zeros_like(x)
                xn[Kc, I, I] = (x0[Kc, I, I] + a * (((x[Kc, I, 0:N] + x[Kc, I, 2:N + 2])
                                                     + (x[Kc, 0:N, I] + x[Kc, 2:N + 2, I]))
                                                    + (x[0:nzl, I, I] + x[2:nzl + 2, I, I]))) * inv
                self.set_bnd(bb, xn)
                new.append(xn)
            self.exchange(*new)
            xs = new
        return xs

    def advect(self, bs, ds, d0s, u, v, w, dt):
        N, t, nzl = self.N, self.t, self.nzl
        I, Kc = slice(1, N + 1), slice(1, nzl + 1)
        Nf = t(N)
        dt0 = t(dt) * Nf
        lo, hi = t(0.5), Nf + t(0.5)
        kk, jj, ii = np.meshgrid(np.arange(1, nzl + 1) + self.kg0, np.arange(1, N + 1), np.arange(1, N + 1),
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
        if kl0.min() < 0 or kl0.max() > nzl:
            raise RuntimeError("SF_ERR_HALO_EXCEEDED (emulator)")
        for b, d, d0 in zip(bs, ds, d0s):
            g = lambda i_, j_, k_: d0[k_, j_, i_]
            d[Kc, I, I] = (s0 * (t0 * (r0 * g(i0, j0, kl0) + r1 * g(i0, j0, kl0 + 1))
                                 + t1 * (r0 * g(i0, j0 + 1, kl0) + r1 * g(i0, j0 + 1, kl0 + 1)))
                           + s1 * (t0 * (r0 * g(i0 + 1, j0, kl0) + r1 * g(i0 + 1, j0, kl0 + 1))
                                   + t1 * (r0 * g(i0 + 1, j0 + 1, kl0) + r1 * g(i0 + 1, j0 + 1, kl0 + 1))))
            self.set_bnd(b, d)
        self.exchange(*ds)

    def project(self, u, v, w, p, div, K):
        N, t, nzl = self.N, self.t, self.nzl
        I, Kc = slice(1, N + 1), slice(1, nzl + 1)
        Nf = t(N)
        c_div, c_grad = t(-0.5) * (t(1) / Nf), t(0.5) * Nf
        p[...] = 0
        div[Kc, I, I] = c_div * (((u[Kc, I, 2:N + 2] - u[Kc, I, 0:N]) + (v[Kc, 2:N + 2, I] - v[Kc, 0:N, I]))
                                 + (w[2:nzl + 2, I, I] - w[0:nzl, I, I]))
        self.set_bnd(0, div)
        self.exchange(div)  # as the product does: div stays in the v0 slot and may be next step's source
        (p_new,) = self.lin_solve(0, [p], [div], 1, 6, K)
        p[...] = p_new
        u[Kc, I, I] = u[Kc, I, I] - c_grad * (p[Kc, I, 2:N + 2] - p[Kc, I, 0:N])
        v[Kc, I, I] = v[Kc, I, I] - c_grad * (p[Kc, 2:N + 2, I] - p[Kc, 0:N, I])
        w[Kc, I, I] = w[Kc, I, I] - c_grad * (p[2:nzl + 2, I, I] - p[0:nzl, I, I])
        self.set_bnd(1, u)
        self.set_bnd(2, v)
        self.set_bnd(3, w)
        self.exchange(u, v, w)

    def step(self, f, dt, diff, visc, K):
        """f: dict of the 8 local fields; returns the dict after vel_step + dens_step (names re-bound as in SPEC)."""
        t, N = self.t, self.N
        Nf = t(N)
        for a, s in (("u", "u0"), ("v", "v0"), ("w", "w0")):
            self.add_source(f[a], f[s], dt)
        f["u"], f["u0"] = f["u0"], f["u"]
        f["v"], f["v0"] = f["v0"], f["v"]
        f["w"], f["w0"] = f["w0"], f["w"]
        a = ((t(dt) * t(visc)) * Nf) * Nf
        f["u"], f["v"], f["w"] = self.lin_solve([1, 2, 3], [f["u"], f["v"], f["w"]], [f["u0"], f["v0"], f["w0"]], a,
                                                t(1) + t(6) * a, K)
        self.project(f["u"], f["v"], f["w"], f["u0"], f["v0"], K)
        f["u"], f["u0"] = f["u0"], f["u"]
        f["v"], f["v0"] = f["v0"], f["v"]
        f["w"], f["w0"] = f["w0"], f["w"]
        self.advect([1, 2, 3], [f["u"], f["v"], f["w"]], [f["u0"], f["v0"], f["w0"]], f["u0"], f["v0"], f["w0"], dt)
        self.project(f["u"], f["v"], f["w"], f["u0"], f["v0"], K)
        self.add_source(f["dens"], f["dens0"], dt)
        f["dens"], f["dens0"] = f["dens0"], f["dens"]
        a = ((t(dt) * t(diff)) * Nf) * Nf
        (f["dens"],) = self.lin_solve(0, [f["dens"]], [f["dens0"]], a, t(1) + t(6) * a, K)
        f["dens"], f["dens0"] = f["dens0"], f["dens"]
        self.advect([0], [f["dens"]], [f["dens0"]], f["u"], f["v"], f["w"], dt)
        return f
