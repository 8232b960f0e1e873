// sf_kernels.hpp — gfx950 (MI355X / CDNA4) kernels of the stable-fluids hot path.
//
// Numerics: docs/SPEC.md, expression by expression (built with -ffp-contract=off; results are
// bit-identical to oracle/stable_fluids_oracle.hpp). Nothing here is derived from the reference,
// which has no grid kernels; the layout idea "one ghost plane of the slowest index"
// is the reference's `buffer = GRIDSIZE*GRIDSIZE` (solver-unidyn.cu:187).
//
// Device layout of one field of one slab (element type T, W = 16/sizeof(T) lanes per vector):
//   planes kl = 0 .. nzl+2G-1 (G ghost planes either side of the nzl interior planes; global k = kg0 + kl)
//   rows   j  = 0 .. N+1
//   row pitch px (multiple of 128 B); cell i lives at row_base + lead + (i-1), with
//   lead*sizeof(T) = 128 B, so interior cell i = 1 starts a 128-byte line in every row and a
//   64-lane wave reading 16 B per lane touches exactly eight whole lines. The shell cell i = 0 sits
//   at lead-1, the shell cell N+1 right after the interior. A plane is one contiguous
//   px*(N+2)-element block — one halo message.
//
// All kernels are HBM-bandwidth bound 7-point stencils / streams: 16-byte coalesced row loads,
// k-marching with the k-1/k/k+1 values of the centre column kept in registers, boundary shells
// (set_bnd) fused into the producing kernel so no separate O(N^2) launch is needed per sweep.
#pragma once
#include <hip/hip_runtime.h>

namespace sfk {

struct Geom {
    int N;        // interior cells per axis
    int nzl;      // interior planes held by this slab (local planes G .. G+nzl-1)
    int G;        // ghost planes per side (1, or 2 when sweep pairs are fused across slabs)
    int np;       // planes stored = nzl + 2G
    int kg0;      // global k of local plane 0 (= first interior k - G)
    int px;       // row pitch, elements
    int lead;     // element offset of cell i = 1 inside a row
    long plane;   // plane stride, elements (= px * (N+2))
    int wall_lo;  // local plane 0 is the physical shell k = 0
    int wall_hi;  // local plane nzl+1 is the physical shell k = N+1
};

template <class T> struct VecT;
template <> struct VecT<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int W = 4;
};
template <> struct VecT<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int W = 2;
};

template <class T>
__device__ __forceinline__ typename VecT<T>::type ldv(const T* p) {
    return *reinterpret_cast<const typename VecT<T>::type*>(p);
}
template <class T>
__device__ __forceinline__ void stv(T* p, typename VecT<T>::type v) {
    *reinterpret_cast<typename VecT<T>::type*>(p) = v;
}

// Value of the neighbouring lane across the whole 64-lane wave, as a DPP move (v_mov_b32 wave_shr:1 /
// wave_shl:1 — a VALU modifier on gfx9-family ISAs incl. gfx950, no LDS round trip as ds_bpermute has).
// lane_up: lane l receives lane l-1 (lane 0 receives 0); lane_dn: lane l receives lane l+1 (lane 63: 0).
// (bound_ctrl: a lane without a source reads 0, so the move needs no initialised destination.)
__device__ __forceinline__ float lane_up(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x138, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_dn(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x130, 0xf, 0xf, true));
}
__device__ __forceinline__ double lane_up(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffLL), 0x138, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x138, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}
__device__ __forceinline__ double lane_dn(double v) {
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_mov_dpp((int)(b & 0xffffffffLL), 0x130, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp((int)(b >> 32), 0x130, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (unsigned int)lo);
}

// Offset of shell cell i = 0 of row (j, kl); cell i is at row0 + i.
__device__ __forceinline__ long row0(const Geom& g, int j, int kl) {
    return (long)kl * g.plane + (long)j * g.px + (g.lead - 1);
}

// Store the nv valid cells of a W-wide vector starting at cell i0 of the row whose i=0 is at `r`.
template <class T, int W>
__device__ __forceinline__ void store_cells(T* __restrict__ f, long r, int i0, const T (&v)[W], int nv) {
    if (W == VecT<T>::W && nv == W) {  // (W < the 16-byte vector: cell by cell)
        typename VecT<T>::type o;
#pragma unroll
        for (int e = 0; e < W; ++e) o[e] = v[e];
        stv(f + r + i0, o);
    } else {
#pragma unroll
        for (int e = 0; e < W; ++e)
            if (e < nv) f[r + i0 + e] = v[e];
    }
}

// set_bnd fused into the producer (SPEC §3 set_bnd): the thread that produced interior cells
// out[0..nv) at (i0.., j, kl) also writes every shell cell that depends only on them.
template <class T, int W>
__device__ __forceinline__ void emit_shells(T* __restrict__ f, const Geom& g, int b, int i0, int j,
                                            int kl, const T (&out)[W], int nv, bool with_i = true) {
    // with_i == false: leave every shell cell with i = 0 or i = N+1 unwritten (intermediate Jacobi
    // sweeps recompute those on the fly instead of spending a whole extra line per row on them)
    const int N = g.N;
    const bool ilo = with_i && (i0 == 1), ihi = with_i && (i0 + nv - 1 == N);
    const bool jlo = (j == 1), jhi = (j == N);
    const int kg = g.kg0 + kl;
    const bool klo = g.wall_lo && kg == 1, khi = g.wall_hi && kg == N;
    if (!(ilo | ihi | jlo | jhi | klo | khi)) return;

    const T sx = (b == 1) ? T(-1) : T(1);
    const T sy = (b == 2) ? T(-1) : T(1);
    const T sz = (b == 3) ? T(-1) : T(1);
    const T half = T(0.5);
    const T third = (T)(1.0 / 3.0);
    const T vlo = out[0];
    T vhi = out[0];
#pragma unroll
    for (int e = 1; e < W; ++e)
        if (e == nv - 1) vhi = out[e];

    const long r = row0(g, j, kl);
    if (ilo) f[r] = sx * vlo;
    if (ihi) f[r + N + 1] = sx * vhi;

#pragma unroll
    for (int sj = 0; sj < 2; ++sj) {
        if (!(sj ? jhi : jlo)) continue;
        const long rj = r + (sj ? (long)g.px : -(long)g.px);
        T t[W];
#pragma unroll
        for (int e = 0; e < W; ++e) t[e] = sy * out[e];
        store_cells<T, W>(f, rj, i0, t, nv);
        if (ilo) f[rj] = half * (sy * vlo + sx * vlo);
        if (ihi) f[rj + N + 1] = half * (sy * vhi + sx * vhi);
    }
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
        if (!(sk ? khi : klo)) continue;
        const long rk = r + (sk ? g.plane : -g.plane);
        T t[W];
#pragma unroll
        for (int e = 0; e < W; ++e) t[e] = sz * out[e];
        store_cells<T, W>(f, rk, i0, t, nv);
        if (ilo) f[rk] = half * (sz * vlo + sx * vlo);
        if (ihi) f[rk + N + 1] = half * (sz * vhi + sx * vhi);
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
            if (!(sj ? jhi : jlo)) continue;
            const long rjk = rk + (sj ? (long)g.px : -(long)g.px);
#pragma unroll
            for (int e = 0; e < W; ++e) t[e] = half * (sz * out[e] + sy * out[e]);
            store_cells<T, W>(f, rjk, i0, t, nv);
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                if (!(si ? ihi : ilo)) continue;
                const T v = si ? vhi : vlo;
                const T ex = half * (sz * v + sy * v);
                const T ey = half * (sz * v + sx * v);
                const T ez = half * (sy * v + sx * v);
                f[rjk + (si ? N + 1 : 0)] = third * ((ex + ey) + ez);
            }
        }
    }
}

// Out-of-line form for kernels that are short of registers: only wall-adjacent threads call it.
template <class T, int W>
__device__ __attribute__((noinline)) void emit_shells_call(T* __restrict__ f, const Geom& g, int b, int i0,
                                                           int j, int kl, typename VecT<T>::type o, bool with_i) {
    T out[W];
#pragma unroll
    for (int e = 0; e < W; ++e) out[e] = o[e];
    emit_shells<T, W>(f, g, b, i0, j, kl, out, W, with_i);
}

// Thread -> (vector column, row). Returns false if the thread is outside the grid.
template <int W>
__device__ __forceinline__ bool thread_cell(const Geom& g, int& i0, int& j, int& nv) {
    i0 = 1 + W * (int)(blockIdx.x * blockDim.x + threadIdx.x);
    j = 1 + (int)(blockIdx.y * blockDim.y + threadIdx.y);
    if (i0 > g.N || j > g.N) return false;
    nv = g.N - i0 + 1;
    nv = nv > W ? W : nv;
    return true;
}

// ---------------------------------------------------------------------------------------------
// add_source: x += dt*s over every stored element (SPEC: "all S^3 entries"; pads hold zeros).
// Pure stream: 3 words per element.
template <class T, int NF>
struct AddSourceArgs {
    T* x[NF];
    const T* s[NF];
    T dt;
    long nvec;  // vectors per field
};

template <class T, int NF>
__global__ void __launch_bounds__(256) add_source_kernel(AddSourceArgs<T, NF> A) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    // one thread per 16 bytes of every field, the whole grid in memory order (what a stream needs to reach
    // ~6 TB/s here, tools/membench.hip); all loads of the thread are requested before the first store
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.nvec) return;
    V a[NF], s[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        a[f] = ldv(A.x[f] + q * W);
        s[f] = ldv(A.s[f] + q * W);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int e = 0; e < W; ++e) a[f][e] = a[f][e] + A.dt * s[f][e];
        stv(A.x[f] + q * W, a[f]);
    }
}

// add_source with the source taken from a resident slot: x += dt*src and, in the same pass, s_copy = src, which
// is exactly "copy src into the x0 slot, then add_source(x, x0)" (4 words per element instead of 5).
template <class T, int NF>
struct AddSourceBoundArgs {
    T* x[NF];
    T* s_copy[NF];
    const T* src[NF];
    T dt;
    long nvec;
};

template <class T, int NF>
__global__ void __launch_bounds__(256) add_source_bound_kernel(AddSourceBoundArgs<T, NF> A) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.nvec) return;
    V a[NF], s[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        a[f] = ldv(A.x[f] + q * W);
        s[f] = ldv(A.src[f] + q * W);
    }
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int e = 0; e < W; ++e) a[f][e] = a[f][e] + A.dt * s[f][e];
        stv(A.x[f] + q * W, a[f]);
        stv(A.s_copy[f] + q * W, s[f]);
    }
}

// out = a + dt*s on up to two plane ranges (the add_source expression): the right-hand side of a folded
// add_source on the ghost planes next to a slab, which the source pair itself does not store.
template <class T, int NF>
struct RhsPlanesArgs {
    T* out[NF];
    const T* a[NF];
    const T* s[NF];
    T dt;
    long off[2];  // element offset of each range
    long nvec;    // vectors per range
};

template <class T, int NF>
__global__ void __launch_bounds__(256) rhs_planes_kernel(RhsPlanesArgs<T, NF> A) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q >= A.nvec) return;
    const long p = A.off[blockIdx.y] + q * W;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        V a = ldv(A.a[f] + p);
        const V s = ldv(A.s[f] + p);
#pragma unroll
        for (int e = 0; e < W; ++e) a[e] = a[e] + A.dt * s[e];
        stv(A.out[f] + p, a);
    }
}

// ---------------------------------------------------------------------------------------------
// lin_solve: one Jacobi sweep + fused set_bnd (SPEC §3 lin_solve). Algorithmic traffic 3 words
// per cell (read x, read x0, write x'). Each thread owns a W-wide column piece and marches kchunk
// planes keeping x[k-1], x[k], x[k+1] in registers; j±1 rows and the two i-neighbours come from
// L1/L2 (they are lines this block or its neighbour has just loaded).
template <class T, int NF>
struct JacobiArgs {
    const T* x[NF];
    const T* x0[NF];
    T* xn[NF];
    int b[NF];
    T a, inv;
    // jacobi2_kernel<..., SRC = true> only (add_source folded into the first sweep pair of diffuse): x is the
    // source s (Stam's initial guess), x0 is the field BEFORE add_source, the right-hand side x0 + dt*s is formed in
    // registers and stored to x0out for the later pairs
    T* x0out[NF];
    T dt;
};

// Register-blocked flat sweep (the production Jacobi kernel).
//   * 1-D grid in memory order, one workgroup = (tx lanes x W cells) x (ty*RJ rows) x RK planes, so the
//     lines in flight form a compact moving window (what a plain copy needs to reach ~6.2 TB/s here);
//   * XCD banding: workgroups are dealt round-robin over the 8 XCDs (b % 8 labels the XCD group), so
//     the j-tiles are cut into 8 contiguous bands and group e only ever works on band e — its j+-1
//     and k+-1 neighbour lines were fetched by the same XCD and hit in its own 4 MiB L2 instead of
//     crossing the fabric. Placement affects speed only, never results;
//   * each thread produces RJ x RK vectors from (RK+2)*RJ + 2*RK loads of x (instead of 5 per
//     vector) and gets its i+-1 neighbours from the adjacent lanes by wave shuffle — measured, the
//     two per-lane dword loads for i+-1 cost as much as the k+-1 vector loads;
//   * the i = 0 / N+1 shell cells are recomputed in registers in all but the first sweep and written
//     only by the last one (they would cost a whole extra 128-byte line per row end otherwise).
struct TileMap {
    int gx;            // tiles along i
    int gy;            // tiles along j
    int band;          // j-tiles per XCD band (0: plain order)
    int nxcd;          // 8 when banded
    int ishell_mem;    // read x[0], x[N+1] from memory (first sweep: x is caller data)
    int ishell_write;  // write the i = 0 / N+1 shell cells of x' (last sweep)
    // A launch covers `ke - kb` logical planes t = 0 .. ; plane t sits at kb + t + (t >= split ? gap : 0), which
    // lets ONE launch sweep the first and the last planes of a slab (the halo-critical ones). split is a multiple
    // of the kernel's plane block; no split: split = INT_MAX, gap = 0.
    int split;
    int gap;
    int rows;   // jacobi2_kernel: row strips per workgroup
    int strip;  // jacobi2_kernel: lanes reserved per strip (>= N/W; N/W itself packs strips densely, a multiple of
                // 64 keeps every strip aligned to wave boundaries at the price of idle lanes)
    int strip_shift;       // log2(strip) when strip is a power of two, else -1 (spares the kernel a division)
    unsigned nvec_magic;   // floor(2^32 / (N/W)) + 1: t / (N/W) == umulhi(t, magic) for the t < 2^16 of the OVL mapping
    // jacobi2_kernel: every thread touches one 128-byte line of the tile pf_dz plane blocks ahead at the same (j, i)
    // position — the lines only that tile brings in: its own rows of x and x0 — so that the workgroup which gets
    // there ~32 workgroups later finds them in L2 / Infinity Cache instead of HBM. 0 = off.
    int pf_dz;
};

__device__ __forceinline__ int plane_of(const TileMap& m, int kb, int t) {
    return kb + t + (t >= m.split ? m.gap : 0);
}

// Workgroup -> (i-tile, j-tile, plane, extra) for the 1-D banded grid. Returns false for padding tiles.
__device__ __forceinline__ bool flat_tile(const TileMap& m, int nk, int& it, int& jt, int& kk, int& f) {
    int r = (int)blockIdx.x;
    if (m.band > 0) {
        const int xcd = r % m.nxcd;
        r /= m.nxcd;
        it = r % m.gx;
        r /= m.gx;
        jt = xcd * m.band + r % m.band;
        r /= m.band;
    } else {
        it = r % m.gx;
        r /= m.gx;
        jt = r % m.gy;
        r /= m.gy;
    }
    kk = r % nk;
    f = r / nk;
    return jt < m.gy;
}

// Thread -> first cell, row and plane for the one-vector-per-thread kernels on the flat grid.
template <int W>
__device__ __forceinline__ bool flat_cell(const Geom& g, const TileMap& m, int kb, int ke, int& i0, int& j,
                                          int& kl, int& nv) {
    int it, jt, kk, f;
    if (!flat_tile(m, ke - kb, it, jt, kk, f)) return false;
    kl = plane_of(m, kb, kk);
    i0 = 1 + W * (it * (int)blockDim.x + (int)threadIdx.x);
    j = 1 + jt * (int)blockDim.y + (int)threadIdx.y;
    if (i0 > g.N || j > g.N) return false;
    nv = g.N - i0 + 1;
    nv = nv > W ? W : nv;
    return true;
}

template <class T, int NF, bool NT, int RJ, int RK>
__global__ void __launch_bounds__(256) jacobi_rb_kernel(Geom g, JacobiArgs<T, NF> A, int kb, int ke,
                                                         TileMap m) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    int it, jt, kk, f;
    {
        int r = (int)blockIdx.x;
        if (m.band > 0) {
            const int xcd = r % m.nxcd;
            r /= m.nxcd;
            it = r % m.gx;
            r /= m.gx;
            jt = xcd * m.band + r % m.band;
            r /= m.band;
        } else {
            it = r % m.gx;
            r /= m.gx;
            jt = r % m.gy;
            r /= m.gy;
        }
        const int nkg = (ke - kb + RK - 1) / RK;
        kk = r % nkg;
        f = r / nkg;  // field index: the slowest grid dimension (uniform per workgroup)
        if (jt >= m.gy) return;
    }
    const int N = g.N;
    const int k0 = plane_of(m, kb, kk * RK);
    {
        const int left = (ke - kb) - kk * RK;  // logical planes left in this launch from this block on
        ke = k0 + (left < RK ? left : RK);     // from here on: physical end of this block's planes
    }
    const int i0 = 1 + W * (it * (int)blockDim.x + (int)threadIdx.x);
    const int j0 = 1 + (jt * (int)blockDim.y + (int)threadIdx.y) * RJ;
    if (i0 > N || j0 > N) return;
    int nv = N - i0 + 1;
    nv = nv > W ? W : nv;
    const T a = A.a, inv = A.inv;
    const T* __restrict__ x = A.x[0];
    const T* __restrict__ x0 = A.x0[0];
    T* __restrict__ xn = A.xn[0];
    int b = A.b[0];
#pragma unroll
    for (int ff = 1; ff < NF; ++ff)
        if (f == ff) {
            x = A.x[ff];
            x0 = A.x0[ff];
            xn = A.xn[ff];
            b = A.b[ff];
        }
    const T sx = (b == 1) ? T(-1) : T(1);

    // addresses: rows beyond N+1 / planes beyond ke are clamped (their results are never stored)
    long rowq[RJ + 2];  // q of rows j0-1 .. j0+RJ in plane 0 (without the plane term)
#pragma unroll
    for (int r = 0; r < RJ + 2; ++r) {
        int j = j0 - 1 + r;
        j = j > N + 1 ? N + 1 : j;
        rowq[r] = (long)j * g.px + (g.lead - 1) + i0;
    }
    long planeq[RK + 2];
#pragma unroll
    for (int r = 0; r < RK + 2; ++r) {
        int kl = k0 - 1 + r;
        kl = kl > ke ? ke : kl;
        planeq[r] = (long)kl * g.plane;
    }

    V X[RK + 2][RJ], Jlo[RK], Jhi[RK], S[RK][RJ];
#pragma unroll
    for (int r = 0; r < RK + 2; ++r)
#pragma unroll
        for (int rj = 0; rj < RJ; ++rj) X[r][rj] = ldv(x + planeq[r] + rowq[rj + 1]);
#pragma unroll
    for (int rk = 0; rk < RK; ++rk) {
        Jlo[rk] = ldv(x + planeq[rk + 1] + rowq[0]);
        Jhi[rk] = ldv(x + planeq[rk + 1] + rowq[RJ + 1]);
#pragma unroll
        for (int rj = 0; rj < RJ; ++rj) {
            const T* sp = x0 + planeq[rk + 1] + rowq[rj + 1];
            S[rk][rj] = NT ? __builtin_nontemporal_load(reinterpret_cast<const V*>(sp)) : ldv(sp);
        }
    }

    const bool has_left = ((int)threadIdx.x & 63) != 0;
    const bool has_right = (((int)threadIdx.x + 1) & 63) != 0 && (int)threadIdx.x + 1 < (int)blockDim.x;
    const bool first_vec = (i0 == 1), last_vec = (i0 + W - 1 >= N);
    // end cells no lane of this wave holds: requested here, together with the vector loads (a load at the point of
    // use would be a dependent memory round trip per output position)
    const bool left_mem = first_vec ? m.ishell_mem != 0 : !has_left;
    const bool right_mem = last_vec ? m.ishell_mem != 0 : !has_right;
    T XL[RK][RJ], XR[RK][RJ];
#pragma unroll
    for (int rk = 0; rk < RK; ++rk)
#pragma unroll
        for (int rj = 0; rj < RJ; ++rj) XL[rk][rj] = XR[rk][rj] = T(0);
    if (left_mem) {
#pragma unroll
        for (int rk = 0; rk < RK; ++rk)
#pragma unroll
            for (int rj = 0; rj < RJ; ++rj) XL[rk][rj] = x[planeq[rk + 1] + rowq[rj + 1] - 1];
    }
    if (right_mem) {
#pragma unroll
        for (int rk = 0; rk < RK; ++rk)
#pragma unroll
            for (int rj = 0; rj < RJ; ++rj) XR[rk][rj] = x[planeq[rk + 1] + rowq[rj + 1] + W];
    }

#pragma unroll
    for (int rk = 0; rk < RK; ++rk) {
        const int kl = k0 + rk;
#pragma unroll
        for (int rj = 0; rj < RJ; ++rj) {
            const int j = j0 + rj;
            V c = X[rk + 1][rj];
            // i-neighbours: adjacent lanes hold the adjacent vectors (shuffles run on every lane)
            const T up = lane_up(c[W - 1]);
            const T dn = lane_dn(c[0]);
            if (kl >= ke || j > N) continue;  // wave-uniform for tx >= 64; lanes of other rows otherwise
            const long q = planeq[rk + 1] + rowq[rj + 1];
            T last = c[0];
#pragma unroll
            for (int e = 1; e < W; ++e)
                if (e == nv - 1) last = c[e];
            T xm, xp;
            if (first_vec)
                xm = m.ishell_mem ? XL[rk][rj] : sx * c[0];
            else
                xm = has_left ? up : XL[rk][rj];
            if (last_vec) {
                if (m.ishell_mem) {
                    xp = XR[rk][rj];
                } else {
                    xp = sx * last;
                    if (nv < W) {  // the shell cell N+1 lies inside this vector: patch it
#pragma unroll
                        for (int e = 1; e < W; ++e)
                            if (e == nv) c[e] = sx * last;
                    }
                }
            } else {
                xp = has_right ? dn : XR[rk][rj];
            }
            const V km = X[rk][rj], kp = X[rk + 2][rj];
            const V jm = (rj == 0) ? Jlo[rk] : X[rk + 1][rj > 0 ? rj - 1 : 0];
            const V jp = (rj == RJ - 1) ? Jhi[rk] : X[rk + 1][rj < RJ - 1 ? rj + 1 : RJ - 1];
            const V s = S[rk][rj];
            T out[W];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const T left = (e == 0) ? xm : c[e - 1];
                const T right = (e == W - 1) ? xp : c[e + 1];
                out[e] = (s[e] + a * (((left + right) + (jm[e] + jp[e])) + (km[e] + kp[e]))) * inv;
            }
            if (NT && nv == W) {
                V o;
#pragma unroll
                for (int e = 0; e < W; ++e) o[e] = out[e];
                __builtin_nontemporal_store(o, reinterpret_cast<V*>(xn + q));
            } else {
                store_cells<T, W>(xn, q - i0, i0, out, nv);
            }
            emit_shells<T, W>(xn, g, b, i0, j, kl, out, nv, m.ishell_write != 0);
        }
    }
}

// Two Jacobi sweeps per pass through HBM (temporal blocking, T = 2).
// x'' = J(J(x)) computed in one kernel: each thread produces RJ x RK output vectors and recomputes, in
// registers, the first-sweep values y = J(x) on the cross-shaped neighbourhood its outputs need
// (dist <= 1 around the RJ x RK block; x itself is read at dist <= 2). Redundant first-sweep work is
// (RJ+2)(RK+2)-4 over RJ*RK vectors (3x at 2x2) while HBM traffic per sweep halves: x, x0 are read once and x''
// is written once for two sweeps. (With the traffic halved the kernel is bound by instruction issue and latency at
// its two waves per SIMD — about 760 vector + 280 scalar instructions per wave and 1,024 cells — so instruction
// count matters: see the notes at the load phase, the grid decode and the wall selects.)
// The first sweep's set_bnd is applied in registers (y on a wall row/plane is +-y of the adjacent
// interior row/plane; its i-shell is sx*y[1], sx*y[N]), so the result is bit-identical to two separate
// sweeps. In the strip mappings a 256-thread workgroup holds whole row strips; the only values that cross waves
// are the end cells of y (and of x, see XLDS), exchanged through a few words of LDS. The overlapped mapping (OVL)
// has no cross-wave values at all.
// Requirements (checked by the launcher): N % W == 0; strips: a row fits one workgroup (N/W <= 256); decomposed
// grids need two ghost planes per side (G = 2).
#ifndef SF_J2_WAVES
#define SF_J2_WAVES 2
#endif
// XLDS: hand the x end cells across wave seams through LDS (one extra barrier) instead of masked per-lane loads.
// Pays when rows start anywhere inside a wave (row width not a multiple of 64 vectors: both ends of most waves
// are seams); with rows of 64 / 128 vectors each wave has at most one seam and the loads are cheaper.
// XZ: the incoming iterate is identically zero (project's p = 0): no x is loaded at all — the same operations
// are applied to literal zeros, so the bits equal a sweep over a zero-filled field — and the caller can skip the
// memset of p.
// OVL: seam-free mapping for rows of any width. The (row pair, vector) items of a plane pair are numbered in memory
// order and every wave takes 60 consecutive ones in lanes 2..61; lanes 0, 1 and 62, 63 hold the two items before /
// after them and only feed the shuffles (x of the outer one, y of the inner one), so nothing crosses waves: no
// LDS, no barrier, and 94 % of the lanes produce output whatever N/W is (a row end inside a wave is handled by the
// same in-register i-shell logic as everywhere else).
#ifndef SF_OVL_OUT_N
#define SF_OVL_OUT_N 60
#endif
constexpr int SF_OVL_OUT = SF_OVL_OUT_N;
constexpr int SF_OVL_LO = (64 - SF_OVL_OUT_N) / 2;  // first output lane
// SRC: the first pair of diffuse with add_source folded in. `x` is the source array s, `x0` the field before
// add_source; the right-hand side rhs = x0 + dt*s (the expression of add_source_kernel) is formed in registers at
// the 12 positions the first sweep needs — s there is already held as x — and stored at the thread's own output
// positions to `x0out`, where the later pairs read it. Saves the add_source pass (4 words per cell) for one extra
// word written here. Only interior cells of rhs are stored: lin_solve never reads rhs on the shell.
template <class T, int NF, bool NT, int RJ, int RK, bool XLDS, bool XZ = false, bool OVL = false, bool SRC = false>
__global__ void __launch_bounds__(256, SF_J2_WAVES) jacobi2_kernel(Geom g, JacobiArgs<T, NF> A, int kb, int ke,
                                                       TileMap m) {
    static_assert(!(OVL && XLDS), "the overlapped mapping has no seams");
    static_assert(!(SRC && XZ), "a source pair has a non-zero iterate");
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    constexpr int NPOS = RJ * RK;
    constexpr int NYPOS = (RK + 2) * (RJ + 2);
    __shared__ T sh_first[4][NPOS];    // [wave][output position]: y of the wave's first cell
    __shared__ T sh_last[4][NPOS];     //                          y of the wave's last cell
    __shared__ T shx_first[4][NYPOS];  // [wave][first-sweep position]: x of the wave's first / last cell
    __shared__ T shx_last[4][NYPOS];
    // 3-D grid (x fastest in dispatch order, so workgroup x of a banded launch runs on XCD x): x = XCD group (or the
    // j-tile when not banded), y = j-tile inside the band, z = plane block (+ field). No division to decode.
    int jt, kk, f;
    {
        jt = m.band > 0 ? (int)blockIdx.x * m.band + (int)blockIdx.y : (int)blockIdx.x;
        if (NF == 1) {
            kk = (int)blockIdx.z;
            f = 0;
        } else {
            const int nkg = (ke - kb + RK - 1) / RK;
            kk = (int)blockIdx.z % nkg;
            f = (int)blockIdx.z / nkg;
        }
    }
    const int N = g.N;
    const int nvec = N / W;
    const int k0 = plane_of(m, kb, kk * RK);
    const int nlogical = ke - kb;  // logical planes of this launch
    {
        const int left = (ke - kb) - kk * RK;
        ke = k0 + (left < RK ? left : RK);  // physical end of this block's planes
    }
    const bool tile_ok = jt < m.gy;  // uniform per workgroup
    // 256 threads in a line hold m.rows row strips of nvec vectors each: rows need not start at a wave
    // boundary, so a wave seam can fall anywhere inside a row (handled through LDS below)
    const int tid = (int)threadIdx.x;
    int vec, j0;
    bool active;
    if (OVL) {
        if (!tile_ok) return;  // no barrier in this mapping
        const int total = ((N + RJ - 1) / RJ) * nvec;
        int t = (jt * 4 + (tid >> 6)) * SF_OVL_OUT + (tid & 63) - SF_OVL_LO;
        active = t >= 0 && t < total && (tid & 63) >= SF_OVL_LO && (tid & 63) < SF_OVL_LO + SF_OVL_OUT;
        t = t < 0 ? 0 : (t >= total ? total - 1 : t);  // feeder / padding lanes run on valid addresses
        const int rg = nvec == 1 ? t : (int)__umulhi((unsigned)t, m.nvec_magic);  // t / nvec
        vec = t - rg * nvec;
        j0 = 1 + rg * RJ;
    } else {
        const int strip = m.strip_shift >= 0 ? (tid >> m.strip_shift) : tid / m.strip;
        vec = tid - strip * m.strip;
        j0 = 1 + (jt * m.rows + strip) * RJ;
        active = tile_ok && strip < m.rows && vec < nvec && j0 <= N;
        // out-of-range threads keep running on clamped (valid) addresses so that every wave reaches the
        // barrier and every DPP source lane is alive; they store nothing
        vec = vec < nvec ? vec : nvec - 1;
        j0 = (tile_ok && strip < m.rows && j0 <= N) ? j0 : N;
    }
    const int i0 = 1 + W * vec;
    const T a = A.a, inv = A.inv;
    const T* __restrict__ x = A.x[0];
    const T* __restrict__ x0 = A.x0[0];
    T* __restrict__ xn = A.xn[0];
    T* __restrict__ x0out = A.x0out[0];
    int b = A.b[0];
#pragma unroll
    for (int ff = 1; ff < NF; ++ff)
        if (f == ff) {
            x = A.x[ff];
            x0 = A.x0[ff];
            xn = A.xn[ff];
            x0out = A.x0out[ff];
            b = A.b[ff];
        }
    const T sx = (b == 1) ? T(-1) : T(1);
    const T sy = (b == 2) ? T(-1) : T(1);
    const T sz = (b == 3) ? T(-1) : T(1);

    // rows j0-2 .. j0+RJ+1 and planes k0-2 .. k0+RK+1, clamped into the stored range
    long rowq[RJ + 4];
#pragma unroll
    for (int r = 0; r < RJ + 4; ++r) {
        int j = j0 - 2 + r;
        j = j < 0 ? 0 : (j > N + 1 ? N + 1 : j);
        rowq[r] = (long)j * g.px + (g.lead - 1) + i0;
    }
    // (one 64-bit multiply for plane k0, then clamped steps of one plane: the stored range is [0, np-1])
    long planeq[RK + 4];
    planeq[2] = (long)k0 * g.plane;
#pragma unroll
    for (int r = 1; r >= 0; --r) planeq[r] = planeq[r + 1] - ((k0 - 2 + r >= 0) ? g.plane : 0L);
#pragma unroll
    for (int r = 3; r < RK + 4; ++r) planeq[r] = planeq[r - 1] + ((k0 - 2 + r <= g.np - 1) ? g.plane : 0L);
    // distance of block-local coordinate a in [-2, R+1] from the output range [0, R-1]
#define SF_DIST(a_, R_) ((a_) < 0 ? -(a_) : ((a_) > (R_)-1 ? (a_) - ((R_)-1) : 0))

    const int lane = tid & 63;
    const int wave = tid >> 6;
    const bool first_vec = (vec == 0), last_vec = (vec == nvec - 1);
    // neighbour vector lives in the same wave (always, for the lanes that matter, in the overlapped mapping)
    const bool has_left = OVL || lane != 0, has_right = OVL || lane != 63;
    const bool multi_wave = !OVL && (64 % m.strip) != 0;  // some row crosses a wave boundary

    // ---- every load of the thread is requested here, before the first use, in the order the first sweep consumes
    // them (loads complete in order): end cells, right-hand sides, then the x vectors position by position, so the
    // arithmetic of the first positions overlaps the flight of the later vectors.
    // End cells of x that no lane of this wave holds (the i-shell cells of caller data in a first sweep; the cell
    // across a wave seam when seams are not handed over through LDS). Loading each one where it is used costs a
    // dependent memory round trip per first-sweep position (12 in a row: the compiler cannot hoist a load out of
    // its divergent branch).
    const bool left_mem = !XZ && (first_vec ? m.ishell_mem != 0 : (!has_left && !XLDS));
    const bool right_mem = !XZ && (last_vec ? m.ishell_mem != 0 : (!has_right && !XLDS));
    const bool mirror_l = first_vec && m.ishell_mem == 0, mirror_r = last_vec && m.ishell_mem == 0;
    T XL[NYPOS], XR[NYPOS];
#pragma unroll
    for (int pos = 0; pos < NYPOS; ++pos) XL[pos] = XR[pos] = T(0);
    if (left_mem) {
#pragma unroll
        for (int c = -1; c <= RK; ++c)
#pragma unroll
            for (int r = -1; r <= RJ; ++r)
                if (SF_DIST(c, RK) + SF_DIST(r, RJ) <= 1)
                    XL[(c + 1) * (RJ + 2) + (r + 1)] = x[planeq[c + 2] + rowq[r + 2] - 1];
    }
    if (right_mem) {
#pragma unroll
        for (int c = -1; c <= RK; ++c)
#pragma unroll
            for (int r = -1; r <= RJ; ++r)
                if (SF_DIST(c, RK) + SF_DIST(r, RJ) <= 1)
                    XR[(c + 1) * (RJ + 2) + (r + 1)] = x[planeq[c + 2] + rowq[r + 2] + W];
    }
    V S[RK + 2][RJ + 2];
#pragma unroll
    for (int c = -1; c <= RK; ++c)
#pragma unroll
        for (int r = -1; r <= RJ; ++r)
            if (SF_DIST(c, RK) + SF_DIST(r, RJ) <= 1) S[c + 1][r + 1] = ldv(x0 + planeq[c + 2] + rowq[r + 2]);
    V X[RK + 4][RJ + 4];
    {
        bool have[RK + 4][RJ + 4] = {};  // resolved at compile time after unrolling
#pragma unroll
        for (int c = -1; c <= RK; ++c)
#pragma unroll
            for (int r = -1; r <= RJ; ++r) {
                if (SF_DIST(c, RK) + SF_DIST(r, RJ) > 1) continue;
#pragma unroll
                for (int t = 0; t < 5; ++t) {  // the stencil of first-sweep position (c, r)
                    const int cc = c + (t == 1 ? -1 : (t == 2 ? 1 : 0)), rr = r + (t == 3 ? -1 : (t == 4 ? 1 : 0));
                    if (have[cc + 2][rr + 2]) continue;
                    have[cc + 2][rr + 2] = true;
                    if (XZ) {
#pragma unroll
                        for (int e = 0; e < W; ++e) X[cc + 2][rr + 2][e] = T(0);
                    } else {
                        X[cc + 2][rr + 2] = ldv(x + planeq[cc + 2] + rowq[rr + 2]);
                    }
                }
            }
    }
    // L2 / Infinity-Cache warm-up for the workgroup that will run pf_dz plane blocks further on (see TileMap). The
    // eight threads whose vectors share a 128-byte line take one (array, plane, row) combination each, so every
    // line of that tile's own rows is requested once; requested last so that no wait for the vectors above includes
    // it; the value is never used.
    static_assert(RJ == 2 && RK == 2, "warm-up line assignment assumes a 2x2 block");
    int pf = 0;  // one dword is enough to bring the line in, whatever T is
    if (!XZ && m.pf_dz > 0 && (kk + m.pf_dz) * RK + RK <= nlogical) {
        const int c = vec & 7;
        const int kpf = plane_of(m, kb, (kk + m.pf_dz) * RK + ((c >> 1) & 1));
        pf = *reinterpret_cast<const int*>(((c & 1) ? x0 : x) + (long)kpf * g.plane + rowq[2 + (c >> 2)]);
    }
    // nothing below may be scheduled above this point and no load below it: all requests are out before any use
    __builtin_amdgcn_sched_barrier(0);
    if (SRC) {
        const T dt = A.dt;
#pragma unroll
        for (int c = -1; c <= RK; ++c)
#pragma unroll
            for (int r = -1; r <= RJ; ++r)
                if (SF_DIST(c, RK) + SF_DIST(r, RJ) <= 1) {
#pragma unroll
                    for (int e = 0; e < W; ++e)
                        S[c + 1][r + 1][e] = S[c + 1][r + 1][e] + dt * X[c + 2][r + 2][e];  // add_source: x + dt*s
                }
    }
    // ---- x end cells cross waves through LDS (cheaper than two masked per-lane loads per position) ---------
    if (XLDS && !XZ && multi_wave) {
#pragma unroll
        for (int c = -1; c <= RK; ++c)
#pragma unroll
            for (int r = -1; r <= RJ; ++r) {
                if (SF_DIST(c, RK) + SF_DIST(r, RJ) > 1) continue;
                if (lane == 0) shx_first[wave][(c + 1) * (RJ + 2) + (r + 1)] = X[c + 2][r + 2][0];
                if (lane == 63) shx_last[wave][(c + 1) * (RJ + 2) + (r + 1)] = X[c + 2][r + 2][W - 1];
            }
        __syncthreads();
    }

    // ---- first sweep: y on the cross-shaped neighbourhood -------------------------------------------
    V Y[RK + 2][RJ + 2];
#pragma unroll
    for (int c = -1; c <= RK; ++c)
#pragma unroll
        for (int r = -1; r <= RJ; ++r) {
            if (SF_DIST(c, RK) + SF_DIST(r, RJ) > 1) continue;
            const V cc = X[c + 2][r + 2];
            const T up = lane_up(cc[W - 1]);
            const T dn = lane_dn(cc[0]);
            const int pos = (c + 1) * (RJ + 2) + (r + 1);
            // straight selects on loop-invariant lane masks (no divergent branches: each costs exec-mask traffic)
            T xm = up, xp = dn;
            if (XZ) {
                xm = T(0);
                xp = T(0);
            } else {
                if (XLDS && multi_wave) {
                    xm = has_left ? up : shx_last[wave > 0 ? wave - 1 : 0][pos];
                    xp = has_right ? dn : shx_first[wave < 3 ? wave + 1 : 3][pos];
                }
                xm = left_mem ? XL[pos] : xm;
                xp = right_mem ? XR[pos] : xp;
                xm = mirror_l ? sx * cc[0] : xm;
                xp = mirror_r ? sx * cc[W - 1] : xp;
            }
            const V km = X[c + 1][r + 2], kp = X[c + 3][r + 2];
            const V jm = X[c + 2][r + 1], jp = X[c + 2][r + 3];
            const V s = S[c + 1][r + 1];
            V y;
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const T left = (e == 0) ? xm : cc[e - 1];
                const T right = (e == W - 1) ? xp : cc[e + 1];
                y[e] = (s[e] + a * (((left + right) + (jm[e] + jp[e])) + (km[e] + kp[e]))) * inv;
            }
            Y[c + 1][r + 1] = y;
        }

    // ---- end cells of y cross waves through LDS (only when a row is wider than one wave) ---------------
    if (multi_wave) {
#pragma unroll
        for (int rk = 0; rk < RK; ++rk)
#pragma unroll
            for (int rj = 0; rj < RJ; ++rj) {
                if (lane == 0) sh_first[wave][rk * RJ + rj] = Y[rk + 1][rj + 1][0];
                if (lane == 63) sh_last[wave][rk * RJ + rj] = Y[rk + 1][rj + 1][W - 1];
            }
        __syncthreads();
    }

    // ---- second sweep --------------------------------------------------------------------------------
    // Most waves touch no wall: two wave-uniform tests (supersets of the per-output conditions) let them skip the
    // per-element wall selects and the shell stores altogether — about a sixth of the kernel's instructions
    const bool near_jk = (j0 <= 1) | (j0 + RJ - 1 >= N) | (g.wall_lo && g.kg0 + k0 <= 1) |
                         (g.wall_hi && g.kg0 + k0 + RK - 1 >= N);
    const bool wave_walls = __builtin_amdgcn_ballot_w64(near_jk) != 0ull;
    const bool wave_shells =
        wave_walls || (m.ishell_write != 0 && __builtin_amdgcn_ballot_w64(first_vec | last_vec) != 0ull);
#pragma unroll
    for (int rk = 0; rk < RK; ++rk) {
        const int kl = k0 + rk;
        const int kg = g.kg0 + kl;
#pragma unroll
        for (int rj = 0; rj < RJ; ++rj) {
            const int j = j0 + rj;
            const V yc = Y[rk + 1][rj + 1];
            const T up = lane_up(yc[W - 1]);
            const T dn = lane_dn(yc[0]);
            T ym = up, yp = dn;
            if (multi_wave) {
                ym = has_left ? up : sh_last[wave > 0 ? wave - 1 : 0][rk * RJ + rj];
                yp = has_right ? dn : sh_first[wave < 3 ? wave + 1 : 3][rk * RJ + rj];
            }
            ym = first_vec ? sx * yc[0] : ym;
            yp = last_vec ? sx * yc[W - 1] : yp;
            if (!active || kl >= ke || j > N) continue;
            V jm = Y[rk + 1][rj], jp = Y[rk + 1][rj + 2], km = Y[rk][rj + 1], kp = Y[rk + 2][rj + 1];
            if (wave_walls) {  // first-sweep set_bnd on the j / k walls
                if (j == 1) jm = sy * yc;
                if (j == N) jp = sy * yc;
                if (g.wall_lo && kg == 1) km = sz * yc;
                if (g.wall_hi && kg == N) kp = sz * yc;
            }
            const V s = S[rk + 1][rj + 1];
            T out[W];
#pragma unroll
            for (int e = 0; e < W; ++e) {
                const T left = (e == 0) ? ym : yc[e - 1];
                const T right = (e == W - 1) ? yp : yc[e + 1];
                out[e] = (s[e] + a * (((left + right) + (jm[e] + jp[e])) + (km[e] + kp[e]))) * inv;
            }
            const long q = planeq[rk + 2] + rowq[rj + 2];
            V o;
#pragma unroll
            for (int e = 0; e < W; ++e) o[e] = out[e];
            if (NT)
                __builtin_nontemporal_store(o, reinterpret_cast<V*>(xn + q));
            else
                stv(xn + q, o);
            if (SRC) stv(x0out + q, s);
            if (wave_shells) emit_shells<T, W>(xn, g, b, i0, j, kl, out, W, m.ishell_write != 0);
        }
    }
    if (!XZ) asm volatile("" ::"v"(pf));  // keeps the warm-up load; it completed long ago (loads return in order)
#undef SF_DIST
}

// Lane vector of the marching kernel (jacobi_sk_kernel below): WL cells = 8 bytes (two floats / one double) — half the
// register state per lane of the 16-byte vectors used elsewhere, which is what lets two waves share a SIMD — four with
// two rows per wave (fp32 four-sweep launches: 16 waves per workgroup, Solver::SK4_TJ / SK4_NW).
// A buffer offset that is out of range of every plane resource under either reading of the range check (with or without
// the instruction's scalar offset, at most a plane, added in): dropped stores, loads that return 0.
// Diagnostic builds of tools/sk_probe.hip only (the library is built without it): -DSF_SK_DIAG=1 drops the stores of
// the marching loop, 3 replaces its loads by register copies, 4 does both — what does a march step cost with no memory
// operation in it? (profiles/r03_marching_kernel_experiments.md §6: 88 % of the full step.) Results are garbage.
#ifndef SF_SK_DIAG
#define SF_SK_DIAG 0
#endif
constexpr unsigned SK_OOB = 0x80000000u;
// Rows of padding the host allocates before / after every field for the marching kernel: the rows of a workgroup's tile
// are addressed as (row 0 of the lane) + r rows, not clamped into the plane, so the first j-block reaches S-1 rows
// below j = 0 and the last one up to NW x TJ - S - 2 rows beyond j = N+1 (of the first / last plane: elsewhere that is
// the neighbouring plane). Nothing is stored there and what is loaded only feeds rows that are not stored.
constexpr int SK_PAD_ROWS_FRONT = 4, SK_PAD_ROWS_BACK = 64;

template <class T, int WL>
struct LaneVec {
    typedef T type __attribute__((ext_vector_type(WL)));
};

// Plane-relative addressing of the marching kernel: a buffer resource whose base is the (wave-uniform) start of one
// plane plus a 32-bit per-lane byte offset (`buffer_load ... offen`): no 64-bit address arithmetic in vector registers.
__device__ __forceinline__ __amdgpu_buffer_rsrc_t plane_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0x7fffffff, 0x00020000);
}
template <class T, int WL>
__device__ __forceinline__ typename LaneVec<T, WL>::type buf_load(__amdgpu_buffer_rsrc_t r, unsigned off,
                                                                  unsigned soff = 0) {
    // soff: a wave-uniform byte offset added by the instruction itself (the `soffset` scalar operand)
    typedef typename LaneVec<T, WL>::type VW;
    constexpr int B = WL * (int)sizeof(T);
    static_assert(B == 4 || B == 8 || B == 16, "lane vector must be 4, 8 or 16 bytes");
    if constexpr (B == 4) {
        return __builtin_bit_cast(VW, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
    } else if constexpr (B == 8) {
        typedef int I2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(VW, (I2)__builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
    } else {
        typedef int I4 __attribute__((ext_vector_type(4)));
        return __builtin_bit_cast(VW, (I4)__builtin_amdgcn_raw_buffer_load_b128(r, off, soff, 0));
    }
}
template <class T, int WL, bool NT>
__device__ __forceinline__ void buf_store(__amdgpu_buffer_rsrc_t r, unsigned off, typename LaneVec<T, WL>::type v,
                                          unsigned soff = 0) {
    constexpr int B = WL * (int)sizeof(T);
    constexpr int AUX = NT ? 2 : 0;  // nt
    if constexpr (B == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, soff, AUX);
    } else if constexpr (B == 8) {
        typedef int I2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(I2, v), r, off, soff, AUX);
    } else {
        typedef int I4 __attribute__((ext_vector_type(4)));
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(I4, v), r, off, soff, AUX);
    }
}
template <class T>
__device__ __forceinline__ void buf_store1(__amdgpu_buffer_rsrc_t r, unsigned off, T v, unsigned soff = 0) {
    if constexpr (sizeof(T) == 4) {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), r, off, soff, 0);
    } else {
        typedef int I2 __attribute__((ext_vector_type(2)));
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(I2, v), r, off, soff, 0);
    }
}

template <class T>
__device__ __forceinline__ T buf_load1(__amdgpu_buffer_rsrc_t r, unsigned off, unsigned soff = 0) {  // out of range: 0
    if constexpr (sizeof(T) == 4) {
        return __builtin_bit_cast(T, __builtin_amdgcn_raw_buffer_load_b32(r, off, soff, 0));
    } else {
        typedef int I2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(T, (I2)__builtin_amdgcn_raw_buffer_load_b64(r, off, soff, 0));
    }
}

// set_bnd fused into the marching kernel: emit_shells with every address formed from a plane resource (wave-uniform)
// and the byte offsets of rows j-1, j, j+1 the lane already holds — no address registers of its own. Same
// expressions, same bits. `o` holds the WL interior cells at row j of plane ko; rm / rc / rp: planes ko-1, ko, ko+1.
template <class T, int WL>
__device__ __forceinline__ void j2k_shells(__amdgpu_buffer_rsrc_t rm, __amdgpu_buffer_rsrc_t rc,
                                           __amdgpu_buffer_rsrc_t rp, unsigned offc, unsigned offm, unsigned offp,
                                           typename LaneVec<T, WL>::type o, T sx, T sy, T sz, bool jlo, bool jhi,
                                           bool klo, bool khi, bool ilo, bool ihi) {
    typedef typename LaneVec<T, WL>::type VW;
    constexpr unsigned LO = (unsigned)(-(int)sizeof(T)), HI = WL * (unsigned)sizeof(T);  // cells i = 0 / N+1
    const T half = T(0.5);
    const T third = (T)(1.0 / 3.0);
    const T vlo = o[0], vhi = o[WL - 1];
    if (ilo) buf_store1<T>(rc, offc + LO, sx * vlo);
    if (ihi) buf_store1<T>(rc, offc + HI, sx * vhi);
#pragma unroll
    for (int sj = 0; sj < 2; ++sj) {
        if (!(sj ? jhi : jlo)) continue;
        const unsigned oj = sj ? offp : offm;
        VW t;
#pragma unroll
        for (int e = 0; e < WL; ++e) t[e] = sy * o[e];
        buf_store<T, WL, false>(rc, oj, t);
        if (ilo) buf_store1<T>(rc, oj + LO, half * (sy * vlo + sx * vlo));
        if (ihi) buf_store1<T>(rc, oj + HI, half * (sy * vhi + sx * vhi));
    }
#pragma unroll
    for (int sk = 0; sk < 2; ++sk) {
        if (!(sk ? khi : klo)) continue;
        const __amdgpu_buffer_rsrc_t rk = sk ? rp : rm;
        VW t;
#pragma unroll
        for (int e = 0; e < WL; ++e) t[e] = sz * o[e];
        buf_store<T, WL, false>(rk, offc, t);
        if (ilo) buf_store1<T>(rk, offc + LO, half * (sz * vlo + sx * vlo));
        if (ihi) buf_store1<T>(rk, offc + HI, half * (sz * vhi + sx * vhi));
#pragma unroll
        for (int sj = 0; sj < 2; ++sj) {
            if (!(sj ? jhi : jlo)) continue;
            const unsigned oj = sj ? offp : offm;
#pragma unroll
            for (int e = 0; e < WL; ++e) t[e] = half * (sz * o[e] + sy * o[e]);
            buf_store<T, WL, false>(rk, oj, t);
#pragma unroll
            for (int si = 0; si < 2; ++si) {
                if (!(si ? ihi : ilo)) continue;
                const T v = si ? vhi : vlo;
                const T ex = half * (sz * v + sy * v);
                const T ey = half * (sz * v + sx * v);
                const T ez = half * (sy * v + sx * v);
                buf_store1<T>(rk, oj + (si ? HI : LO), third * ((ex + ey) + ez));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// S fused sweeps per pass through HBM (S = 2 or 3), k-marching, LDS halo exchange between the waves of a workgroup.
//
// Why (measured, profiles/r02_pair_kernel_issue.md): jacobi2_kernel recomputes the first sweep on a 12-position cross
// per 4 outputs and requests 36 vectors per 4 outputs; at 512^3 it is bound by instruction issue and the L1 path at two
// waves per SIMD (3.6 TB/s of HBM traffic). A first k-marching form (a lane owns a column of rows, j+-1 in its own
// registers, i+-1 by DPP, k+-1 by marching; 2.2x fewer instructions) reached 6.1 TB/s of fabric traffic with a third
// of its reads being overlap rows and chunk ends: memory-bound. So the lever left is bytes requested per sweep:
//   * S sweeps per pass: x, x0 read once and x^(S) written once for S sweeps (3 words per cell / S);
//   * the NW waves of a workgroup are stacked in j, each owning TJ rows of the same 64-lane column: the row above /
//     below a wave's rows — of x and of every intermediate sweep level — comes from its neighbour wave through LDS
//     (two 8-byte rows per level, wave and step, double-buffered, ONE barrier per step), so inside a workgroup no row
//     is loaded twice and no intermediate value is computed twice. Only the workgroup's outer S rows per side (and S
//     lanes per side of each wave, and S-1 planes per chunk end per level) are redundant: they are computed by
//     everybody's code and simply not stored (a wave never branches on its position: the barrier needs lockstep).
//   * per lane: register rings of four planes for x, x0 and each intermediate level (TJ rows of 8 bytes); ring
//     positions are compile-time constants of the four-fold unrolled march (PH = step mod 4).
// Step kk of the march (kk = k0-S+1 .. k1+S-2):   requests x(kk+2), x0(kk+1);
//   level l = 1..S computes plane kk-l+1 from level l-1 (x for l = 1) on planes kk-l, kk-l+1, kk-l+2, rows -1 and TJ
//   from LDS (published one step earlier); level S is stored. Level l starts 2(l-1) steps into the chunk.
// set_bnd between sweeps is applied in registers exactly as in jacobi2_kernel (i: sx*, j/k walls: sy*, sz* of the
// adjacent interior value), so the result is bit-identical to S separate sweeps.
// Mapping: a j-block holds V = NW*TJ - 2S output rows; the (j-block, vector) items are numbered in memory order and a
// workgroup takes P = 64 - 2*ceil(S/WL) consecutive ones in the middle lanes (the overlapped mapping of jacobi2_kernel,
// any row width); the outer lanes only feed the shuffles.
// (left + right) of a two-cell lane vector whose i-neighbours sit in the adjacent lanes: e0 = c1[lane-1] + c1,
// e1 = c0 + c0[lane+1]. For float the lane shift rides on the add itself (v_add_f32 with a DPP source: two
// instructions instead of two DPP moves, a packed add and their wait states). Same operands, same IEEE sums.
// The leading s_nop covers the "VALU write -> DPP read" hazard, which hipcc cannot see inside an asm statement.
__device__ __forceinline__ void lr_sum2(float c0, float c1, float& e0, float& e1) {
    asm("s_nop 1\n\t"
        "v_add_f32_dpp %0, %2, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_add_f32_dpp %1, %3, %3 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "=&v"(e0), "=&v"(e1)
        : "v"(c1), "v"(c0));
}

struct SkMap {
    int ncb;    // column blocks (workgroups per chunk)
    int band;   // workgroups per XCD band (grid = 8 x band x nchunks*NF)
    int kc;     // output planes per chunk
    int njb;    // j-blocks
    unsigned nvec_magic;
    int gap;    // boundary launch of a decomposed grid: chunk 1 starts `gap` planes after the end of chunk 0
};

template <class T, int WL, int S, int TJ, int NW>
struct SkShared {
    typename LaneVec<T, WL>::type edge[2][S][NW][2][64];  // [buffer][level 0..S-1][wave][first/last row][lane]
};

// The FIRST pass of a solve inside the marching kernel (FIRST != 0; four-sweep launches only). Its iterate is not the
// output of an earlier sweep, so level 1 differs:
//   1  caller data: the i = 0 / N+1 shell cells of x are READ (one unconditional dword load per row whose offset is
//      out of range for every lane that is not at a row end), not recomputed;
//   2  diffuse with add_source folded in: x is the source array (Stam's initial guess, i-shell read as in 1), x0 the
//      field before add_source; rhs = x0 + dt * x (the expression of add_source_kernel) replaces x0 for every level
//      the moment its plane arrives and is stored to x0out for the later launches (interior cells of the chunk);
//   3  project: the iterate is identically zero — no x is requested at all, level 1 is evaluated on literal zeros.
// Where a lane of a j-wall workgroup meets the walls. Row t of a tile (t = wave x TJ + r) is j = jb x V + 1 - S + t:
// j = 1 is row S of the first j-block and j = N row jtN of the last one — the same row for every such lane, so "is this
// row at the wall" is a wave-uniform test of t combined with a lane mask, not a per-lane comparison per row.
struct SkWall {
    bool lo, hi;  // the lane belongs to the first / last j-block
    bool hi2;     // ... to the one before the last: its upper halo rows can reach j = N (tile row jtN + V)
    int jtN;      // tile row of j = N in the last j-block (rows beyond it do not exist)
};

// Diagnostic build only (tools/sk_probe.hip, -DSF_SK_STAMP): s_memtime stamps at eight points of every march step,
// summed per wave: where do the cycles of a step go? No stamp executes in libsfgpu.so. (A stamp orders memory
// operations only: hipcc moves vector arithmetic across it, so the split BETWEEN the levels is approximate.)
#ifdef SF_SK_STAMP
__device__ unsigned long long* g_sk_stamp = nullptr;  // [workgroup][wave][10]
#define SF_SK_T(q)                                                \
    do {                                                          \
        const long long t_ = __builtin_amdgcn_s_memtime();        \
        fx.st[q] += (unsigned long long)(t_ - fx.tlast);          \
        fx.tlast = t_;                                            \
    } while (0)
#else
#define SF_SK_T(q) do { } while (0)
#endif

template <class T, int TJ>
struct SkFirst {
    T xs[4][TJ];         // ring of the row-end shell cells of x (modes 1, 2)
    T* __restrict__ prhs;  // mode 2: plane kk of x0out (clamped like px)
    T dt;
    int k0, k1;
#ifdef SF_SK_STAMP
    unsigned long long st[10];
    long long tlast;
#endif
};

template <class T, int WL, bool NT, int S, int TJ, int NW, int PH, int NACT, bool WALLS, bool ISH, bool ROWEND, int FIRST>
__device__ __forceinline__ void jsk_step(const Geom& g, SkShared<T, WL, S, TJ, NW>& sh,
                                         typename LaneVec<T, WL>::type (&xr)[S == 4 ? 3 : 4][TJ],
                                         typename LaneVec<T, WL>::type (&yr)[S - 1][S == 4 ? 3 : 4][TJ],
                                         typename LaneVec<T, WL>::type (&sr)[S == 4 ? 5 : 4][TJ],
                                         unsigned row0, unsigned stcol, const T* __restrict__& px,
                                         const T* __restrict__& ps0, T* __restrict__& pout, int kk, T a, T inv, T sx,
                                         T sy, T sz, SkWall jw, int wave, int lane, bool first_vec, bool last_vec,
                                         SkFirst<T, TJ>& fx) {
    typedef typename LaneVec<T, WL>::type VW;
    static_assert(FIRST == 0 || S == 4, "first passes exist as four-sweep launches only");
    const int N = g.N;
    const int kmax = g.np - 1;
    constexpr unsigned OOBL = SK_OOB;
    constexpr int RB = (PH + 1) & 1, WB = PH & 1;  // LDS buffer read (written one step ago) / written in this step
    // Row addressing: row r of the lane's tile is (row0: the byte offset of its row 0) + r rows, the r rows through
    // the instruction's scalar offset — one register for TJ rows. Row 0 of the first j-block lies S-1 rows below the
    // plane, so every plane resource starts S-1 rows early (jsk_march shifts the field pointers) and row0 counts from there (a buffer offset cannot be
    // negative); rows outside the plane read the padding or the neighbouring plane (SK_PAD_ROWS_*). A lane / row that
    // must not be stored — feeder lanes: stcol is out of range; the outer S rows of the tile: wave-uniform; on a j
    // wall also rows beyond N: per lane — carries the out-of-range offset instead.
    const unsigned pxb = (unsigned)g.px * (unsigned)sizeof(T);
    auto prs = [&](const T* pl) -> __amdgpu_buffer_rsrc_t { return plane_rsrc(pl); };  // (pointers arrive shifted)
    auto rv = [&](int) -> unsigned { return row0; };
    auto rs_ = [&](int r) -> unsigned { return (unsigned)r * pxb; };
    auto row_stored = [&](int r) -> bool {  // wave-uniform
        const int jt = wave * TJ + r;
        return jt >= S && jt < NW * TJ - S;
    };
    // (these offsets are loop-invariant and hipcc holds one per row, shell cell and face through the march; hiding
    // stcol from the optimizer once per step frees those registers, but measured 2 % slower at four rows per wave)
    const unsigned stc = stcol;
    auto st_off = [&](int r) -> unsigned {
        if constexpr (!WALLS) return row_stored(r) ? stc : SK_OOB;
        else return (row_stored(r) && !(jw.hi && wave * TJ + r > jw.jtN)) ? stc : SK_OOB;
    };
    // on a j wall: is row r of this lane j = 1 / j = N? (SkWall: a wave-uniform test of the row and a lane mask)
    auto at_jlo = [&](int r) -> bool { return WALLS && jw.lo && wave * TJ + r == S; };
    auto at_jhi = [&](int r) -> bool {
        constexpr int V = NW * TJ - 2 * S;
        return WALLS && ((jw.hi && wave * TJ + r == jw.jtN) || (jw.hi2 && wave * TJ + r == jw.jtN + V));
    };
    constexpr bool XSH = S == 4;  // x in a three-slot shift register (0 = plane kk-1, 1 = kk, 2 = kk+1), see below
    // (1) requests for the next step: x(kk+2), x0(kk+1). px / ps0 point at those planes and advance by one plane per
    // step (held at the last stored plane: the values requested beyond it are never used)
    const __amdgpu_buffer_rsrc_t rd = prs(px);
    // (held at plane 0 while the index is still negative — a chunk that starts at the first plane reaches S-1
    // planes below it — and at the last stored plane beyond it: values requested outside are never used)
    px += (kk + 2 >= 0 && kk + 2 < kmax) ? g.plane : 0;
    auto request_x = [&]() {
        constexpr int slot = XSH ? 2 : ((PH + 2) & 3);
        if constexpr (FIRST != 3) {
#pragma unroll
#if SF_SK_DIAG == 3 || SF_SK_DIAG == 4
            for (int r = 0; r < TJ; ++r) { xr[slot][r] = xr[0][r]; asm volatile("" : "+v"(xr[slot][r])); }
#else
            for (int r = 0; r < TJ; ++r) xr[slot][r] = buf_load<T, WL>(rd, rv(r), rs_(r));
#endif
        }
        if constexpr ((FIRST == 1 || FIRST == 2) && ROWEND) {
#pragma unroll
            for (int r = 0; r < TJ; ++r) {
                const unsigned off = first_vec ? rv(r) - (unsigned)sizeof(T)
                                               : (last_vec ? rv(r) + WL * (unsigned)sizeof(T) : OOBL);
                fx.xs[(PH + 2) & 3][r] = buf_load1<T>(rd, off, rs_(r));
            }
        }
    };
    {
        const __amdgpu_buffer_rsrc_t rs = prs(ps0);
        ps0 += (kk + 1 >= 0 && kk + 1 < kmax) ? g.plane : 0;
        // S = 4: the request for x is issued after level 1 (below), into the slot level 1 has just released
        if constexpr (!XSH) request_x();
        if constexpr (S == 4) {
            // five planes of x0 are live with four levels (one in flight): not a divisor of the four-fold unrolled
            // march, so x0 is a shift register: slot l serves level l, slot 0 receives the request
#pragma unroll
            for (int l = 4; l >= 1; --l)
#pragma unroll
                for (int r = 0; r < TJ; ++r) sr[l][r] = sr[l - 1][r];
#pragma unroll
#if SF_SK_DIAG == 3 || SF_SK_DIAG == 4
            for (int r = 0; r < TJ; ++r) { sr[0][r] = sr[4][r]; asm volatile("" : "+v"(sr[0][r])); }
#else
            for (int r = 0; r < TJ; ++r) sr[0][r] = buf_load<T, WL>(rs, rv(r), rs_(r));
#endif
        } else {
#pragma unroll
            for (int r = 0; r < TJ; ++r) sr[(PH + 1) & 3][r] = buf_load<T, WL>(rs, rv(r), rs_(r));
        }
    }
    // ring slots of x(kk-1), x(kk), x(kk+1)
    constexpr int XM = XSH ? 0 : ((PH + 3) & 3), XC = XSH ? 1 : ((PH + 0) & 3), XP = XSH ? 2 : ((PH + 1) & 3);
    if constexpr (FIRST == 2) {
        // x0(kk) has arrived (slot 1 after the shift) and x(kk) = the source is the centre plane of level 1: form the
        // right-hand side of this plane once, for every level, and store it where the later launches read it
        const bool inr = kk >= fx.k0 && kk < fx.k1;  // wave-uniform: planes of this chunk
        const __amdgpu_buffer_rsrc_t rr = prs(fx.prhs);
        fx.prhs += (kk >= 0 && kk < kmax) ? g.plane : 0;
#pragma unroll
        for (int r = 0; r < TJ; ++r) {
            VW rhs;
#pragma unroll
            for (int e = 0; e < WL; ++e) rhs[e] = sr[1][r][e] + fx.dt * xr[XC][r][e];
            sr[1][r] = rhs;
            buf_store<T, WL, false>(rr, inr ? st_off(r) : OOBL, rhs, rs_(r));
        }
    }
    // (2) the neighbours' edge rows of every active source level (published in the previous step)
    const int wlo = wave > 0 ? wave - 1 : 0, whi = wave < NW - 1 ? wave + 1 : NW - 1;
    // (read one level ahead of their use — level 1's here, level l+1's while level l computes — so that only two
    // pairs are live at a time: with four levels all eight at once cost the registers that decide between four and
    // five rows per wave)
    VW hm[S], hp[S];
    auto read_halo = [&](int l) {
        if (FIRST == 3 && l == 1) {
#pragma unroll
            for (int e = 0; e < WL; ++e) {
                hm[0][e] = T(0);
                hp[0][e] = T(0);
            }
        } else {
            hm[l - 1] = sh.edge[RB][l - 1][wlo][1][lane];
            hp[l - 1] = sh.edge[RB][l - 1][whi][0][lane];
        }
    };
    if constexpr (S == 4) {
        read_halo(1);
    } else {
#pragma unroll
        for (int l = 1; l <= NACT; ++l) read_halo(l);
    }
    SF_SK_T(0);  // requests issued, halo reads issued
    __builtin_amdgcn_sched_barrier(0);
    // (3) levels 1 .. NACT
#pragma unroll
    for (int l = 1; l <= NACT; ++l) {
        // source level l-1 on planes kk-l (km), kk-l+1 (centre), kk-l+2 (kp); this level's plane is kk-l+1
        const int pl = kk - l + 1;
        const int kg = g.kg0 + pl;
        T* __restrict__ po = pout;  // plane kk-S+1: only used by level S
        const __amdgpu_buffer_rsrc_t rc = prs(po);
        if constexpr (S == 4) {
            if (l + 1 <= NACT) read_halo(l + 1);
        }
        // S = 4: the intermediate levels live in three-slot shift registers (0 = the plane written in this step, 1, 2
        // = the two before it) instead of four-slot rings — a v_mov is free in a request-bound kernel, the fifteen
        // vectors it saves are what lets a wave hold five rows
        constexpr bool YSH = S == 4;
        if constexpr (YSH) {
            if (l < S) {
#pragma unroll
                for (int r = 0; r < TJ; ++r) {
                    yr[l - 1 < S - 1 ? l - 1 : 0][2][r] = yr[l - 1 < S - 1 ? l - 1 : 0][1][r];
                    yr[l - 1 < S - 1 ? l - 1 : 0][1][r] = yr[l - 1 < S - 1 ? l - 1 : 0][0][r];
                }
            }
        }
        VW olast[TJ];  // level S: the rows as stored
#pragma unroll
        for (int r = 0; r < TJ; ++r) {
            VW cc, km, kp, jm, jp;
            if (l == 1 && FIRST == 3) {
#pragma unroll
                for (int e = 0; e < WL; ++e) cc[e] = km[e] = kp[e] = jm[e] = jp[e] = T(0);
            } else if (l == 1) {
                cc = xr[XC][r];
                km = xr[XM][r];
                kp = xr[XP][r];
                jm = r > 0 ? xr[XC][r > 0 ? r - 1 : 0] : hm[0];
                jp = r < TJ - 1 ? xr[XC][r < TJ - 1 ? r + 1 : 0] : hp[0];
            } else {
                const int q = l >= 2 ? l - 2 : 0;
                constexpr int NS = S == 4 ? 3 : 4;
                const int sc = YSH ? 1 : ((PH + 5 - l) & 3), sm = YSH ? 2 : ((PH + 4 - l) & 3),
                          sp = YSH ? 0 : ((PH + 6 - l) & 3);
                cc = yr[q][sc % NS][r];
                km = yr[q][sm % NS][r];
                kp = yr[q][sp % NS][r];
                jm = r > 0 ? yr[q][sc % NS][r > 0 ? r - 1 : 0] : hm[l - 1];
                jp = r < TJ - 1 ? yr[q][sc % NS][r < TJ - 1 ? r + 1 : 0] : hp[l - 1];
            }
            // (left + right): the i-neighbours of the lane's end cells live in the adjacent lanes; at a row end they
            // are the i-shell cells sx * (end cell) instead (ROWEND: does this workgroup hold a row end at all?)
            VW lr;
            // the i-shell cell at a row end: sx * (end cell) — the set_bnd of the previous sweep, recomputed — or, in
            // level 1 of a first pass over caller data, the cell as it stands in memory
            const bool shell_mem = (FIRST == 1 || FIRST == 2) && l == 1;
            const T xsv = shell_mem ? fx.xs[(PH + 0) & 3][r] : T(0);
            if (FIRST == 3 && l == 1) {
#pragma unroll
                for (int e = 0; e < WL; ++e) lr[e] = T(0) + T(0);
            } else if constexpr (WL == 2 && sizeof(T) == 4) {
                T e0, e1;
                lr_sum2(cc[0], cc[1], e0, e1);
                if constexpr (ROWEND) {
                    e0 = first_vec ? (shell_mem ? xsv : sx * cc[0]) + cc[1] : e0;
                    e1 = last_vec ? cc[0] + (shell_mem ? xsv : sx * cc[1]) : e1;
                }
                lr[0] = e0;
                lr[1] = e1;
            } else {
                const T up = lane_up(cc[WL - 1]);
                const T dn = lane_dn(cc[0]);
                const T cm = (ROWEND && first_vec) ? (shell_mem ? xsv : sx * cc[0]) : up;
                const T cp = (ROWEND && last_vec) ? (shell_mem ? xsv : sx * cc[WL - 1]) : dn;
#pragma unroll
                for (int e = 0; e < WL; ++e) {
                    const T left = (e == 0) ? cm : cc[e > 0 ? e - 1 : 0];
                    const T right = (e == WL - 1) ? cp : cc[e < WL - 1 ? e + 1 : WL - 1];
                    lr[e] = left + right;
                }
            }
            if (l >= 2) {  // set_bnd of the source level on the j walls (k walls: after the rows of the producing level)
                if (WALLS) {
                    // (as selects: a wave-uniform branch per row around them measured 7 % slower at 256^3)
                    if (at_jlo(r)) jm = sy * cc;
                    if (at_jhi(r)) jp = sy * cc;
                }
            }
            const VW s = (S == 4) ? sr[l < 5 ? l : 0][r] : sr[(PH + 5 - l) & 3][r];
            // (whole lane vectors: the two cells of a lane pair up in the packed instructions — left to itself the SLP
            // vectoriser pairs cells of different rows and pays for it in register moves)
            const VW o = (s + a * ((lr + (jm + jp)) + (km + kp))) * inv;
            if (l < S) {
                yr[l - 1 < S - 1 ? l - 1 : 0][YSH ? 0 : ((PH + 5 - l) & 3) % (S == 4 ? 3 : 4)][r] = o;
            } else {
                // st_off(r): the row's store offset, or an out-of-range one if this lane / row must not store (feeder
                // lane, outer S rows of the workgroup's tile, row beyond N)
                constexpr unsigned OOB = SK_OOB;
                // No branch around the stores (a "stored / not stored" join makes hipcc wait for every older store at
                // the next s_waitcnt): lanes that must not store carry an offset beyond the resource's range, and the
                // hardware drops an out-of-range buffer store.
                {
                    const unsigned offv = st_off(r);
                    const unsigned so = rs_(r);
                    // the i = 0 / N+1 shell cell of a row end: ONE store instruction per row serves whichever end the
                    // lane sits at (a lane is never both: rows hold >= 2 vectors), and only workgroups that hold a
                    // row end issue it at all. (The last pass of a solve still costs 25-35 % more than a plain one at
                    // 512^3: N^2 x planes x 2 four-byte writes, each alone in its 128-byte line — cell 0 ends the line
                    // before the row, cell N+1 starts the one after it when N % 32 == 0 — are partial writes to HBM.
                    // Non-temporal, whole-32-byte-sector and all-lanes-in-range forms of this store measured the same.)
                    constexpr bool SHELL = ISH && ROWEND;
                    // (offv is the row's offset or out of range: the shell cell sits one cell before / WL cells after it)
                    const unsigned offsh = (SHELL && offv != OOB && first_vec) ? offv - (unsigned)sizeof(T)
                                           : ((SHELL && offv != OOB && last_vec) ? offv + WL * (unsigned)sizeof(T) : OOB);
                    const T osh = first_vec ? o[0] : o[WL - 1];
#if SF_SK_DIAG == 1 || SF_SK_DIAG == 4
                    asm volatile("" ::"v"(o));
#else
                    buf_store<T, WL, NT>(rc, offv, o, so);
#endif
                    if constexpr (SHELL) buf_store1<T>(rc, offsh, sx * osh, so);
                    olast[r] = o;  // (the k faces of a first / last plane are stored after the rows, below)
                    const bool kslo = g.wall_lo && kg == 1, kshi = g.wall_hi && kg == N;
                    if constexpr (WALLS) {
                        // a row next to a j wall (per lane; no lane in most waves and steps, so the branch is skipped):
                        // the j face cell of every interior cell of the row, the i-j edge cells at a row end, and on a
                        // first / last plane the j-k edge cells and the corners — the expressions of emit_shells
                        const bool jlo = at_jlo(r), jhi = at_jhi(r);
                        if ((jlo | jhi) && offv != OOB) {
                            const T half = T(0.5);
                            // (row r-1 / r+1 of the tile: the row below is reached through the scalar offset — offv
                            // alone can be row 0 of the resource, and a buffer offset must not go below zero)
                            const unsigned sj = r > 0 ? so - pxb : so;
                            const unsigned offj = r > 0 ? (jlo ? offv : offv + 2 * pxb) : (jlo ? offv - pxb : offv + pxb);
                            const unsigned offjs = first_vec ? offj - (unsigned)sizeof(T)
                                                             : (last_vec ? offj + WL * (unsigned)sizeof(T) : OOB);
                            VW t;
#pragma unroll
                            for (int e = 0; e < WL; ++e) t[e] = sy * o[e];
                            buf_store<T, WL, false>(rc, offj, t, sj);
                            if constexpr (SHELL) buf_store1<T>(rc, offjs, half * (sy * osh + sx * osh), sj);
                            if (kslo | kshi) {
                                const __amdgpu_buffer_rsrc_t rk = prs(kslo ? po - g.plane : po + g.plane);
#pragma unroll
                                for (int e = 0; e < WL; ++e) t[e] = half * (sz * o[e] + sy * o[e]);
                                buf_store<T, WL, false>(rk, offj, t, sj);
                                if constexpr (SHELL) {
                                    const T third = (T)(1.0 / 3.0);
                                    const T ex = half * (sz * osh + sy * osh);
                                    const T ey = half * (sz * osh + sx * osh);
                                    const T ez = half * (sy * osh + sx * osh);
                                    buf_store1<T>(rk, offjs, third * ((ex + ey) + ez), sj);
                                }
                            }
                        }
                    }
                }
            }
        }
        if (l < S) {
            // set_bnd of this level on the k walls, where the next level reads it: the plane below the first one is
            // sz x (first plane), the plane above the last one sz x (last plane). Wave-uniform and true in one step per
            // level and chunk end, so it is a branch around register moves (no memory operation inside), taken by
            // everybody or nobody: the rows themselves carry no k-wall selects.
            constexpr int NS = S == 4 ? 3 : 4;
            const int s_new = YSH ? 0 : (((PH + 5 - l) & 3) % NS), s_old = YSH ? 1 : (((PH + 4 - l) & 3) % NS);
            const bool fix_lo = g.wall_lo && kg == 1, fix_hi = g.wall_hi && kg == N + 1;
            if (fix_lo | fix_hi) {
                asm volatile("; k-wall planes of level %0" ::"n"(l));  // (keeps the branch a branch: no if-conversion into selects)
                const int q = l - 1 < S - 1 ? l - 1 : 0;
#pragma unroll
                for (int r = 0; r < TJ; ++r) {
                    if (fix_lo) yr[q][s_old][r] = sz * yr[q][s_new][r];
                    if (fix_hi) yr[q][s_new][r] = sz * yr[q][s_old][r];
                }
            }
        } else {
            // first / last plane of a wall slab (wave-uniform, one step per chunk end): the k face of every row and,
            // with the i-shell, its two i-k edge cells — the expressions of emit_shells
            const bool kslo = g.wall_lo && kg == 1, kshi = g.wall_hi && kg == N;
            if (kslo | kshi) {
                constexpr bool SHELL = ISH && ROWEND;
                const __amdgpu_buffer_rsrc_t rlo = prs(po - g.plane), rhi = prs(po + g.plane);
#pragma unroll
                for (int r = 0; r < TJ; ++r) {
                    const unsigned offv = st_off(r), so = rs_(r);
                    const unsigned offsh = (SHELL && offv != SK_OOB && first_vec) ? offv - (unsigned)sizeof(T)
                                           : ((SHELL && offv != SK_OOB && last_vec) ? offv + WL * (unsigned)sizeof(T) : SK_OOB);
                    const T half = T(0.5);
                    const T osh = first_vec ? olast[r][0] : olast[r][WL - 1];
                    const T esh = half * (sz * osh + sx * osh);  // the i-k edge cell of that row end
                    VW t;
#pragma unroll
                    for (int e = 0; e < WL; ++e) t[e] = sz * olast[r][e];
                    if (kslo) {
                        buf_store<T, WL, false>(rlo, offv, t, so);
                        if constexpr (SHELL) buf_store1<T>(rlo, offsh, esh, so);
                    }
                    if (kshi) {
                        buf_store<T, WL, false>(rhi, offv, t, so);
                        if constexpr (SHELL) buf_store1<T>(rhi, offsh, esh, so);
                    }
                }
            }
        }
        if (l == 1) SF_SK_T(1);  // level 1 (waits for x(kk+1), x0(kk))
        if (l == 2) SF_SK_T(3);
        if (l == 3) SF_SK_T(4);
        if (l == 4) SF_SK_T(5);  // level 4 and its stores
        if constexpr (XSH) {
            if (l == 1) {
                // level 1 was the last reader of x(kk-1): shift, and request x(kk+2) into the slot that fell free
                // (three planes of x live instead of four; the request still has levels 2..S and the start of the
                // next step to arrive in)
#pragma unroll
                for (int r = 0; r < TJ; ++r) {
                    xr[0][r] = xr[1][r];
                    xr[1][r] = xr[2][r];
                }
                request_x();
                SF_SK_T(2);  // shift + request of x(kk+2)
            }
        }
    }
    // (4) publish this wave's edge rows: x(kk+1) and every intermediate level computed in this step
    if constexpr (FIRST != 3) {
        constexpr int XN = XSH ? 1 : ((PH + 1) & 3);  // x(kk+1): after the shift it is the centre slot
        sh.edge[WB][0][wave][0][lane] = xr[XN][0];
        sh.edge[WB][0][wave][1][lane] = xr[XN][TJ - 1];
    }
#pragma unroll
    for (int l = 1; l < S; ++l)
        if (l <= NACT) {
            constexpr int NS = S == 4 ? 3 : 4;
            const int sw = (S == 4) ? 0 : (((PH + 5 - l) & 3) % NS);
            sh.edge[WB][l][wave][0][lane] = yr[l - 1][sw][0];
            sh.edge[WB][l][wave][1][lane] = yr[l - 1][sw][TJ - 1];
        }
    if (NACT == S) pout += g.plane;
    SF_SK_T(6);  // edge rows published
    // (5) one barrier per step (LDS only: the global requests stay in flight across it)
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    SF_SK_T(7);  // barrier
#ifdef SF_SK_STAMP
    fx.st[8] += 1;
#endif
}

template <class T, int WL, bool NT, int S, int TJ, int NW, bool WALLS, bool ISH, bool ROWEND, int FIRST>
__device__ __forceinline__ void jsk_march(const Geom& g, SkShared<T, WL, S, TJ, NW>& sh, unsigned row0, unsigned stcol,
                                          const T* __restrict__ x,
                                          const T* __restrict__ x0, T* __restrict__ xn, int k0, int k1, T a, T inv, T sx,
                                          T sy, T sz, SkWall jw, int wave, int lane, bool first_vec, bool last_vec,
                                          T* __restrict__ x0out, T dt) {
    typedef typename LaneVec<T, WL>::type VW;
    static_assert(S >= 2 && S <= 4, "two, three or four fused sweeps");
    // every plane resource starts S-1 rows before its plane (jsk_step): shift the field pointers once
    x -= (long)(S - 1) * g.px;
    x0 -= (long)(S - 1) * g.px;
    xn -= (long)(S - 1) * g.px;
    x0out -= (long)(S - 1) * g.px;
    const int kmax = g.np - 1;
    auto plane_of_k = [&](const T* base, int kl) -> __amdgpu_buffer_rsrc_t {
        kl = kl < 0 ? 0 : (kl > kmax ? kmax : kl);
        return plane_rsrc(base + (long)kl * g.plane);
    };
    constexpr bool XSH = S == 4;
    constexpr int XA = XSH ? 0 : 3, XB = XSH ? 1 : 0, XC = XSH ? 2 : 1;  // slots of x(kk-1), x(kk), x(kk+1)
    const unsigned pxb = (unsigned)g.px * (unsigned)sizeof(T);
    auto rv = [&](int) -> unsigned { return row0; };
    auto rs_ = [&](int r) -> unsigned { return (unsigned)r * pxb; };
    VW xr[XSH ? 3 : 4][TJ], yr[S - 1][S == 4 ? 3 : 4][TJ], sr[S == 4 ? 5 : 4][TJ];
    int kk = k0 - S + 1;  // first step; ring phase 0
    SkFirst<T, TJ> fx;
    fx.dt = dt;
    fx.k0 = k0;
    fx.k1 = k1;
    fx.prhs = x0out + (long)(kk > kmax ? kmax : (kk < 0 ? 0 : kk)) * g.plane;
#ifdef SF_SK_STAMP
    for (int q = 0; q < 10; ++q) fx.st[q] = 0;
    const long long t_begin = __builtin_amdgcn_s_memtime();
#endif
    {
        // planes kk-1, kk, kk+1 of x in ring slots 3, 0, 1; x0(kk) in slot 0
        const __amdgpu_buffer_rsrc_t pa = plane_of_k(x, kk - 1), pb = plane_of_k(x, kk), pc = plane_of_k(x, kk + 1);
        const __amdgpu_buffer_rsrc_t ps = plane_of_k(x0, kk);
#pragma unroll
        for (int r = 0; r < TJ; ++r) {
            if constexpr (FIRST != 3) {
                xr[XA][r] = buf_load<T, WL>(pa, rv(r), rs_(r));
                xr[XB][r] = buf_load<T, WL>(pb, rv(r), rs_(r));
                xr[XC][r] = buf_load<T, WL>(pc, rv(r), rs_(r));
            } else {
#pragma unroll
                for (int q = 0; q < (XSH ? 3 : 4); ++q)
#pragma unroll
                    for (int e = 0; e < WL; ++e) xr[q][r][e] = T(0);
            }
            sr[0][r] = buf_load<T, WL>(ps, rv(r), rs_(r));
        }
        if constexpr ((FIRST == 1 || FIRST == 2) && ROWEND) {
#pragma unroll
            for (int r = 0; r < TJ; ++r) {
                const unsigned off = first_vec ? rv(r) - (unsigned)sizeof(T)
                                               : (last_vec ? rv(r) + WL * (unsigned)sizeof(T) : SK_OOB);
                fx.xs[0][r] = buf_load1<T>(pb, off, rs_(r));
                fx.xs[1][r] = buf_load1<T>(pc, off, rs_(r));
            }
        }
        // the first step reads the edges of x(kk) from buffer 1
        if constexpr (FIRST != 3) {
            sh.edge[1][0][wave][0][lane] = xr[XB][0];
            sh.edge[1][0][wave][1][lane] = xr[XB][TJ - 1];
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    // running plane pointers: x(kk+2), x0(kk+1), clamped into the stored planes [0, kmax], and the output plane k0
    const T* __restrict__ px = x + (long)(kk + 2 > kmax ? kmax : (kk + 2 < 0 ? 0 : kk + 2)) * g.plane;
    const T* __restrict__ ps0 = x0 + (long)(kk + 1 > kmax ? kmax : (kk + 1 < 0 ? 0 : kk + 1)) * g.plane;
    T* __restrict__ pout = xn + (long)k0 * g.plane;
#define SF_SK_STEP(PH_, NACT_)                                                                                       \
    jsk_step<T, WL, NT, S, TJ, NW, PH_, NACT_, WALLS, ISH, ROWEND, FIRST>(g, sh, xr, yr, sr, row0, stcol, px, \
                                                                          ps0, pout, kk, a, inv, sx, sy, sz, jw,       \
                                                                          wave, lane, first_vec, last_vec, fx)
    const int kend = k1 + S - 2;  // last step
#ifdef SF_SK_STAMP
    fx.tlast = __builtin_amdgcn_s_memtime();
#endif
    SF_SK_STEP(0, 1);
    ++kk;
    SF_SK_STEP(1, 1);
    ++kk;
    if constexpr (S == 2) {
        for (;;) {
            SF_SK_STEP(2, 2);
            if (++kk > kend) break;
            SF_SK_STEP(3, 2);
            if (++kk > kend) break;
            SF_SK_STEP(0, 2);
            if (++kk > kend) break;
            SF_SK_STEP(1, 2);
            if (++kk > kend) break;
        }
    } else if constexpr (S == 3) {
        SF_SK_STEP(2, 2);
        ++kk;
        SF_SK_STEP(3, 2);
        ++kk;
        for (;;) {
            SF_SK_STEP(0, 3);
            if (++kk > kend) break;
            SF_SK_STEP(1, 3);
            if (++kk > kend) break;
            SF_SK_STEP(2, 3);
            if (++kk > kend) break;
            SF_SK_STEP(3, 3);
            if (++kk > kend) break;
        }
    } else {
        SF_SK_STEP(2, 2);
        ++kk;
        SF_SK_STEP(3, 2);
        ++kk;
        SF_SK_STEP(0, 3);
        ++kk;
        SF_SK_STEP(1, 3);
        ++kk;
        for (;;) {
            SF_SK_STEP(2, 4);
            if (++kk > kend) break;
            SF_SK_STEP(3, 4);
            if (++kk > kend) break;
            SF_SK_STEP(0, 4);
            if (++kk > kend) break;
            SF_SK_STEP(1, 4);
            if (++kk > kend) break;
        }
    }
#undef SF_SK_STEP
#ifdef SF_SK_STAMP
    fx.st[9] = (unsigned long long)(__builtin_amdgcn_s_memtime() - t_begin);
    if (g_sk_stamp && lane == 0) {
        const long wg = (long)blockIdx.x + (long)gridDim.x * ((long)blockIdx.y + (long)gridDim.y * blockIdx.z);
        for (int q = 0; q < 10; ++q) g_sk_stamp[(wg * NW + wave) * 10 + q] = fx.st[q];
    }
#endif
}


template <class T, int NF, int WL, bool NT, int S, int TJ, int NW, bool ISH, int FIRST = 0>
__global__ void __launch_bounds__(64 * NW, (NW + 3) / 4) jacobi_sk_kernel(Geom g, JacobiArgs<T, NF> A, int kb, int ke,
                                                                         SkMap m) {
    // Every sweep level loses one CELL of validity per side in i, so after S levels ceil(S / WL) lanes per side hold
    // at least one invalid cell and only feed the shuffles
    constexpr int F = (S + WL - 1) / WL;
    constexpr int P = 64 - 2 * F;       // productive lanes of a wave
    constexpr int V = NW * TJ - 2 * S;  // output rows of a j-block
    static_assert(V > 0, "tile too small for the sweep depth");
    __shared__ SkShared<T, WL, S, TJ, NW> sh;
    const int N = g.N;
    const int nvec = N / WL;
    int chunk, f;
    {
        const int nchunk = (ke - kb + m.kc - 1) / m.kc;
        if (NF == 1) {
            chunk = (int)blockIdx.z;
            f = 0;
        } else {
            chunk = (int)blockIdx.z % nchunk;
            f = (int)blockIdx.z / nchunk;
        }
    }
    // Column block: workgroup x of the grid runs on XCD x, and the column blocks of one band (neighbours in j and i, who
    // read each other's halo rows) share an XCD's L2. The bands rarely divide evenly — 18 column blocks in bands of 3
    // leave two XCDs without work — so the band -> XCD assignment rotates with the chunk: over the chunks of a launch
    // every XCD receives the same number of workgroups.
    // A launch whose workgroups all fit the chip at once (band == 0) is as slow as its fullest XCD: there the column
    // blocks are dealt out one by one (the hardware sends consecutive workgroups to consecutive XCDs).
    const int cb = m.band > 0 ? (int)((blockIdx.x + (unsigned)chunk) & 7u) * m.band + (int)blockIdx.y : (int)blockIdx.x;
    if (cb >= m.ncb) return;  // whole workgroups
    const int tid = (int)threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int total = m.njb * nvec;
    int t = cb * P + lane - F;
    const bool active = t >= 0 && t < total && lane >= F && lane < 64 - F;
    t = t < 0 ? 0 : (t >= total ? total - 1 : t);  // feeder / padding lanes run on valid addresses
    const int jb = nvec == 1 ? t : (int)__umulhi((unsigned)t, m.nvec_magic);
    const int vec = t - jb * nvec;
    const int i0 = 1 + WL * vec;
    const int jrow0 = jb * V + 1 - S + wave * TJ;  // j of this lane's row 0 (may lie outside [0, N+1])
    SkWall jw;
    jw.lo = jb == 0;
    jw.hi = jb == m.njb - 1;
    jw.hi2 = jb == m.njb - 2;
    jw.jtN = N - 1 + S - (m.njb - 1) * V;
    const int shift = chunk > 0 ? m.gap : 0;  // (gap != 0 only for the two-chunk boundary launch)
    const int k0 = kb + chunk * m.kc + shift;
    const int k1 = ((kb + chunk * m.kc + m.kc < ke) ? kb + chunk * m.kc + m.kc : ke) + shift;

    const T a = A.a, inv = A.inv;
    const T* __restrict__ x = A.x[0];
    const T* __restrict__ x0 = A.x0[0];
    T* __restrict__ xn = A.xn[0];
    T* __restrict__ x0out = FIRST == 2 ? A.x0out[0] : A.xn[0];
    int b = A.b[0];
#pragma unroll
    for (int ff = 1; ff < NF; ++ff)
        if (f == ff) {
            x = A.x[ff];
            x0 = A.x0[ff];
            xn = A.xn[ff];
            x0out = FIRST == 2 ? A.x0out[ff] : A.xn[ff];
            b = A.b[ff];
        }
    const T sx = (b == 1) ? T(-1) : T(1);
    const T sy = (b == 2) ? T(-1) : T(1);
    const T sz = (b == 3) ? T(-1) : T(1);

    // byte offset of the lane's row 0 from the start of the plane resources, S-1 rows below the plane (jsk_step)
    const unsigned row0 = (unsigned)((jrow0 + S - 1) * g.px + (g.lead - 1) + i0) * (unsigned)sizeof(T);
    const unsigned stcol = active ? row0 : SK_OOB;
    const bool first_vec = (vec == 0), last_vec = (vec == nvec - 1);
    // Does this WORKGROUP touch a j wall? (uniform over the workgroup, from its item range, feeder lanes included —
    // they run on the addresses of their own item, and off the walls rows are addressed without clamping: every wave
    // takes the same instantiation.) j walls live in the first / last j-block. k walls: one wave-uniform test per level
    // and step, after the rows (jsk_step).
    int flo = cb * P - F, fhi = cb * P + P - 1 + F;
    flo = flo < 0 ? 0 : flo;
    fhi = fhi >= total ? total - 1 : fhi;
    const int fb_lo = nvec == 1 ? flo : (int)__umulhi((unsigned)flo, m.nvec_magic);
    const int fb_hi = nvec == 1 ? fhi : (int)__umulhi((unsigned)fhi, m.nvec_magic);
    const bool jwall = fb_lo == 0 || (fb_hi + 1) * V + S >= N;  // a tile row (halo included) is j <= 1 or j >= N
    // ... and a row end (a lane, feeder lanes included, whose vector is the first or last of its row)? Uniform as well.
    const bool rowend = fb_lo != fb_hi || flo - fb_lo * nvec == 0 || fhi - fb_hi * nvec == nvec - 1;
    if (jwall)
        jsk_march<T, WL, NT, S, TJ, NW, true, ISH, true, FIRST>(g, sh, row0, stcol, x, x0, xn, k0, k1, a, inv, sx, sy, sz, jw,
                                                         wave, lane, first_vec, last_vec, x0out, A.dt);
    else if (rowend)
        jsk_march<T, WL, NT, S, TJ, NW, false, ISH, true, FIRST>(g, sh, row0, stcol, x, x0, xn, k0, k1, a, inv, sx, sy, sz, jw,
                                                          wave, lane, first_vec, last_vec, x0out, A.dt);
    else
        jsk_march<T, WL, NT, S, TJ, NW, false, ISH, false, FIRST>(g, sh, row0, stcol, x, x0, xn, k0, k1, a, inv, sx, sy, sz, jw,
                                                           wave, lane, first_vec, last_vec, x0out, A.dt);
}

// ---------------------------------------------------------------------------------------------
// advect: semi-Lagrangian back-trace + trilinear interpolation + fused set_bnd (SPEC §3 advect).
// NF fields share one back-trace (vel_step advects u,v,w through the same velocity).
// Algorithmic traffic: 3 (velocity) + NF (gather, assuming reuse) + NF (write) words per cell.
template <class T, int NF>
struct AdvectArgs {
    T* d[NF];
    const T* d0[NF];
    int b[NF];
    const T* u;
    const T* v;
    const T* w;
    T dt0;
    int* flag;  // set to 1 if a back-trace left the planes this slab stores
    int skip_ishell;  // leave the i = 0 / N+1 shell cells of the result unwritten (nobody reads them: Solver::vel_step_body)
};

template <class T, int NF>
__global__ void __launch_bounds__(256) advect_kernel(Geom g, AdvectArgs<T, NF> A, int kb, int ke, TileMap m) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    typedef T Pair __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    int i0, j, kl, nv;
    if (!flat_cell<W>(g, m, kb, ke, i0, j, kl, nv)) return;
    const int N = g.N;
    const T Nf = (T)N;
    const T lo = T(0.5), hi = Nf + T(0.5);
    const long q = row0(g, j, kl) + i0;
    const V uu = ldv(A.u + q), vv = ldv(A.v + q), ww = ldv(A.w + q);
    const int kg = g.kg0 + kl;
    T out[NF][W];
    bool bad = false;
    // Phase 1: all W back-traces; phase 2: every gather of the thread (4 corner pairs x W cells x NF fields) is
    // requested before the first one is consumed; phase 3: the interpolation. Cells past the row end (ragged last
    // vector) trace from whatever the padding holds: the clamps and the NaN guard keep their addresses inside the
    // buffer, their results are never stored and they cannot raise the halo flag.
    T s1[W], t1[W], r1[W];
    long p00[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        T x = (T)(i0 + e) - A.dt0 * uu[e];
        T y = (T)j - A.dt0 * vv[e];
        T z = (T)kg - A.dt0 * ww[e];
        if (x < lo) x = lo;
        if (x > hi) x = hi;
        if (y < lo) y = lo;
        if (y > hi) y = hi;
        if (z < lo) z = lo;
        if (z > hi) z = hi;
        int ia = (x == x) ? (int)x : 0;
        int ja = (y == y) ? (int)y : 0;
        int ka = (z == z) ? (int)z : 0;
        ia = ia < 0 ? 0 : (ia > N ? N : ia);
        ja = ja < 0 ? 0 : (ja > N ? N : ja);
        ka = ka < 0 ? 0 : (ka > N ? N : ka);
        s1[e] = x - (T)ia;
        t1[e] = y - (T)ja;
        r1[e] = z - (T)ka;
        int kla = ka - g.kg0;  // local plane of k0; k1 = kla + 1 must also be stored
        if (kla < 0 || kla > g.np - 2) {
            bad |= (e < nv);
            kla = kla < 0 ? 0 : g.np - 2;
        }
        p00[e] = row0(g, ja, kla) + ia;  // (i0,j0,k0); +plane: k1; +px: j1
    }
    Pair C[NF][W][4];
#pragma unroll
    for (int e = 0; e < W; ++e)
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const T* __restrict__ d0 = A.d0[f];
            // the i0 / i0+1 samples are adjacent in memory: one element-aligned two-wide load per corner pair
            C[f][e][0] = *reinterpret_cast<const Pair*>(d0 + p00[e]);
            C[f][e][1] = *reinterpret_cast<const Pair*>(d0 + p00[e] + g.plane);
            C[f][e][2] = *reinterpret_cast<const Pair*>(d0 + p00[e] + g.px);
            C[f][e][3] = *reinterpret_cast<const Pair*>(d0 + p00[e] + g.px + g.plane);
        }
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const T s0 = T(1) - s1[e], t0 = T(1) - t1[e], r0 = T(1) - r1[e];
#pragma unroll
        for (int f = 0; f < NF; ++f) {
            const Pair c00 = C[f][e][0], c01 = C[f][e][1], c10 = C[f][e][2], c11 = C[f][e][3];
            out[f][e] = s0 * (t0 * (r0 * c00[0] + r1[e] * c01[0]) + t1[e] * (r0 * c10[0] + r1[e] * c11[0])) +
                        s1[e] * (t0 * (r0 * c00[1] + r1[e] * c01[1]) + t1[e] * (r0 * c10[1] + r1[e] * c11[1]));
        }
    }
    if (bad) atomicOr(A.flag, 1);
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        store_cells<T, W>(A.d[f], q - i0, i0, out[f], nv);
        emit_shells<T, W>(A.d[f], g, A.b[f], i0, j, kl, out[f], nv, A.skip_ishell == 0);
    }
}

// advect, one cell per lane ("wavefront shuffles for the trilinear lookups"): a wave holds 64 consecutive cells of a
// row. The eight samples of a cell are four (j, k) corners x the pair (i0, i0+1); with one cell per lane the i0 sample
// of lane l+1 IS the i0+1 sample of lane l whenever the two back-traces land in consecutive cells of the same row
// (a smooth flow: almost always), so each lane requests its four i0 samples — 64 lanes x 4 bytes, consecutive up to
// the few places where the integer part of a back-trace steps: whole lines instead of 8-byte pairs scattered at a
// 16-byte stride — and takes the i0+1 samples from its neighbour lane by a DPP wave shift. Lanes whose neighbour
// landed elsewhere, and lane 63, load their own i0+1 samples under an exec mask. (Waves of 63 cells whose lane 63 only
// feeds, so that the masked loads are skipped in most waves, measured slower: 204 vs 176 us — rows no longer start on
// a line.) Same values, same expressions as advect_kernel: bit-identical.
template <class T, int NF, bool PAIRS = false>
__global__ void __launch_bounds__(256) advect_row_kernel(Geom g, AdvectArgs<T, NF> A, int kb, int ke, int wpr) {
    typedef T Pair __attribute__((ext_vector_type(2), aligned(sizeof(T))));
    const int lane = (int)threadIdx.x & 63;
    const int wv = __builtin_amdgcn_readfirstlane((int)blockIdx.x * 4 + ((int)threadIdx.x >> 6));
    const int N = g.N;
    const int row = wv / wpr;  // (wave-uniform: scalar arithmetic)
    if (row >= N * (ke - kb)) return;
    const int seg = wv - row * wpr;
    const int kq = row / N;
    const int kl = kb + kq, j = 1 + row - kq * N;
    const int i = 1 + seg * 64 + lane;
    const bool ok = i <= N;
    const int ic = ok ? i : N;  // lanes past the row end repeat its last cell and store nothing
    const T Nf = (T)N;
    const T lo = T(0.5), hi = Nf + T(0.5);
    const long q = row0(g, j, kl) + ic;
    const T uu = A.u[q], vv = A.v[q], ww = A.w[q];
    const int kg = g.kg0 + kl;
    T x = (T)ic - A.dt0 * uu;
    T y = (T)j - A.dt0 * vv;
    T z = (T)kg - A.dt0 * ww;
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    if (y < lo) y = lo;
    if (y > hi) y = hi;
    if (z < lo) z = lo;
    if (z > hi) z = hi;
    int ia = (x == x) ? (int)x : 0;
    int ja = (y == y) ? (int)y : 0;
    int ka = (z == z) ? (int)z : 0;
    ia = ia < 0 ? 0 : (ia > N ? N : ia);
    ja = ja < 0 ? 0 : (ja > N ? N : ja);
    ka = ka < 0 ? 0 : (ka > N ? N : ka);
    const T s1 = x - (T)ia, t1 = y - (T)ja, r1 = z - (T)ka;
    int kla = ka - g.kg0;  // local plane of k0; k1 = kla + 1 must also be stored
    bool bad = false;
    if (kla < 0 || kla > g.np - 2) {
        bad = ok;
        kla = kla < 0 ? 0 : g.np - 2;
    }
    const long p00 = row0(g, ja, kla) + ia;  // (i0,j0,k0); +plane: k1; +px: j1
    // does the next lane's back-trace land in the next cell of the same row? (lane 63 receives 0: never equal)
    const long pn = __builtin_bit_cast(long, lane_dn(__builtin_bit_cast(double, p00)));
    const bool shared = pn == p00 + 1;
    const long off[4] = {0, g.plane, g.px, g.px + g.plane};
    T c0[NF][4], c1[NF][4] = {};
    if constexpr (PAIRS) {
        // every lane loads the (i0, i0+1) pair itself — 2 cells per lane at a one-cell stride, no shifts, no masked
        // loads: the better form in fp64 (16 bytes per lane: 307 vs 415 us at 256^3), the worse one in fp32 (215 vs 171)
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const Pair pr = *reinterpret_cast<const Pair*>(A.d0[f] + p00 + off[c]);
                c0[f][c] = pr[0];
                c1[f][c] = pr[1];
            }
    } else {
#pragma unroll
    for (int f = 0; f < NF; ++f)
#pragma unroll
        for (int c = 0; c < 4; ++c) c0[f][c] = A.d0[f][p00 + off[c]];
    }
    if (!PAIRS && !shared) {
#pragma unroll
        for (int f = 0; f < NF; ++f)
#pragma unroll
            for (int c = 0; c < 4; ++c) c1[f][c] = A.d0[f][p00 + off[c] + 1];
    }
    const T s0 = T(1) - s1, t0 = T(1) - t1, r0 = T(1) - r1;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            if constexpr (!PAIRS) {
                const T nb = lane_dn(c0[f][c]);
                c1[f][c] = shared ? nb : c1[f][c];
            }
        }
        T out[1];
        out[0] = s0 * (t0 * (r0 * c0[f][0] + r1 * c0[f][1]) + t1 * (r0 * c0[f][2] + r1 * c0[f][3])) +
                 s1 * (t0 * (r0 * c1[f][0] + r1 * c1[f][1]) + t1 * (r0 * c1[f][2] + r1 * c1[f][3]));
        if (ok) {
            A.d[f][q] = out[0];
            emit_shells<T, 1>(A.d[f], g, A.b[f], i, j, kl, out, 1, A.skip_ishell == 0);
        }
    }
    if (bad) atomicOr(A.flag, 1);
}

// ---------------------------------------------------------------------------------------------
// project, first half: div = c_div*((du + dv) + dw), set_bnd(0, div). p is zeroed by the caller
// (hipMemsetAsync over the whole field, which also covers set_bnd(0,p) and the ghost planes).
template <class T>
struct ProjectArgs {
    T* u;
    T* v;
    T* w;
    T* p;
    T* div;
    T c_div, c_grad;
    // The i = 0 / N+1 cells of u (project_div) / p (project_sub) are not read from memory but mirrored from the row's
    // end cells — exactly what set_bnd(1, u) / set_bnd(0, p) would have stored there: -u[1], -u[N] / p[1], p[N]. Set
    // when the solve that produced the field left its i-shell unwritten (Solver::vel_step_body: those 2 N^2 x planes
    // isolated four-byte writes per field are partial writes to HBM and cost a 512^3 last pass 18 %).
    int mirror_u, mirror_p;
    int skip_div_ishell;  // div's i-shell is never read by the solve; written only where the slot is compared afterwards
};

template <class T>
__global__ void __launch_bounds__(256) project_div_kernel(Geom g, ProjectArgs<T> A, int kb, int ke,
                                                          TileMap m) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    int i0, j, kl, nv;
    if (!flat_cell<W>(g, m, kb, ke, i0, j, kl, nv)) return;
    const long q = row0(g, j, kl) + i0;
    const V uc = ldv(A.u + q);
    T um = A.u[q - 1];
    const T up = A.u[q + W];
    if (A.mirror_u && i0 == 1) um = T(-1) * uc[0];
    const V vm = ldv(A.v + q - g.px), vp = ldv(A.v + q + g.px);
    const V wm = ldv(A.w + q - g.plane), wp = ldv(A.w + q + g.plane);
    T out[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const T left = (e == 0) ? um : uc[e - 1];
        T right = (e == W - 1) ? up : uc[e + 1];
        if (A.mirror_u && i0 + e == g.N) right = T(-1) * uc[e];  // (a ragged last vector holds cell N+1 itself)
        out[e] = A.c_div * (((right - left) + (vp[e] - vm[e])) + (wp[e] - wm[e]));
    }
    store_cells<T, W>(A.div, q - i0, i0, out, nv);
    emit_shells<T, W>(A.div, g, 0, i0, j, kl, out, nv, A.skip_div_ishell == 0);
}

// project, second half: u -= c_grad*dp/di etc., set_bnd(1,u), (2,v), (3,w).
template <class T>
__global__ void __launch_bounds__(256) project_sub_kernel(Geom g, ProjectArgs<T> A, int kb, int ke,
                                                          TileMap m) {
    constexpr int W = VecT<T>::W;
    typedef typename VecT<T>::type V;
    int i0, j, kl, nv;
    if (!flat_cell<W>(g, m, kb, ke, i0, j, kl, nv)) return;
    const long q = row0(g, j, kl) + i0;
    const T* __restrict__ p = A.p;
    const V pc = ldv(p + q);
    T pm = p[q - 1];
    const T pp = p[q + W];
    if (A.mirror_p && i0 == 1) pm = T(1) * pc[0];
    const V pjm = ldv(p + q - g.px), pjp = ldv(p + q + g.px);
    const V pkm = ldv(p + q - g.plane), pkp = ldv(p + q + g.plane);
    const V uc = ldv(A.u + q), vc = ldv(A.v + q), wc = ldv(A.w + q);
    T ou[W], ov[W], ow[W];
#pragma unroll
    for (int e = 0; e < W; ++e) {
        const T left = (e == 0) ? pm : pc[e - 1];
        T right = (e == W - 1) ? pp : pc[e + 1];
        if (A.mirror_p && i0 + e == g.N) right = T(1) * pc[e];
        ou[e] = uc[e] - A.c_grad * (right - left);
        ov[e] = vc[e] - A.c_grad * (pjp[e] - pjm[e]);
        ow[e] = wc[e] - A.c_grad * (pkp[e] - pkm[e]);
    }
    store_cells<T, W>(A.u, q - i0, i0, ou, nv);
    store_cells<T, W>(A.v, q - i0, i0, ov, nv);
    store_cells<T, W>(A.w, q - i0, i0, ow, nv);
    emit_shells<T, W>(A.u, g, 1, i0, j, kl, ou, nv);
    emit_shells<T, W>(A.v, g, 2, i0, j, kl, ov, nv);
    emit_shells<T, W>(A.w, g, 3, i0, j, kl, ow, nv);
}

// ---------------------------------------------------------------------------------------------
// Stand-alone set_bnd (SPEC §3), three dependent passes reading memory exactly like the oracle.
// Only used by sf_set_bnd(); the step kernels fuse it. pass 0 = faces, 1 = edges, 2 = corners.
template <class T>
__global__ void __launch_bounds__(256) set_bnd_kernel(Geom g, T* __restrict__ x, int b, int pass) {
    const int N = g.N;
    const T sx = (b == 1) ? T(-1) : T(1);
    const T sy = (b == 2) ? T(-1) : T(1);
    const T sz = (b == 3) ? T(-1) : T(1);
    const T half = T(0.5);
    const T third = (T)(1.0 / 3.0);
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const int E = N + 1;
    const int klo_l = 1 - g.kg0;      // local plane of global k = 1   (valid if wall_lo)
    const int khi_l = N - g.kg0;      // local plane of global k = N   (valid if wall_hi)
    auto at = [&](int i, int j, int kl) -> T& { return x[row0(g, j, kl) + i]; };
    if (pass == 0) {
        // i- and j-faces of every local interior plane, k-faces on wall slabs.
        const long per = (long)N * g.nzl;  // (a, kl) pairs
        if (t < per) {
            const int a = 1 + (int)(t % N), kl = g.G + (int)(t / N);
            at(0, a, kl) = sx * at(1, a, kl);
            at(E, a, kl) = sx * at(N, a, kl);
            at(a, 0, kl) = sy * at(a, 1, kl);
            at(a, E, kl) = sy * at(a, N, kl);
        }
        const long nn = (long)N * N;
        if (t < nn) {
            const int i = 1 + (int)(t % N), j = 1 + (int)(t / N);
            if (g.wall_lo) at(i, j, klo_l - 1) = sz * at(i, j, klo_l);
            if (g.wall_hi) at(i, j, khi_l + 1) = sz * at(i, j, khi_l);
        }
    } else if (pass == 1) {
        // z-directed edges on every local interior plane
        if (t < g.nzl) {
            const int kl = g.G + (int)t;
            for (int a = 0; a < 2; ++a)
                for (int c = 0; c < 2; ++c) {
                    const int I = a ? E : 0, In = a ? N : 1, J = c ? E : 0, Jn = c ? N : 1;
                    at(I, J, kl) = half * (at(In, J, kl) + at(I, Jn, kl));
                }
        }
        // x- and y-directed edges on wall planes
        if (t < N) {
            const int s = 1 + (int)t;
            for (int c = 0; c < 2; ++c) {
                if (!(c ? g.wall_hi : g.wall_lo)) continue;
                const int K = c ? khi_l + 1 : klo_l - 1, Kn = c ? khi_l : klo_l;
                for (int a = 0; a < 2; ++a) {
                    const int J = a ? E : 0, Jn = a ? N : 1;
                    at(s, J, K) = half * (at(s, Jn, K) + at(s, J, Kn));
                    const int I = J, In = Jn;
                    at(I, s, K) = half * (at(In, s, K) + at(I, s, Kn));
                }
            }
        }
    } else {
        if (t < 8) {
            const int a = (int)t & 1, c = ((int)t >> 1) & 1, e = ((int)t >> 2) & 1;
            if (e ? g.wall_hi : g.wall_lo) {
                const int I = a ? E : 0, In = a ? N : 1, J = c ? E : 0, Jn = c ? N : 1;
                const int K = e ? khi_l + 1 : klo_l - 1, Kn = e ? khi_l : klo_l;
                at(I, J, K) = third * ((at(In, J, K) + at(I, Jn, K)) + at(I, J, Kn));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Tracer particles (docs/SPEC.md §6): the trilinear sample of advect applied at arbitrary points.
template <class T>
struct TriSampleDev {
    long q;
    long dj, dk;
    T s0, s1, t0, t1, r0, r1;
    __device__ TriSampleDev(const Geom& g, T x, T y, T z) {
        const int N = g.N;
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
        q = row0(g, j0, k0 - g.kg0) + i0;
        dj = g.px;
        dk = g.plane;
    }
    __device__ T operator()(const T* __restrict__ d0) const {
        return s0 * (t0 * (r0 * d0[q] + r1 * d0[q + dk]) + t1 * (r0 * d0[q + dj] + r1 * d0[q + dj + dk])) +
               s1 * (t0 * (r0 * d0[q + 1] + r1 * d0[q + 1 + dk]) +
                     t1 * (r0 * d0[q + 1 + dj] + r1 * d0[q + 1 + dj + dk]));
    }
};

template <class T>
__device__ __forceinline__ T clamp_coord(int N, T x) {
    const T lo = T(0.5), hi = (T)N + T(0.5);
    if (x < lo) x = lo;
    if (x > hi) x = hi;
    return x;
}

template <class T>
__global__ void __launch_bounds__(256) tracers_advect_kernel(Geom g, int n, T* __restrict__ pos,
                                                              const T* __restrict__ u, const T* __restrict__ v,
                                                              const T* __restrict__ w, T dt0) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n) return;
    const int N = g.N;
    const T x = clamp_coord(N, pos[3 * t]), y = clamp_coord(N, pos[3 * t + 1]), z = clamp_coord(N, pos[3 * t + 2]);
    const TriSampleDev<T> S(g, x, y, z);
    const T vx = S(u), vy = S(v), vz = S(w);
    pos[3 * t] = clamp_coord(N, x + dt0 * vx);
    pos[3 * t + 1] = clamp_coord(N, y + dt0 * vy);
    pos[3 * t + 2] = clamp_coord(N, z + dt0 * vz);
}

template <class T>
__global__ void __launch_bounds__(256) tracers_sample_kernel(Geom g, int n, const T* __restrict__ pos,
                                                              const T* __restrict__ dens, const T* __restrict__ u,
                                                              const T* __restrict__ v, const T* __restrict__ w,
                                                              T* __restrict__ dens_out, T* __restrict__ speed_out) {
    const int t = (int)(blockIdx.x * blockDim.x + threadIdx.x);
    if (t >= n) return;
    const TriSampleDev<T> S(g, pos[3 * t], pos[3 * t + 1], pos[3 * t + 2]);
    const T a = S(u), b = S(v), c = S(w);
    dens_out[t] = S(dens);
    speed_out[t] = sqrt((a * a + b * b) + c * c);
}

// ---------------------------------------------------------------------------------------------
// Utility kernels.
template <class T>
__global__ void __launch_bounds__(256) fill_kernel(T* __restrict__ x, T value, long nvec) {
    constexpr int W = VecT<T>::W;
    typename VecT<T>::type v;
#pragma unroll
    for (int e = 0; e < W; ++e) v[e] = value;
    const long stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += stride) stv(x + q * W, v);
}

// Ghost-plane copies between logical slabs of one process: up to 6 (field, direction) segments in one launch.
struct HaloCopyArgs {
    const float4* src[6];
    float4* dst[6];
    int nseg;
    long n16;  // 16-byte units per segment
};

static __global__ void __launch_bounds__(256) halo_copy_kernel(HaloCopyArgs A) {
    const int seg = (int)blockIdx.y;
    if (seg >= A.nseg) return;
    const float4* __restrict__ s = A.src[0];
    float4* __restrict__ d = A.dst[0];
#pragma unroll
    for (int q = 1; q < 6; ++q)
        if (seg == q) {
            s = A.src[q];
            d = A.dst[q];
        }
    const long stride = (long)gridDim.x * blockDim.x;
    for (long q = (long)blockIdx.x * blockDim.x + threadIdx.x; q < A.n16; q += stride) d[q] = s[q];
}

// float4 copy used to quote the achievable HBM rate in the same run as the solver numbers.
static __global__ void __launch_bounds__(256) copy16_kernel(const float4* __restrict__ src,
                                                     float4* __restrict__ dst, long n) {
    const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q < n) dst[q] = src[q];
}

}  // namespace sfk
