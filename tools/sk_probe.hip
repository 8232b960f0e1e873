// tools/sk_probe.hip — where do the cycles of one march step of sfk::jacobi_sk_kernel go?
// Diagnostic harness ("commit the lone-workgroup table"): includes the product's kernel header with
// -DSF_SK_STAMP, which adds s_memtime stamps at five points of every step (no stamp exists in libsfgpu.so), launches
// the plain four-sweep instantiation exactly as Solver::launch_sk_cfg does — whole grid, one chunk layer, eight
// workgroups (one per XCD), one lone workgroup — and prints, per launch shape, the kernel time (HIP events) and the
// mean cycles per step and wave spent in: issuing the requests | level 1 (first use of the planes requested one step
// ago) | levels 2..4 + stores | publishing the edge rows | the barrier.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DSF_SK_STAMP tools/sk_probe.hip -o tools/sk_probe
// Options: -DPROBE_F64 (double, one cell per lane); -DPROBE_TJ=<rows per wave> -DPROBE_NW=<waves per workgroup> (default: the library's 2 x 16; 4 x 8 is the
// round-2 tile); -DSF_SK_DIAG=1|3|4 drops the stores / the loads / both of the marching loop (garbage results: timing
// only); without -DSF_SK_STAMP the launch times carry no stamp overhead (~10 %).
#include "../fluidsolvergpu_amd/csrc/sf_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#ifndef PROBE_TJ
#define PROBE_TJ 2
#endif
#ifndef PROBE_NW
#define PROBE_NW 16
#endif
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

template <bool NT>
void run(int N, const char* what, int gx, int gy, int gz_override) {
#ifdef PROBE_F64
    typedef double T;
#else
    typedef float T;
#endif
    constexpr int WL = 8 / (int)sizeof(T), S = 4, TJ = PROBE_TJ, NW = PROBE_NW, W = 16 / (int)sizeof(T);
    constexpr int V = NW * TJ - 2 * S, P = 64 - 2 * ((S + WL - 1) / WL);
    sfk::Geom g{};
    g.N = N; g.nzl = N; g.G = 1; g.np = N + 2; g.kg0 = 0; g.lead = 128 / (int)sizeof(T);
    g.px = ceil_div(g.lead + N + 1 + W, 128 / (int)sizeof(T)) * (128 / (int)sizeof(T));
    g.plane = (long)g.px * (N + 2);
    g.wall_lo = g.wall_hi = 1;
    const long elems = g.plane * g.np + 256, front = (4L * g.plane + 4L * g.px + 63) / 64 * 64, back = 4L * g.plane + 64L * g.px;
    T *x, *x0, *xn;
    const size_t bytes = (size_t)(front + elems + back) * sizeof(T);
    CK(hipMalloc(&x, bytes)); CK(hipMalloc(&x0, bytes)); CK(hipMalloc(&xn, bytes));
    {
        std::vector<T> h((size_t)(front + elems + back));
        for (size_t q = 0; q < h.size(); ++q) h[q] = (T)((q * 2654435761u >> 8) & 1023) * (1.0f / 1024) - 0.5f;
        CK(hipMemcpy(x, h.data(), bytes, hipMemcpyHostToDevice));
        CK(hipMemcpy(x0, h.data(), bytes, hipMemcpyHostToDevice));
        CK(hipMemset(xn, 0, bytes));
    }
    sfk::SkMap m{};
    const int nvec = N / WL;
    m.njb = ceil_div(N, V);
    m.ncb = ceil_div((long)m.njb * nvec, P);
    m.nvec_magic = 0xFFFFFFFFu / (unsigned)nvec + 1u;
    const int kb = 1, ke = 1 + N, np = N;
    int nchunk = 1;
    {  // Solver::sk_chunks
        double best = -1;
        for (int c = 1; c <= std::max(1, np / 8); ++c) {
            const int kc = ceil_div(np, c);
            const long total = (long)m.ncb * ceil_div(np, kc);
            const double tm = (double)ceil_div(ceil_div(total, 8), 32) * (kc + 2 * S - 2 + 2);
            if (best < 0 || tm < best * 0.999) { best = tm; nchunk = c; }
        }
    }
    m.kc = ceil_div(np, nchunk);
    nchunk = ceil_div(np, m.kc);
    m.band = ceil_div(m.ncb, 8);
    dim3 grid(gx > 0 ? gx : 8, gy > 0 ? gy : m.band, gz_override > 0 ? gz_override : nchunk);
    const long nwg = (long)grid.x * grid.y * grid.z;
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, nwg * NW * 10 * sizeof(unsigned long long)));
    CK(hipMemset(d_st, 0, nwg * NW * 10 * sizeof(unsigned long long)));
#ifdef SF_SK_STAMP
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sfk::g_sk_stamp), &d_st, sizeof(d_st)));
#endif
    sfk::JacobiArgs<T, 1> A{};
    A.x[0] = x + front; A.x0[0] = x0 + front; A.xn[0] = xn + front; A.b[0] = 0; A.a = 0.3f; A.inv = 1.0f / 2.8f;
    A.x0out[0] = xn + front; A.dt = (T)0.1f;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        CK(hipEventRecord(e0));
        hipLaunchKernelGGL((sfk::jacobi_sk_kernel<T, 1, WL, NT, S, TJ, NW, false, 0>), grid, dim3(64 * NW), 0, 0, g, A, kb, ke, m);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    unsigned long long sum_bits = 0;
    if (gz_override == 0 && gx == 0) {  // whole grid: checksum of the output (variants must agree bit for bit)
        std::vector<T> h((size_t)(front + elems + back));
        CK(hipMemcpy(h.data(), xn, bytes, hipMemcpyDeviceToHost));
        for (size_t q = (size_t)front; q < (size_t)(front + elems); ++q) {  // (the field itself, not its padding)
            unsigned long long u = 0;
            memcpy(&u, &h[q], sizeof(T));
            sum_bits = (sum_bits ^ u) * 1099511628211ull;
        }
    }
    std::vector<unsigned long long> st(nwg * NW * 10);
    CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[10] = {}, nw = 0;
    for (long w = 0; w < nwg * NW; ++w) {
        if (st[w * 10 + 8] == 0) continue;  // workgroups beyond ncb return at once / build without stamps
        for (int q = 0; q < 10; ++q) sum[q] += (double)st[w * 10 + q];
        nw += 1;
    }
    printf("%-34s N=%d NT=%d grid %ux%ux%u (%ld wg, kc=%d): %8.1f us", what, N, (int)NT, grid.x, grid.y, grid.z, nwg, m.kc, best * 1e3);
    if (nw > 0) {
        const double n = sum[8];
        printf(" | steps/wave %5.1f | cycles/step: x0 request+halo read %4.0f  level1 %4.0f  x request %4.0f  level2 %4.0f  level3 %4.0f  "
               "level4+stores %4.0f  publish %4.0f  sync %4.0f = %5.0f", n / nw, sum[0] / n, sum[1] / n, sum[2] / n, sum[3] / n, sum[4] / n,
               sum[5] / n, sum[6] / n, sum[7] / n, (sum[0] + sum[1] + sum[2] + sum[3] + sum[4] + sum[5] + sum[6] + sum[7]) / n);
    }
    printf("\n");
    if (sum_bits) printf("    output checksum %016llx\n", sum_bits);
    if (nwg == 1 && nw > 0)  // the lone workgroup, wave by wave (waves w and w + 4 share a SIMD)
        for (int w = 0; w < NW; ++w) {
            const unsigned long long* q = &st[w * 10];
            const double n = (double)q[8];
            printf("      wave %d: x0 request+halo read %4.0f  level1 %4.0f  x request %4.0f  level2 %4.0f  level3 %4.0f  level4+stores %4.0f  "
                   "publish %4.0f  sync %4.0f\n", w, q[0] / n, q[1] / n, q[2] / n, q[3] / n, q[4] / n, q[5] / n, q[6] / n, q[7] / n);
        }
    CK(hipFree(d_st)); CK(hipFree(x)); CK(hipFree(x0)); CK(hipFree(xn));
}

int main(int argc, char** argv) {
    const int sizes[2] = {256, 512};
    for (int N : sizes) {
        if (argc > 1 && N != atoi(argv[1])) continue;
        run<false>(N, "whole grid", 0, 0, 0);
        if (N == 512) run<true>(N, "whole grid", 0, 0, 0);
        run<false>(N, "one chunk layer (8 x band x 1)", 0, 0, 1);
        run<false>(N, "eight workgroups, one per XCD", 8, 1, 1);
        run<false>(N, "one lone workgroup", 1, 1, 1);
    }
    return 0;
}
