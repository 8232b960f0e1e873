// tools/sk_probe.hip — where do the cycles of one march step of sfk::jacobi_sk_kernel go?
// Diagnostic harness ("commit the lone-workgroup table"): includes the product's kernel header with
// -DSF_SK_STAMP, which adds s_memtime stamps at five points of every step (no stamp exists in libsfgpu.so), launches
// the plain four-sweep instantiation exactly as Solver::launch_sk_cfg does — whole grid, one chunk layer, eight
// workgroups (one per XCD), one lone workgroup — and prints, per launch shape, the kernel time (HIP events) and the
// mean cycles per step and wave spent in: issuing the requests | level 1 (first use of the planes requested one step
// ago) | levels 2..4 + stores | publishing the edge rows | the barrier.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -DSF_SK_STAMP tools/sk_probe.hip -o tools/sk_probe
#include "../fluidsolvergpu_amd/csrc/sf_kernels.hpp"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

static int ceil_div(long a, long b) { return (int)((a + b - 1) / b); }

template <bool NT>
void run(int N, const char* what, int gx, int gy, int gz_override) {
    typedef float T;
    constexpr int WL = 2, S = 4, TJ = 4, NW = 8, W = 4;
    constexpr int V = NW * TJ - 2 * S, P = 64 - 2 * ((S + WL - 1) / WL);
    sfk::Geom g{};
    g.N = N; g.nzl = N; g.G = 1; g.np = N + 2; g.kg0 = 0; g.lead = 32;
    g.px = ceil_div(g.lead + N + 1 + W, 32) * 32;
    g.plane = (long)g.px * (N + 2);
    g.wall_lo = g.wall_hi = 1;
    const long elems = g.plane * g.np + 256, front = (4L * g.px + 63) / 64 * 64, back = 64L * g.px;
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
    m.band = ceil_div(m.ncb, 8);
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
    dim3 grid(gx > 0 ? gx : 8, gy > 0 ? gy : m.band, gz_override > 0 ? gz_override : nchunk);
    const long nwg = (long)grid.x * grid.y * grid.z;
    unsigned long long* d_st;
    CK(hipMalloc(&d_st, nwg * NW * 8 * sizeof(unsigned long long)));
    CK(hipMemset(d_st, 0, nwg * NW * 8 * sizeof(unsigned long long)));
    CK(hipMemcpyToSymbol(HIP_SYMBOL(sfk::g_sk_stamp), &d_st, sizeof(d_st)));
    sfk::JacobiArgs<T, 1> A{};
    A.x[0] = x + front; A.x0[0] = x0 + front; A.xn[0] = xn + front; A.b[0] = 0; A.a = 0.3f; A.inv = 1.0f / 2.8f;
    A.x0out[0] = xn + front; A.dt = 0.1f;
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
    std::vector<unsigned long long> st(nwg * NW * 8);
    CK(hipMemcpy(st.data(), d_st, st.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    double sum[8] = {}, nw = 0;
    for (long w = 0; w < nwg * NW; ++w) {
        if (st[w * 8 + 5] == 0) continue;  // workgroups beyond ncb return at once
        for (int q = 0; q < 8; ++q) sum[q] += (double)st[w * 8 + q];
        nw += 1;
    }
    const double steps = sum[5] / nw;
    printf("%-34s N=%d NT=%d grid %ux%ux%u (%ld wg, kc=%d): %8.1f us | steps/wave %5.1f | cycles per step: request %5.0f  level1 %5.0f  "
           "levels2-4 %5.0f  publish %5.0f  barrier %5.0f  = %6.0f  (whole march %7.0f/step)\n",
           what, N, (int)NT, grid.x, grid.y, grid.z, nwg, m.kc, best * 1e3, steps, sum[0] / sum[5], sum[1] / sum[5], sum[2] / sum[5],
           sum[3] / sum[5], sum[4] / sum[5], (sum[0] + sum[1] + sum[2] + sum[3] + sum[4]) / sum[5], sum[6] / sum[5]);
    CK(hipFree(d_st)); CK(hipFree(x)); CK(hipFree(x0)); CK(hipFree(xn));
}

int main(int argc, char** argv) {
    const int sizes[2] = {256, 512};
    for (int N : sizes) {
        run<false>(N, "whole grid", 0, 0, 0);
        if (N == 512) run<true>(N, "whole grid", 0, 0, 0);
        run<false>(N, "one chunk layer (8 x band x 1)", 0, 0, 1);
        run<false>(N, "eight workgroups, one per XCD", 8, 1, 1);
        run<false>(N, "one lone workgroup", 1, 1, 1);
    }
    return 0;
}
