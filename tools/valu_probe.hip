// tools/valu_probe.hip — what does one fp32 vector instruction cost on gfx950, per wave and per SIMD?
// The marching Jacobi kernel (sfk::jacobi_sk_kernel) runs two waves per SIMD and hipcc's SLP vectoriser turns its
// two-cell lane vectors into v_pk_add_f32 / v_pk_mul_f32; this probe measures the issue cost of those forms next to
// the scalar ones, independent and as a dependent chain, with 1, 2 and 4 waves per SIMD (one workgroup per CU).
// Build: hipcc -O3 --offload-arch=gfx950 tools/valu_probe.hip -o tools/valu_probe
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s line %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

enum Mode { ADD_IND, PK_ADD_IND, PK_MUL_IND, ADD_DEP, PK_ADD_DEP, DPP_ADD_IND, CNDMASK_IND, FMA_IND, PK_FMA_IND, MOV_IND,
            PK_ADD_DEP2, ADD_DEP2, CNDMASK_E64, BFI_IND, AND_OR_IND, MUL_IND, PK_ADD_NOP, MAX_IND, NMODES };
static const char* NAMES[NMODES] = {"v_add_f32 x16 independent", "v_pk_add_f32 x16 independent", "v_pk_mul_f32 x16 independent",
                                    "v_add_f32 dependent chain", "v_pk_add_f32 dependent chain", "v_add_f32_dpp wave_shr x16 independent",
                                    "v_cndmask_b32 x16 independent", "v_fma_f32 x16 independent", "v_pk_fma_f32 x16 independent",
                                    "v_mov_b32 x16 independent", "v_pk_add_f32 two interleaved chains", "v_add_f32 two interleaved chains",
                                    "v_cndmask_b32_e64 (SGPR-pair mask) x16 independent", "v_bfi_b32 x16 independent",
                                    "v_and_or_b32 x16 independent", "v_mul_f32 x16 independent",
                                    "v_pk_add_f32 + s_nop 0 x16 independent", "v_max_f32 x16 independent"};

template <int MODE>
__global__ void probe(long long* out, float* sink, int iters, float seed) {
    f2 a[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) a[q] = f2{seed + q, seed - q};
    const f2 b = f2{seed * 0.5f, seed * 0.25f};
    const unsigned long long msk = __builtin_amdgcn_read_exec() ^ (unsigned long long)(iters & 1);  // wave-uniform mask
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                if (MODE == ADD_IND) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(b.x));
                if (MODE == PK_ADD_IND) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[q]) : "v"(b));
                if (MODE == PK_MUL_IND) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[q]) : "v"(b));
                if (MODE == ADD_DEP) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[0].x) : "v"(b.x));
                if (MODE == PK_ADD_DEP) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[0]) : "v"(b));
                if (MODE == PK_ADD_DEP2) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[q & 1]) : "v"(b));
                if (MODE == ADD_DEP2) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[q & 1].x) : "v"(b.x));
                if (MODE == DPP_ADD_IND)
                    asm volatile("v_add_f32_dpp %0, %1, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"
                                 : "+v"(a[q].x) : "v"(a[(q + 8) & 15].y));
                if (MODE == CNDMASK_IND) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[q].x) : "v"(b.x));
                if (MODE == FMA_IND) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(a[q].x) : "v"(b.x));
                if (MODE == PK_FMA_IND) asm volatile("v_pk_fma_f32 %0, %0, %1, %1" : "+v"(a[q]) : "v"(b));
                if (MODE == MOV_IND) asm volatile("v_mov_b32 %0, %1" : "+v"(a[q].x) : "v"(b.x));
                if (MODE == CNDMASK_E64) asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(a[q].x) : "v"(b.x), "s"(msk));
                if (MODE == BFI_IND) asm volatile("v_bfi_b32 %0, %1, %2, %0" : "+v"(a[q].x) : "v"(b.y), "v"(b.x));
                if (MODE == AND_OR_IND) asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(a[q].x) : "v"(b.y), "v"(b.x));
                if (MODE == MUL_IND) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(b.x));
                if (MODE == PK_ADD_NOP) asm volatile("v_pk_add_f32 %0, %0, %1\n\ts_nop 0" : "+v"(a[q]) : "v"(b));
                if (MODE == MAX_IND) asm volatile("v_max_f32 %0, %0, %1" : "+v"(a[q].x) : "v"(b.x));
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
#pragma unroll
    for (int q = 0; q < 16; ++q) s += a[q].x + a[q].y;
    if (s == 12345.678f) sink[0] = s;
    if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
double run(int waves_per_simd, int iters) {
    const int nb = 256, threads = 256 * waves_per_simd;
    long long* d;
    float* sink;
    CK(hipMalloc(&d, sizeof(long long) * nb * threads / 64));
    CK(hipMalloc(&sink, 4));
    hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(threads), 0, 0, d, sink, iters, 1.0f);
    hipLaunchKernelGGL(probe<MODE>, dim3(nb), dim3(threads), 0, 0, d, sink, iters, 1.0f);
    CK(hipDeviceSynchronize());
    std::vector<long long> h(nb * threads / 64);
    CK(hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
    std::sort(h.begin(), h.end());
    CK(hipFree(d));
    CK(hipFree(sink));
    return (double)h[h.size() / 2] / ((double)iters * 64.0);  // cycles per instruction as seen by one wave (median)
}

template <int MODE>
void row() {
    const int iters = 2000;
    const double c1 = run<MODE>(1, iters), c2 = run<MODE>(2, iters), c4 = run<MODE>(4, iters);
    printf("%-42s | %6.2f | %6.2f (%5.2f /SIMD) | %6.2f (%5.2f /SIMD)\n", NAMES[MODE], c1, c2, c2 / 2, c4, c4 / 4);
}

int main() {
    printf("s_memtime ticks per instruction as seen by ONE wave (median over waves), one workgroup per CU\n");
    printf("%-42s | 1 w/SIMD | 2 waves/SIMD          | 4 waves/SIMD\n", "instruction stream");
    row<ADD_IND>(); row<PK_ADD_IND>(); row<PK_MUL_IND>(); row<FMA_IND>(); row<PK_FMA_IND>(); row<MOV_IND>();
    row<DPP_ADD_IND>(); row<CNDMASK_IND>(); row<CNDMASK_E64>(); row<BFI_IND>(); row<AND_OR_IND>(); row<MUL_IND>();
    row<MAX_IND>(); row<PK_ADD_NOP>(); row<ADD_DEP>(); row<PK_ADD_DEP>(); row<ADD_DEP2>(); row<PK_ADD_DEP2>();
    return 0;
}
