// Probe: what do stream-ordering packets cost between two back-to-back kernels on MI355X?
// Build: hipcc -O3 --offload-arch=gfx950 -o tools/evgap tools/evgap.hip ; run: tools/evgap
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

__global__ void busy(float* p, long n) {
    long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = p[i] * 1.0001f + 1.0f;
}

int main() {
    const long n = 64L << 20;  // 256 MB read + write: ~90 us
    float* d;
    CK(hipMalloc(&d, n * sizeof(float)));
    CK(hipMemset(d, 0, n * sizeof(float)));
    float* d2;
    CK(hipMalloc(&d2, 1 << 20));
    hipStream_t a, b;
    CK(hipStreamCreateWithFlags(&a, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&b, hipStreamNonBlocking));
    const int R = 200;
    for (int fence = 0; fence < 2; ++fence) {
        const unsigned fl = hipEventDisableTiming | (fence ? 0u : (unsigned)hipEventDisableSystemFence);
        std::vector<hipEvent_t> ev(R + 1), evb(R + 1);
        for (auto& e : ev) CK(hipEventCreateWithFlags(&e, fl));
        for (auto& e : evb) CK(hipEventCreateWithFlags(&e, fl));
        hipEvent_t done;  // recorded once on stream b, long complete
        CK(hipEventCreateWithFlags(&done, fl));
        hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, b, d2, 1L << 18);
        CK(hipEventRecord(done, b));
        CK(hipDeviceSynchronize());
        for (int mode = 0; mode < 7; ++mode) {
            CK(hipDeviceSynchronize());
            auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < R; ++r) {
                switch (mode) {
                case 0: break;
                case 1: CK(hipStreamWaitEvent(a, done, 0)); break;                       // satisfied wait
                case 2: CK(hipEventRecord(ev[r], a)); break;                             // record
                case 3: CK(hipStreamWaitEvent(a, done, 0)); CK(hipEventRecord(ev[r], a)); break;
                case 4: break;                                                           // stopEvent on the kernel
                case 5:  // the real pattern: small kernel on b waits for the previous big one, big waits for small
                    CK(hipStreamWaitEvent(a, evb[r], 0));
                    CK(hipEventRecord(ev[r], a));
                    CK(hipStreamWaitEvent(b, ev[r], 0));
                    hipLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, b, d2, 1L << 18);
                    CK(hipEventRecord(evb[r + 1], b));
                    break;
                case 6:  // same with stop events instead of records
                    CK(hipStreamWaitEvent(a, evb[r], 0));
                    if (r > 0) CK(hipStreamWaitEvent(b, ev[r - 1], 0));
                    hipExtLaunchKernelGGL(busy, dim3(1024), dim3(256), 0, b, nullptr, evb[r + 1], 0, d2, 1L << 18);
                    break;
                }
                if (mode == 4 || mode == 6)
                    hipExtLaunchKernelGGL(busy, dim3((unsigned)(n / 256)), dim3(256), 0, a, nullptr, ev[r], 0, d, n);
                else
                    hipLaunchKernelGGL(busy, dim3((unsigned)(n / 256)), dim3(256), 0, a, d, n);
            }
            CK(hipDeviceSynchronize());
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / R;
            static const char* names[] = {"back-to-back", "+ satisfied wait", "+ record", "+ wait + record", "stopEvent on kernel",
                                          "two-stream pattern (records)", "two-stream pattern (stop events)"};
            printf("fence=%d  %-34s %8.2f us per launch\n", fence, names[mode], us);
            if (mode == 5 || mode == 6) { CK(hipEventRecord(evb[0], b)); }
        }
        CK(hipEventRecord(evb[0], b));
    }
    return 0;
}
