// What does a timing event cost on the stream?  Chains of small dependent kernels with (a) nothing between them, (b) a hipEventRecord between each pair,
// (c) the events attached to the launches themselves (hipExtLaunchKernelGGL start / stop events: no marker packet of their own).
// build: hipcc -O2 --offload-arch=gfx950 -o scratch/ubench/event_cost scratch/ubench/event_cost.hip
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void small(unsigned* p, int n) { int i = blockIdx.x * blockDim.x + threadIdx.x; if (i < n) p[i] += 1; }
__global__ void big(unsigned* p, size_t n) { for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = p[i] * 3 + 1; }
static double now() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main() {
    unsigned* d; size_t N = 64u << 20; hipMalloc(&d, N * 4); hipMemset(d, 0, N * 4);
    hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    const int K = 8, REP = 200;
    std::vector<hipEvent_t> ev(2 * K); for (auto& e : ev) hipEventCreate(&e);
    auto run = [&](int mode, bool with_big) {
        hipStreamSynchronize(s);
        double t0 = now();
        for (int r = 0; r < REP; ++r) {
            for (int k = 0; k < K; ++k) {
                if (mode == 1) hipEventRecord(ev[2 * k], s);
                if (with_big && k == 0) {
                    if (mode == 2) hipExtLaunchKernelGGL(big, dim3(2048), dim3(256), 0, s, ev[0], ev[1], 0, d, N);
                    else hipLaunchKernelGGL(big, dim3(2048), dim3(256), 0, s, d, N);
                } else {
                    if (mode == 2) hipExtLaunchKernelGGL(small, dim3(64), dim3(256), 0, s, ev[2 * k], ev[2 * k + 1], 0, d, 16384);
                    else hipLaunchKernelGGL(small, dim3(64), dim3(256), 0, s, d, 16384);
                }
                if (mode == 1) hipEventRecord(ev[2 * k + 1], s);
            }
        }
        hipStreamSynchronize(s);
        double dt = now() - t0;
        float ms0 = 0, ms_span = 0;
        if (mode) { hipEventElapsedTime(&ms0, ev[0], ev[1]); hipEventElapsedTime(&ms_span, ev[0], ev[2 * K - 1]); }
        printf("mode %d (%s) big=%d: %.2f us per kernel slot; event time of slot 0: %.2f us, span of the %d slots of the last rep by events: %.2f us\n", mode,
               mode == 0 ? "no events" : mode == 1 ? "hipEventRecord around every kernel" : "events attached to the launches (hipExtLaunchKernelGGL)", (int)with_big, dt * 1e3 / (REP * K), ms0 * 1e3, K, ms_span * 1e3);
    };
    for (int rep = 0; rep < 2; ++rep) for (int b = 0; b < 2; ++b) for (int m = 0; m < 3; ++m) run(m, b != 0);
    return 0;
}
