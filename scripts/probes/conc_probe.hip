// How many kernels from different HIP streams does the device really run at once?  K streams, M back-to-back launches each of a
// one-workgroup kernel that spins for `us` microseconds (s_memrealtime, 100 MHz): concurrency = K * M * us / wall.
// Once with plain launches, once with the M launches of a stream captured into a graph (what bbme_estimate replays).
//   hipcc --offload-arch=gfx950 -O2 -o conc_probe conc_probe.hip ; GPU_MAX_HW_QUEUES=16 ./conc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
__global__ void spin(unsigned ticks, unsigned *out)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (out && threadIdx.x == 0 && ticks == 0xffffffffu) *out = 1;
}
int main(int argc, char **argv)
{
    const int us = argc > 1 ? atoi(argv[1]) : 50, M = argc > 2 ? atoi(argv[2]) : 40, wgs = argc > 3 ? atoi(argv[3]) : 1;
    printf("GPU_MAX_HW_QUEUES=%s, kernel = %d workgroup(s) x 64 threads spinning %d us, %d launches per stream\n",
           getenv("GPU_MAX_HW_QUEUES") ? getenv("GPU_MAX_HW_QUEUES") : "(unset)", wgs, us, M);
    for (int graph = 0; graph < 2; ++graph)
        for (int K : {1, 2, 3, 4, 6, 8, 12, 16}) {
            std::vector<hipStream_t> s(K);
            std::vector<hipGraphExec_t> g(K, nullptr);
            for (auto &x : s) OK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
            if (graph)
                for (int k = 0; k < K; ++k) {
                    hipGraph_t gr;
                    OK(hipStreamBeginCapture(s[k], hipStreamCaptureModeThreadLocal));
                    for (int m = 0; m < M; ++m) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s[k], us * 100u, nullptr);
                    OK(hipStreamEndCapture(s[k], &gr));
                    OK(hipGraphInstantiate(&g[k], gr, nullptr, nullptr, 0));
                    OK(hipGraphDestroy(gr));
                }
            double best = 1e9;
            for (int rep = 0; rep < 3; ++rep) {
                OK(hipDeviceSynchronize());
                const auto t0 = std::chrono::steady_clock::now();
                if (graph) for (int k = 0; k < K; ++k) OK(hipGraphLaunch(g[k], s[k]));
                else for (int m = 0; m < M; ++m) for (int k = 0; k < K; ++k) hipLaunchKernelGGL(spin, dim3(wgs), dim3(64), 0, s[k], us * 100u, nullptr);
                for (auto &x : s) OK(hipStreamSynchronize(x));
                best = std::min(best, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count());
            }
            printf("%s K=%2d streams: wall %8.3f ms, per-kernel slot %6.1f us, concurrency %.2f\n", graph ? "graph" : "plain", K, best * 1e3,
                   best * 1e6 / M, K * M * (double)us / (best * 1e6));
            for (auto &x : g) if (x) OK(hipGraphExecDestroy(x));
            for (auto &x : s) OK(hipStreamDestroy(x));
        }
    return 0;
}
