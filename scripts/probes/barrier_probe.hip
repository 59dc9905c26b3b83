// What does a kernel boundary cost against a grid barrier inside one persistent kernel?  (the
// regulariser is ~70 launches of which ~40 are a handful of dependent memory trips; would a level's sweeps in ONE launch pay?)
// P phases; a phase = every workgroup takes one dependent agent-scope (sc1) load trip, like the first trip of a solver launch.
//   (a) P launches of a G-workgroup kernel, captured in a hipGraph (what bbme_estimate replays);
//   (b) ONE launch of G workgroups with a grid barrier between the phases: one no-return atomicAdd per workgroup on a counter,
//       then lane 0 polls the counter (agent scope) with s_sleep; every spin is capped, a cap raises a flag and ends the kernel.
//   hipcc --offload-arch=gfx950 -O2 -o barrier_probe barrier_probe.hip ; ./barrier_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned trip(const unsigned *buf, unsigned idx, unsigned n)
{
    return __hip_atomic_load(buf + (idx * 2654435761u) % n, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__global__ __launch_bounds__(256) void phase_kernel(const unsigned *buf, unsigned n, unsigned phase, unsigned *out)
{
    const unsigned v = trip(buf, blockIdx.x * 256u + threadIdx.x + phase * 7919u, n);
    if (v == 0xdeadbeefu) out[0] = v;                      // never true: keeps the load
}

__global__ __launch_bounds__(256) void persistent_kernel(const unsigned *buf, unsigned n, unsigned phases, unsigned *counter, unsigned *out)
{
    __shared__ unsigned s_ok;
    for (unsigned p = 0; p < phases; ++p) {
        const unsigned v = trip(buf, blockIdx.x * 256u + threadIdx.x + p * 7919u, n);
        if (v == 0xdeadbeefu) out[0] = v;
        // grid barrier: everything this workgroup wrote is at the memory side (the solver drains its stores the same way), then
        // one arrival per workgroup; lane 0 waits until all gridDim.x have arrived in this phase
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            atomicAdd(counter, 1u);
            const unsigned target = (p + 1) * gridDim.x;
            unsigned ok = 0;
            for (unsigned spin = 0; spin < 200000u; ++spin) {
                if (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= target) { ok = 1; break; }
                __builtin_amdgcn_s_sleep(2);
            }
            s_ok = ok;
            if (!ok) out[1] = 1;                            // the cap: reported, and every workgroup leaves
        }
        __syncthreads();
        if (!s_ok) return;
    }
}

int main()
{
    const unsigned n = 1u << 24;
    unsigned *buf, *counter, *out;
    OK(hipMalloc(&buf, n * 4)); OK(hipMemset(buf, 0, n * 4));
    OK(hipMalloc(&counter, 4)); OK(hipMalloc(&out, 8)); OK(hipMemset(out, 0, 8));
    hipStream_t s;
    OK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    const unsigned P = 16;
    for (int G : {8, 32, 64, 128, 256, 512}) {
        hipGraph_t gr; hipGraphExec_t ge;
        OK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
        for (unsigned p = 0; p < P; ++p) hipLaunchKernelGGL(phase_kernel, dim3(G), dim3(256), 0, s, buf, n, p, out);
        OK(hipStreamEndCapture(s, &gr));
        OK(hipGraphInstantiate(&ge, gr, nullptr, nullptr, 0));
        OK(hipGraphDestroy(gr));
        double best_a = 1e9, best_b = 1e9;
        for (int rep = 0; rep < 6; ++rep) {
            OK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 20; ++i) OK(hipGraphLaunch(ge, s));
            OK(hipStreamSynchronize(s));
            best_a = std::min(best_a, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 20);
        }
        for (int rep = 0; rep < 6; ++rep) {
            OK(hipStreamSynchronize(s));
            auto t0 = std::chrono::steady_clock::now();
            for (int i = 0; i < 20; ++i) {
                OK(hipMemsetAsync(counter, 0, 4, s));
                hipLaunchKernelGGL(persistent_kernel, dim3(G), dim3(256), 0, s, buf, n, P, counter, out);
            }
            OK(hipStreamSynchronize(s));
            best_b = std::min(best_b, std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() / 20);
        }
        unsigned flags[2];
        OK(hipMemcpy(flags, out, 8, hipMemcpyDeviceToHost));
        printf("%3d workgroups x 256 threads, %u phases: graph of %u launches %7.2f us (%5.2f us per phase) | one persistent launch (+ its memset) %7.2f us "
               "(%5.2f us per phase)%s\n", G, P, P, best_a * 1e6, best_a * 1e6 / P, best_b * 1e6, best_b * 1e6 / P, flags[1] ? "  BARRIER CAP HIT" : "");
        OK(hipGraphExecDestroy(ge));
    }
    return 0;
}
