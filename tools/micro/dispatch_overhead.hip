// dispatch_overhead.hip — what ONE dependent kernel dispatch costs on this part, whatever the kernel does: chains of EMPTY kernels on one
// stream (each depends on the previous one, as rollout -> finish -> rollout ... do), timed (a) by the dispatch's own begin/end timestamps
// (hipExtLaunchKernel start/stop events = what rocprofv3 reports as the kernel's duration) and (b) by wall clock per kernel over the chain.
// An MI355X has 8 XCDs with private L2s: between two dependent dispatches the command processor releases (L2 write-back) and acquires
// (invalidate) — a fixed cost every kernel of a control step pays, and the reason the finish kernel reads ~4.3 us in profiles/ even when
// it returns at once (tools/ablate.py finish_s0).   hipcc -O2 --offload-arch=gfx950 tools/micro/dispatch_overhead.hip -o build/dispatch_overhead
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_empty(float *p) { if (p == nullptr && threadIdx.x == 12345) p[0] = 0.f; }
__global__ void k_touch(float *p) { if (threadIdx.x == 0) p[blockIdx.x] += 1.0f; } // one dword per workgroup: dirty lines in every XCD's L2

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

int main()
{
    hipStream_t st;
    CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    float *buf;
    CK(hipMalloc((void **)&buf, sizeof(float) * 4096));
    CK(hipMemset(buf, 0, sizeof(float) * 4096));
    const int n = 400;
    std::vector<hipEvent_t> ev(2 * n);
    for (auto &e : ev) CK(hipEventCreate(&e));
    printf("[\n");
    bool first = true;
    for (int touch = 0; touch < 2; ++touch)
        for (int grid : {1, 192, 1024}) {
            for (int lds : {0, 26 * 1024}) {
                for (int i = 0; i < 50; ++i) hipLaunchKernelGGL(touch ? k_touch : k_empty, dim3(grid), dim3(256), lds, st, buf);
                CK(hipStreamSynchronize(st));
                const auto t0 = std::chrono::steady_clock::now();
                for (int i = 0; i < n; ++i)
                    hipExtLaunchKernelGGL(touch ? k_touch : k_empty, dim3(grid), dim3(256), lds, st, ev[2 * i], ev[2 * i + 1], 0, buf);
                CK(hipStreamSynchronize(st));
                const double wall = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / n;
                double evs = 0.0;
                for (int i = 0; i < n; ++i) { float ms; CK(hipEventElapsedTime(&ms, ev[2 * i], ev[2 * i + 1])); evs += ms * 1e3; }
                printf("%s {\"kernel\": \"%s\", \"workgroups\": %d, \"threads\": 256, \"lds_bytes\": %d, \"event_begin_to_end_us\": %.2f, \"wall_per_dependent_kernel_us\": %.2f}",
                       first ? "" : ",\n", touch ? "one dword store per workgroup" : "empty", grid, lds, evs / n, wall);
                first = false;
            }
        }
    printf("\n]\n");
    return 0;
}
