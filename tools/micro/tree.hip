// Micro-benchmark of the record combine in isolation (rocprofv3 --kernel-trace): k_finish_cols on column-major
// tile records (the product path), and the 16:1 fold used above 1024 records.
#include "../../mppi-tf_amd/csrc/mppi_kernels.hip.h"
#include <cstdio>
using namespace mppi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_writer(float* recs, int nb, int ncol) {  // stands in for the rollout kernel's record stores (column-major)
    for (int c = threadIdx.x; c < ncol; c += blockDim.x) recs[(size_t)c * nb + blockIdx.x] = c == 0 ? 10.f + (blockIdx.x % 7) : 0.001f * c;
}

int main() {
    const int HA = 192, ncol = 194, nb = 1024;
    float *recs, *part2, *U0, *U1, *u, *dbg; unsigned long long* step;
    CK(hipMalloc(&recs, sizeof(float) * nb * ncol)); CK(hipMalloc(&part2, sizeof(float) * 64 * ncol));
    CK(hipMalloc(&U0, 4 * (HA + 3))); CK(hipMalloc(&U1, 4 * (HA + 3))); CK(hipMalloc(&u, 64)); CK(hipMalloc(&dbg, 64)); CK(hipMalloc(&step, 8));
    CK(hipMemset(U0, 0, 4 * (HA + 3))); CK(hipMemset(U1, 0, 4 * (HA + 3))); CK(hipMemset(step, 0, 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 50; ++rep) {
        hipLaunchKernelGGL(k_writer, dim3(nb), dim3(256), 0, st, recs, nb, ncol);
        hipLaunchKernelGGL(k_finish_cols, dim3(HA), dim3(kThreads), 0, st, recs, 1, nb, nb, HA, 3, -1.0f, U0, U1, u, (float*)nullptr, 1, step, dbg);
        hipLaunchKernelGGL(k_writer, dim3(nb), dim3(256), 0, st, recs, nb, ncol);
        hipLaunchKernelGGL(k_combine_group, dim3(64), dim3(kThreads), 0, st, recs, 1, nb, nb, HA, -1.0f, part2);
        hipLaunchKernelGGL(k_finish_cols, dim3(HA), dim3(kThreads), 0, st, part2, ncol, 1, 8, HA, 3, -1.0f, U0, U1, u, (float*)nullptr, 1, step, dbg);
    }
    CK(hipStreamSynchronize(st));
    printf("done\n");
    return 0;
}
