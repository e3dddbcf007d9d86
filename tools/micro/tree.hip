// Micro-benchmark of the record tree (k_combine_group + k_finish) in isolation (rocprofv3 --kernel-trace).
#include "../../mppi-tf_amd/csrc/mppi_kernels.hip.h"
#include <cstdio>
#include <vector>
using namespace mppi;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_writer(float* recs, int stride) {  // stands in for the rollout kernel's record stores
    float* r = recs + (size_t)blockIdx.x * stride;
    for (int c = threadIdx.x; c < stride; c += blockDim.x) r[c] = c == 0 ? 10.f + (blockIdx.x % 7) : 0.001f * c;
}
__global__ void k_reader_simple(const float* recs, int n, float* out) { // 64 blocks: each thread sums 16 strided values
    int stride = 194; float acc = 0;
    for (int b = 0; b < 16; ++b) acc += recs[(size_t)(blockIdx.x * 16 + b) * stride + threadIdx.x % stride];
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    const int HA = 192, stride = 194, nb = 1024, ng = 64;
    float *recs, *part2, *U, *u, *dbg, *Uu, *out; unsigned long long* step;
    CK(hipMalloc(&recs, sizeof(float) * nb * stride)); CK(hipMalloc(&part2, sizeof(float) * ng * stride));
    CK(hipMalloc(&U, 4 * HA)); CK(hipMalloc(&u, 64)); CK(hipMalloc(&dbg, 64)); CK(hipMalloc(&Uu, 4 * HA)); CK(hipMalloc(&step, 8)); CK(hipMalloc(&out, 4 * 64 * 256));
    CK(hipMemset(U, 0, 4 * HA)); CK(hipMemset(step, 0, 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 60; ++rep) {
        const bool with_writer = rep < 30;
        if (with_writer) hipLaunchKernelGGL(k_writer, dim3(nb), dim3(256), 0, st, recs, stride);
        hipLaunchKernelGGL(k_reader_simple, dim3(ng), dim3(256), 0, st, recs, nb, out);
        if (with_writer) hipLaunchKernelGGL(k_writer, dim3(nb), dim3(256), 0, st, recs, stride);
        hipLaunchKernelGGL(k_combine_group, dim3(ng), dim3(kThreads), 0, st, recs, nb, HA, -1.0f, part2);
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishThreads), finish_lds_bytes(HA), st, part2, ng, HA, 3, -1.0f, U, u, (float*)nullptr, 1, step, dbg, Uu);
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishThreads), finish_lds_bytes(HA), st, part2, 2, HA, 3, -1.0f, U, u, (float*)nullptr, 1, step, dbg, Uu);
        hipLaunchKernelGGL(k_finish, dim3(1), dim3(kFinishThreads), finish_lds_bytes(HA), st, part2, 2, HA, 3, -1.0f, U, u, (float*)nullptr, 0, step, (float*)nullptr, (float*)nullptr);

    }
    CK(hipStreamSynchronize(st));
    printf("done\n");
    return 0;
}
