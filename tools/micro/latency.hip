// Micro-benchmarks of small-kernel latency on gfx950 (rocprofv3 --kernel-trace gives durations).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty() {}
__global__ void k_ldst(const float* in, float* out) { out[threadIdx.x] = in[threadIdx.x] + 1.f; }
__global__ void k_chain4(const int* idx, float* out) {  // 4 dependent loads
    int i = idx[threadIdx.x & 63]; i = idx[i]; i = idx[i]; i = idx[i]; out[threadIdx.x] = (float)i;
}
__global__ void k_sync6(const float* in, float* out) {
    __shared__ float s[1024];
    float v = in[threadIdx.x];
    for (int r = 0; r < 6; ++r) { s[threadIdx.x] = v; __syncthreads(); v = s[(threadIdx.x + 64) & 1023] + 1.f; __syncthreads(); }
    out[threadIdx.x] = v;
}
__global__ void k_dbl(const float* in, float* out) {
    double a = in[threadIdx.x];
    for (int r = 0; r < 16; ++r) a = a * 1.0000001 + (double)in[(threadIdx.x + r) & 1023];
    out[threadIdx.x] = (float)a;
}
__global__ void k_expf(const float* in, float* out) { out[threadIdx.x] = expf(in[threadIdx.x]); }
__global__ void k_rmw(unsigned long long* c) { if (threadIdx.x == 0) c[0] = c[0] + 1ull; }
__global__ void k_sload(const float* __restrict__ c, float* out) { out[threadIdx.x] = c[0] + c[17] + c[300]; }

int main() {
    float *in, *out; int* idx; unsigned long long* ctr;
    CK(hipMalloc(&in, 4096 * 4)); CK(hipMalloc(&out, 4096 * 4)); CK(hipMalloc(&idx, 64 * 4)); CK(hipMalloc(&ctr, 8));
    std::vector<int> h(64); for (int i = 0; i < 64; ++i) h[i] = (i * 7 + 3) & 63;
    CK(hipMemcpy(idx, h.data(), 256, hipMemcpyHostToDevice)); CK(hipMemset(in, 0, 4096 * 4)); CK(hipMemset(ctr, 0, 8));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int rep = 0; rep < 50; ++rep) {
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, st);
        hipLaunchKernelGGL(k_empty, dim3(1), dim3(1024), 0, st);
        hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st);
        hipLaunchKernelGGL(k_ldst, dim3(1), dim3(1024), 0, st, in, out);
        hipLaunchKernelGGL(k_chain4, dim3(1), dim3(1024), 0, st, idx, out);
        hipLaunchKernelGGL(k_sync6, dim3(1), dim3(1024), 0, st, in, out);
        hipLaunchKernelGGL(k_dbl, dim3(1), dim3(1024), 0, st, in, out);
        hipLaunchKernelGGL(k_expf, dim3(1), dim3(1024), 0, st, in, out);
        hipLaunchKernelGGL(k_rmw, dim3(1), dim3(64), 0, st, ctr);
        hipLaunchKernelGGL(k_sload, dim3(1), dim3(1024), 0, st, in, out);
        hipLaunchKernelGGL(k_sload, dim3(64), dim3(256), 0, st, in, out);
    }
    CK(hipStreamSynchronize(st));
    printf("done\n");
    return 0;
}
