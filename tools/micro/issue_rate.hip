// Instruction issue-rate micro-benchmark for gfx950: N independent chains per thread, many waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int ITERS = 2048;

__global__ void k_fmul(float* out, float s) { float a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i;
  for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = a[i] * s; }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
__global__ void k_xor(unsigned* out, unsigned s) { unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = (a[i] ^ s) + it; }
  unsigned r = 0; for (int i = 0; i < 8; ++i) r += a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
__global__ void k_mad64(unsigned* out, unsigned s) { unsigned a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x + i;
  for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) { unsigned long long m = (unsigned long long)a[i] * 0xD2511F53u; a[i] = (unsigned)(m >> 32) ^ (unsigned)m; } }
  unsigned r = 0; for (int i = 0; i < 8; ++i) r += a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
__global__ void k_log(float* out, float s) { float a[8]; for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 0.001f + i + 2.f;
  for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = __builtin_amdgcn_logf(a[i]) + s; }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i]; out[blockIdx.x * blockDim.x + threadIdx.x] = r; }
__global__ void k_pkmul(float* out, float s) { typedef float f2 __attribute__((ext_vector_type(2))); f2 a[8]; for (int i = 0; i < 8; ++i) a[i] = (f2){threadIdx.x * 0.001f + i, 1.f};
  f2 ss = {s, s}; for (int it = 0; it < ITERS; ++it) { _Pragma("unroll") for (int i = 0; i < 8; ++i) a[i] = a[i] * ss; }
  float r = 0; for (int i = 0; i < 8; ++i) r += a[i].x + a[i].y; out[blockIdx.x * blockDim.x + threadIdx.x] = r; }

int main() {
  float* o; CK(hipMalloc(&o, 4 * 1024 * 1024 * 4)); hipStream_t st; CK(hipStreamCreate(&st));
  const int grids[3] = {1024, 2048, 4096};   // x256 threads = 4 / 8 / 16 waves per CU... (256 CUs)
  for (int rep = 0; rep < 3; ++rep) for (int gi = 0; gi < 3; ++gi) { int g = grids[gi];
    hipLaunchKernelGGL(k_fmul, dim3(g), dim3(256), 0, st, o, 1.0001f);
    hipLaunchKernelGGL(k_xor, dim3(g), dim3(256), 0, st, (unsigned*)o, 12345u);
    hipLaunchKernelGGL(k_mad64, dim3(g), dim3(256), 0, st, (unsigned*)o, 12345u);
    hipLaunchKernelGGL(k_log, dim3(g), dim3(256), 0, st, o, 1.5f);
    hipLaunchKernelGGL(k_pkmul, dim3(g), dim3(256), 0, st, o, 1.0001f);
  }
  CK(hipStreamSynchronize(st)); printf("done\n"); return 0; }
