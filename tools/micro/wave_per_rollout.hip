// wave_per_rollout.hip — SURVEY §7 asked for a number: north_star's literal layout ("one wavefront per rollout, trajectory
// and running cost in LDS") against one LANE per rollout, on BASELINE configs[2] (point_mass3d, K=65536, H=64).
// Both kernels here are plain restatements of rollout + cost (same Philox noise, same per-step arithmetic from
// mppi_device.hip.h; no tile soft-min record, no producer/consumer split), so the comparison isolates the layout:
//   k_wave_per_rollout: wave = rollout, lane = time step. Noise (one horizon group per lane 0..15, shared through LDS),
//       v = u + eps and the action cost are lane-parallel; the recurrence x_{t+1} = A x_t + B v_t is sequential in t —
//       every step broadcasts v_t from lane t (v_readlane) and all 64 lanes compute the same 6 floats — and writes the
//       trajectory to LDS; state costs are lane-parallel again, summed with a wave reduction.
//   k_lane_per_rollout: lane = rollout, everything lane-local, sequential in t.
// Prints both durations and the largest relative difference of the sample costs (the per-rollout sums differ only in
// summation order).   Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off -std=c++17 -I include tools/micro/wave_per_rollout.hip -o build/wave_per_rollout
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <vector>
#include "../../mppi-tf_amd/csrc/mppi_kernels.hip.h"
using namespace mppi;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
constexpr int A = 3, S = 6;

__global__ __launch_bounds__(256) void k_wave_per_rollout(const DevConsts *__restrict__ C, const float *__restrict__ x0,
                                                          const float *__restrict__ U, float *__restrict__ cost)
{
    __shared__ float z_s[4][16][4 * A];   // per wave: 16 horizon groups x 12 normals
    __shared__ float traj[4][65][S];      // per wave: x_1 .. x_H (+ x_0)
    const int w = threadIdx.x >> 6, t = threadIdx.x & 63;
    const int k = blockIdx.x * 4 + w;
    const int H = C->H; // 64: one lane per time step
    if (t < 16) {
        float z[4 * A];
        normals_group<A>(C->seed, (unsigned long long)k, (unsigned long long)t, z);
#pragma unroll
        for (int i = 0; i < 4 * A; ++i) z_s[w][t][i] = z[i];
    }
    __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): LDS of one wave is in order
    __builtin_amdgcn_wave_barrier();
    float zt[A], e[A], u[A], v[A];
#pragma unroll
    for (int i = 0; i < A; ++i) { zt[i] = z_s[w][t >> 2][(t & 3) * A + i]; u[i] = U[t * A + i]; }
    scale_noise<A, true>(C, zt, e);
#pragma unroll
    for (int i = 0; i < A; ++i) v[i] = u[i] + e[i];
    const float ac = action_cost<A, true>(C, u, e);
    float x[S];
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x0[i];
    for (int tt = 0; tt < H; ++tt) { // the sequential part: 64 lanes, one useful result
        float vt[A];
#pragma unroll
        for (int i = 0; i < A; ++i) vt[i] = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v[i]), tt));
        pm_step<A>(C, x, vt);
        if (t == 0) {
#pragma unroll
            for (int i = 0; i < S; ++i) traj[w][tt + 1][i] = x[i];
        }
    }
    __builtin_amdgcn_s_waitcnt(0xc07f);
    __builtin_amdgcn_wave_barrier();
    float xt[S];
#pragma unroll
    for (int i = 0; i < S; ++i) xt[i] = traj[w][t + 1][i];
    const float sc = state_cost<S, false>(C, xt);
    float c = wave_sum(sc + ac);
    c = c + state_cost<S, false>(C, x); // terminal cost
    if (t == 0) cost[k] = c;
}

__global__ __launch_bounds__(256) void k_lane_per_rollout(const DevConsts *__restrict__ C, const float *__restrict__ x0,
                                                          const float *__restrict__ U, float *__restrict__ cost)
{
    const int k = blockIdx.x * 256 + threadIdx.x;
    const int H = C->H;
    float x[S], c = 0.0f, z[4 * A];
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x0[i];
    for (int g = 0; g < H / 4; ++g) {
        normals_group<A>(C->seed, (unsigned long long)k, (unsigned long long)g, z);
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            float zt[A], e[A], u[A], v[A];
#pragma unroll
            for (int i = 0; i < A; ++i) { zt[i] = z[tl * A + i]; u[i] = U[(4 * g + tl) * A + i]; }
            scale_noise<A, true>(C, zt, e);
#pragma unroll
            for (int i = 0; i < A; ++i) v[i] = u[i] + e[i];
            const float ac = action_cost<A, true>(C, u, e);
            pm_step<A>(C, x, v);
            c = c + (state_cost<S, false>(C, x) + ac);
        }
    }
    c = c + state_cost<S, false>(C, x);
    cost[k] = c;
}

int main()
{
    const int K = 65536, H = 64;
    DevConsts c{};
    c.K_local = K; c.H = H; c.s = S; c.a = A; c.lambda = 1.f; c.neg_inv_lambda = -1.f; c.gamma = 1.f; c.upsilon = 1.f; c.dt = 0.1f;
    c.bp = 0.005f; c.bq = 0.1f; c.seed = 1;
    const float goal[S] = {1, 0, .5f, 0, .75f, 0};
    for (int i = 0; i < S; ++i) { c.goal[i] = goal[i]; c.qdiag[i] = 1.f; }
    for (int i = 0; i < A; ++i) { c.sigma[i * kMaxA + i] = 0.25f; c.sigma_inv[i * kMaxA + i] = 4.f; }
    DevConsts *dC; float *x0, *U, *c1, *c2;
    CK(hipMalloc((void **)&dC, sizeof(c))); CK(hipMemcpy(dC, &c, sizeof(c), hipMemcpyHostToDevice));
    CK(hipMalloc((void **)&x0, S * 4)); CK(hipMemset(x0, 0, S * 4));
    CK(hipMalloc((void **)&U, H * A * 4)); CK(hipMemset(U, 0, H * A * 4));
    CK(hipMalloc((void **)&c1, K * 4)); CK(hipMalloc((void **)&c2, K * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms[2] = {0, 0};
    for (int which = 0; which < 2; ++which) {
        for (int r = 0; r < 12; ++r) {
            CK(hipEventRecord(e0, 0));
            if (which == 0) hipLaunchKernelGGL(k_wave_per_rollout, dim3(K / 4), dim3(256), 0, 0, dC, x0, U, c1);
            else hipLaunchKernelGGL(k_lane_per_rollout, dim3(K / 256), dim3(256), 0, 0, dC, x0, U, c2);
            CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
            float m; CK(hipEventElapsedTime(&m, e0, e1));
            if (r >= 2) ms[which] += m / 10;
        }
    }
    std::vector<float> h1(K), h2(K);
    CK(hipMemcpy(h1.data(), c1, K * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h2.data(), c2, K * 4, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int i = 0; i < K; ++i) worst = std::max(worst, std::fabs((double)h1[i] - h2[i]) / std::fabs((double)h2[i]));
    std::printf("{\"workload\": \"point_mass3d K=65536 H=64, rollout + cost only\", \"wave_per_rollout_us\": %.1f, \"lane_per_rollout_us\": %.1f, "
                "\"ratio\": %.1f, \"max_rel_cost_diff\": %.2e}\n", ms[0] * 1e3, ms[1] * 1e3, ms[0] / ms[1], worst);
    return 0;
}
