// mppi_device.hip.h — device-side arithmetic of the MPPI control step for gfx950 (CDNA4).
//
// Layout of one control step on the GPU:
//   k_rollout_tile : one workgroup owns a TILE of R rollouts for the whole horizon.
//       phase A  all waves fill the tile's noise eps[tau*a][R] in LDS (Philox4x32-10 through
//                rocRAND's engine, or a coalesced read of injected noise from HBM)
//       phase B  one lane per rollout runs the H-step recurrence + cost out of LDS
//       phase C  tile-local soft-min: (beta_b, eta_b, V_b[tau*a]) -> one partial record
//   k_finish       : fixed-order combine of the partial records with exp(-(beta_b-beta)/λ)
//                    rescale, U' = U + V/η, emit u, shift (or emit a shard record).
// The noise never leaves the CU in Philox mode: it is written to LDS once and read twice.
//
// The arithmetic follows the reference op by op (file:line cited per function, relative to
// the reference checkout) and is compiled with -ffp-contract=off, so rollout costs are
// bit-identical to an unfused fp32 evaluation of the reference's matmul chain.
#pragma once

#include <hip/hip_runtime.h>
#include <rocrand/rocrand_philox4x32_10.h>
#include <rocrand/rocrand_normal.h>

#include <stdint.h>

#include "mppi_c.h"

namespace mppi {

constexpr int kMaxS = MPPI_MAX_S;
constexpr int kMaxA = MPPI_MAX_A;
constexpr int kThreads = 256;       // 4 wavefronts of 64
constexpr int kFinishThreads = 1024;

// Device-resident constants of one controller (one copy in HBM, read through scalar loads:
// every access below is wave-uniform).
struct DevConsts {
    int K_local;      // samples owned by this handle
    int k_offset;     // global index of the first owned sample (Philox subsequence base)
    int H, s, a;
    int q_full;       // 1: Q is a dense [s,s] matrix (Py StaticCost), 0: diagonal (C++ Diag(in_Q))
    int action_cost_kind;
    int model_kind;
    float lambda;
    float neg_inv_lambda; // {{-1.f/m_lambda}}  controller_base.cpp:171
    float gamma, upsilon;
    float py_ncoef;       // λ(1-1/υ)            cost_base.py:157
    float dt;
    float bp, bq;         // ((dt*dt)/2)/m , dt/m   model_base.cpp:72-79
    unsigned long long seed;
    float goal[kMaxS];
    float qdiag[kMaxS];
    float sigma[kMaxA * kMaxA];     // [a,a] row-major, FIXED stride kMaxA, zero padded
    float sigma_inv[kMaxA * kMaxA]; // [a,a] row-major, FIXED stride kMaxA, zero padded
    float qfull[kMaxS * kMaxS];     // [s,s] row-major, FIXED stride kMaxS, zero padded
};

// ----------------------------------------------------------------------------------------
// Noise: Philox4x32-10 evaluated AT a counter through rocRAND's own engine. Equivalent to
//   rocrand_init(seed, subsequence, 4*block, &st); rocrand_normal4(&st);
// but costs one 10-round evaluation instead of two (rocrand4() pre-computes the next block).
// Replaces RandomNormal(seed=1) of controller_base.cpp:196-199 (TF's stream itself is
// unpinned).
struct PhiloxAt : rocrand_device::philox4x32_10_engine {
    __device__ PhiloxAt(unsigned long long seed, unsigned long long subsequence, unsigned long long offset)
        : rocrand_device::philox4x32_10_engine(seed, subsequence, offset) {}
    __device__ uint4 block() const { return m_state.result; }
};

// z[0..A) standard normals of (global sample gk, control step `step`, horizon index t).
// Counter layout: subsequence = gk, block = (step*H + t)*ceil(A/4) + q  — a function of the
// GLOBAL sample index only, so results do not depend on how K is sharded over GPUs.
template <int A>
__device__ __forceinline__ void normals_at(unsigned long long seed, unsigned long long gk,
                                           unsigned long long step_h_plus_t, float (&z)[A])
{
    constexpr int A4 = (A + 3) / 4;
#pragma unroll
    for (int q = 0; q < A4; ++q) {
        PhiloxAt eng(seed, gk, 4ull * (step_h_plus_t * A4 + q));
        const float4 n = rocrand_device::detail::normal_distribution4(eng.block());
        if (4 * q + 0 < A) z[4 * q + 0] = n.x;
        if (4 * q + 1 < A) z[4 * q + 1] = n.y;
        if (4 * q + 2 < A) z[4 * q + 2] = n.z;
        if (4 * q + 3 < A) z[4 * q + 3] = n.w;
    }
}

// eps = Σ · z   (controller_base.cpp:201 BatchMatMulV2(sigma, rng): Σ multiplies z directly)
template <int A>
__device__ __forceinline__ void scale_noise(const DevConsts *__restrict__ C, const float (&z)[A], float (&e)[A])
{
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < A; ++j) acc = acc + C->sigma[i * kMaxA + j] * z[j];
        e[i] = acc;
    }
}

// ----------------------------------------------------------------------------------------
// model_base.cpp:53-82  x' = A x + (B/m) v with A = I⊗[[1,dt],[0,1]], B = I⊗[[dt²/2],[dt]]/m.
// The dense rows reduce to (products with the structural zeros are exact):
//   free_p = p + dt*q ; free_q = q ; act_p = bp*v ; act_q = bq*v ; x' = free + act.
template <int A>
__device__ __forceinline__ void pm_free_step(const DevConsts *__restrict__ C, const float (&x)[2 * A], float (&fr)[2 * A])
{
#pragma unroll
    for (int j = 0; j < A; ++j) {
        fr[2 * j] = x[2 * j] + C->dt * x[2 * j + 1];
        fr[2 * j + 1] = x[2 * j + 1];
    }
}

template <int A>
__device__ __forceinline__ void pm_action_step(const DevConsts *__restrict__ C, const float (&v)[A], float (&ac)[2 * A])
{
#pragma unroll
    for (int j = 0; j < A; ++j) {
        ac[2 * j] = C->bp * v[j];
        ac[2 * j + 1] = C->bq * v[j];
    }
}

template <int A>
__device__ __forceinline__ void pm_step(const DevConsts *__restrict__ C, float (&x)[2 * A], const float (&v)[A])
{
    float fr[2 * A], ac[2 * A];
    pm_free_step<A>(C, x, fr);
    pm_action_step<A>(C, v, ac);
#pragma unroll
    for (int i = 0; i < 2 * A; ++i) x[i] = fr[i] + ac[i];
}

// cost_base.cpp:56-61 mStateCost: diff = x-g ; left = Q·diff ; cost = diffᵀ·left.
template <int S, bool QFULL>
__device__ __forceinline__ float state_cost(const DevConsts *__restrict__ C, const float (&x)[S])
{
    float diff[S], left[S];
#pragma unroll
    for (int i = 0; i < S; ++i) diff[i] = x[i] - C->goal[i];
    if (QFULL) {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < S; ++j) acc = acc + C->qfull[i * kMaxS + j] * diff[j];
            left[i] = acc;
        }
    } else {
#pragma unroll
        for (int i = 0; i < S; ++i) left[i] = C->qdiag[i] * diff[i];
    }
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) acc = acc + diff[i] * left[i];
    return acc;
}

// cost_base.cpp:63-68 (C++: λ·uᵀ(Σ⁻¹ε), u = NOMINAL action) or cost_base.py:114-170 (γ/υ form).
template <int A>
__device__ __forceinline__ float action_cost(const DevConsts *__restrict__ C, const float (&u)[A], const float (&e)[A])
{
    float rhsN[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < A; ++j) acc = acc + C->sigma_inv[i * kMaxA + j] * e[j];
        rhsN[i] = acc;
    }
    float mix = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) mix = mix + u[i] * rhsN[i];
    if (C->action_cost_kind == MPPI_ACTION_COST_CPP) return C->lambda * mix;

    float rhsA[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < A; ++j) acc = acc + C->sigma_inv[i * kMaxA + j] * u[j];
        rhsA[i] = acc;
    }
    mix = 2.0f * mix;
    float n = 0.0f, ac = 0.0f;
#pragma unroll
    for (int i = 0; i < A; ++i) n = n + e[i] * rhsN[i];
#pragma unroll
    for (int i = 0; i < A; ++i) ac = ac + u[i] * rhsA[i];
    ac = C->gamma * ac;
    mix = C->gamma * mix;
    n = C->py_ncoef * n;
    const float control = ac + mix;
    return 0.5f * (control + n);
}

// ----------------------------------------------------------------------------------------
// wavefront (64-lane) butterflies: fixed order, deterministic.
__device__ __forceinline__ float wave_min(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fminf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_max(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
    return v;
}
__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

} // namespace mppi
