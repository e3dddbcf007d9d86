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
    int state_cost_kind;  // MPPI_STATE_COST_*
    float ell[7];         // ElipseCost: a, b, cx, cy, speed, m_state, m_vel (elipse_cost.py:10-46)
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

// The per-step helpers below take the constants through a template type CT: DevConsts in HBM, or a kernel-local copy
// of just the fields a wave needs (PcProducerConsts / PcConsumerConsts). A barrier is a memory fence, so constants read
// through a global pointer are re-fetched (s_load + s_waitcnt) after every __syncthreads; a local copy made before the
// first barrier lives in SGPRs for the whole kernel. Same field names and indexing, same arithmetic.
// Where tile b keeps its record. Records are column-major [2 + H*a][nbp] (the finish kernel reads whole columns), so
// a tile's 2 + H*a values are scattered dwords, one per column. Workgroups go to the 8 XCDs round-robin and every XCD
// has its own write-back L2: with slot = (b mod 8) * nbp/8 + b / 8 the 16 floats of a 64-byte line belong to 16 tiles
// of the SAME XCD, whose L2 merges them into one line write (r02: WRITE_SIZE of k_rollout_pc at C3 6.3 MB -> see
// DESIGN §3.3). nbp = the record count padded to a multiple of 128 (16 per line x 8 XCDs); slots no tile owns hold a
// neutral record (beta = kPadBeta, eta = 0, V = 0: weight exp(-(kPadBeta - beta)/lambda) = 0) written once at create.
constexpr float kPadBeta = 3.0e38f; // finite: a group of pads alone combines to (kPadBeta, 0, 0), not to NaN
__host__ __device__ inline int record_pad(int n) { return (n + 127) & ~127; }
__host__ __device__ inline int record_slot(int b, int nbp) { return (b & 7) * (nbp >> 3) + (b >> 3); }

template <int A>
struct PcProducerConsts {
    int action_cost_kind;
    float lambda, gamma, py_ncoef;
    float sigma[A * kMaxA], sigma_inv[A * kMaxA]; // row i at [i*kMaxA + j], j < A
    template <bool DIAG>
    __device__ __forceinline__ void load(const DevConsts *__restrict__ C)
    {
        action_cost_kind = C->action_cost_kind;
        lambda = C->lambda; gamma = C->gamma; py_ncoef = C->py_ncoef;
#pragma unroll
        for (int i = 0; i < A; ++i) {
#pragma unroll
            for (int j = 0; j < A; ++j) {
                if (!DIAG || i == j) {
                    sigma[i * kMaxA + j] = C->sigma[i * kMaxA + j];
                    sigma_inv[i * kMaxA + j] = C->sigma_inv[i * kMaxA + j];
                }
            }
        }
    }
};
template <int S>
struct PcConsumerConsts {
    float dt, bp, bq, neg_inv_lambda;
    float goal[S], qdiag[S];
    __device__ __forceinline__ void load(const DevConsts *__restrict__ C)
    {
        dt = C->dt; bp = C->bp; bq = C->bq; neg_inv_lambda = C->neg_inv_lambda;
#pragma unroll
        for (int i = 0; i < S; ++i) { goal[i] = C->goal[i]; qdiag[i] = C->qdiag[i]; }
    }
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

// Standard normals of one GROUP of 4 consecutive horizon steps of one sample.
// Counter layout (noise of sample gk, control step `step`, horizon steps 4g..4g+3):
//   subsequence = gk (the GLOBAL sample index: results do not depend on how K is sharded)
//   Philox block = (step*ceil(H/4) + g)*A + q , q = 0..A-1   -> 4A uniforms -> 4A normals
//   normal m = 4q+{0,1,2,3} (Box-Muller pairs (x,y),(z,w) of block q) is z[t = 4g + m/A][j = m%A]
// Every Philox output word is used (A blocks per 4 steps instead of 4).
// Box-Muller as rocRAND's box_muller(x, y) (rocrand_normal.h:53-68):
//   u = 2^-32 + x·2^-32 ; v = 2π·2^-32 + y·2π·2^-32 ; (sin v, cos v)·sqrt(-2 ln u)
// with the hardware-rate log and sqrt (v_log_f32·ln2, v_sqrt_f32; ~1 ulp) in place of libm-accurate
// logf/sqrtf, whose range/denormal fix-ups cost ~20 of the ~35 instructions of a pair and buy nothing
// here: u >= 2^-32 is never denormal and the noise only has to be N(0,1). rocRAND itself already uses
// the fast __sincosf. -DMPPI_ROCRAND_NORMALS selects rocRAND's normal_distribution4 verbatim.
__device__ __forceinline__ float2 box_muller_hw(unsigned int x, unsigned int y)
{
    // u and the angle as single fused multiply-adds (within 1 ulp of the two-step form above); the angle is kept in
    // REVOLUTIONS, which is what v_sin_f32 / v_cos_f32 take (2π·2^-32·(y+1) would only be divided by 2π again), and
    // -2·ln2 is one constant: 13 instead of 18 instructions per pair in a VALU-issue-bound kernel.
    const float u = __builtin_fmaf((float)x, ROCRAND_2POW32_INV, ROCRAND_2POW32_INV);
    const float rev = __builtin_fmaf((float)y, ROCRAND_2POW32_INV, ROCRAND_2POW32_INV);
    const float s = __builtin_amdgcn_sqrtf(__builtin_amdgcn_logf(u) * -1.3862943611198906f);
    return float2{__builtin_amdgcn_sinf(rev) * s, __builtin_amdgcn_cosf(rev) * s};
}

// The Philox4x32-10 block function (Random123; constants as rocrand_philox4x32_10.h:60-65) at the counter
// rocrand_init(seed, subsequence, 4*block) would reach: counter = {lo(block), hi(block), lo(subseq), hi(subseq)},
// key = seed. Bit-identical to rocRAND's engine (PhiloxAt above, kept for -DMPPI_ROCRAND_NORMALS and checked
// by the noise parity test); written out so each round is 2 x v_mad_u64_u32 + 2 x v_bitop3_b32 (3-input xor)
// instead of the 6 instructions hipcc emits for the header's two-step xor — the kernel is VALU-issue bound.
__device__ __forceinline__ uint4 philox4x32_10_block(unsigned long long seed, unsigned long long subsequence,
                                                     unsigned long long block)
{
    unsigned int c0 = (unsigned int)block, c1 = (unsigned int)(block >> 32);
    unsigned int c2 = (unsigned int)subsequence, c3 = (unsigned int)(subsequence >> 32);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long m0 = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * c0;
        const unsigned long long m1 = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * c2;
        const unsigned int n0 = __builtin_amdgcn_bitop3_b32((unsigned int)(m1 >> 32), c1, k0, 0x96);
        const unsigned int n2 = __builtin_amdgcn_bitop3_b32((unsigned int)(m0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (unsigned int)m1; c2 = n2; c3 = (unsigned int)m0;
        k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
    }
    return uint4{c0, c1, c2, c3};
}

// The same block function for the rollout kernels' counters, where the BLOCK index (and the key) is wave-uniform and the
// subsequence is a per-lane sample index below 2^32 (k_offset and K are ints): in the first three rounds half of the state
// is then uniform, so those products run on the scalar unit and the three-input xors become a scalar xor + ONE v_xor_b32 with
// an SGPR operand — instead of a v_bitop3_b32 (1.7x the issue cost of v_xor) that needs two v_mov to bring its two scalar
// operands into VGPRs (VOP3 reads one SGPR). Bit-identical to philox4x32_10_block(seed, subsequence, block) by construction
// (same integer arithmetic, regrouped xors); checked against rocRAND's engine by the noise parity tests.
__device__ __forceinline__ uint4 philox4x32_10_block_ub(unsigned long long seed, unsigned int subsequence_lo, unsigned long long block_uniform)
{
    const unsigned int c0s = (unsigned int)block_uniform, c1s = (unsigned int)(block_uniform >> 32);
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
    // round 0: c0, c1 uniform; c2 = the lane's sample; c3 = 0
    const unsigned long long m0s = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * c0s; // scalar
    unsigned long long m1 = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * subsequence_lo;
    const unsigned int a0 = (unsigned int)(m1 >> 32) ^ (c1s ^ k0), a1 = (unsigned int)m1;
    const unsigned int a2s = (unsigned int)(m0s >> 32) ^ k1, a3s = (unsigned int)m0s;      // uniform
    k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
    // round 1: c0 = a0, c1 = a1 per lane; c2 = a2s, c3 = a3s uniform
    unsigned long long m0 = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * a0;
    const unsigned long long m1s = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * a2s;        // scalar
    const unsigned int b0 = a1 ^ ((unsigned int)(m1s >> 32) ^ k0), b1s = (unsigned int)m1s;
    const unsigned int b2 = (unsigned int)(m0 >> 32) ^ (a3s ^ k1), b3 = (unsigned int)m0;
    k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
    // round 2: c1 = b1s uniform, the rest per lane
    m0 = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * b0;
    m1 = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * b2;
    unsigned int c0 = (unsigned int)(m1 >> 32) ^ (b1s ^ k0), c1 = (unsigned int)m1;
    unsigned int c2 = __builtin_amdgcn_bitop3_b32((unsigned int)(m0 >> 32), b3, k1, 0x96), c3 = (unsigned int)m0;
    k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
#pragma unroll
    for (int r = 3; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * c0;
        const unsigned long long p1 = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * c2;
        const unsigned int n0 = __builtin_amdgcn_bitop3_b32((unsigned int)(p1 >> 32), c1, k0, 0x96);
        const unsigned int n2 = __builtin_amdgcn_bitop3_b32((unsigned int)(p0 >> 32), c3, k1, 0x96);
        c0 = n0; c1 = (unsigned int)p1; c2 = n2; c3 = (unsigned int)p0;
        k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
    }
    return uint4{c0, c1, c2, c3};
}

// A consecutive Philox blocks of one sample, ROUND-MAJOR: round r of every block before round r+1 of any (r04). One block is a chain
// of 10 dependent rounds, each two independent 32x32->64 products followed by the two three-input xors that need them; written block
// after block (philox4x32_10_block_ub in a loop) hipcc keeps that order and a wave offers the SIMD two independent instructions at a
// time — whenever fewer than four waves of the SIMD are runnable (barriers, the light consumer wave, the tile tails) the chain's
// latency is exposed (SQ_ACTIVE_INST_VALU 12.2 us against 9.4 us at ideal issue rates at C3). With the A blocks of a horizon group
// advanced in lock step there are 2A independent products per round. The same integer arithmetic per block: bit-identical to
// philox4x32_10_block_ub(seed, subsequence_lo, block0_uniform + q) by construction, checked against rocRAND's engine by the noise tests.
template <int A>
__device__ __forceinline__ void philox4x32_10_blocks_ub(unsigned long long seed, unsigned int subsequence_lo, unsigned long long block0_uniform,
                                                        uint4 (&out)[A])
{
    unsigned int k0 = (unsigned int)seed, k1 = (unsigned int)(seed >> 32);
    unsigned int c0[A], c1[A], c2[A], c3[A];
    // round 0: c0, c1 uniform (the block index); c2 = the lane's sample (the same product for every block); c3 = 0
    const unsigned long long m1 = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * subsequence_lo;
    unsigned int a0[A], a2s[A], a3s[A];
    const unsigned int a1 = (unsigned int)m1;
#pragma unroll
    for (int q = 0; q < A; ++q) {
        const unsigned long long blk = block0_uniform + (unsigned long long)q;
        const unsigned int c0s = (unsigned int)blk, c1s = (unsigned int)(blk >> 32);
        const unsigned long long m0s = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * c0s; // scalar
        a0[q] = (unsigned int)(m1 >> 32) ^ (c1s ^ k0);
        a2s[q] = (unsigned int)(m0s >> 32) ^ k1; a3s[q] = (unsigned int)m0s;              // uniform
    }
    k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
    // round 1: c0 = a0, c1 = a1 per lane; c2 = a2s, c3 = a3s uniform
    unsigned int b0[A], b1s[A], b2[A], b3[A];
#pragma unroll
    for (int q = 0; q < A; ++q) {
        const unsigned long long m0 = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * a0[q];
        const unsigned long long m1s = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * a2s[q];   // scalar
        b0[q] = a1 ^ ((unsigned int)(m1s >> 32) ^ k0); b1s[q] = (unsigned int)m1s;
        b2[q] = (unsigned int)(m0 >> 32) ^ (a3s[q] ^ k1); b3[q] = (unsigned int)m0;
    }
    k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
    // round 2: c1 = b1s uniform, the rest per lane
#pragma unroll
    for (int q = 0; q < A; ++q) {
        const unsigned long long m0 = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * b0[q];
        const unsigned long long m1b = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * b2[q];
        c0[q] = (unsigned int)(m1b >> 32) ^ (b1s[q] ^ k0); c1[q] = (unsigned int)m1b;
        c2[q] = __builtin_amdgcn_bitop3_b32((unsigned int)(m0 >> 32), b3[q], k1, 0x96); c3[q] = (unsigned int)m0;
    }
    k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
#pragma unroll
    for (int r = 3; r < 10; ++r) {
        unsigned long long p0[A], p1[A];
#pragma unroll
        for (int q = 0; q < A; ++q) {
            p0[q] = (unsigned long long)ROCRAND_PHILOX_M4x32_0 * c0[q];
            p1[q] = (unsigned long long)ROCRAND_PHILOX_M4x32_1 * c2[q];
        }
#pragma unroll
        for (int q = 0; q < A; ++q) {
            const unsigned int n0 = __builtin_amdgcn_bitop3_b32((unsigned int)(p1[q] >> 32), c1[q], k0, 0x96);
            const unsigned int n2 = __builtin_amdgcn_bitop3_b32((unsigned int)(p0[q] >> 32), c3[q], k1, 0x96);
            c0[q] = n0; c1[q] = (unsigned int)p1[q]; c2[q] = n2; c3[q] = (unsigned int)p0[q];
        }
        k0 += ROCRAND_PHILOX_W32_0; k1 += ROCRAND_PHILOX_W32_1;
#if !defined(MPPI_PHILOX_FREE_ORDER)
        // Keep the rounds round-major: ONE empty asm statement that takes and returns the multiplicands of every block's next round.
        // (A __builtin_amdgcn_sched_barrier does not do it: pure arithmetic is not ordered against it when the block's DAG is
        // linearised, and hipcc emitted block 0's whole chain between the barriers and the other blocks behind the last one.)
        if constexpr (A == 2) asm volatile("" : "+v"(c0[0]), "+v"(c2[0]), "+v"(c0[1]), "+v"(c2[1]));
        if constexpr (A == 3) asm volatile("" : "+v"(c0[0]), "+v"(c2[0]), "+v"(c0[1]), "+v"(c2[1]), "+v"(c0[2]), "+v"(c2[2]));
        if constexpr (A == 4) asm volatile("" : "+v"(c0[0]), "+v"(c2[0]), "+v"(c0[1]), "+v"(c2[1]), "+v"(c0[2]), "+v"(c2[2]), "+v"(c0[3]), "+v"(c2[3]));
#endif
    }
#pragma unroll
    for (int q = 0; q < A; ++q) out[q] = uint4{c0[q], c1[q], c2[q], c3[q]};
}

// The 4 standard normals of ONE Philox block of a sample (block uniform or not): the single place every rollout kernel that
// draws block by block (k_rollout_mlp2, k_rollout_mlp32, k_rollout_nnauv32) gets them from, so that the rollout and the tile
// record (mlp_tile_record -> normals_group) always agree — also in the -DMPPI_ROCRAND_NORMALS variant build.
__device__ __forceinline__ float4 normals_of_block(unsigned long long seed, unsigned long long gk, unsigned long long block)
{
#if defined(MPPI_ROCRAND_NORMALS)
    PhiloxAt eng(seed, gk, 4ull * block);
    return rocrand_device::detail::normal_distribution4(eng.block());
#else
    const uint4 r = philox4x32_10_block(seed, gk, block);
    const float2 n0 = box_muller_hw(r.x, r.y), n1 = box_muller_hw(r.z, r.w);
    return float4{n0.x, n0.y, n1.x, n1.y};
#endif
}

// (A/B timing only, tools/ablate.py philox_block_major: r03's order — one block's ten rounds after the other)
template <int A>
__device__ __forceinline__ void normals_group_ub_block_major(unsigned long long seed, unsigned int gk_lo, unsigned long long group_index_uniform, float (&z)[4 * A])
{
#pragma unroll
    for (int q = 0; q < A; ++q) {
        const uint4 r = philox4x32_10_block_ub(seed, gk_lo, group_index_uniform * A + q);
        const float2 n0 = box_muller_hw(r.x, r.y), n1 = box_muller_hw(r.z, r.w);
        z[4 * q + 0] = n0.x; z[4 * q + 1] = n0.y; z[4 * q + 2] = n1.x; z[4 * q + 3] = n1.y;
    }
}

// normals_group for a wave-uniform group index and a sample index below 2^32 (the rollout kernels' case)
template <int A>
__device__ __forceinline__ void normals_group_ub(unsigned long long seed, unsigned int gk_lo, unsigned long long group_index_uniform,
                                                 float (&z)[4 * A])
{
#if defined(MPPI_ROCRAND_NORMALS)
#pragma unroll
    for (int q = 0; q < A; ++q) {
        PhiloxAt eng(seed, (unsigned long long)gk_lo, 4ull * (group_index_uniform * A + q));
        const float4 n = rocrand_device::detail::normal_distribution4(eng.block());
        z[4 * q + 0] = n.x; z[4 * q + 1] = n.y; z[4 * q + 2] = n.z; z[4 * q + 3] = n.w;
    }
#else
    uint4 r[A];
    philox4x32_10_blocks_ub<A>(seed, gk_lo, group_index_uniform * A, r);
#pragma unroll
    for (int q = 0; q < A; ++q) {
        const float2 n0 = box_muller_hw(r[q].x, r[q].y), n1 = box_muller_hw(r[q].z, r[q].w);
        z[4 * q + 0] = n0.x; z[4 * q + 1] = n0.y; z[4 * q + 2] = n1.x; z[4 * q + 3] = n1.y;
    }
#endif
}

template <int A>
__device__ __forceinline__ void normals_group(unsigned long long seed, unsigned long long gk,
                                              unsigned long long group_index, float (&z)[4 * A])
{
#pragma unroll
    for (int q = 0; q < A; ++q) {
#if defined(MPPI_ROCRAND_NORMALS)
        PhiloxAt eng(seed, gk, 4ull * (group_index * A + q));
        const float4 n = rocrand_device::detail::normal_distribution4(eng.block());
        z[4 * q + 0] = n.x; z[4 * q + 1] = n.y; z[4 * q + 2] = n.z; z[4 * q + 3] = n.w;
#else
        const uint4 r = philox4x32_10_block(seed, gk, group_index * A + q);
        const float2 n0 = box_muller_hw(r.x, r.y), n1 = box_muller_hw(r.z, r.w);
        z[4 * q + 0] = n0.x; z[4 * q + 1] = n0.y; z[4 * q + 2] = n1.x; z[4 * q + 3] = n1.y;
#endif
    }
}

// eps = Σ · z   (controller_base.cpp:201 BatchMatMulV2(sigma, rng): Σ multiplies z directly).
// DIAG: Σ (and Σ⁻¹) are diagonal — the reference's default Σ = c·I. The dense row sum then adds
// exact zeros (0·z_j) to ONE non-zero product, so evaluating only that product is bit-identical
// (up to the sign of a zero) and saves 2(A²-A) operations per (k,t).
template <int A, bool DIAG = false, class CT = DevConsts>
__device__ __forceinline__ void scale_noise(const CT *__restrict__ C, const float (&z)[A], float (&e)[A])
{
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (DIAG) {
            e[i] = C->sigma[i * kMaxA + i] * z[i];
        } else {
            float acc = C->sigma[i * kMaxA] * z[0]; // 0 + x is x (up to the sign of a zero): the first add is dropped
#pragma unroll
            for (int j = 1; j < A; ++j) acc = acc + C->sigma[i * kMaxA + j] * z[j];
            e[i] = acc;
        }
    }
}

// ----------------------------------------------------------------------------------------
// model_base.cpp:53-82  x' = A x + (B/m) v with A = I⊗[[1,dt],[0,1]], B = I⊗[[dt²/2],[dt]]/m.
// The dense rows reduce to (products with the structural zeros are exact):
//   free_p = p + dt*q ; free_q = q ; act_p = bp*v ; act_q = bq*v ; x' = free + act.
template <int A, class CT = DevConsts>
__device__ __forceinline__ void pm_free_step(const CT *__restrict__ C, const float (&x)[2 * A], float (&fr)[2 * A])
{
#pragma unroll
    for (int j = 0; j < A; ++j) {
        fr[2 * j] = x[2 * j] + C->dt * x[2 * j + 1];
        fr[2 * j + 1] = x[2 * j + 1];
    }
}

template <int A, class CT = DevConsts>
__device__ __forceinline__ void pm_action_step(const CT *__restrict__ C, const float (&v)[A], float (&ac)[2 * A])
{
#pragma unroll
    for (int j = 0; j < A; ++j) {
        ac[2 * j] = C->bp * v[j];
        ac[2 * j + 1] = C->bq * v[j];
    }
}

// FMA: the opt-in contracted form (MPPI_FLAG_FP_CONTRACT): every multiply that feeds an add becomes one fused multiply-add — fewer
// instructions in an issue-bound kernel, one rounding less per pair; NOT the reference's op-by-op rounding (sample costs then agree
// with an fp64 evaluation to ~1e-6 relative instead of bit for bit with the unfused fp32 one).
template <int A, class CT = DevConsts, bool FMA = false>
__device__ __forceinline__ void pm_step(const CT *__restrict__ C, float (&x)[2 * A], const float (&v)[A])
{
    if constexpr (FMA) {
#pragma unroll
        for (int j = 0; j < A; ++j) {
            const float fr = __builtin_fmaf(C->dt, x[2 * j + 1], x[2 * j]);
            x[2 * j] = __builtin_fmaf(C->bp, v[j], fr);
            x[2 * j + 1] = __builtin_fmaf(C->bq, v[j], x[2 * j + 1]);
        }
    } else {
        float fr[2 * A], ac[2 * A];
        pm_free_step<A>(C, x, fr);
        pm_action_step<A>(C, v, ac);
#pragma unroll
        for (int i = 0; i < 2 * A; ++i) x[i] = fr[i] + ac[i];
    }
}

// ----------------------------------------------------------------------------------------
// The point-mass step and the diagonal quadratic cost in PACKED form (two fp32 per v_pk_* instruction), written out instead of left
// to the SLP vectoriser (which finds some pairs, builds them with v_mov, and changes its mind with the shape of the loop around them).
// Axes go in pairs: P = (p_2i, p_2i+1), Q = (q_2i, q_2i+1); an odd last axis travels as L = (p, q). Per element these are exactly
// pm_step's and state_cost<S, false>'s operations in their order (IEEE add and multiply per lane, no contraction), and the cost's sum runs
// over the state index 0, 1, 2, ... as the reference's diffT·(Q·diff) does: bit-identical to the scalar forms above and below.
typedef float v2f __attribute__((ext_vector_type(2)));
template <int A>
struct PmPack {
    static constexpr int NPR = A / 2;
    static constexpr bool ODD = (A & 1) != 0;
    static constexpr int NA = NPR ? NPR : 1;
    float dt;
    v2f dt2, bp2, bq2, bpq;
    v2f gP[NA], gQ[NA], qP[NA], qQ[NA], gL, qL;
    template <class CT>
    __device__ __forceinline__ void load(const CT *__restrict__ C)
    {
        dt = C->dt;
        dt2 = v2f{C->dt, C->dt}; bp2 = v2f{C->bp, C->bp}; bq2 = v2f{C->bq, C->bq}; bpq = v2f{C->bp, C->bq};
#pragma unroll
        for (int i = 0; i < NPR; ++i) {
            gP[i] = v2f{C->goal[4 * i], C->goal[4 * i + 2]}; gQ[i] = v2f{C->goal[4 * i + 1], C->goal[4 * i + 3]};
            qP[i] = v2f{C->qdiag[4 * i], C->qdiag[4 * i + 2]}; qQ[i] = v2f{C->qdiag[4 * i + 1], C->qdiag[4 * i + 3]};
        }
        if constexpr (ODD) { gL = v2f{C->goal[2 * A - 2], C->goal[2 * A - 1]}; qL = v2f{C->qdiag[2 * A - 2], C->qdiag[2 * A - 1]}; }
    }
};
template <int A>
struct PmState {
    static constexpr int NA = (A / 2) ? (A / 2) : 1;
    v2f P[NA], Q[NA], L;
    __device__ __forceinline__ void from(const float (&x)[2 * A])
    {
#pragma unroll
        for (int i = 0; i < A / 2; ++i) { P[i] = v2f{x[4 * i], x[4 * i + 2]}; Q[i] = v2f{x[4 * i + 1], x[4 * i + 3]}; }
        if constexpr (A & 1) L = v2f{x[2 * A - 2], x[2 * A - 1]};
    }
};
// x <- A x + (B/m) v  (pm_step): fr_p = p + dt q ; x_p = fr_p + bp v ; x_q = q + bq v
template <int A>
__device__ __forceinline__ void pm_step_packed(const PmPack<A> &K, PmState<A> &s, const float (&v)[A])
{
#pragma unroll
    for (int i = 0; i < A / 2; ++i) {
        const v2f V = v2f{v[2 * i], v[2 * i + 1]};
        const v2f acp = K.bp2 * V, acq = K.bq2 * V;
        const v2f m = K.dt2 * s.Q[i];
        const v2f f = s.P[i] + m;
        s.P[i] = f + acp;
        s.Q[i] = s.Q[i] + acq;
    }
    if constexpr (A & 1) {
        const v2f ac = K.bpq * v2f{v[A - 1], v[A - 1]};
        const float m = K.dt * s.L.y;
        const v2f fr = v2f{s.L.x + m, s.L.y};
        s.L = fr + ac;
    }
}
// diffT (Q diff), Q diagonal, summed over the state index in order (state_cost<S, false>)
template <int A>
__device__ __forceinline__ float state_cost_packed(const PmPack<A> &K, const PmState<A> &s)
{
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < A / 2; ++i) {
        const v2f dP = s.P[i] - K.gP[i], dQ = s.Q[i] - K.gQ[i];
        const v2f tP = dP * (K.qP[i] * dP), tQ = dQ * (K.qQ[i] * dQ);
        acc = i == 0 ? tP.x : acc + tP.x; // (the scalar form starts from diff[0]*left[0], not from 0 + it)
        acc = acc + tQ.x;
        acc = acc + tP.y;
        acc = acc + tQ.y;
    }
    if constexpr (A & 1) {
        const v2f dL = s.L - K.gL;
        const v2f tL = dL * (K.qL * dL);
        acc = A == 1 ? tL.x : acc + tL.x;
        acc = acc + tL.y;
    }
    return acc;
}

// cost_base.cpp:56-61 mStateCost: diff = x-g ; left = Q·diff ; cost = diffᵀ·left.
template <int S, bool QFULL, class CT = DevConsts, bool FMA = false>
__device__ __forceinline__ float state_cost(const CT *__restrict__ C, const float (&x)[S])
{
    if constexpr (FMA && !QFULL) { // contracted diagonal form (see pm_step)
        float acc = 0.0f;
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float d = x[i] - C->goal[i];
            const float l = C->qdiag[i] * d;
            acc = i == 0 ? d * l : __builtin_fmaf(d, l, acc);
        }
        return acc;
    }
    float diff[S], left[S];
#pragma unroll
    for (int i = 0; i < S; ++i) diff[i] = x[i] - C->goal[i];
    if constexpr (QFULL) {
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
    float acc = diff[0] * left[0];
#pragma unroll
    for (int i = 1; i < S; ++i) acc = acc + diff[i] * left[i];
    return acc;
}

// costs/elipse_cost.py:48-85 ElipseCost.state_cost, state = (x, vx, y, vy, ...): in the reference's operation order, every
// operation rounded on its own (correctly rounded divide and square root: hipcc's default for fp32):
//   v = sqrt(vx² + vy²) ; dx = (x-cx)/a ; dy = (y-cy)/b ; m_state·|dx² + dy² - 1| + m_vel·(v - speed)²
template <int S, class CT = DevConsts>
__device__ __forceinline__ float state_cost_ellipse(const CT *__restrict__ C, const float (&x)[S])
{
    static_assert(S >= 4, "the elliptic cost reads (x, vx, y, vy)");
    const float vx2 = x[1] * x[1], vy2 = x[3] * x[3];
    const float v = sqrtf(vx2 + vy2); // correctly rounded (not __fsqrt_rn: that is the native ~1 ulp v_sqrt_f32 in this toolchain)
    const float dx = (x[0] - C->ell[2]) / C->ell[0];
    const float dy = (x[2] - C->ell[3]) / C->ell[1];
    const float dx2 = dx * dx, dy2 = dy * dy;
    float d = (dx2 + dy2) - 1.0f;
    d = fabsf(d);
    d = C->ell[5] * d;
    const float dvv = v - C->ell[4];
    float dv = dvv * dvv;
    dv = C->ell[6] * dv;
    return d + dv;
}

// Kernel-local constants of the two other cost_base forms the producer/consumer kernel's consumer serves (COST = 1, 2): the
// elliptic cost's seven numbers; a dense Q, compact [S][S], with the goal. Same arithmetic as state_cost_ellipse / state_cost<S, true>.
struct PcEllipseConsts {
    float ell[7];
    __device__ __forceinline__ void load(const DevConsts *__restrict__ C)
    {
#pragma unroll
        for (int i = 0; i < 7; ++i) ell[i] = C->ell[i];
    }
};
template <int S>
struct PcDenseQConsts {
    float goal[S], q[S * S];
    __device__ __forceinline__ void load(const DevConsts *__restrict__ C)
    {
#pragma unroll
        for (int i = 0; i < S; ++i) {
            goal[i] = C->goal[i];
#pragma unroll
            for (int j = 0; j < S; ++j) q[i * S + j] = C->qfull[i * kMaxS + j];
        }
    }
};
template <int S>
__device__ __forceinline__ float state_cost_dense(const PcDenseQConsts<S> *__restrict__ C, const float (&x)[S])
{
    float diff[S], left[S];
#pragma unroll
    for (int i = 0; i < S; ++i) diff[i] = x[i] - C->goal[i];
#pragma unroll
    for (int i = 0; i < S; ++i) {
        float acc = 0.0f;
#pragma unroll
        for (int j = 0; j < S; ++j) acc = acc + C->q[i * S + j] * diff[j];
        left[i] = acc;
    }
    float acc = diff[0] * left[0];
#pragma unroll
    for (int i = 1; i < S; ++i) acc = acc + diff[i] * left[i];
    return acc;
}

// the cost_base slot of the tile kernel and the helpers: the state cost the controller was built with (wave-uniform branch)
template <int S, bool QFULL>
__device__ __forceinline__ float state_cost_of(const DevConsts *__restrict__ C, const float (&x)[S])
{
    if constexpr (S >= 4) {
        if (C->state_cost_kind == MPPI_STATE_COST_ELLIPSE) return state_cost_ellipse<S>(C, x);
    }
    return state_cost<S, QFULL>(C, x);
}

// cost_base.cpp:63-68 (C++: λ·uᵀ(Σ⁻¹ε), u = NOMINAL action) or cost_base.py:114-170 (γ/υ form).
// KIND: the action-cost form when the caller has already branched on it (MPPI_ACTION_COST_CPP / _PY), -1 = read C->action_cost_kind here.
// k_rollout_pc's producers branch ONCE around their whole horizon loop: the per-step test it replaces split every horizon group of the
// unrolled loop into basic blocks (two scalar branches per step) that the scheduler could not move the Philox rounds across.
template <int A, bool DIAG = false, class CT = DevConsts, bool FMA = false, int KIND = -1>
__device__ __forceinline__ float action_cost(const CT *__restrict__ C, const float (&u)[A], const float (&e)[A])
{
    const bool cpp_form = KIND >= 0 ? KIND == MPPI_ACTION_COST_CPP : C->action_cost_kind == MPPI_ACTION_COST_CPP;
    if constexpr (FMA && DIAG) { // contracted form of the diagonal-Sigma case (see pm_step); both cost kinds
        float mix = u[0] * (C->sigma_inv[0] * e[0]);
#pragma unroll
        for (int i = 1; i < A; ++i) mix = __builtin_fmaf(u[i], C->sigma_inv[i * kMaxA + i] * e[i], mix);
        if (cpp_form) return C->lambda * mix;
        float n = e[0] * (C->sigma_inv[0] * e[0]), ac = u[0] * (C->sigma_inv[0] * u[0]);
#pragma unroll
        for (int i = 1; i < A; ++i) {
            n = __builtin_fmaf(e[i], C->sigma_inv[i * kMaxA + i] * e[i], n);
            ac = __builtin_fmaf(u[i], C->sigma_inv[i * kMaxA + i] * u[i], ac);
        }
        const float control = __builtin_fmaf(C->gamma, ac, C->gamma * (2.0f * mix));
        return 0.5f * __builtin_fmaf(C->py_ncoef, n, control);
    }
    float rhsN[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (DIAG) { // see scale_noise: the off-diagonal terms are exact zeros
            rhsN[i] = C->sigma_inv[i * kMaxA + i] * e[i];
        } else {
            float acc = C->sigma_inv[i * kMaxA] * e[0];
#pragma unroll
            for (int j = 1; j < A; ++j) acc = acc + C->sigma_inv[i * kMaxA + j] * e[j];
            rhsN[i] = acc;
        }
    }
    float mix = u[0] * rhsN[0];
#pragma unroll
    for (int i = 1; i < A; ++i) mix = mix + u[i] * rhsN[i];
    if (cpp_form) return C->lambda * mix;

    float rhsA[A];
#pragma unroll
    for (int i = 0; i < A; ++i) {
        if (DIAG) {
            rhsA[i] = C->sigma_inv[i * kMaxA + i] * u[i];
        } else {
            float acc = C->sigma_inv[i * kMaxA] * u[0];
#pragma unroll
            for (int j = 1; j < A; ++j) acc = acc + C->sigma_inv[i * kMaxA + j] * u[j];
            rhsA[i] = acc;
        }
    }
    mix = 2.0f * mix;
    float n = e[0] * rhsN[0], ac = u[0] * rhsA[0];
#pragma unroll
    for (int i = 1; i < A; ++i) n = n + e[i] * rhsN[i];
#pragma unroll
    for (int i = 1; i < A; ++i) ac = ac + u[i] * rhsA[i];
    ac = C->gamma * ac;
    mix = C->gamma * mix;
    n = C->py_ncoef * n;
    const float control = ac + mix;
    return 0.5f * (control + n);
}

// ----------------------------------------------------------------------------------------
// wavefront (64-lane) butterflies: fixed order, deterministic.
__device__ __forceinline__ float wave_min_bpermute(float v) // the LDS-crossbar form (ds_bpermute_b32: 24 issue cycles + its latency per level)
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
__device__ __forceinline__ float wave_sum_bpermute(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d_bpermute(double v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = v + __shfl_xor(v, off, 64);
    return v;
}

// DPP butterflies (no LDS crossbar): quad swaps, half-row and row mirrors, then the four row
// sums through readlane. Fixed association -> deterministic; every lane returns the total.
template <int CTRL>
__device__ __forceinline__ float dpp_mov(float v)
{
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v)
{
    v = v + dpp_mov<0xB1>(v);  // quad_perm [1,0,3,2]
    v = v + dpp_mov<0x4E>(v);  // quad_perm [2,3,0,1]
    v = v + dpp_mov<0x141>(v); // row_half_mirror
    v = v + dpp_mov<0x140>(v); // row_mirror
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return (s0 + s1) + (s2 + s3);
}

// min over the 64 lanes, every lane gets it: the same DPP ladder (min is exact in any order)
__device__ __forceinline__ float wave_min_dpp(float v)
{
    v = fminf(v, dpp_mov<0xB1>(v));
    v = fminf(v, dpp_mov<0x4E>(v));
    v = fminf(v, dpp_mov<0x141>(v));
    v = fminf(v, dpp_mov<0x140>(v));
    const float s0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 0));
    const float s1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 16));
    const float s2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 32));
    const float s3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), 48));
    return fminf(fminf(s0, s1), fminf(s2, s3));
}
// 64-lane sum of doubles on the same DPP ladder (two dword moves per level), every lane gets the total: the finish kernels'
// eta and V sums (24 ds_bpermute round trips per workgroup before)
template <int CTRL>
__device__ __forceinline__ double dpp_mov_d(double v)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xF, 0xF, true), hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xF, 0xF, true);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
__device__ __forceinline__ double readlane_d(double v, int lane)
{
    const long long b = __builtin_bit_cast(long long, v);
    const int lo = __builtin_amdgcn_readlane((int)b, lane), hi = __builtin_amdgcn_readlane((int)(b >> 32), lane);
    return __builtin_bit_cast(double, ((long long)hi << 32) | (long long)(unsigned int)lo);
}
__device__ __forceinline__ double wave_sum_d(double v)
{
    v = v + dpp_mov_d<0xB1>(v);
    v = v + dpp_mov_d<0x4E>(v);
    v = v + dpp_mov_d<0x141>(v);
    v = v + dpp_mov_d<0x140>(v);
    return (readlane_d(v, 0) + readlane_d(v, 16)) + (readlane_d(v, 32) + readlane_d(v, 48));
}

// The tile soft-min of every rollout kernel (beta_b = min, eta_b = sum over a tile's 64 lanes) runs on these: no LDS
// crossbar on the critical tail of a tile (12 dependent ds_bpermute round trips before, ~0.5 us per tile). One fixed
// association for every kernel, so kernels that must agree on a record (k_rollout_pc / k_rollout_tile) still do.
__device__ __forceinline__ float wave_min(float v) { return wave_min_dpp(v); }
__device__ __forceinline__ float wave_sum(float v) { return wave_sum_dpp(v); }

// ----------------------------------------------------------------------------------------
// Transposing butterfly: sums N per-lane values ACROSS the 64 lanes for N independent columns at once. Each level
// pairs lanes (l, l^p) and halves the live registers: of a register pair (lo, hi) the lane whose role bit is 0 ends up
// with lo_self + lo_partner, the other with hi_self + hi_partner. After the 6 levels lane l holds, in out[m], the
// 64-lane total of column   n = 64*m + lane_column(l).   Fixed association -> deterministic.
//
// Instruction choice, from tools/micro/valu_issue.hip on this part (cycles per wave-instruction per SIMD, several
// waves resident): v_add/v_mul 2.3, v_fma 2.5, DPP forms 4.2, v_cndmask_b32 with the mask in an SGPR pair 4.2 but with
// the mask in VCC (what hipcc picks for `role ? a : b`) 19.4, ds_bpermute_b32 (= __shfl_xor) 24. The first version of
// this butterfly (quad swaps and row mirrors + two selects per pair, __shfl_xor for the strides 16 and 32) cost
// ~3100 cycles per producer wave at N = 72, a quarter of the rollout kernel; this one ~550:
//   level 0, stride 32: v_permlane32_swap (lanes 32-63 of lo <-> lanes 0-31 of hi), then lo + hi      role = lane bit 5
//   level 1, stride 16: v_permlane16_swap (odd rows of lo <-> even rows of hi), then lo + hi            role = bit 4
//   level 2, stride 8 : two v_add_f32_dpp row_ror:8 writing complementary banks (bank = 4 lanes)        role = bit 3
//   level 3, stride 4 : two v_add_f32_dpp, row_ror:12 on banks {0,2} / row_ror:4 on banks {1,3}         role = bit 2
//   level 4, stride 2 : lo + quad_perm[2,3,0,1](lo), hi + ..(hi), bitwise select by a lane-mask register role = bit 1
//   level 5, stride 1 : the same with quad_perm[1,0,3,2]                                                 role = bit 0
// No selects on a condition register anywhere; the big strides come first, while there are many registers.
// The swaps and the bank-masked adds are inline asm: hipcc mis-compiles `r[0] + r[1]` of __builtin_amdgcn_permlane32_swap
// (ROCm 7.2: emits v_add_f32 v1, v1, v1) and has no builtin for a DPP add that leaves the other banks of its
// destination alone. Each statement opens with the wait states its DPP / permlane reads need after a VALU write
// (2; hipcc does not look inside an asm string).
__device__ __forceinline__ int lane_column(int lane)
{
    return ((lane >> 5) & 1) | (((lane >> 4) & 1) << 1) | (((lane >> 3) & 1) << 2) | (((lane >> 2) & 1) << 3) |
           (((lane >> 1) & 1) << 4) | ((lane & 1) << 5);
}

// lane-mask registers for the bitwise selects of levels 4 and 5: all ones where the role bit is set. Made opaque so that
// hipcc keeps the and/or form (v_bfi_b32) instead of rebuilding a v_cndmask on VCC.
struct ButterflyMasks {
    unsigned m1, m0;
    __device__ __forceinline__ explicit ButterflyMasks(int lane)
    {
        m1 = (lane & 2) ? 0xffffffffu : 0u;
        m0 = (lane & 1) ? 0xffffffffu : 0u;
        asm volatile("" : "+v"(m1), "+v"(m0));
    }
};

__device__ __forceinline__ float bit_select(unsigned mask, float if_set, float if_clear)
{
    return __builtin_bit_cast(float, (__builtin_bit_cast(unsigned, if_set) & mask) | (__builtin_bit_cast(unsigned, if_clear) & ~mask));
}

template <int LEVEL>
__device__ __forceinline__ float tfold_pair(float lo, float hi, const ButterflyMasks &bm)
{
    if constexpr (LEVEL == 0) {
        asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        return lo + hi;
    } else if constexpr (LEVEL == 1) {
        asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(lo), "+v"(hi));
        return lo + hi;
    } else if constexpr (LEVEL == 2) {
        float out;
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0x3\n\t"
            "v_add_f32_dpp %0, %2, %2 row_ror:8 row_mask:0xf bank_mask:0xc"
            : "=&v"(out) : "v"(lo), "v"(hi));
        return out;
    } else if constexpr (LEVEL == 3) { // row_ror:n = lane i reads lane (i - n) mod 16 of its row
        float out;
        asm("s_nop 1\n\t"
            "v_add_f32_dpp %0, %1, %1 row_ror:12 row_mask:0xf bank_mask:0x5\n\t"
            "v_add_f32_dpp %0, %2, %2 row_ror:4 row_mask:0xf bank_mask:0xa"
            : "=&v"(out) : "v"(lo), "v"(hi));
        return out;
    } else if constexpr (LEVEL == 4) {
        return bit_select(bm.m1, hi + dpp_mov<0x4E>(hi), lo + dpp_mov<0x4E>(lo)); // quad_perm [2,3,0,1]
    } else {
        return bit_select(bm.m0, hi + dpp_mov<0xB1>(hi), lo + dpp_mov<0xB1>(lo)); // quad_perm [1,0,3,2]
    }
}

template <int N, int LEVEL>
__device__ __forceinline__ void tfold(const float (&in)[N], float (&out)[(N + 1) / 2], const ButterflyMasks &bm)
{
#pragma unroll
    for (int j = 0; j < (N + 1) / 2; ++j) out[j] = tfold_pair<LEVEL>(in[2 * j], (2 * j + 1 < N) ? in[2 * j + 1] : 0.0f, bm);
}

// in[N] per lane -> out[ceil(N/64)] per lane, out[m] = total of column 64*m + lane_column(lane)
template <int N>
__device__ __forceinline__ void wave_transpose_sum(const float (&in)[N], float (&out)[(N + 63) / 64], int lane)
{
    constexpr int N1 = (N + 1) / 2, N2 = (N1 + 1) / 2, N3 = (N2 + 1) / 2, N4 = (N3 + 1) / 2, N5 = (N4 + 1) / 2, N6 = (N5 + 1) / 2;
    static_assert(N6 == (N + 63) / 64, "ceil-halving six times equals ceil(N/64)");
    const ButterflyMasks bm(lane);
    float a1[N1], a2[N2], a3[N3], a4[N4], a5[N5];
    tfold<N, 0>(in, a1, bm);
    tfold<N1, 1>(a1, a2, bm);
    tfold<N2, 2>(a2, a3, bm);
    tfold<N3, 3>(a3, a4, bm);
    tfold<N4, 4>(a4, a5, bm);
    tfold<N5, 5>(a5, out, bm);
}

} // namespace mppi
