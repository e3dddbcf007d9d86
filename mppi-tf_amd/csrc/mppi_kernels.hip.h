// mppi_kernels.hip.h — the HIP kernels of the MPPI control step (gfx950). See mppi_device.hip.h
// for the arithmetic and DESIGN.md §3 for the data layout and the roofline of each kernel.
#pragma once

#include "mppi_device.hip.h"
#include "mppi_ablate.hip.h" // timing-study macros: plain code in the shipped build

#include <type_traits>

namespace mppi {

enum NoiseSrc { SRC_PHILOX = 0, SRC_HBM = 1 };
enum TileMode {
    MODE_ROLLOUT = 0,     // phase A + B (costs) + C (tile soft-min record)
    MODE_COSTS_GIVEN = 1, // phase A + C with costs read from `cost` (mUpdate on given costs; normalize pass 2)
    MODE_COST_ONLY = 2,   // phase A + B, no record (mBuildModelGraph alone; normalize pass 1)
    MODE_NOISE_ONLY = 3   // phase A + the noise export only: reads neither x nor cost, writes only noise_out (MPPI_DBG_NOISE)
};

// LDS floats a tile needs: eps[HA][R+1] (padded: lane-per-rollout reads AND lane-per-column reads
// are both conflict-free on the 32-bank ds_read_b32 path), w[R], U[HA], 8 scratch.
__host__ __device__ inline size_t tile_lds_floats(int HA, int R) { return (size_t)HA * (R + 1) + R + HA + 8; }

// ----------------------------------------------------------------------------------------
// k_rollout_tile: rows A3–A9 of SURVEY §8a fused (mPrepareAction/mPrepareNoise slicing,
// model step, step cost, terminal cost, tile-local min/exp/sum and weighted noise).
// grid = ceil(K_local / R) workgroups of 256 threads; dynamic LDS = tile_lds_floats(H*A, R)*4.
// Tiles are independent and share no HBM operand but x[s] and U[H,a] (≤ 1.5 KB), so the
// blockIdx→tile map needs no XCD remap.
template <int A, int R, bool QFULL, int SRC, int MODE>
__global__ __launch_bounds__(kThreads) void k_rollout_tile(
    const DevConsts *__restrict__ C, const float *__restrict__ x_dev, const float *__restrict__ U_dev,
    const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr,
    float *__restrict__ cost, float *__restrict__ partials, float *__restrict__ noise_out, const int rsb, const int rsc)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * A;
    constexpr int RP = R + 1;
    const int H = C->H;
    const int HA = H * A;
    const int K = C->K_local;
    float *eps_s = smem;
    float *w_s = eps_s + (size_t)HA * RP;
    float *U_s = w_s + R;
    float *red_s = U_s + HA;

    const int tid = threadIdx.x;
    const int k0 = blockIdx.x * R;

    for (int i = tid; i < HA; i += kThreads) U_s[i] = U_dev[i];

    // ---- phase A: the tile's noise -> LDS, eps_s[(t*A+j)*RP + kl] --------------------------
    if (SRC == SRC_PHILOX) {
        constexpr int TQ = kThreads / R; // groups (of 4 horizon steps) generated concurrently per rollout
        const int kl = tid % R;
        const int NG = (H + 3) / 4;
        const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)(k0 + kl);
        const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
        const unsigned long long seed = C->seed;
        for (int g = tid / R; g < NG; g += TQ) {
            float z[4 * A];
            MPPI_NORMALS_GROUP(A, seed, gk, base + (unsigned long long)g, z);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                const int t = 4 * g + tl;
                if (t < H) {
                    float zz[A], e[A];
#pragma unroll
                    for (int j = 0; j < A; ++j) zz[j] = z[tl * A + j];
                    scale_noise<A>(C, zz, e);
#pragma unroll
                    for (int j = 0; j < A; ++j) eps_s[(t * A + j) * RP + kl] = e[j];
                }
            }
        }
    } else {
        // injected noise: coalesced dword read of the tile's contiguous [R, H*A] slab
        const size_t slab = (size_t)k0 * HA;
        const int n = R * HA;
        for (int i = tid; i < n; i += kThreads) {
            const int kl = i / HA, c = i - kl * HA;
            eps_s[c * RP + kl] = (k0 + kl < K) ? eps_hbm[slab + i] : 0.0f;
        }
    }
    __syncthreads();

    if (noise_out != nullptr) { // debug export (MPPI_DBG_NOISE): the noise this step used
        const size_t slab = (size_t)k0 * HA;
        const int n = R * HA;
        for (int i = tid; i < n; i += kThreads) {
            const int kl = i / HA, c = i - kl * HA;
            if (k0 + kl < K) noise_out[slab + i] = eps_s[c * RP + kl];
        }
    }
    if (MODE == MODE_NOISE_ONLY) return;

    // ---- phase B: one lane per rollout, wave 0 ---------------------------------------------
    if (tid < 64) {
        const bool valid = (tid < R) && (k0 + tid < K);
        const int kl = tid < R ? tid : R - 1;
        float c = 0.0f;
        if (MODE != MODE_COSTS_GIVEN) {
            float x[S];
#pragma unroll
            for (int i = 0; i < S; ++i) x[i] = x_dev[i];
            for (int t = 0; t < MPPI_ABL_STEPS(H); ++t) {
                float u[A], e[A], v[A];
#pragma unroll
                for (int j = 0; j < A; ++j) {
                    u[j] = U_s[t * A + j];              // mPrepareAction  controller_base.cpp:205-208
                    e[j] = eps_s[(t * A + j) * RP + kl]; // mPrepareNoise   controller_base.cpp:210-213
                    v[j] = u[j] + e[j];                  // to_apply        controller_base.cpp:258
                }
                pm_step<A>(C, x, v);
                const float sc = state_cost_of<S, QFULL>(C, x); // cost on the POST-step state (quadratic or elliptic)
                const float ac = action_cost<A>(C, u, e);
                const float tmp = sc + ac;                    // Step_cost_result cost_base.cpp:49
                c = c + tmp;                                  // path_cost        controller_base.cpp:268
            }
            c = c + state_cost_of<S, QFULL>(C, x); // terminal: x_H counted a second time, :271-272
            if (valid) cost[k0 + tid] = c;
        } else {
            c = valid ? cost[k0 + tid] : 0.0f;
        }
        if (MODE != MODE_COST_ONLY) {
            // tile-local mBeta / mExpArg / mExp / mNabla (controller_base.cpp:166-182)
            const float beta = wave_min(valid ? c : INFINITY);
            const float arg = C->neg_inv_lambda * (c - beta);
            const float ek = valid ? expf(arg) : 0.0f;
            const float eta = wave_sum(ek);
            if (tid < R) w_s[tid] = ek;
            if (tid == 0) { red_s[0] = beta; red_s[1] = eta; }
        }
    }
    if (MODE == MODE_COST_ONLY) return;
    __syncthreads();

    // ---- phase C: V_b[c] = Σ_k e_k·eps[k,c] in fixed k order (mWeightedNoise, :188-192) -----
    // record element (b, col) lives at partials[b*rsb + col*rsc]: column-major (rsb = 1) for the finish kernel
    float *rec = partials + (size_t)record_slot(blockIdx.x, rsc) * rsb;
    if (tid == 0) { rec[0] = red_s[0]; rec[(size_t)rsc] = red_s[1]; }
    for (int c = tid; c < MPPI_ABL_WSUM_COLS(HA); c += kThreads) {
        const float *row = eps_s + (size_t)c * RP;
        float acc = 0.0f;
#pragma unroll 8
        for (int kl = 0; kl < R; ++kl) acc = acc + w_s[kl] * row[kl];
        rec[(size_t)(2 + c) * rsc] = acc;
    }
}

// ----------------------------------------------------------------------------------------
// k_rollout_pc: the hot configuration (Philox noise, rollout + tile record) as a
// producer/consumer pipeline inside one workgroup of 4 wavefronts that owns 64 rollouts:
//   waves 1..3 (producers): lane = rollout. Producer p draws the noise of horizon groups
//       g = 3i+p (4 steps per group, A Philox blocks), keeps eps in REGISTERS for the weighted
//       sum, and hands the consumer per step the perturbed action v = u_t+eps and the action
//       cost (both depend on nothing but (u_t, eps)) through a double-buffered LDS chunk;
//   wave 0 (consumer): lane = rollout, runs the sequential H-step recurrence + state cost out
//       of the LDS chunks (12 steps per chunk, one s_barrier per chunk), then the tile soft-min;
//   producers finish with V_b[t,j] = Σ_k e_k·eps[k,t,j] straight from registers (DPP butterflies).
// Why: with the noise parked in LDS for the weighted sum a tile costs 50 KB, so only 3 tiles fit
// a CU while K=65536 gives every CU exactly 4; and the recurrence ran on 1 of 4 waves while the
// other 3 idled at the barrier (22 of 45 µs). Here LDS is 26 KB/tile, registers ~110/lane ->
// 4 workgroups (16 waves) per CU in one round, all waves busy.
// Arithmetic per (k,t) is the same op sequence as k_rollout_tile: sample costs stay bit-identical.
// NSLOT = max horizon groups per producer = ceil(ceil(H/4)/3)  (6 for H<=72, 11 for H<=132).
// compile-time loop: the slot index must be a constant so that eps_r[slot] stays in registers
// (a rolled loop would index the array dynamically and push it to scratch).
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f)
{
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// Wave priority by progress (s_setprio): the SIMD arbiter favours the oldest wave, so the workgroups sharing a CU
// finish one after the other and the last one runs alone, latency-bound. Lowering a wave's priority as it advances
// lets the ones behind catch up, so that all of them finish together. Only when the whole grid is resident at once
// (`balance`, at most 4 workgroups per CU): with several rounds of workgroups the age order staggers their phases,
// which is what keeps the SIMDs busy there (K=2^20: 224 us without, 244 us with priorities).
// virtual progress = chunk index + a head start of 3/4 chunk per generation of age (the arbiter still breaks ties
// by age inside one level): level = 3 - floor(4*(i + 3(3-gen)/4)/n), gen = blockIdx/256 = arrival order on the CU.
// Measured at K=65536, H=64 (tools/timeline.py): workgroup end times 10.5 .. 19.4 us after the first start without,
// 13.2 .. 15.3 us with; kernel 21.9 -> 19.2 us together with the SIMD-true role placement below.
// r05: the head start of a generation (quarter chunks) arrives with the launch — `balance` = 1 | bias(gen 0) << 8 | bias(gen 1) << 12 | bias(gen 2) << 16 |
// bias(gen 3) << 20, default 9, 6, 3, 0 = r04's 3 (3 - gen) — so that it can be tuned per handle (MPPI_TUNE_PC_BALANCE) without a rebuild.
__device__ __forceinline__ void pc_set_prio(int i, int n, int gen, int balance, int boost = 0)
{
    const int bias = (balance >> (8 + 4 * min(gen, 3))) & 15; // quarter chunks
    const int lvl = min(3, boost + 3 - min(3, (16 * i + 4 * bias) / (4 * n)));
    switch (lvl) {
    case 0: __builtin_amdgcn_s_setprio(0); break;
    case 1: __builtin_amdgcn_s_setprio(1); break;
    case 2: __builtin_amdgcn_s_setprio(2); break;
    default: __builtin_amdgcn_s_setprio(3); break;
    }
}

// NP producers + 1 consumer = NP+1 wavefronts per workgroup; a chunk = one horizon group (4 steps)
// from every producer = 4·NP steps; two chunk buffers of (A+1) floats per (step, lane).
// floats per (step, lane) slot of a chunk: the A perturbed actions + the action cost, padded so that ONE LDS instruction moves
// a slot (ds_write_b64 / _b128 by the producer, ds_read_b64 / _b128 by the consumer: a quarter of the LDS instructions and
// waits of the dword layout [step][value][lane]); a_dim = 4 (5 values) keeps the dword layout
__host__ __device__ constexpr int pc_slot_floats(int A) { return A == 1 ? 2 : (A == 2 || A == 3) ? 4 : A + 1; }
__host__ __device__ inline size_t pc_lds_floats(int A, int NP) { return (size_t)2 * 4 * NP * pc_slot_floats(A) * 64; }

// COST: the cost_base form of the consumer — 0 quadratic with a diagonal Q (the BASELINE configurations), 1 ElipseCost (s >= 4),
// 2 quadratic with a dense Q. A template parameter, so the instances of the hot configuration are what they were.
// 3: the diagonal quadratic cost with CONTRACTED arithmetic (MPPI_FLAG_FP_CONTRACT: fused multiply-adds in the model step, the state cost and the
// action cost — not bit-identical to the reference's op-by-op rounding; tests/test_parity_gpu.py holds it to an fp64 evaluation instead).
enum { PC_COST_DIAG = 0, PC_COST_ELLIPSE = 1, PC_COST_DENSE = 2, PC_COST_DIAG_FMA = 3 };
// PASS (r04): the two passes of the Python reference's normalizeCost=True (controller_base.py:468-474: c' = (c - min c)/(max c - min c) before the
// soft-min) on this kernel. The normalised update is the plain one at the temperature lambda (max - min) (exp(-(c' - min c')/lambda) =
// exp(-(c - min c)/(lambda (max - min)))), which needs every cost before any weight:
//   PC_PASS_PLAIN (0)   the step's one pass (every BASELINE configuration: these instances are what they were);
//   PC_PASS_COSTS (1)   pass 1: the plain pass, and the consumer also leaves its tile's (min, max) cost in tile_mm[b], tile_mm[rsc + b];
//   PC_PASS_WEIGHTS (2) pass 2: NO rollouts — every workgroup reduces the tile pairs to the global range (the same values in the same order
//                       everywhere: the same bits), the consumer takes its costs from `cost`, the producers regenerate the noise (same Philox
//                       counters) for the weighted sums; records at the range's temperature. tile_mm == NULL: the temperature is already in
//                       mm_out[2] (a K-sharded handle, whose range the ranks agreed on). Replaces a full second rollout pass and the
//                       single-workgroup k_cost_minmax launch between the two (42 -> 33 us per normalised step at configs[2]'s shape).
enum { PC_PASS_PLAIN = 0, PC_PASS_COSTS = 1, PC_PASS_WEIGHTS = 2 };
template <int A, int NP, int NSLOT, bool DIAG, int COST = PC_COST_DIAG, int PASS = PC_PASS_PLAIN>
__global__ __launch_bounds__(64 * (NP + 1), (NSLOT * 4 * A <= 80 ? NP + 1 : 2)) void k_rollout_pc(
    const DevConsts *__restrict__ C, const float *__restrict__ x_dev, const float *__restrict__ U_dev,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int rsb, const int rsc, const int balance, float *__restrict__ tile_mm, float *__restrict__ mm_out)
{
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * A;
    constexpr int NW = NP + 1;
    constexpr int CS = 4 * NP;                // steps per chunk
    constexpr int SLOT = pc_slot_floats(A);   // floats per (step, lane)
    constexpr bool PACKED = SLOT != A + 1 || A == 3; // [CS][64 lanes][SLOT], one LDS instruction per slot; else [CS][(A+1)][64 lanes] dwords
    constexpr int CH = CS * SLOT * 64;        // floats per chunk buffer
    typedef float slot_t __attribute__((ext_vector_type(SLOT == 2 ? 2 : 4)));
    constexpr int NREG = NSLOT * 4 * A;       // noise values a producer lane keeps
    constexpr bool FMA = COST == PC_COST_DIAG_FMA;
    const int H = C->H;
    const int K = C->K_local;
    const int NG = (H + 3) / 4;               // horizon groups
    const int nch = (NG + NP - 1) / NP;       // chunks
    float *buf = smem;                        // [2][CS][(A+1)][64]
    float *w_s = smem;                        // [64] weights: reuses buffer 0 once every chunk is consumed
    MPPI_TL_DECL();

    const int tid = threadIdx.x;
    // Role placement. A workgroup's waves are spread over the CU's 4 SIMDs and the 4 workgroups that share a CU
    // (observed dispatch: block b -> XCD b%8, then a CU of it; blocks b, b+256, b+512, b+768 meet on one CU) should
    // put their light consumer wave on 4 DIFFERENT SIMDs, so that every SIMD runs 1 consumer + NP producers. The
    // hardware rotates the SIMD order of successive workgroups itself (measured with HW_ID: consumers chosen by wave
    // index landed 2+2+0+0), so with one wave per SIMD the role comes from the SIMD id the wave actually runs on:
    // consumer = the wave on SIMD gen%4 (gen = b/256). Falls back to the wave index when the 4 waves are not on 4
    // distinct SIMDs, and for grids of several rounds (no fixed set of co-resident workgroups there).
    // Placement only affects speed, never results.
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int gen = (int)(blockIdx.x >> 8);
    int wave = (wave_hw + NW - gen % NW) % NW; // SGPR: scalar branches, scalar loads of U
    if (NW == 4 && balance) {
        __shared__ int simd_s[4];
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_REG_HW_ID[5:4]
        if ((tid & 63) == 0) simd_s[wave_hw] = simd;
        __syncthreads();
        const int s0 = simd_s[0], s1 = simd_s[1], s2 = simd_s[2], s3 = simd_s[3];
        if (((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15) wave = (simd + 4 - (gen & 3)) & 3;
        wave = __builtin_amdgcn_readfirstlane(wave);
    }
    const int lane = tid & 63;
    const int k0 = blockIdx.x * 64;
    const bool valid = (k0 + lane) < K;
    float *rec = partials + (size_t)record_slot(blockIdx.x, rsc) * rsb; // element (b, col) at partials[b*rsb + col*rsc]
    MPPI_TL_WHERE(wave);
    float nil_range = 0.0f; // PC_PASS_WEIGHTS: -1/(lambda (max - min)) over ALL tiles
    if constexpr (PASS == PC_PASS_WEIGHTS) {
        if (tile_mm != nullptr) {
            __shared__ float mn_s[NW], mx_s[NW];
            float mn = INFINITY, mx = -INFINITY;
            for (int i = tid; i < (int)gridDim.x; i += 64 * NW) { mn = fminf(mn, tile_mm[i]); mx = fmaxf(mx, tile_mm[rsc + i]); }
            mn = wave_min(mn); mx = wave_max(mx);
            if (lane == 0) { mn_s[wave_hw] = mn; mx_s[wave_hw] = mx; }
            __syncthreads();
            mn = mn_s[0]; mx = mx_s[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) { mn = fminf(mn, mn_s[w]); mx = fmaxf(mx, mx_s[w]); }
            nil_range = C->neg_inv_lambda / (mx - mn); // k_cost_minmax's expression
            if (blockIdx.x == 0 && tid == 0) { mm_out[0] = mn; mm_out[1] = mx - mn; mm_out[2] = nil_range; } // what the finish and the weights' export read
        } else {
            nil_range = mm_out[2];
        }
    }

    if (wave != 0) {
        // ------------------------------------------------------------------ producers
        const int p = wave - 1;
        const unsigned int gk = (unsigned int)C->k_offset + (unsigned int)(k0 + lane); // the global sample index: below 2^31 + 64
        const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
        const unsigned long long seed = C->seed;
        float eps_r[NREG];
        PcProducerConsts<A> pcst; // SGPR-resident copy: no constant re-fetch after the barriers
        pcst.template load<DIAG>(C);
        const PcProducerConsts<A> *PC = &pcst;
        MPPI_STAMP(16 + 10 * p + 9);
        auto produce = [&](auto ic, auto kindc) { // kindc: the action-cost form, resolved once below
            constexpr int i = decltype(ic)::value;
            constexpr int KIND = decltype(kindc)::value;
            const int g = NP * i + p;
            if (balance) pc_set_prio(i, nch, gen, balance);
            if (PASS == PC_PASS_WEIGHTS && i < nch && g < NG) { // the noise alone: nothing is published, no chunk barrier
                float z[4 * A];
                MPPI_NORMALS_GROUP_UB(A, seed, gk, base + (unsigned long long)g, z);
#pragma unroll
                for (int tl = 0; tl < 4; ++tl) {
                    float zz[A], e[A];
#pragma unroll
                    for (int j = 0; j < A; ++j) zz[j] = z[tl * A + j];
                    scale_noise<A, DIAG>(PC, zz, e);
#pragma unroll
                    for (int j = 0; j < A; ++j) eps_r[(i * 4 + tl) * A + j] = e[j];
                }
            }
            if (PASS != PC_PASS_WEIGHTS && i < nch) { // chunk i exists (wave-uniform)
                float *cb = buf + (i & 1) * CH + (size_t)(4 * p) * SLOT * 64;
                if (g < NG) {
                    // the nominal actions of the group's 4 steps: scalar loads issued ahead of the Philox rounds that
                    // hide them (mPrepareAction controller_base.cpp:205-208); steps past the horizon are never consumed
                    float ug[4][A];
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        const int tt = min(4 * g + tl, H - 1);
#pragma unroll
                        for (int j = 0; j < A; ++j) ug[tl][j] = U_dev[tt * A + j];
                    }
                    float z[4 * A];
                    MPPI_NORMALS_GROUP_UB(A, seed, gk, base + (unsigned long long)g, z);
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        const int t = 4 * g + tl;
                        float zz[A], e[A], u[A];
#pragma unroll
                        for (int j = 0; j < A; ++j) zz[j] = z[tl * A + j];
                        scale_noise<A, DIAG>(PC, zz, e);
                        // steps past the horizon (ragged last group) need no mask: the butterfly sums every column on its own and the
                        // columns with t >= H are never stored (r03: the e * keep multiply that used to zero them cost 0.3 us at C3)
                        float slot[SLOT];
#pragma unroll
                        for (int j = 0; j < SLOT; ++j) slot[j] = 0.0f;
#pragma unroll
                        for (int j = 0; j < A; ++j) {
                            u[j] = ug[tl][j];
                            eps_r[(i * 4 + tl) * A + j] = e[j];
                            slot[j] = u[j] + e[j]; // to_apply, :258
                        }
                        slot[A] = action_cost<A, DIAG, PcProducerConsts<A>, FMA, KIND>(PC, u, e);
                        if constexpr (PACKED) {
                            slot_t sv;
#pragma unroll
                            for (int j = 0; j < SLOT; ++j) sv[j] = slot[j];
                            *static_cast<slot_t *>(__builtin_assume_aligned(cb + (tl * 64 + lane) * SLOT, SLOT * 4)) = sv;
                        } else {
#pragma unroll
                            for (int j = 0; j <= A; ++j) cb[(tl * (A + 1) + j) * 64 + lane] = slot[j];
                        }
                    }
                }
                MPPI_STAMP(16 + 10 * p + (i < 7 ? i : 6));
                __syncthreads(); // chunk i published
            }
            if (!(i < nch && g < NG)) {
#pragma unroll
                for (int r = 0; r < 4 * A; ++r) eps_r[i * 4 * A + r] = 0.0f;
            }
        };
        // the action-cost form (C++ reference / Python gamma-upsilon form) is decided once, around the whole horizon loop — in the instances of
        // the step's one pass with the diagonal cost (the BASELINE configurations); the others keep the per-step test (half the code to compile)
        if constexpr (PASS == PC_PASS_PLAIN && (COST == PC_COST_DIAG || COST == PC_COST_DIAG_FMA)) {
            if (MPPI_PC_KIND_ONCE(pcst.action_cost_kind == MPPI_ACTION_COST_CPP)) static_for<0, NSLOT>([&](auto ic) { produce(ic, std::integral_constant<int, MPPI_PC_KIND_OF(MPPI_ACTION_COST_CPP)>{}); });
            else static_for<0, NSLOT>([&](auto ic) { produce(ic, std::integral_constant<int, MPPI_PC_KIND_OF(MPPI_ACTION_COST_PY)>{}); });
        } else {
            static_for<0, NSLOT>([&](auto ic) { produce(ic, std::integral_constant<int, -1>{}); });
        }
        __syncthreads(); // weights published by the consumer
        MPPI_STAMP(16 + 10 * p + 7);
        // phase C from registers: V_b[t,j] = Σ_k e_k·eps[k,t,j]  (mWeightedNoise, controller_base.cpp:188-192)
        const float w = w_s[lane];
#pragma unroll
        for (int r = 0; r < NREG; ++r) eps_r[r] = w * eps_r[r];
        float tot[(NREG + 63) / 64];
        MPPI_WAVE_TRANSPOSE_SUM(NREG, eps_r, tot, lane);
        const int colbase = lane_column(lane);
#pragma unroll
        for (int m = 0; m < (NREG + 63) / 64; ++m) {
            const int n = 64 * m + colbase;        // register index this lane owns the total of
            const int i = n / (4 * A), rem = n - i * (4 * A);
            const int tl = rem / A, j = rem - tl * A;
            const int t = 4 * (NP * i + p) + tl;
            if (n < NREG && t < H) rec[(size_t)(2 + t * A + j) * rsc] = tot[m];
        }
        MPPI_STAMP(16 + 10 * p + 8);
    } else {
        // ------------------------------------------------------------------ consumer
        float x[S];
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = x_dev[i];
        PcConsumerConsts<S> ccst;
        ccst.load(C);
        const PcConsumerConsts<S> *CC = &ccst;
        PcEllipseConsts ecst;
        PcDenseQConsts<S> qcst;
        if constexpr (COST == PC_COST_ELLIPSE) ecst.load(C);
        if constexpr (COST == PC_COST_DENSE) qcst.load(C);
        auto cost_of = [&](const float (&xs)[S]) {
            if constexpr (COST == PC_COST_ELLIPSE) return state_cost_ellipse<S>(&ecst, xs);
            else if constexpr (COST == PC_COST_DENSE) return state_cost_dense<S>(&qcst, xs);
            else return state_cost<S, false, PcConsumerConsts<S>, FMA>(CC, xs);
        };
        float c = 0.0f;
        MPPI_STAMP(0);
        MPPI_STAMP_RT(62);
        if constexpr (PASS == PC_PASS_WEIGHTS) c = cost[valid ? k0 + lane : 0]; // pass 1 left this step's costs there
        if constexpr (PASS != PC_PASS_WEIGHTS) {
        __syncthreads(); // chunk 0 published
        MPPI_STAMP(1);
        for (int ch = 0; ch < nch; ++ch) {
            if (balance) pc_set_prio(ch, nch, gen, balance, MPPI_PC_CONSUMER_BOOST);
            const float *cb = buf + (ch & 1) * CH;
            const int tend = min(CS, H - ch * CS);
            for (int tl = 0; tl < MPPI_ABL_CHUNK_STEPS(tend, ch); ++tl) {
                float v[A], ac;
                if constexpr (PACKED) {
                    const slot_t sv = *static_cast<const slot_t *>(__builtin_assume_aligned(cb + (tl * 64 + lane) * SLOT, SLOT * 4));
#pragma unroll
                    for (int j = 0; j < A; ++j) v[j] = sv[j];
                    ac = sv[A];
                } else {
#pragma unroll
                    for (int j = 0; j < A; ++j) v[j] = cb[(tl * (A + 1) + j) * 64 + lane];
                    ac = cb[(tl * (A + 1) + A) * 64 + lane];
                }
                pm_step<A, PcConsumerConsts<S>, FMA>(CC, x, v);
                const float sc = cost_of(x);                  // cost on the POST-step state
                const float tmp = sc + ac;                    // Step_cost_result cost_base.cpp:49
                c = c + tmp;                                  // path_cost        controller_base.cpp:268
            }
            // barrier budget: producers run nch (one per chunk) + 1 (weights); the consumer 1 + (nch-1) + 1.
            // After the last chunk nothing is published any more: the producers already sit at the weights barrier.
            MPPI_STAMP(2 + (ch < 7 ? ch : 6));
            if (ch + 1 < nch) __syncthreads(); // chunk ch consumed / chunk ch+1 published
        }
        c = c + cost_of(x); // terminal: x_H counted a second time, :271-272
        MPPI_STORE_COST(valid, cost + k0 + lane, c);
        } // PASS != PC_PASS_WEIGHTS
        // tile-local mBeta / mExpArg / mExp / mNabla (controller_base.cpp:166-182)
        const float beta = wave_min(valid ? c : INFINITY);
        if constexpr (PASS == PC_PASS_COSTS) { // the tile's cost range for the second pass
            const float cmax = wave_max(valid ? c : -INFINITY);
            if (lane == 0) { tile_mm[blockIdx.x] = beta; tile_mm[(size_t)rsc + blockIdx.x] = cmax; }
        }
        const float arg = (PASS == PC_PASS_WEIGHTS ? nil_range : CC->neg_inv_lambda) * (c - beta);
        const float ek = valid ? expf(arg) : 0.0f;
        const float eta = wave_sum(ek);
        w_s[lane] = ek;
        if (lane == 0) { rec[0] = beta; rec[(size_t)rsc] = eta; }
        MPPI_STAMP(9);
        MPPI_STAMP_RT(63);
        __syncthreads(); // weights published
        MPPI_TL_DUMP(valid, cost + k0 + lane);
    }
}

// ----------------------------------------------------------------------------------------
// k_combine_group: first level of the record tree. Workgroup j folds records
// [j*kGroup, (j+1)*kGroup) into ONE record with the same (beta, eta, V) algebra as k_finish:
//   beta_j = min beta_b ; r_b = exp(-(beta_b-beta_j)/λ) ; eta_j = Σ r_b eta_b ; V_j = Σ r_b V_b
// in fixed b order. A single-workgroup combine of ~1000 tile records is a chain of ~200
// dependent L2 round trips per thread (measured 76 µs at K=65536); this level runs on
// ceil(nb/kGroup) CUs with kGroup independent loads in flight per thread instead.
// neutral records in every slot (see record_slot): launched at create and when a handle changes its record count
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ void k_fill_records(float *__restrict__ recs, int nbp, int ncol)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < nbp * ncol) recs[i] = i < nbp ? kPadBeta : 0.0f;
}
#endif

constexpr int kGroup = 16;
// Scalars arrive as kernel arguments (SGPRs at wave start) rather than through DevConsts: these
// kernels are a chain of dependent memory round trips, and every hop removed is ~0.5-1 µs.
// Input element (b, col) at recs[b*sb + col*sc]; output records are row-major [ng, 2+HA].
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ __launch_bounds__(kThreads) void k_combine_group(
    const float *__restrict__ recs, int sb, int sc, int nb, int HA, float neg_inv_lambda, float *__restrict__ out, const float *__restrict__ nil_dev)
{
    if (nil_dev != nullptr) neg_inv_lambda = nil_dev[0]; // two-pass normalizeCost: the temperature of this step (k_cost_minmax)
    const int stride = 2 + HA;
    const int b0 = blockIdx.x * kGroup;
    const int n = min(kGroup, nb - b0);
    const int tid = threadIdx.x;
    float bb[kGroup], r[kGroup];
    float beta = INFINITY;
    // loads are UNCONDITIONAL on clamped indices and masked afterwards: a load under a runtime
    // predicate makes hipcc branch around it and wait per element (16 serial round trips, measured 6 µs).
    // The first column's values are requested before the betas so both round trips overlap.
    float v_first[kGroup];
    const int col_first = min(tid, HA);
#pragma unroll
    for (int b = 0; b < kGroup; ++b) v_first[b] = recs[(size_t)(b0 + min(b, n - 1)) * sb + (size_t)(1 + col_first) * sc];
#pragma unroll
    for (int b = 0; b < kGroup; ++b) bb[b] = recs[(size_t)(b0 + min(b, n - 1)) * sb];
#pragma unroll
    for (int b = 0; b < kGroup; ++b) {
        bb[b] = b < n ? bb[b] : INFINITY;
        beta = fminf(beta, bb[b]);
    }
#pragma unroll
    for (int b = 0; b < kGroup; ++b) r[b] = b < n ? expf(neg_inv_lambda * (bb[b] - beta)) : 0.0f;
    if (tid == 0) out[(size_t)blockIdx.x * stride] = beta;
    for (int col = tid; col < HA + 1; col += kThreads) {
        float v[kGroup];
#pragma unroll
        for (int b = 0; b < kGroup; ++b) v[b] = col == tid ? v_first[b] : recs[(size_t)(b0 + min(b, n - 1)) * sb + (size_t)(1 + col) * sc];
        double acc = 0.0;
#pragma unroll
        for (int b = 0; b < kGroup; ++b) acc += (double)r[b] * (double)v[b]; // r[b] = 0 beyond n
        out[(size_t)blockIdx.x * stride + 1 + col] = (float)acc;
    }
}
#endif

// ----------------------------------------------------------------------------------------
// k_finish_cols: the whole combine + update in ONE launch, one workgroup per horizon-action column c:
//   beta = min_b beta_b ; r_b = exp(-(beta_b-beta)/λ) ; eta = Σ r_b eta_b ; V_c = Σ r_b V_b[c]      (double sums)
// and either a shard record (beta, eta, V) for the exchange and/or the update
//   U'[c] = U[c] + V_c/eta ; u = U'[0]   (mBuildUpdateGraph controller_base.cpp:223, mGetNew :326-329).
// Every workgroup derives beta and eta itself from the same loads in the same fixed order (identical
// bits everywhere), so no workgroup waits for another. The shift (mShift + mInit0, :310-324) costs
// nothing: U' is written to the OTHER of two U buffers whose tail holds a_dim permanent zeros, and the
// next step reads that buffer from offset a_dim. With the tile records stored column-major the loads are
// fully coalesced. Replaces a 16-way fold kernel plus a single-workgroup finish (4.6 + 5.2 µs).
// Record element (b, col) at recs[b*sb + col*sc]. grid = HA workgroups of 256 threads.
// column_combine: (beta, eta, V_c) over nb records, valid on thread 0. ld(b, j) returns element j of record b
// (0 = beta_b, 1 = eta_b, 2 = V_b[c]) and must be safe for every b < nb; the same code (and therefore the same
// summation order and bits) serves the tile records in HBM and the gathered shard records in LDS.
template <class Load>
__device__ __forceinline__ void column_combine(Load ld, int nb, float neg_inv_lambda, float *red_f, double (*red_d)[kThreads / 64],
                                               float &beta_out, double &eta_out, double &V_out)
{
    const int tid = threadIdx.x;
    constexpr int PER = 4; // records per thread held in registers (nb <= 1024 in one pass)
    float bb[PER], ee[PER], vv[PER];
    MPPI_FINISH_STOP(0, 0.f); // (timing-study builds only: mppi_ablate.hip.h)
#pragma unroll
    for (int i = 0; i < PER; ++i) { // unconditional clamped loads (see k_combine_group)
        const int b = min(tid + i * kThreads, nb - 1);
        bb[i] = ld(b, 0);
        ee[i] = ld(b, 1);
        vv[i] = ld(b, 2);
    }
    MPPI_FINISH_STOP(1, ((bb[0] + ee[0] + vv[0]) + (bb[1] + ee[1] + vv[1])) + ((bb[2] + ee[2] + vv[2]) + (bb[3] + ee[3] + vv[3])));
    float bmin = INFINITY;
#pragma unroll
    for (int i = 0; i < PER; ++i) bmin = fminf(bmin, (tid + i * kThreads < nb) ? bb[i] : INFINITY);
    for (int b = tid + PER * kThreads; b < nb; b += kThreads) bmin = fminf(bmin, ld(b, 0));
    bmin = wave_min(bmin);
    if ((tid & 63) == 0) red_f[tid >> 6] = bmin;
    __syncthreads();
    float beta = red_f[0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) beta = fminf(beta, red_f[w]);
    MPPI_FINISH_STOP(2, beta + ee[0] + vv[0]);

    double se = 0.0, sv = 0.0;
#pragma unroll
    for (int i = 0; i < PER; ++i) {
        if (tid + i * kThreads < nb) {
            const float r = expf(neg_inv_lambda * (bb[i] - beta));
            se += (double)r * (double)ee[i];
            sv += (double)r * (double)vv[i];
        }
    }
    for (int b = tid + PER * kThreads; b < nb; b += kThreads) {
        const float r = expf(neg_inv_lambda * (ld(b, 0) - beta));
        se += (double)r * (double)ld(b, 1);
        sv += (double)r * (double)ld(b, 2);
    }
    se = wave_sum_d(se);
    sv = wave_sum_d(sv);
    if ((tid & 63) == 0) { red_d[0][tid >> 6] = se; red_d[1][tid >> 6] = sv; }
    __syncthreads();
    double eta = red_d[0][0], V = red_d[1][0];
#pragma unroll
    for (int w = 1; w < kThreads / 64; ++w) { eta += red_d[0][w]; V += red_d[1][w]; }
    beta_out = beta; eta_out = eta; V_out = V;
}

#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
// Behind a PRE-LAUNCHED rollout (k_step_pc<.., STEP_PRE>, mppi_step.hip.h): the sequence travels between the steps as 8-byte {value, tag}
// granules — the finish of step n reads U from granules `in` (complete: its rollout waited for every one of them), writes the SHIFTED U' as
// granules `out` tagged `tag` (what step n+1's rollout, already resident on the other stream, is polling for) beside the plain U_out, and
// takes the Philox step counter from the host's mirror (`step`): nothing it reads was written by a kernel of the other stream as plain stores.
struct FinishPre {
    const unsigned long long *in;
    unsigned long long *out;
    unsigned tag;
    unsigned long long step;
};

__global__ __launch_bounds__(kThreads) void k_finish_cols(
    const float *__restrict__ recs, int sb, int sc, int nb, int HA, int a, float neg_inv_lambda,
    const float *__restrict__ U_in, float *__restrict__ U_out, float *__restrict__ u_out,
    float *__restrict__ record_out, int apply, unsigned long long *__restrict__ step_ctr, float *__restrict__ dbg,
    const float *__restrict__ clip, const float *__restrict__ nil_dev, const unsigned long long *__restrict__ verdict, unsigned seq,
    const FinishPre pre)
{
    __shared__ float red_f[kThreads / 64];
    __shared__ double red_d[2][kThreads / 64];
    const int c = blockIdx.x, tid = threadIdx.x;
    // behind an ARMED rollout launch (mppi_step.hip.h): the update applies only if tile 0 accepted launch `seq` (its verdict word {1, seq});
    // an aborted launch leaves U, u and the step counter as they were
    if (verdict != nullptr && verdict[0] != (((unsigned long long)seq << 32) | 1ull)) return;
    if (nil_dev != nullptr) neg_inv_lambda = nil_dev[0]; // two-pass normalizeCost: the temperature of this step (k_cost_minmax)
    if (pre.in != nullptr) __builtin_amdgcn_s_setprio(3); // (the next step's rollout is resident around these waves, drawing noise — and waiting for them)
    const float u_old = pre.in != nullptr ? __uint_as_float((unsigned)__hip_atomic_load(pre.in + c, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) : U_in[c];
    const unsigned long long step_old = pre.in != nullptr ? pre.step : step_ctr[0];
    float lo = -INFINITY, hi = INFINITY; // clip_act (controller_base.py:500-504): [a_min | a_max], NULL = off
    if (clip != nullptr) { lo = clip[c % a]; hi = clip[a + c % a]; }
    float beta;
    double eta, V;
    column_combine([&](int b, int j) { return recs[(size_t)b * sb + (size_t)(j == 2 ? 2 + c : j) * sc]; },
                   nb, neg_inv_lambda, red_f, red_d, beta, eta, V);
    if (tid == 0) {
        if (record_out != nullptr) {
            record_out[2 + c] = (float)V;
            if (c == 0) { record_out[0] = beta; record_out[1] = (float)eta; }
        }
        if (c == 0 && dbg != nullptr) { dbg[0] = beta; dbg[1] = (float)eta; }
        if (apply) {
            const float un = fminf(fmaxf(u_old + (float)(V / eta), lo), hi);
            U_out[c] = un;            // U' ; the next step reads U_out + a (the shifted sequence)
            if (c < a) u_out[c] = un; // mGetNew
            if (c == 0) step_ctr[0] = step_old + 1ull;
            if (pre.out != nullptr) { // the shifted sequence as granules: column c is row c - a of the next step, the last a rows are zero (mShift / mInit0)
                const int dst = c >= a ? c - a : HA - a + c;
                const float val = c >= a ? un : 0.0f;
                __hip_atomic_store(pre.out + dst, ((unsigned long long)pre.tag << 32) | (unsigned long long)__float_as_uint(val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}
#endif

// ----------------------------------------------------------------------------------------
// k_finish_cols_xchg: the K-sharded step's finish with the record exchange INSIDE the kernel.
// Workgroup c folds this shard's tile records to (beta_g, eta_g, V_g[c]) as above, then
//   send : stores the three values as 8-byte {value, seq} packets into slot (seq&1, c, rank) of EVERY rank's inbox
//          (peer memory mapped over xGMI; one naturally-aligned 64-bit store carries value and flag together, so no
//          fence and no separate flag — the idea of RCCL's LL protocol);
//   recv : spins (system-scope loads of its own uncached inbox) until slot (seq&1, c, g) carries seq for every g;
//   then combines the G records in rank order with the same column_combine and applies U' — replicated, bit-identical
//   on every rank, and bit-identical to the all-gather path (same floats through the same code).
// No workgroup waits on anything a peer sends only after receiving from it (every send precedes the wait), so there is
// no circular wait. Slot reuse is safe with two parities: a rank sends seq+2 only after it has received seq+1 from
// everyone, which they sent after they finished reading seq. Every spin has a wall-clock deadline; what happens on
// expiry is described at the kernel below (zero update + flag, never garbage).
constexpr int kMaxPeers = 16;
struct XchgPeers { unsigned long long *inbox[kMaxPeers]; };

// Returns the packet; *ok = false if the deadline passed first (the flag words are then raised: `status` in mapped host
// memory for the host, `status_dev` in device memory for later launches).
__device__ __forceinline__ unsigned long long xchg_wait(const unsigned long long *slot, unsigned seq, long long timeout_ticks,
                                                        unsigned *status, unsigned *status_dev, bool *ok)
{
    const long long t0 = wall_clock64();
    for (;;) {
        const unsigned long long got = __hip_atomic_load(slot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned)(got >> 32) == seq) { *ok = true; return got; }
        if (wall_clock64() - t0 > timeout_ticks) {
            __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_store(status_dev, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            *ok = false;
            return got;
        }
        __builtin_amdgcn_s_sleep(1);
    }
}

// A missed deadline must not put garbage into the controller: a workgroup whose packets did not all arrive applies a
// ZERO update to its column (U'[c] = U[c]; the sequence still shifts and the Philox step counter still advances, so
// the handle's host-side bookkeeping and the noise streams of the ranks stay aligned) and raises the flag. Once the
// flag is up (`status_dev`, read at the start of every launch) later launches skip the exchange altogether — no sends,
// no spins, zero update — instead of burning one deadline per queued step; mppi_shard_p2p_step refuses further steps
// as soon as the host sees the flag (MPPI_ERR_EXCHANGE), and ShardedController re-synchronises U and the step counter
// from rank 0 before it continues on the all-gather path (distributed.py).
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ __launch_bounds__(kThreads) void k_finish_cols_xchg(
    const float *__restrict__ recs, int sb, int sc, int nb, int HA, int a, float neg_inv_lambda,
    const float *__restrict__ U_in, float *__restrict__ U_out, float *__restrict__ u_out,
    unsigned long long *__restrict__ step_ctr, float *__restrict__ dbg,
    XchgPeers peers, int G, int rank, unsigned seq, long long timeout_ticks, unsigned *status, unsigned *status_dev,
    const float *__restrict__ clip)
{
    __shared__ float red_f[kThreads / 64];
    __shared__ double red_d[2][kThreads / 64];
    __shared__ float mine[3];
    __shared__ float theirs[3][kMaxPeers];
    __shared__ int bad_s;
    const int c = blockIdx.x, tid = threadIdx.x;
    const float u_old = U_in[c];
    const unsigned long long step_old = step_ctr[0];
    const unsigned dead = __hip_atomic_load(status_dev, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // requested early, used late
    float lo = -INFINITY, hi = INFINITY;
    if (clip != nullptr) { lo = clip[c % a]; hi = clip[a + c % a]; }
    float beta;
    double eta, V;
    column_combine([&](int b, int j) { return recs[(size_t)b * sb + (size_t)(j == 2 ? 2 + c : j) * sc]; },
                   nb, neg_inv_lambda, red_f, red_d, beta, eta, V);
    if (tid == 0) { mine[0] = beta; mine[1] = (float)eta; mine[2] = (float)V; bad_s = dead ? 1 : 0; }
    __syncthreads();
    const size_t slot0 = ((size_t)(seq & 1u) * HA + c) * G;
    if (!dead && tid < 3 * G) {
        const int p = tid / 3, j = tid - 3 * p;
        const unsigned long long pkt = ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(mine[j]);
        __hip_atomic_store(peers.inbox[p] + (slot0 + rank) * 3 + j, pkt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        bool ok;
        const unsigned long long got = xchg_wait(peers.inbox[rank] + (slot0 + p) * 3 + j, seq, timeout_ticks, status, status_dev, &ok);
        theirs[j][p] = __uint_as_float((unsigned)got);
        if (!ok) bad_s = 1;
    }
    __syncthreads();
    const bool bad = bad_s != 0; // workgroup-uniform
    if (!bad) column_combine([&](int b, int j) { return theirs[j][b]; }, G, neg_inv_lambda, red_f, red_d, beta, eta, V);
    if (tid == 0) {
        if (c == 0 && dbg != nullptr && !bad) { dbg[0] = beta; dbg[1] = (float)eta; }
        const float un = bad ? u_old : fminf(fmaxf(u_old + (float)(V / eta), lo), hi);
        U_out[c] = un;
        if (c < a) u_out[c] = un;
        if (c == 0) step_ctr[0] = step_old + 1ull;
    }
}
#endif

// k_savgol: filterSeq (controller_base.py:277-291): out[t,j] = Σ_i rows[t,i]·in[start[t]+i, j]. rows holds, for
// every t, the `window` Savitzky-Golay weights that evaluate the least-squares polynomial of the window starting at
// start[t] at position t (centre for interior rows, off-centre for the 'interp' edges); built on the host in fp64.
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ void k_savgol(const float *__restrict__ in, float *__restrict__ out, const float *__restrict__ rows,
                         const int *__restrict__ start, int H, int a, int window)
{
    const int idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= H * a) return;
    const int t = idx / a, j = idx - t * a;
    const float *r = rows + (size_t)t * window;
    const float *src = in + (size_t)start[t] * a + j;
    double acc = 0.0;
    for (int i = 0; i < window; ++i) acc += (double)r[i] * (double)src[(size_t)i * a];
    out[idx] = (float)acc;
}
#endif

// k_xchg_probe: the exchange's self-test, run once after the inboxes are attached and before the first step: the
// same packets, stores and spins on a separate probe region of the inbox (after the 2*HA*G*3 step slots), with a
// known payload. got[g] = the value received from rank g (the host checks got[g] == payload(g, seq)).
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ void k_xchg_probe(XchgPeers peers, size_t probe_off, int G, int rank, unsigned seq, float payload,
                             long long timeout_ticks, unsigned *status, unsigned *status_dev, float *got)
{
    const int p = threadIdx.x;
    if (p >= G) return;
    const size_t slot0 = probe_off + (size_t)(seq & 1u) * G;
    const unsigned long long pkt = ((unsigned long long)seq << 32) | (unsigned long long)__float_as_uint(payload);
    __hip_atomic_store(peers.inbox[p] + slot0 + rank, pkt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    bool ok;
    const unsigned long long r = xchg_wait(peers.inbox[rank] + slot0 + p, seq, timeout_ticks, status, status_dev, &ok);
    got[p] = ok ? __uint_as_float((unsigned)r) : __uint_as_float(0x7fc00000u);
}
#endif

// ----------------------------------------------------------------------------------------
// Learned model_base (BASELINE configs[3]/[4]):
//   in = ([x;v]-Xmean)/Xstd ; h1 = relu(W1ᵀin+b1) ; h2 = relu(W2ᵀh1+b2) ; y = W3ᵀh2+b3 ;
//   x' = x + (y·Ystd + Ymean)          (nn_model.py:215-239,289-304 convention; 2 x 256 hidden units)
// Device-side weights, Keras layout [in x out] row-major.
constexpr int kHid = 256;
struct MlpDev {
    const float *W1, *b1, *W2, *b2, *W3, *b3; // the {256, 256, s} network of k_rollout_mlp / _mlp2 / _bx3 (NULL otherwise)
    float xmean[kMaxS + kMaxA], xstd[kMaxS + kMaxA], ymean[kMaxS], ystd[kMaxS];
    int n_layers, widths[4];                  // every network: Dense layers l = 0 .. n_layers-1, [in x out] row-major
    const float *Wl[4], *bl[4];
    int ld[4];                                // row stride of Wl[l]: widths[l], except an output layer padded to an even width (NNAUVModel: 13 -> 14)
};

typedef float f32x16 __attribute__((ext_vector_type(16)));

// One sample per thread, plain loops in the reference order (mul and add rounded separately,
// input index ascending): the slow, exactly-ordered evaluation behind mppi_model_step for MLP handles.
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ void k_mlp_step_ref(const DevConsts *__restrict__ C, const MlpDev *__restrict__ M,
                               const float *__restrict__ x, int kx, const float *__restrict__ v, int k,
                               float *__restrict__ scratch, float *__restrict__ out_next)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int s = C->s, a = C->a, nin = s + a;
    float *cur = scratch + (size_t)i * 2 * kHid, *nxt = cur + kHid;
    const float *xi = x + (size_t)(kx == 1 ? 0 : i) * s;
    for (int j = 0; j < s; ++j) cur[j] = (xi[j] - M->xmean[j]) / M->xstd[j];
    for (int j = 0; j < a; ++j) cur[s + j] = (v[(size_t)i * a + j] - M->xmean[s + j]) / M->xstd[s + j];
    int width = nin;
    for (int l = 0; l < M->n_layers; ++l) {
        const int out_w = M->widths[l];
        const float *W = M->Wl[l], *b = M->bl[l];
        for (int o = 0; o < out_w; ++o) {
            float acc = 0.0f;
            for (int j = 0; j < width; ++j) acc = acc + cur[j] * W[j * out_w + o];
            acc = acc + b[o];
            nxt[o] = (l + 1 < M->n_layers && acc < 0.0f) ? 0.0f : acc; // relu on hidden layers
        }
        float *t = cur; cur = nxt; nxt = t;
        width = out_w;
    }
    for (int o = 0; o < s; ++o) out_next[(size_t)i * s + o] = xi[o] + (cur[o] * M->ystd[o] + M->ymean[o]);
}
#endif

// Tile record of the MLP rollout kernels: every wave holds the same 64 costs; beta, eta by wave 0, and
// V_b[t,i] = Σ_k e_k·eps[k,t,i] with wave w regenerating the noise of horizon groups g = w, w+8, ... from the Philox
// counters (cheap next to H steps of MFMA).
template <int A, bool DIAG, int NWAVES = 8>
__device__ __forceinline__ void mlp_tile_record(const DevConsts *__restrict__ C, float c, bool valid, int w, int lane, int kk,
                                                int H, int NG, int SRC, const float *__restrict__ eps_hbm,
                                                unsigned long long seed, unsigned long long gk, unsigned long long base,
                                                float *__restrict__ rec, int rsc)
{
    const int HA = H * A;
    const float beta = wave_min(valid ? c : INFINITY);
    const float ek = valid ? expf(C->neg_inv_lambda * (c - beta)) : 0.0f;
    const float eta = wave_sum(ek);
    if (w == 0 && lane == 0) { rec[0] = beta; rec[(size_t)rsc] = eta; }
    // the 4 A columns of a horizon group go through ONE transposing butterfly (wave_transpose_sum: ~2.3 instructions per column where a
    // 64-lane sum per column costs 12, and one store instruction per group): the regenerated noise is a fifth of a 13-state pipeline kernel's
    // instructions, and this was a third of that
    const int col = lane_column(lane), ctl = col / A, ci = col - ctl * A; // the column of a group this lane ends up with the total of
    for (int g = w; g < NG; g += NWAVES) {
        float zz[4 * A], prod[4 * A];
        if (SRC == SRC_PHILOX) normals_group<A>(seed, gk, base + (unsigned long long)g, zz);
#pragma unroll
        for (int tl = 0; tl < 4; ++tl) {
            const int t = 4 * g + tl;
            float z1[A], e[A];
            if (SRC == SRC_PHILOX) {
#pragma unroll
                for (int i = 0; i < A; ++i) z1[i] = zz[tl * A + i];
                scale_noise<A, DIAG>(C, z1, e);
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) e[i] = t < H ? eps_hbm[(size_t)kk * HA + t * A + i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < A; ++i) prod[tl * A + i] = ek * e[i]; // (steps past the horizon: summed like the others, never stored)
        }
        float tot[1];
        wave_transpose_sum<4 * A>(prod, tot, lane);
        const int t = 4 * g + ctl;
        if (col < 4 * A && t < H) rec[(size_t)(2 + t * A + ci) * rsc] = tot[0];
    }
}

// k_rollout_mlp: one workgroup of 8 wavefronts owns 64 rollouts for the whole horizon.
// Transposed formulation hᵀ = Wᵀ·inᵀ so that the rollout index sits on the LANE of every MFMA
// operand and result and everything per-rollout (state, cost, noise) is lane-local:
//   A operand = weights, WEIGHT-STATIONARY in registers: wave w owns hidden units [32w,32w+32) of both
//       layers: 128+1 VGPRs of W2 (k pairs (2s,2s+1) in the lane halves; bias as k=256 against a row of
//       ones) and 5 VGPRs of W1 (k = 0..8 inputs, 9 = bias);
//   B operand = activations [k][rollout], two column blocks of 32 rollouts per weight register (each
//       A register feeds two MFMAs): layer 1 from the lanes' own normalised inputs (one cross-half
//       exchange), layer 2 from h1ᵀ in LDS ([256][64] fp32 = 64 KB, conflict-free row reads);
//   v_mfma_f32_32x32x2_f32: exact fp32 (a k-ordered fmaf chain), 64 FLOP/clk/SIMD = the fp32 peak;
//       two independent accumulators per wave keep the pipe issuing back to back.
//   layer 3 (256 -> s) is 2 % of the FLOPs: VALU partial dot products over the wave's 32 units straight
//       from the accumulator registers, lane halves and the 8 waves summed in fixed order through LDS.
// Lane l of EVERY wave owns rollout l of the tile (state, running cost): the per-step scalar work
// (normalise, state update, costs: ~1/10 of the MFMA time) is replicated across the 8 waves instead of
// broadcast, and covers 64 rollouts per instruction. Noise comes from a double-buffered LDS block that
// one wave per horizon group fills (Philox once per workgroup, not once per wave). The weighted-noise
// sum regenerates eps from the Philox counters at the end (cheap next to H steps of MFMA).
// Two s_barriers per horizon step. LDS: 64 KB h1 + 12 KB partial y + 6 KB W3 + 6 KB noise.
// Tried and dropped (round 1): phase-shifting the two column blocks so one block's scalar chain issues in
// the MFMA gaps of the other (one micro-step per MFMA slot, pinned with sched_barrier). It was correct but
// ran at 76 TFLOP/s against 122: with 134 weight + 32 accumulator registers the extra live ranges spilled
// (256 B/lane of scratch inside the MFMA stream) and each pinned micro-step's LDS latency stalled the
// in-order wave past its next MFMA. It needs the weights partly in LDS before it can pay.
constexpr int kMlpThreads = 512;
constexpr int kMlpR = 64;
__host__ __device__ inline size_t mlp_lds_floats(int S, int A)
{
    return (size_t)kHid * kMlpR + 8 * S * kMlpR + kHid * S + 2 * 4 * A * kMlpR + 64;
}

template <int A, bool DIAG>
__global__ __launch_bounds__(kMlpThreads, 2) void k_rollout_mlp(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr bool QFULL = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * A, NIN = S + A;
    constexpr int K1 = (NIN + 2) / 2 * 2; // inputs + bias, padded to the MFMA's k pairs (10 for s=6,a=3)
    constexpr int R = kMlpR;
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    float *h1_s = smem;                 // [kHid][R]
    float *y_s = h1_s + kHid * R;       // [8][S][R]
    float *w3_s = y_s + 8 * S * R;      // [kHid][S]
    float *z_s = w3_s + kHid * S;       // [2][4*A][R] standard normals of the current / next horizon group

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
    const int k0 = blockIdx.x * R;
    const bool valid = (k0 + lane) < K;
    const int kk = valid ? k0 + lane : K - 1; // clamp: out-of-range lanes recompute the last sample, masked later

    // ---- stationary weights -> registers
    float a2[kHid / 2 + 1], a1[K1 / 2];
    const int unit = 32 * w + j;
#pragma unroll
    for (int s2 = 0; s2 < kHid / 2; ++s2) a2[s2] = M->W2[(size_t)(2 * s2 + hh) * kHid + unit];
    a2[kHid / 2] = hh == 0 ? M->b2[unit] : 0.0f;
#pragma unroll
    for (int s1 = 0; s1 < K1 / 2; ++s1) {
        const int kin = 2 * s1 + hh;
        a1[s1] = kin < NIN ? M->W1[(size_t)kin * kHid + unit] : (kin == NIN ? M->b1[unit] : 0.0f);
    }
    for (int i = tid; i < kHid * S; i += kMlpThreads) w3_s[i] = M->W3[i];

    float x[S];
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];
    float c = 0.0f;
    const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    if (SRC == SRC_PHILOX && w == 0) { // horizon group 0
        float z[4 * A];
        normals_group<A>(seed, gk, base, z);
#pragma unroll
        for (int m = 0; m < 4 * A; ++m) z_s[m * R + lane] = z[m];
    }
    __syncthreads();

    for (int t = 0; t < H; ++t) {
        float u[A], e[A], v[A];
        if (SRC == SRC_PHILOX) {
            const float *zb = z_s + ((t >> 2) & 1) * (4 * A * R) + (t & 3) * A * R + lane;
            float zz[A];
#pragma unroll
            for (int i = 0; i < A; ++i) zz[i] = zb[i * R];
            scale_noise<A, DIAG>(C, zz, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(C, u, e);

        // normalised inputs of this lane's rollout (+ the bias input 1, + zero padding)
        float in[K1];
#pragma unroll
        for (int i = 0; i < S; ++i) in[i] = (x[i] - M->xmean[i]) / M->xstd[i];
#pragma unroll
        for (int i = 0; i < A; ++i) in[S + i] = (v[i] - M->xmean[S + i]) / M->xstd[S + i];
        in[NIN] = 1.0f;
#pragma unroll
        for (int i = NIN + 1; i < K1; ++i) in[i] = 0.0f;

        // ---- layer 1: h1ᵀ[32w.., cb*32..] = relu(W1ᵀ in + b1). Column block cb's B operand for k pair s1 is
        // in[2 s1 + hh] of rollout 32cb+j: the lane's own value when cb == hh, its partner's (lane^32) otherwise.
        f32x16 acc0 = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        f32x16 acc1 = acc0;
#pragma unroll
        for (int s1 = 0; s1 < K1 / 2; ++s1) {
            const float mine = hh ? in[2 * s1 + 1] : in[2 * s1];     // in[2 s1 + hh] of my rollout (block hh)
            const float send = hh ? in[2 * s1] : in[2 * s1 + 1];     // in[2 s1 + (1-hh)]: what my partner needs
            const float theirs = __shfl_xor(send, 32, 64);           // partner's in[2 s1 + hh] (block 1-hh)
            const float b0 = hh ? theirs : mine;                     // column block 0
            const float b1 = hh ? mine : theirs;                     // column block 1
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s1], b0, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s1], b1, acc1, 0, 0, 0);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            h1_s[(32 * w + row) * R + j] = acc0[r] < 0.0f ? 0.0f : acc0[r];
            h1_s[(32 * w + row) * R + 32 + j] = acc1[r] < 0.0f ? 0.0f : acc1[r];
        }
        __syncthreads();

        // the next horizon group's normals, once per workgroup (wave g%8), into the other half of the buffer
        if (SRC == SRC_PHILOX && (t & 3) == 0) {
            const int gn = (t >> 2) + 1;
            if (gn < NG && (gn & 7) == w) {
                float z[4 * A];
                normals_group<A>(seed, gk, base + (unsigned long long)gn, z);
                float *zd = z_s + (gn & 1) * (4 * A * R) + lane;
#pragma unroll
                for (int m = 0; m < 4 * A; ++m) zd[m * R] = z[m];
            }
        }

        // ---- layer 2: h2ᵀ[32w.., :] = relu(W2ᵀ h1 + b2): 128 k pairs + the bias pair, two column blocks.
        // Batches of 4 k pairs: 8 LDS reads in flight, then 8 MFMAs on two independent accumulators; the
        // scheduling fence keeps the compiler from hoisting all 256 reads at once (that spills).
        acc0 = (f32x16){0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc1 = acc0;
        const float *h1p = h1_s + hh * R + j;
#pragma unroll
        for (int sb = 0; sb < kHid / 2; sb += 4) {
            float bq0[4], bq1[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { bq0[q] = h1p[2 * (sb + q) * R]; bq1[q] = h1p[2 * (sb + q) * R + 32]; }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[sb + q], bq0[q], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[sb + q], bq1[q], acc1, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        {
            const float one = hh == 0 ? 1.0f : 0.0f;
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[kHid / 2], one, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[kHid / 2], one, acc1, 0, 0, 0);
        }

        // ---- layer 3 partial over this wave's 32 units (16 per lane half), both column blocks, VALU
        float py0[S], py1[S];
#pragma unroll
        for (int n = 0; n < S; ++n) { py0[n] = 0.0f; py1[n] = 0.0f; }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = (r & 3) + 8 * (r >> 2) + 4 * hh;
            const float hv0 = acc0[r] < 0.0f ? 0.0f : acc0[r];
            const float hv1 = acc1[r] < 0.0f ? 0.0f : acc1[r];
            const float *w3 = w3_s + (32 * w + row) * S;
#pragma unroll
            for (int n = 0; n < S; ++n) {
                const float wv = w3[n];
                py0[n] = __builtin_fmaf(hv0, wv, py0[n]);
                py1[n] = __builtin_fmaf(hv1, wv, py1[n]);
            }
        }
#pragma unroll
        for (int n = 0; n < S; ++n) { // halves hold different rows: lane half hh keeps column block hh
            const float keep = hh ? py1[n] : py0[n];
            const float send = hh ? py0[n] : py1[n];
            y_s[(w * S + n) * R + lane] = keep + __shfl_xor(send, 32, 64);
        }
        __syncthreads();

        // ---- y = Σ_waves partial + b3 (fixed order), state update, costs
#pragma unroll
        for (int n = 0; n < S; ++n) {
            float y = y_s[(0 * S + n) * R + lane];
#pragma unroll
            for (int ww = 1; ww < 8; ++ww) y = y + y_s[(ww * S + n) * R + lane];
            y = y + M->b3[n];
            x[n] = x[n] + (y * M->ystd[n] + M->ymean[n]);
        }
        const float sc = state_cost<S, QFULL>(C, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    }
    c = c + state_cost<S, QFULL>(C, x); // terminal cost, controller_base.cpp:271-272
    if (w == 0 && valid) cost[k0 + lane] = c;
    if (MODE == MODE_COST_ONLY) return;

    mlp_tile_record<A, DIAG>(C, c, valid, w, lane, kk, H, NG, SRC, eps_hbm, seed, gk, base,
                             partials + (size_t)record_slot(blockIdx.x, rsc) * rsb, rsc); // element (b, col) at partials[b*rsb + col*rsc]
}

} // namespace mppi
#include "mppi_mlp2.hip.h"
#include "mppi_bx3.hip.h"
#include "mppi_mlp_small.hip.h"
#include "mppi_mlp32.hip.h"
#include "mppi_mlp32b.hip.h"
namespace mppi {

// min / max of the costs (Py normalizeCost, controller_base.py:468-474): out[0]=min, out[1]=max-min
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
// nil_out != NULL: also out[2] = *nil_out = neg_inv_lambda / (max - min) — the soft-min temperature at which the RAW costs weigh
// as the normalised ones do at lambda (exp(-(c'-min c')/lambda) = exp(-(c-min c)/(lambda (max-min)))): the second rollout pass of
// the two-pass normalizeCost path reads it from a DevConsts copy, the finish from out[2]
__global__ __launch_bounds__(1024) void k_cost_minmax(const float *__restrict__ cost, int K, float *__restrict__ out,
                                                      float neg_inv_lambda, float *__restrict__ nil_out, float *__restrict__ range_out)
{
    __shared__ float mn_s[16], mx_s[16];
    const int tid = threadIdx.x, nt = blockDim.x;
    float mn = INFINITY, mx = -INFINITY;
    const int K4 = K >> 2; // 16-byte loads (hipMalloc'd buffer), the tail one by one
    const float4 *c4 = reinterpret_cast<const float4 *>(cost);
    for (int i = tid; i < K4; i += nt) {
        const float4 v = c4[i];
        mn = fminf(fminf(mn, v.x), fminf(v.y, fminf(v.z, v.w)));
        mx = fmaxf(fmaxf(mx, v.x), fmaxf(v.y, fmaxf(v.z, v.w)));
    }
    for (int i = 4 * K4 + tid; i < K; i += nt) { const float c = cost[i]; mn = fminf(mn, c); mx = fmaxf(mx, c); }
    mn = wave_min(mn); mx = wave_max(mx);
    if ((tid & 63) == 0) { mn_s[tid >> 6] = mn; mx_s[tid >> 6] = mx; }
    __syncthreads();
    if (tid == 0) {
        for (int w = 1; w < nt / 64; ++w) { mn = fminf(mn, mn_s[w]); mx = fmaxf(mx, mx_s[w]); }
        out[0] = mn; out[1] = mx - mn;
        if (nil_out != nullptr) { const float v = neg_inv_lambda / (mx - mn); out[2] = v; *nil_out = v; }
        if (range_out != nullptr) { range_out[0] = -mn; range_out[1] = mx; } // a shard's own {-min, max}: both reduce over the ranks with MAX
    }
}

// K-sharded normalizeCost (mppi_shard_partial_normalized): the GLOBAL {-min, max} the ranks agreed on becomes this handle's
// {min, max - min, -1/(lambda (max - min))}, exactly what k_cost_minmax leaves on an unsharded handle
__global__ void k_range_apply(const float *__restrict__ range, float *__restrict__ out, float neg_inv_lambda, float *__restrict__ nil_out)
{
    if (threadIdx.x == 0 && blockIdx.x == 0) {
        const float mn = -range[0], mx = range[1];
        out[0] = mn; out[1] = mx - mn;
        const float v = neg_inv_lambda / (mx - mn);
        out[2] = v;
        if (nil_out != nullptr) *nil_out = v;
    }
}
#endif

// c' = (c - min)/(max - min)  (norm_arg with normalize=True, controller_base.py:468-474)
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ void k_cost_normalize(const float *__restrict__ cost, int K, const float *__restrict__ mm, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < K) out[i] = (cost[i] - mm[0]) / mm[1];
}
#endif

// Per-sample intermediates of the update for inspection (mExpArg, mExp, mWeights :170-186)
#if defined(MPPI_UNIT_CAPI) // non-template kernel: defined in ONE translation unit (mppi_capi.hip)
__global__ void k_weights(const DevConsts *__restrict__ C, const float *__restrict__ cost, int K,
                          const float *__restrict__ beta_eta, float *__restrict__ arg_out,
                          float *__restrict__ exp_out, float *__restrict__ w_out, const float *__restrict__ nil_dev)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= K) return;
    // (nil_dev: the step's temperature of the two-pass normalizeCost path, d_mm[2])
    const float arg = (nil_dev != nullptr ? nil_dev[0] : C->neg_inv_lambda) * (cost[i] - beta_eta[0]);
    const float e = expf(arg);
    if (arg_out) arg_out[i] = arg;
    if (exp_out) exp_out[i] = e;
    if (w_out) w_out[i] = e / beta_eta[1];
}
#endif

// ----------------------------------------------------------------------------------------
// The reference's public graph helpers as kernels (one thread per sample) — the SAME device
// functions the tile kernel runs, so the reference's known-answer vectors exercise them.
// Shapes outside the instantiated set run the zero-padded <kMaxS,kMaxA> instance (adding
// exact zeros does not change any sum).
template <int A>
__global__ void k_model_step(const DevConsts *__restrict__ C, const float *__restrict__ x, int kx,
                             const float *__restrict__ v, int k, int s, int a,
                             float *__restrict__ out_free, float *__restrict__ out_action, float *__restrict__ out_next)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    constexpr int S = 2 * A;
    float xs[S], vs[A], fr[S], ac[S];
    const float *xi = x + (size_t)(kx == 1 ? 0 : i) * s;
#pragma unroll
    for (int j = 0; j < S; ++j) xs[j] = j < s ? xi[j] : 0.0f;
#pragma unroll
    for (int j = 0; j < A; ++j) vs[j] = j < a ? v[(size_t)i * a + j] : 0.0f;
    pm_free_step<A>(C, xs, fr);
    pm_action_step<A>(C, vs, ac);
    for (int j = 0; j < s; ++j) {
        if (out_free && (kx != 1 || i == 0)) out_free[(size_t)(kx == 1 ? 0 : i) * s + j] = fr[j];
        if (out_action) out_action[(size_t)i * s + j] = ac[j];
        if (out_next) out_next[(size_t)i * s + j] = fr[j] + ac[j];
    }
}

template <int S, int A, bool QFULL>
__global__ void k_costs(const DevConsts *__restrict__ C, const float *__restrict__ x, const float *__restrict__ u,
                        const float *__restrict__ eps, int k, int s, int a,
                        float *__restrict__ out_state, float *__restrict__ out_action, float *__restrict__ out_step)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    float sc = 0.0f, ac = 0.0f;
    if (x != nullptr) {
        float xs[S];
#pragma unroll
        for (int j = 0; j < S; ++j) xs[j] = j < s ? x[(size_t)i * s + j] : 0.0f;
        sc = state_cost_of<S, QFULL>(C, xs);
    }
    if (u != nullptr) {
        float us[A], es[A];
#pragma unroll
        for (int j = 0; j < A; ++j) { us[j] = j < a ? u[j] : 0.0f; es[j] = j < a ? eps[(size_t)i * a + j] : 0.0f; }
        ac = action_cost<A>(C, us, es);
    }
    if (out_state) out_state[i] = sc;
    if (out_action) out_action[i] = ac;
    if (out_step) out_step[i] = sc + ac;
}

} // namespace mppi
