// mppi_mlp32.hip.h — k_rollout_mlp32: the reference's Dense(32, relu) x 1..3 + Dense(s) network (nn_model.py:54-60) on
// the matrix cores, weights stationary in registers. Included by mppi_kernels.hip.h.
//
// k_rollout_mlp_small (one rollout per lane, weights through the scalar cache) is starved at BASELINE sizes: K=65536 is
// 1024 waves = ONE per SIMD, every s_load wait exposed (0.36 ms per step, 0.38 of the vector peak). A 32-wide layer is
// exactly one v_mfma_f32_32x32x2_f32 tile (32 units x 32 rollouts, 16 k pairs), and — the point of this kernel — the
// accumulator layout of one layer IS the B-operand layout of the next: lane (j, hh) holds in accumulator register r the
// unit u(r, hh) = 8 (r >> 2) + 4 hh + (r & 3) of rollout j, and the B operand of k pair s wants "input 2 s + hh of rollout
// j" in lane (j, hh). Numbering the next layer's inputs so that input (2 s + hh) is unit u(s, hh) makes accumulator
// register s, after relu, the B operand of MFMA s: no LDS, no shuffle, no barrier between layers. The weights (A operands:
// lane (m, hh) holds W[u(s, hh)][m]) and the biases (the C operand of a layer's first MFMA: lane (., hh), register r holds
// b[u(r, hh)]) are loaded once per wave: 5 + 16 + 2 x (16 + 16) registers.
//   * one wave = 32 rollouts (both lane halves carry rollout j's state, as in k_rollout_mlp2); a workgroup = 2 waves = one
//     64-rollout tile record; K=65536 gives 2 waves per SIMD, which is what fills the MFMA -> relu dependency gaps;
//   * the output layer (s of 32 rows used) stays on the vector ALU straight from the accumulators, W3 rows broadcast from
//     LDS, lane halves combined with v_permlane32_swap;
//   * per step and wave: 5 + 16 (n_hidden - 1) MFMAs (64 cycles each) + ~300 vector instructions.
// Arithmetic: fmaf chains in the layer's k order (units in u(s, hh) order instead of ascending; bias first instead of
// last): within the same tolerance class as the other MLP kernels, held to the same tests.
#pragma once

#include "mppi_mfma32.hip.h"

namespace mppi {

constexpr int kMlp32Threads = 128;
constexpr int kMlp32R = 64; // rollouts per workgroup (2 waves x 32)

template <int A>
__global__ __launch_bounds__(kMlp32Threads) void k_rollout_mlp32(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr int S = 2 * A, NIN = S + A, SP = S / 2, HID = 32;
    constexpr int K1H = (NIN + 1) / 2; // k pairs of layer 1
    constexpr bool QFULL = false, DIAG = false;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float w3_s[HID * 8];     // output-layer rows, padded to 8
    __shared__ float z_s[2][4 * A][32];                               // per wave: the normals of one horizon group
    __shared__ float cost_s[kMlp32R];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
    const int k0 = blockIdx.x * kMlp32R;
    const int kk = min(k0 + 32 * w + j, K - 1); // rollouts past K recompute the last sample, outside every sum
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
    const int n_hidden = M->n_layers - 1;
    auto unit_of = [](int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); };

    // ---- stationary operands
    float a1[K1H];
#pragma unroll
    for (int s1 = 0; s1 < K1H; ++s1) a1[s1] = (2 * s1 + hh) < NIN ? M->Wl[0][(2 * s1 + hh) * HID + j] : 0.0f;
    f32x16 b1t, bht[2];
    float ah[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b1t[r] = M->bl[0][unit_of(r, hh)];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const bool have = l + 2 <= n_hidden; // hidden-to-hidden layer l exists
        const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
        for (int s = 0; s < 16; ++s) ah[l][s] = have ? Wl[unit_of(s, hh) * HID + j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bht[l][r] = have ? bl[unit_of(r, hh)] : 0.0f;
    }
    const float *W3g = M->Wl[n_hidden], *b3g = M->bl[n_hidden];
    for (int i = tid; i < HID * 8; i += kMlp32Threads) w3_s[i] = (i & 7) < S ? W3g[(i >> 3) * S + (i & 7)] : 0.0f;
    float xm[NIN], xr[NIN], b3v[S], ysd[S], ymn[S];
#pragma unroll
    for (int i = 0; i < NIN; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; }
#pragma unroll
    for (int i = 0; i < S; ++i) { b3v[i] = b3g[i]; ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    float x[S], c = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];
    __syncthreads();

    // the layers are single asm statements on fixed accumulator registers (mppi_mfma32.hip.h): MFMAs, wait states and relu together
    // output layer + state update + step cost, from the (relu'd) accumulators of the last hidden layer
    auto finish = [&](const f32x16 &hacc, float ac) {
        f32x2 py[SP];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + (r & 3); // + 4 hh through the lane's base
            const float *wp = w3_s + (4 * hh + row) * 8;
            const f32x4 lo = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp, 16));
            const f32x4 hi = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp + 4, 16));
            const float wv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
            const f32x2 h2 = {hacc[r], hacc[r]};
#pragma unroll
            for (int p2 = 0; p2 < SP; ++p2) {
                const f32x2 w2 = {wv[2 * p2], wv[2 * p2 + 1]};
                py[p2] = r == 0 ? h2 * w2 : __builtin_elementwise_fma(h2, w2, py[p2]);
            }
        }
#pragma unroll
        for (int n = 0; n < S; ++n) { // the other half's rows: lower + upper, in every lane
            float a = (n & 1) ? py[n / 2].y : py[n / 2].x, b = a;
            permlane32_swap(a, b); // a = the lower half's partial in all lanes, b = the upper half's
            const float y = (a + b) + b3v[n];
            x[n] = x[n] + (y * ysd[n] + ymn[n]);
        }
        const float sc = state_cost<S, QFULL>(C, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    };

    for (int t = 0; t < H; ++t) {
        if (SRC == SRC_PHILOX && (t & 3) == 0) { // this wave's normals of the group: block q by the half with q & 1 == hh
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < A; ++q) {
                if ((q & 1) == hh) {
                    const float4 n = normals_of_block(seed, gk, (base + (unsigned long long)(t >> 2)) * A + q);
                    z_s[w][4 * q + 0][j] = n.x; z_s[w][4 * q + 1][j] = n.y; z_s[w][4 * q + 2][j] = n.z; z_s[w][4 * q + 3][j] = n.w;
                }
            }
            __builtin_amdgcn_s_waitcnt(0xc07f); // lgkmcnt(0): one wave's LDS accesses complete in order
            __builtin_amdgcn_wave_barrier();
        }
        float u[A], e[A], v[A];
        if (SRC == SRC_PHILOX) {
            float z1[A];
#pragma unroll
            for (int i = 0; i < A; ++i) z1[i] = z_s[w][(t & 3) * A + i][j];
            scale_noise<A, DIAG>(C, z1, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(C, u, e);
        // layer 1: the B operand of lane (j, hh), k pair s1, is input 2 s1 + hh of rollout j — the lane's own value
        float bv[K1H];
#pragma unroll
        for (int s1 = 0; s1 < K1H; ++s1) {
            const int i0 = 2 * s1, i1 = 2 * s1 + 1;
            const float r0 = i0 < S ? x[i0] : v[i0 - S], m0 = xm[i0], q0 = xr[i0];
            const float r1 = i1 < NIN ? (i1 < S ? x[i1] : v[i1 - S]) : 0.0f, m1 = i1 < NIN ? xm[i1] : 0.0f, q1 = i1 < NIN ? xr[i1] : 0.0f;
            bv[s1] = hh ? (r1 - m1) * q1 : (r0 - m0) * q0;
        }
        f32x16 acc0;
        mfma32_layer1<K1H>(acc0, a1, bv, b1t);
        if (n_hidden >= 2) {
            f32x16 acc1;
            mfma32_hidden_64_80(acc1, acc0, ah[0], bht[0]);
            if (n_hidden >= 3) {
                mfma32_hidden_80_64(acc0, acc1, ah[1], bht[1]);
                finish(acc0, ac);
            } else {
                finish(acc1, ac);
            }
        } else {
            finish(acc0, ac);
        }
    }
    c = c + state_cost<S, QFULL>(C, x); // terminal cost, controller_base.cpp:271-272
    // lane l of BOTH waves now stands for rollout k0 + l of the tile
    if (hh == 0) cost_s[32 * w + j] = c;
    __syncthreads();
    const float ct = cost_s[lane];
    const bool valid = (k0 + lane) < K;
    const int kt = valid ? k0 + lane : K - 1;
    if (w == 0 && valid) cost[k0 + lane] = ct;
    if (MODE == MODE_COST_ONLY) return;
    mlp_tile_record<A, DIAG, 2>(C, ct, valid, w, lane, kt, H, NG, SRC, eps_hbm, seed, (unsigned long long)C->k_offset + (unsigned long long)kt,
                                base, partials + (size_t)record_slot(blockIdx.x, rsc) * rsb, rsc);
}

// k_rollout_mlp32_pc: the same network as a two-wave pipeline per 64-rollout tile (r04; the split of k_rollout_nnauv_pc, mppi_gen.hip.h,
// applied to the point-mass shapes). In k_rollout_mlp32 a wave is 32 rollouts and everything per rollout — noise, costs, the output layer,
// the state update: ~300 vector instructions a step — runs on 64 lanes for 32 results, next to f32 MFMAs that do not overlap with them.
//   wave N (network): lane = rollout. Inputs (its state, the perturbed action from C), the Dense stack on TWO column blocks of 32 rollouts per
//       weight register (mfma32x2_*: one v_permlane32_swap of the lane's (even, odd) inputs yields the B operands of both blocks), the s
//       outputs from the accumulators, x' = x + denorm(y);
//   wave C (cost):    lane = rollout. The cost of the state the previous step produced, and the noise: v = u + eps and the action cost of the
//       NEXT step.
// N -> C: the state; C -> N: the perturbed action, a step ahead; one workgroup barrier per step. Two tiles per workgroup, roles by SIMD as in
// the 13-state pipelines. Same arithmetic per rollout as k_rollout_mlp32 (the output layer's half sums in the same order): same bars.
constexpr int kMlp32PcThreads = 256;

template <int A, bool DIAG>
__global__ __launch_bounds__(kMlp32PcThreads, 2) void k_rollout_mlp32_pc(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int SRC, const int MODE, const int rsb, const int rsc, const int n_tiles, const int balance)
{
    constexpr int S = 2 * A, NIN = S + A, SP = S / 2, HID = 32, K1H = (NIN + 1) / 2, W3LD = 8;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float w3_s[HID * W3LD]; // output-layer rows, padded to 8
    __shared__ float xs_s[2][2][S][64];       // [tile of the workgroup][step parity][state the step produced][rollout]   N -> C
    __shared__ float act_s[2][2][A + 1][64];  // [tile][step parity][perturbed action v, action cost][rollout]            C -> N (v), C keeps the cost
    __shared__ float cost_s[2][64];
    __shared__ __attribute__((aligned(16))) float cst_s[5][16]; // xmean, 1/xstd, b3, ystd, ymean: wave-uniform, read back by broadcast every step
    __shared__ int simd_s[4];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pair = wave_hw >> 1, role = wave_hw & 1; // role 0 = network, 1 = cost
    {
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_REG_HW_ID[5:4]
        if (lane == 0) simd_s[wave_hw] = simd;
        __syncthreads();
        const int s0 = simd_s[0], s1 = simd_s[1], s2 = simd_s[2], s3 = simd_s[3];
        if (balance && ((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15) {
            const int gen = (int)(blockIdx.x >> 8);
            pair = simd & 1;
            role = ((simd >> 1) ^ gen) & 1;
        }
        pair = __builtin_amdgcn_readfirstlane(pair);
        role = __builtin_amdgcn_readfirstlane(role);
    }
    const int tile = 2 * (int)blockIdx.x + pair;
    const bool tile_ok = tile < n_tiles; // the second tile of the last workgroup may not exist: its waves still keep every barrier
    const int k0 = tile * 64;
    const bool valid = tile_ok && (k0 + lane) < K;
    const int kk = min(k0 + lane, K - 1); // rollouts past K recompute the last sample, outside every sum
    const int n_hidden = M->n_layers - 1;
    const float *W3g = M->Wl[n_hidden];
    for (int i = tid; i < HID * W3LD; i += kMlp32PcThreads) w3_s[i] = (i & 7) < S ? W3g[(i >> 3) * S + (i & 7)] : 0.0f;
    if (tid < 16) {
        const float *b3c = M->bl[n_hidden];
        cst_s[0][tid] = tid < NIN ? M->xmean[tid] : 0.0f;
        cst_s[1][tid] = tid < NIN ? 1.0f / M->xstd[tid] : 0.0f;
        cst_s[2][tid] = tid < S ? b3c[tid] : 0.0f;
        cst_s[3][tid] = tid < S ? M->ystd[tid] : 0.0f;
        cst_s[4][tid] = tid < S ? M->ymean[tid] : 0.0f;
    }

    if (role == 0) {
        // ================================================================================= wave N: network, state
        if (balance) __builtin_amdgcn_s_setprio(3); // (one round of the grid only: beyond it the age order staggers the workgroups' phases, as in k_rollout_pc)
        // the wave a step waits for goes first on its SIMD (the other kind fills the gaps)
        const int j = lane & 31, hh = lane >> 5;
        auto unit_of = [](int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); };
        float a1[K1H];
#pragma unroll
        for (int s1 = 0; s1 < K1H; ++s1) a1[s1] = (2 * s1 + hh) < NIN ? M->Wl[0][(2 * s1 + hh) * HID + j] : 0.0f;
        f32x16 b1t, bht[2];
        float ah[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) b1t[r] = M->bl[0][unit_of(r, hh)];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const bool have = l + 2 <= n_hidden;
            const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
            for (int s1 = 0; s1 < 16; ++s1) ah[l][s1] = have ? Wl[unit_of(s1, hh) * HID + j] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) bht[l][r] = have ? bl[unit_of(r, hh)] : 0.0f;
        }
        float x[S];
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = x_dev[i];
        __syncthreads(); // w3_s, cst_s, and the perturbed action of step 0

        // output layer of both column blocks + the state update (state + de-normalised delta)
        auto finish = [&](const f32x16 &hA, const f32x16 &hB) {
            f32x2 pA[SP], pB[SP];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *wp = w3_s + (8 * (r >> 2) + 4 * hh + (r & 3)) * W3LD;
                const f32x4 lo = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp, 16));
                const f32x4 hi = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp + 4, 16));
                const float wv[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
                const f32x2 a2 = {hA[r], hA[r]}, b2 = {hB[r], hB[r]};
#pragma unroll
                for (int p2 = 0; p2 < SP; ++p2) {
                    const f32x2 w2 = {wv[2 * p2], wv[2 * p2 + 1]};
                    pA[p2] = r == 0 ? a2 * w2 : __builtin_elementwise_fma(a2, w2, pA[p2]);
                    pB[p2] = r == 0 ? b2 * w2 : __builtin_elementwise_fma(b2, w2, pB[p2]);
                }
            }
#pragma unroll
            for (int n = 0; n < S; ++n) {
                float lo = (n & 1) ? pA[n / 2].y : pA[n / 2].x, up = (n & 1) ? pB[n / 2].y : pB[n / 2].x;
                permlane32_swap(lo, up); // lane l: lo = the lower half's partial of ITS rollout, up = the upper half's
                const float y = (lo + up) + cst_s[2][n];
                x[n] = x[n] + (y * cst_s[3][n] + cst_s[4][n]);
            }
        };

        for (int t = 0; t < H; ++t) {
            float in[2 * K1H];
#pragma unroll
            for (int i = 0; i < S; ++i) in[i] = x[i];
#pragma unroll
            for (int i = 0; i < A; ++i) in[S + i] = act_s[pair][t & 1][i][lane]; // to_apply of step t (wave C prepared it a step ahead)
            if (2 * K1H > NIN) in[2 * K1H - 1] = 0.0f;
            float ba[K1H], bb[K1H];
#pragma unroll
            for (int s1 = 0; s1 < K1H; ++s1) {
                float ev = (in[2 * s1] - cst_s[0][2 * s1]) * cst_s[1][2 * s1], od = (in[2 * s1 + 1] - cst_s[0][2 * s1 + 1]) * cst_s[1][2 * s1 + 1];
                permlane32_swap(ev, od); // ev: (even, odd) input of rollouts 0..31 in the lane halves; od: of rollouts 32..63
                ba[s1] = ev; bb[s1] = od;
            }
            f32x16 accA, accB;
            mfma32x2_layer1<K1H>(accA, accB, a1, ba, bb, b1t);
            if (n_hidden >= 2) {
                f32x16 hA, hB;
                mfma32x2_hidden_lo_hi(hA, hB, accA, accB, ah[0], bht[0]);
                if (n_hidden >= 3) {
                    mfma32x2_hidden_hi_lo(accA, accB, hA, hB, ah[1], bht[1]);
                    finish(accA, accB);
                } else {
                    finish(hA, hB);
                }
            } else {
                finish(accA, accB);
            }
#pragma unroll
            for (int i = 0; i < S; ++i) xs_s[pair][t & 1][i][lane] = x[i];
            __syncthreads(); // step t handed over
        }
    } else {
        // ================================================================================= wave C: cost, noise
        PcConsumerConsts<S> ccst; // goal, Q: a kernel-local copy (no constant re-fetch behind the per-step barrier)
        ccst.load(C);
        PcProducerConsts<A> pcst;
        pcst.template load<DIAG>(C);
        const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
        const unsigned long long seed = C->seed;
        const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
        float c = 0.0f, z[4 * A];
        auto prepare = [&](int t) { // v = u + eps and the action cost of step t -> act_s[t & 1]
            float e[A], u[A];
            if (SRC == SRC_PHILOX) {
                if ((t & 3) == 0) normals_group<A>(seed, gk, base + (unsigned long long)(t >> 2), z);
                float z1[A];
                const int tl = t & 3; // wave-uniform: four statically indexed copies instead of a dynamically indexed register array
                if (tl == 0) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[i]; }
                else if (tl == 1) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[A + i]; }
                else if (tl == 2) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[2 * A + i]; }
                else { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[3 * A + i]; }
                scale_noise<A, DIAG>(&pcst, z1, e);
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
            }
#pragma unroll
            for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; act_s[pair][t & 1][i][lane] = u[i] + e[i]; }
            act_s[pair][t & 1][A][lane] = action_cost<A, DIAG>(&pcst, u, e);
        };
        prepare(0);
        __syncthreads(); // w3_s, cst_s, and the perturbed action of step 0
        float x[S];
        for (int t = 0; t < H; ++t) {
            if (t >= 1) { // the state step t-1 produced
#pragma unroll
                for (int i = 0; i < S; ++i) x[i] = xs_s[pair][(t - 1) & 1][i][lane];
                const float sc = state_cost<S, false>(&ccst, x);                 // cost on the POST-step state
                const float step_c = sc + act_s[pair][(t - 1) & 1][A][lane];     // Step_cost_result cost_base.cpp:49
                c = c + step_c;                                                  // path_cost        controller_base.cpp:268
            }
            if (t + 1 < H) prepare(t + 1);
            __syncthreads(); // step t handed over
        }
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = xs_s[pair][(H - 1) & 1][i][lane];
        const float sc = state_cost<S, false>(&ccst, x);
        c = c + (sc + act_s[pair][(H - 1) & 1][A][lane]);
        c = c + sc; // terminal cost: x_H counted a second time, controller_base.cpp:271-272
        cost_s[pair][lane] = c;
        if (valid) cost[k0 + lane] = c;
    }
    __syncthreads();
    if (MODE == MODE_COST_ONLY || !tile_ok) return;
    const float ct = cost_s[pair][lane];
    mlp_tile_record<A, DIAG, 2>(C, ct, valid, role, lane, kk, H, NG, SRC, eps_hbm, C->seed, (unsigned long long)C->k_offset + (unsigned long long)kk,
                                step_ctr[0] * (unsigned long long)NG, partials + (size_t)record_slot(tile, rsc) * rsb, rsc);
}

} // namespace mppi
