// mppi_mlp_small.hip.h — k_rollout_mlp_small: learned model_base with the reference's own network shapes,
// Dense(HID, relu) x 1..3 + Dense(s_dim), HID = 16 or 32 (nn_model.py:54-60: three hidden layers of 32; :307-330).
// Included by mppi_kernels.hip.h.
//
// At these widths a 32x32 MFMA tile would be mostly padding and the f32-input MFMA has no rate advantage over packed
// vector math on gfx950 (both 64 FLOP/clk/SIMD, and they share the pipe: tools/micro/mfma_f32_shadow.hip). So:
//   * one rollout per LANE, a wave = a 64-rollout tile (one record), activations in registers as output PAIRS;
//   * the weights are wave-uniform: they come through the scalar cache (kernel-argument pointers, so hipcc emits
//     s_load_dwordx16) and enter v_pk_fma_f32 as an SGPR pair — no LDS, no barrier, no vector loads in the step:
//         (h[2o], h[2o+1]) += in_i * (W[i][2o], W[i][2o+1])        i ascending, as orc_mlp_step sums
//     A {32,32,32,s} step is 1264 v_pk_fma per wave (2528 weights = 10 KB of scalar-cache traffic);
//   * noise, costs, state update and the tile record are the point-mass tile kernel's helpers, lane-local.
// Bound: vector issue (v_pk_fma_f32 at ~4.2 cycles per wave-instruction = 61 FLOP/clk/SIMD).
#pragma once

namespace mppi {

constexpr int kMlpSmallMaxLayers = 4;

struct MlpSmallArgs { // by value in the kernel-argument segment: uniform, read with scalar loads
    const float *W[kMlpSmallMaxLayers];
    const float *b[kMlpSmallMaxLayers];
    int n_layers; // Dense layers in all: n_layers - 1 hidden (relu) + the linear output layer
};

typedef float f32x2s __attribute__((ext_vector_type(2)));

// one Dense layer on output pairs: out[o2] = b + sum_i in[i] * W[i][2 o2 .. 2 o2 + 1], i ascending
template <int IN, int OUT2, bool RELU>
__device__ __forceinline__ void dense_pairs(const float *__restrict__ W, const float *__restrict__ b, const float (&in)[IN],
                                            f32x2s (&out)[OUT2])
{
    constexpr int OUT = 2 * OUT2;
#pragma unroll
    for (int i = 0; i < IN; ++i) {
        const f32x2s x2 = {in[i], in[i]};
#pragma unroll
        for (int o2 = 0; o2 < OUT2; ++o2) {
            const f32x2s w2 = {W[i * OUT + 2 * o2], W[i * OUT + 2 * o2 + 1]};
            out[o2] = i == 0 ? x2 * w2 : __builtin_elementwise_fma(x2, w2, out[o2]);
        }
    }
#pragma unroll
    for (int o2 = 0; o2 < OUT2; ++o2) {
        out[o2] = out[o2] + f32x2s{b[2 * o2], b[2 * o2 + 1]};
        if (RELU) { out[o2].x = fmaxf(out[o2].x, 0.0f); out[o2].y = fmaxf(out[o2].y, 0.0f); }
    }
}

template <int A, int HID>
__global__ __launch_bounds__(64) void k_rollout_mlp_small(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const MlpSmallArgs P, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr int S = 2 * A, NIN = S + A, H2 = HID / 2;
    constexpr bool QFULL = false, DIAG = false;
    static_assert(HID % 2 == 0 && S % 2 == 0, "outputs are handled as pairs");
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int lane = threadIdx.x;
    const int k0 = blockIdx.x * 64;
    const bool valid = (k0 + lane) < K;
    const int kk = valid ? k0 + lane : K - 1; // lanes past K recompute the last sample, outside every sum
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
    const int n_hidden = P.n_layers - 1;

    float xm[NIN], xr[NIN];
#pragma unroll
    for (int i = 0; i < NIN; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; }
    float x[S], c = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];

    float z[4 * A];
    for (int t = 0; t < H; ++t) {
        // the weights are re-read every step (10 KB, scalar-cache resident): left loop-invariant, hipcc hoists as many as
        // it can out of the horizon loop and spills the SGPRs through v_writelane / v_readlane. An opaque zero OFFSET
        // (not an opaque pointer: the loads must keep their kernel-argument provenance to stay scalar) pins them here.
        int zoff; // = 0, "computed" from t: not volatile (a volatile asm counts as a memory clobber, and clobbered loads
                  // are not scalarised), not hoistable (its input changes every iteration)
        asm("s_mov_b32 %0, 0" : "=s"(zoff) : "s"(t));
        const float *Wp[kMlpSmallMaxLayers], *bp[kMlpSmallMaxLayers];
#pragma unroll
        for (int l = 0; l < kMlpSmallMaxLayers; ++l) { Wp[l] = P.W[l] + zoff; bp[l] = P.b[l] + zoff; }
        if (SRC == SRC_PHILOX && (t & 3) == 0) normals_group<A>(seed, gk, base + (unsigned long long)(t >> 2), z);
        float u[A], e[A], v[A];
        if (SRC == SRC_PHILOX) {
            float z1[A];
#pragma unroll
            for (int i = 0; i < A; ++i) { // z[(t & 3) * A + i] without a dynamically indexed register array
                float zi = z[i];
#pragma unroll
                for (int tl = 1; tl < 4; ++tl) zi = (t & 3) == tl ? z[tl * A + i] : zi;
                z1[i] = zi;
            }
            scale_noise<A, DIAG>(C, z1, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(C, u, e);

        float in[NIN];
#pragma unroll
        for (int i = 0; i < S; ++i) in[i] = (x[i] - xm[i]) * xr[i];
#pragma unroll
        for (int i = 0; i < A; ++i) in[S + i] = (v[i] - xm[S + i]) * xr[S + i];
        f32x2s ha[H2], hb[H2];
        float hin[HID];
        dense_pairs<NIN, H2, true>(Wp[0], bp[0], in, ha);
        for (int l = 1; l < n_hidden; ++l) { // wave-uniform trip count, the same code for every hidden layer
#pragma unroll
            for (int o2 = 0; o2 < H2; ++o2) { hin[2 * o2] = ha[o2].x; hin[2 * o2 + 1] = ha[o2].y; }
            const float *Wl = l == 1 ? Wp[1] : Wp[2], *bl = l == 1 ? bp[1] : bp[2];
            dense_pairs<HID, H2, true>(Wl, bl, hin, hb);
#pragma unroll
            for (int o2 = 0; o2 < H2; ++o2) ha[o2] = hb[o2];
        }
#pragma unroll
        for (int o2 = 0; o2 < H2; ++o2) { hin[2 * o2] = ha[o2].x; hin[2 * o2 + 1] = ha[o2].y; }
        f32x2s y[S / 2];
        const float *Wo = n_hidden == 1 ? Wp[1] : (n_hidden == 2 ? Wp[2] : Wp[3]);
        const float *bo = n_hidden == 1 ? bp[1] : (n_hidden == 2 ? bp[2] : bp[3]);
        dense_pairs<HID, S / 2, false>(Wo, bo, hin, y);
#pragma unroll
        for (int i = 0; i < S; ++i) {
            const float yi = (i & 1) ? y[i / 2].y : y[i / 2].x;
            x[i] = x[i] + (yi * M->ystd[i] + M->ymean[i]);
        }
        const float sc = state_cost<S, QFULL>(C, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    }
    c = c + state_cost<S, QFULL>(C, x); // terminal cost, controller_base.cpp:271-272
    if (valid) cost[k0 + lane] = c;
    if (MODE == MODE_COST_ONLY) return;
    mlp_tile_record<A, DIAG, 1>(C, c, valid, 0, lane, kk, H, NG, SRC, eps_hbm, seed, gk, base,
                                partials + (size_t)record_slot(blockIdx.x, rsc) * rsb, rsc);
}

} // namespace mppi
