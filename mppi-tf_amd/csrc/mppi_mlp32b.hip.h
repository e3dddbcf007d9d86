// mppi_mlp32b.hip.h — k_rollout_mlp32_bx3: the reference's Dense(32, relu) x 1..3 + Dense(s) network (nn_model.py:54-60) on the
// BF16 matrix cores at fp32-class accuracy (opt-in MPPI_FLAG_MLP_BF16X3, as for the 2x256 network). Included by mppi_kernels.hip.h.
//
// Why a second Dense(32) kernel: k_rollout_mlp32's exact-fp32 MFMA shares the pipe with the vector ALU — 37 MFMAs of
// 64 cycles AND ~300 vector instructions per step, one after the other. The bf16 MFMA is a real matrix-core instruction: 32 cycles
// for 8x the k depth, and the vector instructions of the SIMD's other wave run beside it. With every fp32 operand split x = hi + lo
// (two bf16) and three products per term (a_lo b_hi + a_hi b_lo + a_hi b_hi, fp32 accumulate) a 32-wide layer is 2 k-blocks x 3 = 6
// MFMAs (192 cycles instead of 1024), and the whole network — the output layer too — runs on the matrix core:
//   * as in k_rollout_mlp32 a layer's accumulator registers are the next layer's B operand: registers 8 kb .. 8 kb + 7 of lane
//     (j, hh), relu'd and split, ARE the B fragment of k-block kb (hidden units 16 kb + 8 (e >> 2) + 4 hh + (e & 3), e = 0..7 — the
//     order the stationary A fragments are loaded in); no LDS, no shuffle, no barrier between layers;
//   * output layer: A rows m < 16 carry output (m & 3) + 4 (m >> 3) — every output twice, once per lane half — so that after the
//     MFMAs register n of EVERY lane is output n of its rollout: no lane-half exchange, no W3 in LDS;
//   * b1 rides as input k = s + a against a constant 1; the other biases are the C operand of a layer's first MFMA;
//   * the MFMAs are the compiler's builtins here (nothing has to be pinned to the accumulation registers: 56 weight + 48 bias
//     registers), so hipcc sees every hazard itself.
// One wave = 32 rollouts (both lane halves carry rollout j's state), a workgroup = 2 waves = one 64-rollout tile record.
#pragma once

namespace mppi {

template <int A>
__global__ __launch_bounds__(kMlp32Threads) void k_rollout_mlp32_bx3(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr int S = 2 * A, NIN = S + A, HID = 32;
    static_assert(NIN + 1 <= 16 && S <= 8, "inputs + bias fit one k-block; outputs fit the 8 registers both lane halves share");
    constexpr bool QFULL = false, DIAG = false;
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    __shared__ float z_s[2][4 * A][32]; // per wave: the normals of one horizon group
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
    auto frag = [](const int (&v)[4]) { return __builtin_bit_cast(bf16x8, i32x4{v[0], v[1], v[2], v[3]}); };
    auto k_unit = [&](int kb, int e) { return 16 * kb + 8 * (e >> 2) + 4 * hh + (e & 3); }; // hidden unit behind k slot 8 hh + e of k-block kb
    auto row_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * hh; };                     // accumulator register r of this lane half

    // ---- stationary operands (hi, lo)
    bf16x8 a1h, a1l, ahh[2][2], ahl[2][2], aoh[2], aol[2];
    f32x16 bht[2], bot;
    {
        int hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) { // layer 1: natural k order 8 hh + e; k = NIN is the bias row
            float v2[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int k = 8 * hh + 2 * q + o;
                v2[o] = k < NIN ? M->Wl[0][k * HID + j] : (k == NIN ? M->bl[0][j] : 0.0f);
            }
            split_pair(v2[0], v2[1], hi[q], lo[q]);
        }
        a1h = frag(hi); a1l = frag(lo);
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const bool have = l + 2 <= n_hidden; // hidden-to-hidden layer l exists
            const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    split_pair(have ? Wl[k_unit(kb, 2 * q) * HID + j] : 0.0f, have ? Wl[k_unit(kb, 2 * q + 1) * HID + j] : 0.0f, hi[q], lo[q]);
                ahh[l][kb] = frag(hi); ahl[l][kb] = frag(lo);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) bht[l][r] = have ? bl[row_of(r)] : 0.0f;
        }
        const float *W3g = M->Wl[n_hidden], *b3g = M->bl[n_hidden];
        const int out_j = j < 16 ? (j & 3) + 4 * (j >> 3) : S; // the output this lane's A row carries (S: none)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                split_pair(out_j < S ? W3g[k_unit(kb, 2 * q) * S + (out_j < S ? out_j : 0)] : 0.0f,
                           out_j < S ? W3g[k_unit(kb, 2 * q + 1) * S + (out_j < S ? out_j : 0)] : 0.0f, hi[q], lo[q]);
            aoh[kb] = frag(hi); aol[kb] = frag(lo);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) { // row (r & 3) + 8 (r >> 2) + 4 hh carries output (r & 3) + 4 (r >> 2) for r < 8
            const int out_r = (r & 3) + 4 * (r >> 2);
            bot[r] = (r < 8 && out_r < S) ? b3g[out_r < S ? out_r : 0] : 0.0f;
        }
    }
    // the lane's 8 layer-1 inputs are k = 8 hh + e: mean / reciprocal deviation of THOSE inputs; the bias input is (1 - 0) * 1
    float xms[8], xrs[8], ysd[S], ymn[S];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m0 = e < NIN ? M->xmean[e < NIN ? e : 0] : 0.0f, r0 = e < NIN ? 1.0f / M->xstd[e < NIN ? e : 0] : 1.0f;
        const float m1 = 8 + e < NIN ? M->xmean[8 + e < NIN ? 8 + e : 0] : 0.0f, r1 = 8 + e < NIN ? 1.0f / M->xstd[8 + e < NIN ? 8 + e : 0] : 1.0f;
        xms[e] = hh ? m1 : m0;
        xrs[e] = hh ? r1 : r0;
    }
#pragma unroll
    for (int i = 0; i < S; ++i) { ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    float x[S], c = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];
    // wave-uniform noise / cost constants as local copies: read once, SGPR-resident (through the DevConsts pointer they are
    // re-fetched with scalar loads every step)
    PcProducerConsts<A> pcst;
    pcst.template load<DIAG>(C);
    const PcProducerConsts<A> *PC = &pcst;
    PcConsumerConsts<S> ccst;
    ccst.load(C);
    const PcConsumerConsts<S> *CC = &ccst;
    __syncthreads();

    // relu + split of a layer's 16 accumulator registers = the next layer's two B fragments
    auto next_b = [&](const f32x16 &acc, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            int hi[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                // relu as ONE instruction hipcc can see (fmaxf costs a canonicalising second v_max; an inline-asm v_max is invisible to its
                // hazard recognizer, which then lets it read the builtin MFMA's result too early — measured: costs off by 4e-3)
                const float ra = __builtin_amdgcn_fmed3f(acc[8 * kb + 2 * q], 0.0f, __builtin_inff());
                const float rb = __builtin_amdgcn_fmed3f(acc[8 * kb + 2 * q + 1], 0.0f, __builtin_inff());
                split_pair(ra, rb, hi[q], lo[q]);
            }
            bh[kb] = frag(hi); bl[kb] = frag(lo);
        }
    };
    auto layer = [&](const bf16x8 (&wh)[2], const bf16x8 (&wl)[2], const bf16x8 (&bh)[2], const bf16x8 (&bl)[2], f32x16 acc) {
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wl[kb], bh[kb], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kb], bl[kb], acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(wh[kb], bh[kb], acc, 0, 0, 0);
        }
        return acc;
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
            scale_noise<A, DIAG>(PC, z1, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(PC, u, e);
        // layer 1: the B fragment of lane (j, hh) is inputs 8 hh .. 8 hh + 7 of rollout j
        auto raw = [&](int k) { return k < S ? x[k < S ? k : 0] : (k < NIN ? v[k < NIN && k >= S ? k - S : 0] : (k == NIN ? 1.0f : 0.0f)); };
        int ih[4], il[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float in2[2];
#pragma unroll
            for (int o = 0; o < 2; ++o) {
                const int ee = 2 * q + o;
                const float sel = hh ? raw(8 + ee) : raw(ee);
                in2[o] = (sel - xms[ee]) * xrs[ee];
            }
            split_pair(in2[0], in2[1], ih[q], il[q]);
        }
        const bf16x8 inh = frag(ih), inl = frag(il);
        f32x16 acc = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1l, inh, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, inl, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1h, inh, acc, 0, 0, 0);
        bf16x8 bh[2], bl[2];
        if (n_hidden >= 2) {
            next_b(acc, bh, bl);
            acc = layer(ahh[0], ahl[0], bh, bl, bht[0]);
            if (n_hidden >= 3) {
                next_b(acc, bh, bl);
                acc = layer(ahh[1], ahl[1], bh, bl, bht[1]);
            }
        }
        next_b(acc, bh, bl);
        acc = layer(aoh, aol, bh, bl, bot); // register n (< 8) of every lane: output n of rollout j, bias included
#pragma unroll
        for (int n = 0; n < S; ++n) x[n] = x[n] + (acc[n] * ysd[n] + ymn[n]);
        const float sc = state_cost<S, QFULL>(CC, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    }
    c = c + state_cost<S, QFULL>(CC, x); // terminal cost, controller_base.cpp:271-272
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

} // namespace mppi
