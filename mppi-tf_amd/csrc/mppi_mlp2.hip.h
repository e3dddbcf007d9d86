// mppi_mlp2.hip.h — k_rollout_mlp2: the learned 2x256 MLP model_base in exact fp32 on the matrix cores, second design
// (BASELINE configs[3]/[4]). Included by mppi_kernels.hip.h.
//
// The fact this kernel is built around (tools/micro/mfma_f32_shadow.hip, profiles/r02_mfma_f32_shadow.json): the
// f32-input MFMA runs at the f32 vector rate and does NOT overlap with the vector ALU. Beside v_mfma_f32_32x32x2_f32
// (64 cycles) every vector instruction of the same wave adds its own 4-5 cycles, every [MFMA, MFMA, vector lump]
// excursion ~18 more, and inside a lump every instruction of the wave — an s_waitcnt, an s_nop, a ds_read — costs an
// issue slot like a v_fma; only BETWEEN back-to-back MFMAs are LDS, scalar and wait instructions free. So the bound of
// this path is (MFMA cycles + vector cycles), not the MFMA peak alone, and the design minimises vector instructions and
// excursions instead of trying to hide them:
//   * ONE wave per SIMD (4 waves, 256 threads, the unified 512-entry register file): wave w owns hidden units
//     [64w, 64w+64) of both layers = two 32-row M-tiles = 2 x 128 stationary W2 registers = exactly the accumulator
//     half (a0-a255), pinned there through inline-asm MFMAs — hipcc left alone keeps them in v0-v255 and spills; the
//     layer-2 bias is the accumulators' initial value (8 LDS reads per set and step) instead of a 129th k pair;
//   * TWO sets of 32 rollouts (one 32-column MFMA tile each: 2 x 16 accumulator registers per set), each with its own
//     h1 image in LDS (2 x 32 KB), software-pipelined against each other: while the 258 layer-2 MFMAs of one set
//     stream, the other set's step is finished and its next one prepared in ~13 LUMPS of vector work (8 of layer 3 at
//     4 accumulator registers each, partial-sum exchange, state update + costs + next inputs, relu) and 10 layer-1
//     MFMAs. Both workgroup barriers of a step fall in mid-stream with the next B operands already requested;
//   * packed math where it halves the count (v_pk_fma_f32 costs one v_fma_f32 here): layer 3 is 1 v_max + 3 v_pk_fma
//     per accumulator register; output pairs travel through LDS side by side (one 8-byte read per pair);
//   * LDS requests trickle (a few per k pair, issued between the two MFMAs and the lump — behind a lump each would
//     hold the next MFMA back), and hipcc is made to wait for a lump's operands one k pair early;
//   * the workgroup is persistent: it walks over tiles blockIdx.x, + gridDim.x, ... with the weights loaded once.
// History (r02, K=65536 H=64, fp32 peak 157.3): k_rollout_mlp 122 TFLOP/s (0.77; 8 waves in lock-step phases) -> this
// design with 64-rollout sets 108 (weight spills) -> 32-rollout sets, work spread one piece per MFMA "to hide it" 120
// (the opposite of what the hardware wants: 258 excursions) -> lumps 131 -> waits/requests out of the lumps 135 ->
// one-statement swaps, 4-register lumps 137-138 -> bias as accumulator init 141 (harness; 143 in bench.py's timing). Measured floor of the MFMAs alone: 156 (tools/micro/mlp2_bench.hip,
// ablation 2047).
// Arithmetic is k_rollout_mlp's up to the reciprocal (multiplication by 1/sigma; v_mfma_f32_32x32x2_f32 = a k-ordered
// fmaf chain; the cross-wave sum of layer 3 has 4 terms instead of 8): the parity tests hold it to the same bars.
// LDS at a_dim = 3: 2 x 32 KB images + partial sums 6 KB + W3 8 KB + noise 6 KB + W1 10 KB + U = ~96 KB.
#pragma once

// Timing-only ablations for tools/micro/mlp2_bench.hip (results are wrong with any bit set; the product builds with 0):
// 1 no layer 3 / state update of the other set, 2 no preparation (inputs, layer 1, relu, image) of the other set,
// 4 no mid-stream barriers, 8 no noise generation; single lumps: 16 layer 3, 32 partial-sum store, 64 state update and
// cost, 128 inputs of the next step, 256 layer 1, 512 relu, 1024 the LDS requests of the lumps,
// 2048 h1 image writes.
#ifndef MPPI_MLP2_ABL
#define MPPI_MLP2_ABL 0
#endif

namespace mppi {

// (MPPI_MLP2_STAMP_AT / MPPI_MLP2_TRACE_*: the timing-study layer, mppi_ablate.hip.h — nothing in the shipped build)

constexpr int kMlp2Threads = 256;
constexpr int kMlp2R = 64; // rollouts per workgroup: two sets of 32
__host__ __device__ inline size_t mlp2_lds_floats(int S, int A, int H)
{
    return (size_t)2 * kHid * 32 + 2 * 4 * S * 32 + kHid * 8 + 2 * 2 * 4 * A * 32 + (size_t)((S + A + 2) / 2 * 2) * kHid +
           (size_t)(H * A + 3) / 4 * 4 + kHid + 64;
}

// lanes 32-63 of a <-> lanes 0-31 of b (v_permlane32_swap; asm: see the transposing butterfly in mppi_device.hip.h)
__device__ __forceinline__ void permlane32_swap(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}
// the odd 16-lane rows of a <-> the even 16-lane rows of b (v_permlane16_swap)
__device__ __forceinline__ void permlane16_swap(float &a, float &b)
{
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
}

template <int A, bool DIAG, int SRC>
__global__ __launch_bounds__(kMlp2Threads, 1) void k_rollout_mlp2(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int MODE, const int rsb, const int rsc)
{
    constexpr bool QFULL = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * A, NIN = S + A;
    constexpr int K1 = (NIN + 2) / 2 * 2; // inputs + bias, padded to the MFMA's k pairs (10 for s=6, a=3)
    constexpr int NKP = kHid / 2;         // k pairs of layer 2 (its bias is the accumulators' initial value, read from LDS)
    constexpr int R = 32;                 // rollouts of a set = columns of one MFMA tile
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    float *h1_s = smem;                        // [2 sets][kHid][R]
    float *y_s = h1_s + 2 * kHid * R;          // [2 sets][4 waves][S/2][R][2]: output pairs (2p, 2p+1) of a rollout adjacent
    float *w3_s = y_s + 2 * 4 * S * R;         // [kHid][8]: rows padded to two 16-byte reads
    float *z_s = w3_s + kHid * 8;              // [2 sets][2 buffers][4*A][R] standard normals of a horizon group
    float *w1_s = z_s + 2 * 2 * 4 * A * R;     // [K1][kHid]: W1, the b1 row, zero padding
    float *b2_s = w1_s + K1 * kHid;            // [kHid] layer-2 bias
    float *u_s = b2_s + kHid;                  // [H*A] the nominal controls (LDS, not s_load: a scalar load's return
                                               // is waited for with lgkmcnt(0), which would drain the B-operand reads)

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
    bool tile_is_first = true; (void)tile_is_first;
    int k0 = 0; // first rollout of the tile at hand (the workgroup walks over tiles blockIdx.x, + gridDim.x, ...)

    // ---- stationary W2 -> registers: a2[mt][s2] = W2[2 s2 + hh][64 w + 32 mt + j]
    // (W2 only: the ten layer-1 weights per lane are used once per half-step and are re-read from LDS a few slots
    // ahead of their use)
    float a2[2][NKP];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int unit = 64 * w + 32 * mt + j;
#pragma unroll
        for (int s2 = 0; s2 < kHid / 2; ++s2) a2[mt][s2] = M->W2[(size_t)(2 * s2 + hh) * kHid + unit];
    }
    for (int i = tid; i < K1 * kHid; i += kMlp2Threads) {
        const int kin = i / kHid, unit = i % kHid;
        w1_s[i] = kin < NIN ? M->W1[(size_t)kin * kHid + unit] : (kin == NIN ? M->b1[unit] : 0.0f);
    }
    static_assert(S <= 8, "W3 rows are staged as 8 floats");
    for (int i = tid; i < kHid * 8; i += kMlp2Threads) w3_s[i] = (i & 7) < S ? M->W3[(i >> 3) * S + (i & 7)] : 0.0f;
    for (int i = tid; i < HA; i += kMlp2Threads) u_s[i] = U_dev[i];
    for (int i = tid; i < kHid; i += kMlp2Threads) b2_s[i] = M->b2[i];

    // wave-uniform constants, read once (a barrier would otherwise force a re-fetch per step)
    float xm[NIN], xr[NIN], b3v[S], ysd[S], ymn[S];
#pragma unroll
    for (int i = 0; i < NIN; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; }
#pragma unroll
    for (int i = 0; i < S; ++i) { b3v[i] = M->b3[i]; ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    PcProducerConsts<A> pcst;
    pcst.template load<DIAG>(C);
    const PcProducerConsts<A> *PC = &pcst;
    PcConsumerConsts<S> ccst;
    ccst.load(C);
    const PcConsumerConsts<S> *CC = &ccst;
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned long long koff = (unsigned long long)C->k_offset;
    // sample index (within this handle's shard) of rollout j of set q, clamped: columns past K recompute the last sample
    auto kk_of = [&](int q) { return min(k0 + R * q + j, K - 1); };

    // per-lane state of rollout j of BOTH sets (replicated in the two lane halves and the 4 waves: every lane needs the
    // layer-1 input of its column)
    float xA[S], xB[S], x0[S], cA = 0.0f, cB = 0.0f, acA = 0.0f, acB = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x0[i] = x_dev[i];
    f32x16 accA[2], accB[2]; // [mt]: M-tile mt x the set's 32 columns
    // LDS indices (in floats) with everything but a compile-time constant in ONE register per array; the constant goes
    // into the DS instruction's 16-bit offset. Opaque to hipcc on purpose (left alone it precomputes every address as a
    // loop invariant and spills them), and kept as INDICES into smem (an opaque pointer would lose its LDS address
    // space and turn into flat accesses). Accumulator register r of M-tile mt is row 64 w + 32 mt + (r & 3) + 8 (r >> 2)
    // + 4 hh of the layer.
    int img_row0[2] = {(0 * kHid + 64 * w + 4 * hh) * R + j, (1 * kHid + 64 * w + 4 * hh) * R + j};
    int w3_row0 = (int)(w3_s - smem) + (64 * w + 4 * hh) * 8;
    int w1_row0 = (int)(w1_s - smem) + hh * kHid + 64 * w + j;
    int y_wr0[2] = {(int)(y_s - smem) + (0 * 4 + w) * S * R + 2 * j + hh, (int)(y_s - smem) + (1 * 4 + w) * S * R + 2 * j + hh};
    int y_rd0 = (int)(y_s - smem) + 2 * j;
    int z_rd0 = (int)(z_s - smem) + j;
    int b_rd0 = hh * R + j;
    int b2_rd0 = (int)(b2_s - smem) + 64 * w + 4 * hh;
    asm volatile("" : "+v"(img_row0[0]), "+v"(img_row0[1]), "+v"(w3_row0), "+v"(w1_row0), "+v"(y_wr0[0]), "+v"(y_wr0[1]),
                 "+v"(y_rd0), "+v"(z_rd0), "+v"(b_rd0), "+v"(b2_rd0));
    constexpr auto crow_of = [](int pi) { return 32 * (pi >> 4) + (pi & 3) + 8 * ((pi & 15) >> 2); };

    // One v_mfma_f32_32x32x2_f32 with the stationary weight pinned to the accumulator half of the register file ("a"):
    // left to itself hipcc keeps the weights in v0-v255 with everything else, runs out, and reloads ~30 of them from
    // scratch in front of their MFMAs; through the intrinsic it also moves the accumulators into a0-a63. Hazards inside
    // the string: s_nop 1 (layer 1 and the bias pair only) covers a B operand written by a vector instruction just before; the
    // accumulate chain (D -> same C) needs none; every VALU reader of an accumulator is several MFMAs downstream of the
    // last write (the prologue and the epilogue pad for themselves).
    auto mfma_acc = [&](f32x16 &acc, float a, float b, auto in_agpr, auto first) {
        if constexpr (decltype(in_agpr)::value) { // layer 2: B comes from LDS (hipcc waits for it), no pad needed
            if constexpr (decltype(first)::value) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "a"(a), "v"(b));
            else asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
        } else {
            if constexpr (decltype(first)::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
            else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
        }
    };
    using std::integral_constant;
    constexpr integral_constant<bool, true> yes{};
    constexpr integral_constant<bool, false> no{};
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    // 8- and 16-byte LDS reads at smem[idx]: clang derives the alignment of a cast float* from the array it points
    // into (4), and the backend then splits the read into b32 pairs with one address register each
    auto lds2 = [&](int idx) { return *static_cast<const f32x2 *>(__builtin_assume_aligned(smem + idx, 8)); };
    auto lds4 = [&](int idx) { return *static_cast<const f32x4 *>(__builtin_assume_aligned(smem + idx, 16)); };
    constexpr int SP = S / 2; // S = 2 A is even: outputs are handled as pairs (v_pk_*_f32 costs what one v_fma_f32 does here)

    // ---------------------------------------------------------------------------------------------------------------
    // The work of one step of a set besides its layer 2, as a few LUMPS of vector instructions: see the header for why
    // (nothing vector hides behind an f32 MFMA, and every MFMA -> VALU -> MFMA excursion costs ~9 cycles on top).
    // LDS instructions are free beside the MFMAs: every lump's operands are requested one or two k pairs ahead.
    struct StepRegs {           // values that live across lumps
        f32x2 py[SP];           // layer-3 partial sums of this lane's rows
        f32x2 wrow[8][SP];      // W3 rows in flight (ring of 8 accumulator registers)
        f32x2 yv[SP][4];        // partial sums of the 4 waves
        float zz[A], u[A], e[A], v[A];
        float a1[2][K1 / 2], b1[K1 / 2];
    };
    // layer 3, accumulator register pi: request its W3 row (16-byte / 8-byte reads at base + constant)
    auto l3_request = [&](auto pic, StepRegs &g) {
        constexpr int pi = decltype(pic)::value, r = pi & 7;
        const int idx = w3_row0 + crow_of(pi) * 8;
        if constexpr (SP >= 2) {
            const f32x4 lo = lds4(idx);
            g.wrow[r][0] = lo.xy;
            g.wrow[r][1] = lo.zw;
        } else {
            g.wrow[r][0] = lds2(idx);
        }
        if constexpr (SP >= 3) g.wrow[r][2] = lds2(idx + 4);
    };
    // py[n] += relu(h2) * W3[row][n] for accumulator registers pi0 .. pi0 + 3: 4 v_max + 4 SP v_pk_fma
    auto l3_lump = [&](auto qc, auto pi0c, StepRegs &g) {
        constexpr int q = decltype(qc)::value, pi0 = decltype(pi0c)::value;
        f32x16 (&acc)[2] = q ? accB : accA;
        float hv[4]; // the relus first: a v_pk_* that reads the register just written needs a wait state
#pragma unroll
        for (int i = 0; i < 4; ++i) asm("v_max_f32 %0, 0, %1" : "=v"(hv[i]) : "v"(acc[(pi0 + i) >> 4][(pi0 + i) & 15])); // fmaxf costs a canonicalising second v_max
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int pi = pi0 + i;
            const f32x2 h2 = {hv[i], hv[i]};
#pragma unroll
            for (int p2 = 0; p2 < SP; ++p2) {
                if (pi == 0) g.py[p2] = h2 * g.wrow[pi & 7][p2];
                else g.py[p2] = __builtin_elementwise_fma(h2, g.wrow[pi & 7][p2], g.py[p2]);
            }
        }
    };
    // Inside a lump EVERY instruction of the wave costs an issue slot (~5 cycles at one wave per SIMD): an s_waitcnt
    // there is as dear as a v_pk_fma. These make hipcc wait for a lump's LDS operands one k pair early, in front of
    // the MFMAs, where the wait is free.
    auto l3_arrived = [&](auto pi0c, StepRegs &g) {
        constexpr int pi0 = decltype(pi0c)::value;
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int p2 = 0; p2 < SP; ++p2) asm volatile("" : "+v"(g.wrow[(pi0 + i) & 7][p2]));
    };
    auto fin_arrived = [&](StepRegs &g) {
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2)
#pragma unroll
            for (int ww = 0; ww < 4; ++ww) asm volatile("" : "+v"(g.yv[p2][ww]));
    };
    auto prep_arrived = [&](StepRegs &g) {
#pragma unroll
        for (int i = 0; i < A; ++i) {
            if constexpr (SRC == SRC_PHILOX) asm volatile("" : "+v"(g.zz[i]));
            else asm volatile("" : "+v"(g.e[i]));
            asm volatile("" : "+v"(g.u[i]));
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int s1 = 0; s1 < K1 / 2; ++s1) asm volatile("" : "+v"(g.a1[mt][s1]));
    };
    // the lane halves hold different rows of the same column. One half-swap of (py[n], py[n+1]) and one add leave the
    // total of output n in the lower half and of output n+1 in the upper half: one 64-lane store writes both.
    auto l3_store = [&](auto qc, StepRegs &g) {
        constexpr int q = decltype(qc)::value;
        float t[SP];
        // a = {py[n].lower, py[n+1].lower}, b = {py[n].upper, py[n+1].upper} after the swap; all swaps behind ONE pad
        // (vector write -> v_permlane*: 2 wait states), the adds far enough behind them to need none
        if constexpr (SP == 3) {
            asm("s_nop 1\n\tv_permlane32_swap_b32 %3, %4\n\tv_permlane32_swap_b32 %5, %6\n\tv_permlane32_swap_b32 %7, %8\n\t"
                "v_add_f32 %0, %3, %4\n\tv_add_f32 %1, %5, %6\n\tv_add_f32 %2, %7, %8"
                : "=&v"(t[0]), "=&v"(t[1]), "=&v"(t[2]), "+v"(g.py[0].x), "+v"(g.py[0].y), "+v"(g.py[1].x), "+v"(g.py[1].y),
                  "+v"(g.py[2].x), "+v"(g.py[2].y));
        } else {
#pragma unroll
            for (int p2 = 0; p2 < SP; ++p2) {
                float a = g.py[p2].x, b = g.py[p2].y;
                permlane32_swap(a, b);
                t[p2] = a + b;
            }
        }
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2) smem[y_wr0[q] + p2 * 2 * R] = t[p2];
    };
    // F: request the partial sums of the 4 waves (piece i of 2 SP: two 8-byte reads); then y = their sum (fixed order)
    // + b3, state update, cost of the step
    auto fin_request = [&](auto qc, auto ic, StepRegs &g) {
        constexpr int q = decltype(qc)::value, i = decltype(ic)::value, p2 = i >> 1;
#pragma unroll
        for (int ww = 2 * (i & 1); ww < 2 * (i & 1) + 2; ++ww) g.yv[p2][ww] = lds2(y_rd0 + ((q * 4 + ww) * SP + p2) * 2 * R);
    };
    auto fin_lump = [&](auto qc, StepRegs &g) {
        constexpr int q = decltype(qc)::value;
        float (&x)[S] = q ? xB : xA;
        float &c = q ? cB : cA;
        const float ac = q ? acB : acA;
        f32x2 y[SP]; // stage by stage over the SP independent chains: a v_pk_* reading the previous one's result stalls
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2) y[p2] = g.yv[p2][0] + g.yv[p2][1];
#pragma unroll
        for (int ww = 2; ww < 4; ++ww)
#pragma unroll
            for (int p2 = 0; p2 < SP; ++p2) y[p2] = y[p2] + g.yv[p2][ww];
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2) y[p2] = y[p2] + f32x2{b3v[2 * p2], b3v[2 * p2 + 1]};
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2) y[p2] = y[p2] * f32x2{ysd[2 * p2], ysd[2 * p2 + 1]};
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2) y[p2] = y[p2] + f32x2{ymn[2 * p2], ymn[2 * p2 + 1]};
#pragma unroll
        for (int p2 = 0; p2 < SP; ++p2) {
            const f32x2 xn = f32x2{x[2 * p2], x[2 * p2 + 1]} + y[p2];
            x[2 * p2] = xn.x; x[2 * p2 + 1] = xn.y;
        }
        const float sc = state_cost<S, QFULL>(CC, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    };
    // P: request noise and nominal control of step t; the layer-1 weights one by one; then the inputs of layer 1
    auto prep_request = [&](auto qc, int t, StepRegs &g) {
        constexpr int q = decltype(qc)::value;
        if constexpr (SRC == SRC_PHILOX) {
#pragma unroll
            for (int i = 0; i < A; ++i) g.zz[i] = smem[z_rd0 + ((q * 2 + ((t >> 2) & 1)) * 4 * A + (t & 3) * A + i) * R];
        } else {
            const int kk = kk_of(q);
#pragma unroll
            for (int i = 0; i < A; ++i) g.e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) g.u[i] = u_s[t * A + i];
    };
    auto a1_request = [&](auto ic, StepRegs &g) {
        constexpr int i = decltype(ic)::value, mt = i / (K1 / 2), s1 = i % (K1 / 2);
        g.a1[mt][s1] = smem[w1_row0 + 2 * s1 * kHid + 32 * mt];
    };
    auto prep_lump = [&](auto qc, StepRegs &g) {
        constexpr int q = decltype(qc)::value;
        float (&x)[S] = q ? xB : xA;
        if constexpr (SRC == SRC_PHILOX) scale_noise<A, DIAG>(PC, g.zz, g.e);
#pragma unroll
        for (int i = 0; i < A; ++i) g.v[i] = g.u[i] + g.e[i];
        (q ? acB : acA) = action_cost<A, DIAG>(PC, g.u, g.e);
        // normalised inputs (multiplication by 1/sigma: this path is tolerance-bound, not order-bound), as the k pairs of
        // layer 1: (in[2 s1], in[2 s1 + 1]); the bias input is 1 = (1 - 0) * 1. Layer 1's B operand of lane (j, hh) is
        // in[2 s1 + hh] of rollout j — the lane's own value.
        auto raw = [&](int i) { return i < S ? x[i] : (i < NIN ? g.v[i - S] : (i == NIN ? 1.0f : 0.0f)); };
#pragma unroll
        for (int s1 = 0; s1 < K1 / 2; ++s1) {
            const int i0 = 2 * s1, i1 = 2 * s1 + 1;
            const f32x2 r2 = {raw(i0), raw(i1)};
            const f32x2 m2 = {i0 < NIN ? xm[i0] : 0.0f, i1 < NIN ? xm[i1] : 0.0f};
            const f32x2 s2 = {i0 < NIN ? xr[i0] : 1.0f, i1 < NIN ? xr[i1] : 1.0f};
            const f32x2 in2 = (r2 - m2) * s2;
            g.b1[s1] = hh ? in2.y : in2.x;
        }
    };
    auto l1_mfmas = [&](auto qc, StepRegs &g) {
        constexpr int q = decltype(qc)::value;
        f32x16 (&acc)[2] = q ? accB : accA;
        static_for<0, K1 / 2>([&](auto s1c) {
            constexpr int s1 = decltype(s1c)::value;
            mfma_acc(acc[0], g.a1[0][s1], g.b1[s1], no, integral_constant<bool, s1 == 0>{});
            mfma_acc(acc[1], g.a1[1][s1], g.b1[s1], no, integral_constant<bool, s1 == 0>{});
        });
    };
    // relu of the set's 32 accumulator registers, in place; then their h1 image writes, two at a time
    auto relu_lump = [&](auto qc) {
        constexpr int q = decltype(qc)::value;
        f32x16 (&acc)[2] = q ? accB : accA;
        asm("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\t"
            "v_max_f32 %4, 0, %4\n\tv_max_f32 %5, 0, %5\n\tv_max_f32 %6, 0, %6\n\tv_max_f32 %7, 0, %7\n\t"
            "v_max_f32 %8, 0, %8\n\tv_max_f32 %9, 0, %9\n\tv_max_f32 %10, 0, %10\n\tv_max_f32 %11, 0, %11\n\t"
            "v_max_f32 %12, 0, %12\n\tv_max_f32 %13, 0, %13\n\tv_max_f32 %14, 0, %14\n\tv_max_f32 %15, 0, %15"
            : "+v"(acc[0][0]), "+v"(acc[0][1]), "+v"(acc[0][2]), "+v"(acc[0][3]), "+v"(acc[0][4]), "+v"(acc[0][5]), "+v"(acc[0][6]),
              "+v"(acc[0][7]), "+v"(acc[0][8]), "+v"(acc[0][9]), "+v"(acc[0][10]), "+v"(acc[0][11]), "+v"(acc[0][12]),
              "+v"(acc[0][13]), "+v"(acc[0][14]), "+v"(acc[0][15]));
        asm("v_max_f32 %0, 0, %0\n\tv_max_f32 %1, 0, %1\n\tv_max_f32 %2, 0, %2\n\tv_max_f32 %3, 0, %3\n\t"
            "v_max_f32 %4, 0, %4\n\tv_max_f32 %5, 0, %5\n\tv_max_f32 %6, 0, %6\n\tv_max_f32 %7, 0, %7\n\t"
            "v_max_f32 %8, 0, %8\n\tv_max_f32 %9, 0, %9\n\tv_max_f32 %10, 0, %10\n\tv_max_f32 %11, 0, %11\n\t"
            "v_max_f32 %12, 0, %12\n\tv_max_f32 %13, 0, %13\n\tv_max_f32 %14, 0, %14\n\tv_max_f32 %15, 0, %15"
            : "+v"(acc[1][0]), "+v"(acc[1][1]), "+v"(acc[1][2]), "+v"(acc[1][3]), "+v"(acc[1][4]), "+v"(acc[1][5]), "+v"(acc[1][6]),
              "+v"(acc[1][7]), "+v"(acc[1][8]), "+v"(acc[1][9]), "+v"(acc[1][10]), "+v"(acc[1][11]), "+v"(acc[1][12]),
              "+v"(acc[1][13]), "+v"(acc[1][14]), "+v"(acc[1][15]));
    };
    auto image_store = [&](auto qc, auto pi0c) {
        constexpr int q = decltype(qc)::value, pi0 = decltype(pi0c)::value;
        f32x16 (&acc)[2] = q ? accB : accA;
#pragma unroll
        for (int pi = pi0; pi < pi0 + 2; ++pi) smem[img_row0[q] + crow_of(pi) * R] = acc[pi >> 4][pi & 15];
    };
    // layer 2's bias as the initial value of the set's accumulators (accumulator register r of M-tile mt is row
    // 32 mt + 8 (r >> 2) + 4 hh + (r & 3) of the wave's 64: four consecutive rows per 16-byte read), piece i of 8
    auto acc_init = [&](auto qc, auto ic) {
        constexpr int q = decltype(qc)::value, i = decltype(ic)::value, mt = i >> 2, g4 = i & 3;
        f32x16 (&acc)[2] = q ? accB : accA;
        const f32x4 b4 = lds4(b2_rd0 + 32 * mt + 8 * g4);
        acc[mt][4 * g4 + 0] = b4.x; acc[mt][4 * g4 + 1] = b4.y; acc[mt][4 * g4 + 2] = b4.z; acc[mt][4 * g4 + 3] = b4.w;
    };
    // The standard normals of horizon group gn for BOTH sets -> buffer gn & 1. One Philox block (4 normals: one action
    // dimension of the group's 4 steps) per lane: unit 2 * wave + hh is block (set, q) = (unit / A, unit % A), so 2 A of
    // the workgroup's 8 half-waves do ~140 vector instructions each instead of one wave doing 4 A blocks in a row.
    auto noise_groups = [&](int gn) {
        const int unit = 2 * w + hh;
        if (2 * w < 2 * A) { // wave-uniform
            const int set = unit / A, q = unit - set * A;
            if (unit < 2 * A) {
                const float4 n = normals_of_block(seed, koff + (unsigned long long)kk_of(set), (base + (unsigned long long)gn) * A + q);
                float *zd = z_s + ((set * 2 + (gn & 1)) * 4 * A + 4 * q) * R + j; // normals 4 q .. 4 q + 3 of normals_group<A>
                zd[0 * R] = n.x; zd[1 * R] = n.y; zd[2 * R] = n.z; zd[3 * R] = n.w;
            }
        }
    };

    // One half-step: the layer-2 MFMAs of set Q stream; between them the OTHER set O finishes its last step (FIN) and
    // prepares step t_prep up to its h1 image (PREP): vector work in a few lumps, LDS requests a few per k pair (a
    // burst of them delays the B-operand reads queued behind it). Two workgroup barriers, both in mid-stream.
    constexpr int KP_L3 = 5;                   // 8 lumps of layer 3 at k pairs 5, 9, .., 33; row pi is requested at k pair pi
    constexpr int KP_ST = 33;                  // partial sums -> LDS (same lump as the last of layer 3)
    constexpr int KP_BAR1 = 36;                // barrier; then 2 SP requests of partial sums, one per k pair
    constexpr int KP_A1 = 26;                  // K1 requests of layer-1 weights, one per k pair
    constexpr int KP_FIN = KP_BAR1 + 2 * SP + 3; // state update, cost; inputs of the next step (requested a k pair before)
    constexpr int KP_L1 = KP_FIN + 1;          // layer 1
    constexpr int KP_RELU = KP_L1 + 2;         // relu; the 32 image writes follow, two per k pair
    constexpr int KP_BAR2 = KP_RELU + 19;
    constexpr int KP_NOISE = KP_BAR2 + 1;
    constexpr int KP_ACC = KP_BAR2 + 2;        // 8 requests: layer-2 bias into the set's accumulators for its next half-step
    static_assert(KP_A1 + K1 <= KP_FIN && KP_ACC + 8 < NKP - 8, "the schedule of a half-step");
    auto half_step = [&](auto Qc, auto finc, auto prepc, int t_prep) {
        constexpr int Q = decltype(Qc)::value, O = 1 - Q;
        constexpr bool do_fin = decltype(finc)::value && !(MPPI_MLP2_ABL & 1), do_prep = decltype(prepc)::value && !(MPPI_MLP2_ABL & 2);
        integral_constant<int, O> Oc;
        f32x16 (&acc)[2] = Q ? accB : accA;
        StepRegs g;
        float bq[4]; // B operands of k pairs kp .. kp+3 (ring): h1[2 kp + hh][j]
#pragma unroll
        for (int i = 0; i < 3; ++i) bq[i] = smem[b_rd0 + (Q * kHid + 2 * i) * R];
        MPPI_MLP2_TRACE_DECL();
        static_for<0, NKP>([&](auto kpc) {
            constexpr int kp = decltype(kpc)::value;
            const float b = bq[kp % 4];
            if constexpr (do_fin && kp >= KP_L3 && kp < KP_L3 + 32 && ((kp - KP_L3) & 3) == 0) l3_arrived(integral_constant<int, kp - KP_L3>{}, g);
            if constexpr (kp == KP_FIN) {
                if constexpr (do_fin) fin_arrived(g);
                if constexpr (do_prep) prep_arrived(g);
            }
            __builtin_amdgcn_sched_barrier(0);
            mfma_acc(acc[0], a2[0][kp], b, yes, no); // (k pair 0 accumulates on the bias: acc_init)
            mfma_acc(acc[1], a2[1][kp], b, yes, no);
            __builtin_amdgcn_sched_barrier(0); // keep what follows in one piece, behind the MFMAs
            // ---- LDS requests: a few per k pair, and in FRONT of the lump — there they issue while the second MFMA
            // runs; behind the lump each would hold the next MFMA back by its issue slot
            if constexpr (kp + 3 < kHid / 2) bq[(kp + 3) % 4] = smem[b_rd0 + (Q * kHid + 2 * (kp + 3)) * R];
            if constexpr (!(MPPI_MLP2_ABL & 1024)) {
                if constexpr (do_fin && kp < 32) l3_request(integral_constant<int, kp>{}, g);
                if constexpr (do_fin && kp > KP_BAR1 && kp <= KP_BAR1 + 2 * SP) fin_request(Oc, integral_constant<int, kp - KP_BAR1 - 1>{}, g);
                if constexpr (do_prep && kp >= KP_A1 && kp < KP_A1 + K1) a1_request(integral_constant<int, kp - KP_A1>{}, g);
                if constexpr (do_prep && kp == KP_FIN - 2) prep_request(Oc, t_prep, g);
                if constexpr (do_prep && kp > KP_RELU && kp <= KP_RELU + 16 && !(MPPI_MLP2_ABL & 2048)) image_store(Oc, integral_constant<int, 2 * (kp - KP_RELU - 1)>{});
                if constexpr (do_prep && kp >= KP_ACC && kp < KP_ACC + 8) acc_init(Oc, integral_constant<int, kp - KP_ACC>{}); // its image is written: the registers are free
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- vector lumps
            if constexpr (do_fin && kp >= KP_L3 && kp < KP_L3 + 32 && ((kp - KP_L3) & 3) == 0 && !(MPPI_MLP2_ABL & 16))
                l3_lump(Oc, integral_constant<int, kp - KP_L3>{}, g);
            if constexpr (do_fin && kp == KP_ST && !(MPPI_MLP2_ABL & 32)) l3_store(Oc, g);
            if constexpr (kp == KP_BAR1 || kp == KP_BAR2) {
                if constexpr (!(MPPI_MLP2_ABL & 4)) __syncthreads(); // the partial sums / the h1 image of set O are complete
            }
            if constexpr (kp == KP_FIN) {
                if constexpr (do_fin && !(MPPI_MLP2_ABL & 64)) fin_lump(Oc, g);
                if constexpr (do_prep && !(MPPI_MLP2_ABL & 128)) prep_lump(Oc, g);
            }
            if constexpr (do_prep && kp == KP_L1 && !(MPPI_MLP2_ABL & 256)) l1_mfmas(Oc, g);
            if constexpr (do_prep && kp == KP_RELU && !(MPPI_MLP2_ABL & 512)) relu_lump(Oc);
            if constexpr (kp == KP_NOISE && Q == 1 && SRC == SRC_PHILOX && do_prep && !(MPPI_MLP2_ABL & 8)) {
                // both sets' next horizon group, other buffer: its last readers were the preparations of step
                // 4 (gn - 1) - 1 of both sets, which lie behind this half-step's barriers
                if ((t_prep & 3) == 1) {
                    const int gn = (t_prep >> 2) + 1;
                    if (gn < NG) noise_groups(gn);
                }
            }
            MPPI_MLP2_TRACE_AT(Q, kp);
            __builtin_amdgcn_sched_barrier(0);
        });
        if constexpr (Q == 0 && decltype(finc)::value && decltype(prepc)::value) MPPI_MLP2_TRACE_FLUSH(t_prep == 10 && tile_is_first && blockIdx.x == 0 && lane == 0);
    };

    // The stationary weights (258 global loads per lane, plus W1 / W3 / U staged in LDS) are this workgroup's for its
    // whole life: it walks over tiles, one per CU at a time (one 512-register workgroup fits a CU).
    const int n_tiles = (K + kMlp2R - 1) / kMlp2R;
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    k0 = tile * kMlp2R;
    cA = 0.0f; cB = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) { xA[i] = x0[i]; xB[i] = x0[i]; }
    // ---- prologue: noise of group 0 of both sets; step 0 of set 0 up to its image
    if constexpr (SRC == SRC_PHILOX) noise_groups(0);
    __syncthreads();
    MPPI_MLP2_STAMP_AT(0);
    {
        integral_constant<int, 0> c0;
        StepRegs g;
        prep_request(c0, 0, g);
        static_for<0, K1>([&](auto ic) { a1_request(ic, g); });
        prep_lump(c0, g);
        l1_mfmas(c0, g);
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(accA[0]), "+v"(accA[1])); // MFMA D -> VALU reader: 16 passes + 2
        relu_lump(c0);
        static_for<0, 16>([&](auto ic) { image_store(c0, integral_constant<int, 2 * decltype(ic)::value>{}); });
        static_for<0, 8>([&](auto ic) { acc_init(c0, ic); });
    }
    __syncthreads();
    {
        integral_constant<int, 0> s0;
        integral_constant<int, 1> s1;
        // layer 2 of set 0, step t | set 1: finish step t-1, prepare step t        (first: nothing to finish yet)
        // layer 2 of set 1, step t | set 0: finish step t,   prepare step t+1      (last: nothing left to prepare)
        half_step(s0, no, yes, 0);
        for (int t = 0; t + 1 < H; ++t) {
            half_step(s1, yes, yes, t + 1);
            half_step(s0, yes, yes, t + 1);
        }
        half_step(s1, yes, no, H);
    }
    MPPI_MLP2_STAMP_AT(1);
    // ---- epilogue: set 1's last step
    {
        integral_constant<int, 1> c1;
        StepRegs g;
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(accB[0]), "+v"(accB[1]));
        static_for<0, 8>([&](auto ic) {
            constexpr int pi0 = 4 * decltype(ic)::value;
            static_for<0, 4>([&](auto rc) { l3_request(integral_constant<int, pi0 + decltype(rc)::value>{}, g); });
            l3_lump(c1, integral_constant<int, pi0>{}, g);
        });
        l3_store(c1, g);
        __syncthreads();
        static_for<0, 2 * SP>([&](auto ic) { fin_request(c1, ic, g); });
        fin_lump(c1, g);
    }
    cA = cA + state_cost<S, QFULL>(CC, xA); // terminal cost, controller_base.cpp:271-272
    cB = cB + state_cost<S, QFULL>(CC, xB);
    // lane l now stands for rollout k0 + l of the tile: set l >> 5, column l & 31
    const float c = hh ? cB : cA;
    const bool valid = (k0 + lane) < K;
    const int kk = valid ? k0 + lane : K - 1;
    if (w == 0 && valid) cost[k0 + lane] = c;
    if (MODE == MODE_COST_ONLY) continue;
    mlp_tile_record<A, DIAG, 4>(C, c, valid, w, lane, kk, H, NG, SRC, eps_hbm, seed, koff + (unsigned long long)kk, base,
                                partials + (size_t)record_slot(tile, rsc) * rsb, rsc); // element (b, col) at partials[b*rsb + col*rsc]
    } // tiles
}

} // namespace mppi
