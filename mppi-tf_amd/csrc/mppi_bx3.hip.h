// mppi_bx3.hip.h — k_rollout_mlp_bx3: the learned 2x256 MLP model_base on the BF16 matrix cores at fp32-class accuracy
// (opt-in MPPI_FLAG_MLP_BF16X3). Included by mppi_kernels.hip.h. Second design (r03); the first — 8 waves in lock-step phases, two
// barriers per step, 1.57 ms at BASELINE configs[3] against 1.22-1.25 ms for this one — is gone.
//
// Arithmetic: every fp32 operand is split x = hi + lo into two bf16 values (hi = bf16(x), lo = bf16(x - hi), together 16 mantissa
// bits) and a product sum is evaluated as three v_mfma_f32_32x32x16_bf16 into one fp32 accumulator (a_lo b_hi + a_hi b_lo + a_hi b_hi;
// bf16 x bf16 is exact in fp32, the dropped lo x lo term is 2^-18 relative): sample costs after 64 recurrent steps within 9e-7
// (relative) of fp64, against 4e-7 for exact fp32; the parity tests hold it to 2e-5. b1 rides as input k = NIN against a constant 1.
// Structure: k_rollout_mlp2's (mppi_mlp2.hip.h) — ONE wave per SIMD, wave w owns hidden units [64w, 64w+64) of both
// layers (2 M-tiles x 16 k-blocks x (hi, lo) x 4 registers = exactly a0-a255, pinned there by inline-asm MFMAs), TWO sets
// of 32 rollouts per workgroup software-pipelined against each other, persistent workgroups, all cross-wave hand-offs
// through workgroup barriers. What differs from the f32 kernel, and why (tools/micro/mfma_bf16_shadow.hip,
// profiles/r03_mfma_bf16_shadow.json): the bf16 MFMA (32 cycles) runs on the matrix core proper and up to ~5 plain
// vector instructions of the SAME wave issue in its shadow for free — v_pk_*_f32 do not (17 cycles for the first, they
// serialise with the MFMA), LDS reads nearly do. So instead of a few big lumps the other set's step is cut into ~95
// PIECES of 5-8 plain (unpacked) vector instructions, one after each MFMA of the running set's stream of 96:
//   layer 3 ON THE MATRIX CORE TOO (relu + split of the accumulator registers: they are the A fragments of
//   out[rollout][n] = sum_u h2[rollout][u] W3[u][n], W3 stationary as 4 x (hi, lo) B fragments, 12 MFMAs; the first cut
//   did it on the vector ALU with W3 rows from LDS: 24 B per lane and accumulator register, LDS-bandwidth-bound, 21 % of
//   the kernel), barrier, cross-wave sum + state update + costs, noise + next inputs (normalise, split into bf16 hi/lo),
//   6 layer-1 MFMAs, relu + split + image stores (one pair of accumulator registers per piece), barrier.
// The h1 image is [part][k-block][lane half][rollout] x 8 bf16, in the k order of an
// accumulator tile, so a layer-1 accumulator register block IS a layer-2 B fragment (one ds_write_b128 / ds_read_b128
// each); here a B fragment feeds both M-tiles of the wave (half the LDS reads per MFMA of the first design).
// Layer 2's bias is the accumulators' initial value (8 LDS reads per set and step, once the set's image is written).
#pragma once

// (MPPI_BX3_CUT / MPPI_BX3_ABL / MPPI_BX3_STAMP_*: the timing-study layer, mppi_ablate.hip.h — 1000 / 0 / nothing in the shipped build)

namespace mppi {

constexpr int kBx3Threads = 256;
constexpr int kBx3R = 64; // rollouts per workgroup: two sets of 32
__host__ __device__ inline size_t bx3_lds_floats(int S, int A, int H)
{
    return (size_t)2 * 8192 + 2 * 4 * S * 32 + 2 * 2 * 4 * A * 32 + kHid + (size_t)(H * A + 3) / 4 * 4 + 64;
}

typedef int i32x4 __attribute__((ext_vector_type(4)));

// two floats -> one register of two bf16 (round to nearest even), low half = a
__device__ __forceinline__ int pk_bf16(float a, float b)
{
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    typedef __bf16 bf16x2_ __attribute__((ext_vector_type(2)));
    return __builtin_bit_cast(int, __builtin_convertvector((f32x2_){a, b}, bf16x2_)); // v_cvt_pk_bf16_f32 (as asm hipcc pads it with s_nop)
}
// (hi, lo) split of a pair: hi = bf16(x), lo = bf16(x - hi)
__device__ __forceinline__ void split_pair(float a, float b, int &hi, int &lo)
{
    hi = pk_bf16(a, b);
    const float ah = __builtin_bit_cast(float, hi << 16), bh = __builtin_bit_cast(float, hi & (int)0xffff0000);
    lo = pk_bf16(a - ah, b - bh);
}

template <int A, bool DIAG, int SRC>
__global__ __launch_bounds__(kBx3Threads, 1) void k_rollout_mlp_bx3(
    const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int MODE, const int rsb, const int rsc)
{
    constexpr bool QFULL = false;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * A, NIN = S + A, SP = S / 2;
    static_assert(NIN + 1 <= 16 && S <= 8, "inputs + bias fit one k-block; W3 rows are staged as 8 floats");
    constexpr int R = 32;  // rollouts of a set = columns of one MFMA tile
    constexpr int NKB = 16; // k-blocks of layer 2
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    float *img_s = smem;                     // [2 sets][2 parts][16 k-blocks][2 halves][32 rollouts] x 16 B = 2 x 32 KB
    float *y_s = img_s + 2 * 8192;           // [2 sets][4 waves][S][R]: layer-3 partial sums of the waves
    float *z_s = y_s + 2 * 4 * S * R;        // [2 sets][2 buffers][4*A][R] standard normals of a horizon group
    float *b2_s = z_s + 2 * 2 * 4 * A * R;   // [kHid] layer-2 bias
    float *u_s = b2_s + kHid;                // [H*A] nominal controls

    const int tid = threadIdx.x;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63, j = lane & 31, hh = lane >> 5;
    int k0 = 0;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    auto lds2 = [&](int idx) { return *static_cast<const f32x2 *>(__builtin_assume_aligned(smem + idx, 8)); };
    auto lds4 = [&](int idx) { return *static_cast<const f32x4 *>(__builtin_assume_aligned(smem + idx, 16)); };
    auto ldsq = [&](int idx) { return *static_cast<const i32x4 *>(__builtin_assume_aligned(smem + idx, 16)); };
    auto stsq = [&](int idx, const i32x4 &v) { *static_cast<i32x4 *>(__builtin_assume_aligned(smem + idx, 16)) = v; };

    // ---- stationary fragments. Layer 2: element e of lane (j, hh) of k-block kb is W2[16 kb + 8 (e >> 2) + 4 hh + (e & 3)][unit]
    // (the k order of an accumulator tile); layer 1: natural k order 8 hh + e, k = NIN is the bias row against a constant 1.
    // Every fragment is born as ONE 128-bit value (a ds_read_b128 of what the lane itself just wrote): assembled from four
    // scalars, hipcc copies the four registers into a fresh tuple in front of every MFMA that uses them.
    i32x4 a2h[2][NKB], a2l[2][NKB], a1h[2], a1l[2], w3h[4], w3l[4];
    {
        int st_idx = (w * 64 + lane) * 4; // this lane's 16-byte slot of round-trip buffer entry 0 (floats); entry n at + n * 1024
        asm volatile("" : "+v"(st_idx));
        int ld_idx = st_idx;
        asm volatile("" : "+v"(ld_idx)); // opaque: no store-to-load forwarding
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int unit = 64 * w + 32 * mt + j;
#pragma unroll
            for (int kb = 0; kb < NKB; ++kb) {
                i32x4 fh, fl;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int ka = 16 * kb + 8 * ((2 * q) >> 2) + 4 * hh + ((2 * q) & 3);
                    int hi, lo;
                    split_pair(M->W2[(size_t)ka * kHid + unit], M->W2[(size_t)(ka + 1) * kHid + unit], hi, lo);
                    fh[q] = hi; fl[q] = lo;
                }
                stsq(st_idx + (2 * (kb & 7)) * 1024, fh);
                stsq(st_idx + (2 * (kb & 7) + 1) * 1024, fl);
                if ((kb & 7) == 7) { // 16 entries x 4 waves x 1 KB = the 64 KB of the two images
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
                    for (int k2 = kb - 7; k2 <= kb; ++k2) {
                        a2h[mt][k2] = ldsq(ld_idx + (2 * (k2 & 7)) * 1024);
                        a2l[mt][k2] = ldsq(ld_idx + (2 * (k2 & 7) + 1) * 1024);
                    }
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
            }
        }
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int unit = 64 * w + 32 * mt + j;
            i32x4 fh, fl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v2[2];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int k = 8 * hh + 2 * q + o;
                    v2[o] = k < NIN ? M->W1[(size_t)k * kHid + unit] : (k == NIN ? M->b1[unit] : 0.0f);
                }
                int hi, lo;
                split_pair(v2[0], v2[1], hi, lo);
                fh[q] = hi; fl[q] = lo;
            }
            stsq(st_idx + (2 * mt) * 1024, fh);
            stsq(st_idx + (2 * mt + 1) * 1024, fl);
        }
        // layer 3: B fragment of k-block f = 2 mt + s (the wave's units 32 mt + 16 s ..+15 in accumulator order), column n = j
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            i32x4 fh, fl;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float v2[2];
#pragma unroll
                for (int o = 0; o < 2; ++o) {
                    const int e = 2 * q + o, unit = 64 * w + 16 * f + 8 * (e >> 2) + 4 * hh + (e & 3);
                    v2[o] = j < S ? M->W3[(size_t)unit * S + (j < S ? j : 0)] : 0.0f;
                }
                int hi, lo;
                split_pair(v2[0], v2[1], hi, lo);
                fh[q] = hi; fl[q] = lo;
            }
            stsq(st_idx + (4 + 2 * f) * 1024, fh);
            stsq(st_idx + (4 + 2 * f + 1) * 1024, fl);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            a1h[mt] = ldsq(ld_idx + (2 * mt) * 1024);
            a1l[mt] = ldsq(ld_idx + (2 * mt + 1) * 1024);
        }
#pragma unroll
        for (int f = 0; f < 4; ++f) {
            w3h[f] = ldsq(ld_idx + (4 + 2 * f) * 1024);
            w3l[f] = ldsq(ld_idx + (4 + 2 * f + 1) * 1024);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __syncthreads(); // the buffer is the image area
    }
    for (int i = tid; i < kHid; i += kBx3Threads) b2_s[i] = M->b2[i];
    for (int i = tid; i < HA; i += kBx3Threads) u_s[i] = U_dev[i];

    // wave-uniform constants, read once (a barrier would otherwise force a re-fetch per step)
    float b3v[S], ysd[S], ymn[S];
#pragma unroll
    for (int i = 0; i < S; ++i) { b3v[i] = M->b3[i]; ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    // the lane's 8 layer-1 inputs are k = 8 hh + e: mean / reciprocal deviation of THOSE inputs, per lane; the bias input
    // is (1 - 0) * 1, the padding (0 - 0) * 1
    float xms[8], xrs[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float m0 = e < NIN ? M->xmean[e] : 0.0f, r0 = e < NIN ? 1.0f / M->xstd[e] : 1.0f;
        const float m1 = 8 + e < NIN ? M->xmean[8 + e < NIN ? 8 + e : 0] : 0.0f, r1 = 8 + e < NIN ? 1.0f / M->xstd[8 + e < NIN ? 8 + e : 0] : 1.0f;
        xms[e] = hh ? m1 : m0;
        xrs[e] = hh ? r1 : r0;
    }
    PcProducerConsts<A> pcst;
    pcst.template load<DIAG>(C);
    const PcProducerConsts<A> *PC = &pcst;
    PcConsumerConsts<S> ccst;
    ccst.load(C);
    const PcConsumerConsts<S> *CC = &ccst;
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned long long koff = (unsigned long long)C->k_offset;
    auto kk_of = [&](int q) { return min(k0 + R * q + j, K - 1); };

    // per-lane state of rollout j of BOTH sets (replicated in the two lane halves and the 4 waves)
    float xA[S], xB[S], x0[S], cA = 0.0f, cB = 0.0f, acA = 0.0f, acB = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x0[i] = x_dev[i];
    f32x16 accA[2], accB[2];
    // LDS indices (floats), everything but a compile-time constant in ONE register per array (see mppi_mlp2.hip.h)
    int img_rd0 = (hh * 32 + j) * 4;
    int img_wr0 = (hh * 32 + j) * 4 + 4 * w * 256;
    int y_wr0[2] = {(int)(y_s - smem) + ((0 * 4 + w) * S + (j < S ? j : 0)) * R + 4 * hh, (int)(y_s - smem) + ((1 * 4 + w) * S + (j < S ? j : 0)) * R + 4 * hh};
    int y_rd0 = (int)(y_s - smem) + j;
    int z_rd0 = (int)(z_s - smem) + j;
    int b2_rd0 = (int)(b2_s - smem) + 64 * w + 4 * hh;
    asm volatile("" : "+v"(img_rd0), "+v"(img_wr0), "+v"(y_wr0[0]), "+v"(y_wr0[1]), "+v"(y_rd0), "+v"(z_rd0), "+v"(b2_rd0));

    using std::integral_constant;
    constexpr integral_constant<bool, true> yes{};
    constexpr integral_constant<bool, false> no{};

    // The MFMAs are inline asm: the A fragments of layer 2 pinned to a0-a255 ("a"), accumulators in v. hipcc does not see
    // them as MFMAs, so their hazards are kept by construction and checked by tools/check_mfma_hazards.py: a vector
    // reader of an accumulator is always at least two MFMAs (16 passes) behind its last write; the B fragments come
    // from LDS (hipcc waits) or, in layer 1, from vector instructions in front of an s_nop 1.
    auto mfma2 = [&](f32x16 &acc, const i32x4 &a, const i32x4 &b) {
        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "a"(a), "v"(b));
    };
    auto mfma1 = [&](f32x16 &acc, const i32x4 &a, const i32x4 &b, auto first) {
        if constexpr (decltype(first)::value) asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&v"(acc) : "v"(a), "v"(b));
        else asm volatile("s_nop 1\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b));
    };

    struct StepRegs {       // values that live across pieces
        f32x16 yacc;        // layer 3: [rollout (register, lane half)][output n = lane & 31]
        i32x4 ah, al;       // ... its A fragment under way: relu + split of 8 accumulator registers
        float yv[S][4];     // partial sums of the 4 waves
        float zz[A], u[A], e[A], v[A];
        float in[8];        // the lane's 8 normalised layer-1 inputs
        i32x4 bh, bl;       // ... split: layer 1's B fragment
        int hi4[4], lo4[4]; // relu + split of 8 accumulator registers under way
        float ra, rb, rah, rbh; // ... of the pair at hand, between the two halves of its piece
        float sc;           // state cost under way
        float acn;          // action cost of the step being prepared (the finish still needs the last step's)
    };
    // ---- layer 3. Pair piece pp (0..15): relu + split of accumulator registers 2 pp, 2 pp + 1 of the set; every 4th pair
    // completes the A fragment of k-block f = pp >> 2 (rows = the set's rollouts, k = the wave's units 16 f ..+15)
    auto l3_pair = [&](auto qc, auto ppc, StepRegs &g) {
        constexpr int q = decltype(qc)::value, pp = decltype(ppc)::value, mt = pp >> 3, r0 = 2 * (pp & 7), f4 = pp & 3;
        f32x16 (&acc)[2] = q ? accB : accA;
        float a, b;
        asm("v_max_f32 %0, 0, %1" : "=v"(a) : "v"(acc[mt][r0])); // fmaxf costs a canonicalising second v_max
        asm("v_max_f32 %0, 0, %1" : "=v"(b) : "v"(acc[mt][r0 + 1]));
        split_pair(a, b, g.hi4[f4], g.lo4[f4]);
        if constexpr (f4 == 3) {
            g.ah = i32x4{g.hi4[0], g.hi4[1], g.hi4[2], g.hi4[3]};
            g.al = i32x4{g.lo4[0], g.lo4[1], g.lo4[2], g.lo4[3]};
            asm volatile("" : "+v"(g.ah), "+v"(g.al));
        } else {
            asm volatile("" : "+v"(g.hi4[f4]), "+v"(g.lo4[f4]));
        }
    };
    auto l3_mfma = [&](auto fc, auto ic, StepRegs &g) { // product i of 3 of k-block f
        constexpr int f = decltype(fc)::value, i = decltype(ic)::value;
        if constexpr (i == 0) mfma1(g.yacc, g.al, w3h[f], integral_constant<bool, f == 0>{});
        else if constexpr (i == 1) mfma1(g.yacc, g.ah, w3l[f], no);
        else mfma1(g.yacc, g.ah, w3h[f], no);
    };
    // lanes n = j < S hold output n of the set's 32 rollouts: register r is rollout (r & 3) + 8 (r >> 2) + 4 hh
    auto l3_store = [&](auto qc, StepRegs &g) {
        constexpr int q = decltype(qc)::value;
        if (j < S) {
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                const f32x4 v4 = {g.yacc[4 * g4], g.yacc[4 * g4 + 1], g.yacc[4 * g4 + 2], g.yacc[4 * g4 + 3]};
                *static_cast<f32x4 *>(__builtin_assume_aligned(smem + y_wr0[q] + 8 * g4, 16)) = v4;
            }
        }
    };
    // ---- finish: y = sum of the 4 waves' partial sums (fixed order) + b3, state update, cost of the step
    auto fin_request = [&](auto qc, auto nc, StepRegs &g) { // output n of the 4 waves
        constexpr int q = decltype(qc)::value, n = decltype(nc)::value;
#pragma unroll
        for (int ww = 0; ww < 4; ++ww) g.yv[n][ww] = smem[y_rd0 + ((q * 4 + ww) * S + n) * R];
    };
    auto fin_piece = [&](auto qc, auto nc, StepRegs &g) { // output n
        constexpr int q = decltype(qc)::value, n = decltype(nc)::value;
        float (&x)[S] = q ? xB : xA;
        float y = g.yv[n][0] + g.yv[n][1];
        y = y + g.yv[n][2];
        y = y + g.yv[n][3];
        y = y + b3v[n];
        x[n] = x[n] + (y * ysd[n] + ymn[n]);
        asm volatile("" : "+v"(x[n])); // the piece stays HERE: left alone hipcc sinks it to the first use of x[n] (one lump of ~80 instructions)
    };
    auto cost_piece = [&](auto qc, auto ic, StepRegs &g) { // dimension i of (x - goal)^T Q (x - goal), as state_cost sums it
        constexpr int q = decltype(qc)::value, i = decltype(ic)::value;
        float (&x)[S] = q ? xB : xA;
        const float d = x[i] - CC->goal[i];
        const float term = d * (CC->qdiag[i] * d);
        g.sc = i == 0 ? term : g.sc + term;
        if constexpr (i == S - 1) {
            float &c = q ? cB : cA;
            const float tmp = g.sc + (q ? acB : acA); // cost on the POST-step state + the action cost of the step
            c = c + tmp;
            asm volatile("" : "+v"(c));
        } else {
            asm volatile("" : "+v"(g.sc));
        }
    };
    // ---- preparation of step t: noise and nominal control, the lane's 8 normalised inputs, split
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
    auto noise_piece = [&](auto qc, auto ic, StepRegs &g) { // 0: scale the noise, perturbed control; 1: action cost
        constexpr int q = decltype(qc)::value, i = decltype(ic)::value;
        if constexpr (i == 0) {
            if constexpr (SRC == SRC_PHILOX) scale_noise<A, DIAG>(PC, g.zz, g.e);
#pragma unroll
            for (int k = 0; k < A; ++k) g.v[k] = g.u[k] + g.e[k];
#pragma unroll
            for (int k = 0; k < A; ++k) asm volatile("" : "+v"(g.v[k]), "+v"(g.e[k]));
        } else {
            g.acn = action_cost<A, DIAG>(PC, g.u, g.e);
            asm volatile("" : "+v"(g.acn));
        }
    };
    auto input_piece = [&](auto qc, auto ec, StepRegs &g) { // inputs e, e + 1 of the lane: k = 8 hh + e
        constexpr int q = decltype(qc)::value, e0 = decltype(ec)::value;
        float (&x)[S] = q ? xB : xA;
        if constexpr (e0 == 0) (q ? acB : acA) = g.acn; // behind the cost pieces of the step just finished
        auto raw = [&](int k) { return k < S ? x[k < S ? k : 0] : (k < NIN ? g.v[k < NIN && k >= S ? k - S : 0] : (k == NIN ? 1.0f : 0.0f)); };
#pragma unroll
        for (int e = e0; e < e0 + 2; ++e) {
            const float lo = raw(e), hi = raw(8 + e);
            const float sel = hh ? hi : lo;
            g.in[e] = (sel - xms[e]) * xrs[e];
        }
        asm volatile("" : "+v"(g.in[e0]), "+v"(g.in[e0 + 1]));
    };
    auto insplit_piece = [&](auto ec, StepRegs &g) {
        constexpr int e0 = decltype(ec)::value;
        int hi, lo;
        split_pair(g.in[e0], g.in[e0 + 1], hi, lo);
        asm volatile("" : "+v"(hi), "+v"(lo));
        g.bh[e0 >> 1] = hi; g.bl[e0 >> 1] = lo;
    };
    auto l1_mfma = [&](auto qc, auto ic, StepRegs &g) { // MFMA i of 6: (mt, product)
        constexpr int q = decltype(qc)::value, i = decltype(ic)::value, mt = i & 1, pr = i >> 1;
        f32x16 (&acc)[2] = q ? accB : accA;
        if constexpr (pr == 0) mfma1(acc[mt], a1l[mt], g.bh, yes);
        else if constexpr (pr == 1) mfma1(acc[mt], a1h[mt], g.bl, no);
        else mfma1(acc[mt], a1h[mt], g.bh, no);
    };
    // relu + split of accumulator registers 2 pp, 2 pp + 1 of the set (pp = 0..15); every 4th pair completes a fragment
    // (8 registers = the B fragment of k-block 4 w + 2 mt + s) and stores it
    auto relu_piece = [&](auto qc, auto ppc, auto halfc, StepRegs &g) {
        constexpr int q = decltype(qc)::value, pp = decltype(ppc)::value, mt = pp >> 3, r0 = 2 * (pp & 7), f = (pp & 3), half = decltype(halfc)::value;
        f32x16 (&acc)[2] = q ? accB : accA;
        if constexpr (half == 0) { // relu, hi = bf16 of the pair, hi back as floats
            asm("v_max_f32 %0, 0, %1" : "=v"(g.ra) : "v"(acc[mt][r0]));
            asm("v_max_f32 %0, 0, %1" : "=v"(g.rb) : "v"(acc[mt][r0 + 1]));
            g.hi4[f] = pk_bf16(g.ra, g.rb);
            g.rah = __builtin_bit_cast(float, g.hi4[f] << 16);
            g.rbh = __builtin_bit_cast(float, g.hi4[f] & (int)0xffff0000);
            asm volatile("" : "+v"(g.ra), "+v"(g.rb), "+v"(g.rah), "+v"(g.rbh), "+v"(g.hi4[f]));
        } else { // lo = bf16 of what hi left; the 4th pair completes and stores the fragment
            g.lo4[f] = pk_bf16(g.ra - g.rah, g.rb - g.rbh);
            asm volatile("" : "+v"(g.lo4[f]));
            if constexpr (f == 3) {
                constexpr int s = (pp & 7) >> 2;
                const i32x4 h4 = {g.hi4[0], g.hi4[1], g.hi4[2], g.hi4[3]}, l4 = {g.lo4[0], g.lo4[1], g.lo4[2], g.lo4[3]};
                stsq(img_wr0 + ((q * 2 + 0) * 16 + 2 * mt + s) * 256, h4);
                stsq(img_wr0 + ((q * 2 + 1) * 16 + 2 * mt + s) * 256, l4);
            }
        }
    };
    // layer 2's bias as the initial value of the set's accumulators (register r of M-tile mt is row 32 mt + 8 (r >> 2) + 4 hh
    // + (r & 3) of the wave's 64: four consecutive rows per 16-byte read), piece i of 8
    auto acc_init = [&](auto qc, auto ic) {
        constexpr int q = decltype(qc)::value, i = decltype(ic)::value, mt = i >> 2, g4 = i & 3;
        f32x16 (&acc)[2] = q ? accB : accA;
        const f32x4 b4 = lds4(b2_rd0 + 32 * mt + 8 * g4);
        acc[mt][4 * g4 + 0] = b4.x; acc[mt][4 * g4 + 1] = b4.y; acc[mt][4 * g4 + 2] = b4.z; acc[mt][4 * g4 + 3] = b4.w;
    };
    // The standard normals of horizon group gn for BOTH sets -> buffer gn & 1 (as k_rollout_mlp2: one Philox block per lane
    // of 2 A of the workgroup's 8 half-waves)
    auto noise_groups = [&](int gn) {
        const int unit = 2 * w + hh;
        if (2 * w < 2 * A) { // wave-uniform
            const int set = unit / A, q = unit - set * A;
            if (unit < 2 * A) {
                const float4 n = normals_of_block(seed, koff + (unsigned long long)kk_of(set), (base + (unsigned long long)gn) * A + q);
                float *zd = z_s + ((set * 2 + (gn & 1)) * 4 * A + 4 * q) * R + j;
                zd[0 * R] = n.x; zd[1 * R] = n.y; zd[2 * R] = n.z; zd[3 * R] = n.w;
            }
        }
    };

    // ---- One half-step: the 96 layer-2 MFMAs of set Q stream (k-block kb = m / 6: (mt, product) = (m & 1, (m % 6) >> 1));
    // after MFMA m comes piece m of the OTHER set O: it finishes its last step (FIN) and prepares step t_prep (PREP).
    constexpr int M_L3 = 2;                        // 16 pair pieces (relu + split of the set's layer-2 accumulators); the 3 layer-3 MFMAs of
                                                   // k-block f ride in the 3 gaps after its 4th pair
    constexpr int M_ST = M_L3 + 16 + 3 + 2;        // partial sums -> LDS (two MFMAs behind the last of layer 3)
    constexpr int M_BAR1 = M_ST + 1;               // barrier
    constexpr int M_FRQ = M_BAR1 + 1;              // SP gaps: the 4 waves' partial sums of two outputs each
    constexpr int M_FIN = M_FRQ + SP + 1;          // S pieces: y, state update
    constexpr int M_COST = M_FIN + S;              // S pieces: state cost by dimension
    constexpr int M_IN = M_COST + S;               // 4 x (normalise a pair of inputs, split it)
    constexpr int M_L1 = M_IN + 8;                 // 6 layer-1 MFMAs
    constexpr int RELU_GAPS = S <= 6 ? 2 : 1;      // a relu + split piece (8 vector instructions) over two gaps where the schedule has them
    constexpr int M_RELU = M_L1 + 6 + 2;           // 16 pieces: relu + split of a register pair, image stores
    constexpr int M_BAR2 = M_RELU + 16 * RELU_GAPS; // barrier: the image of set O is complete
    constexpr int M_NG = M_BAR2 + 1;               // next horizon group's noise (every 4th step)
    constexpr int M_NEXT = M_BAR2 + 1 > 6 * 14 ? M_BAR2 + 1 : 6 * 14; // first B fragments of the next half-step (set O's new image): behind the
                                                   // barrier, and not before k-block 14 (ring entries 0 and 1 serve k-blocks 12 and 13 until then)
    constexpr int M_ACC = M_BAR2 + 1;              // layer-2 bias into set O's accumulators (its image is written: they are free), 2 of 8 reads per gap
    // the noise of the next step does not wait for the finish: requested during layer 3, scaled in the gaps of its last k-block
    constexpr int M_PRQ = M_L3 + 8, M_NOISE = M_L3 + 16;
    static_assert(M_ACC + 4 <= 6 * NKB && M_NEXT < 6 * NKB, "the schedule of a half-step");
    i32x4 bqh[4], bql[4]; // B fragments (hi, lo) of k-blocks kb .. kb + 2: ring of 4, so that 16 and 17 = the next half-step's 0 and 1
    auto b_request = [&](int set, auto kbc) {
        constexpr int kb = decltype(kbc)::value;
        bqh[kb & 3] = ldsq(img_rd0 + ((set * 2 + 0) * 16 + (kb & 15)) * 256);
        bql[kb & 3] = ldsq(img_rd0 + ((set * 2 + 1) * 16 + (kb & 15)) * 256);
    };
    auto half_step = [&](auto Qc, auto finc, auto prepc, int t_prep) {
        constexpr int Q = decltype(Qc)::value, O = 1 - Q;
        constexpr bool do_fin = decltype(finc)::value && !(MPPI_BX3_ABL & 1), do_prep = decltype(prepc)::value && !(MPPI_BX3_ABL & 2);
        integral_constant<int, O> Oc;
        f32x16 (&acc)[2] = Q ? accB : accA;
        StepRegs g;
        static_for<0, 6 * NKB>([&](auto mc) {
            constexpr int m = decltype(mc)::value, kb = m / 6, i = m % 6, mt = i & 1, pr = i >> 1;
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (pr == 0) mfma2(acc[mt], a2l[mt][kb], bqh[kb & 3]);
            else if constexpr (pr == 1) mfma2(acc[mt], a2h[mt][kb], bql[kb & 3]);
            else mfma2(acc[mt], a2h[mt][kb], bqh[kb & 3]);
            __builtin_amdgcn_sched_barrier(0);
            // ---- LDS requests
            if constexpr (i == 0 && kb + 2 < NKB && !(MPPI_BX3_ABL & 32)) b_request(Q, integral_constant<int, kb + 2>{});
            if constexpr (m < MPPI_BX3_CUT) {
            if constexpr (do_fin && m >= M_FRQ && m < M_FRQ + SP) {
                fin_request(Oc, integral_constant<int, 2 * (m - M_FRQ)>{}, g);
                fin_request(Oc, integral_constant<int, 2 * (m - M_FRQ) + 1>{}, g);
            }
            if constexpr (do_prep && m == M_PRQ) prep_request(Oc, t_prep, g);
            if constexpr (do_prep && m >= M_ACC && m < M_ACC + 4) {
                acc_init(Oc, integral_constant<int, 2 * (m - M_ACC)>{});
                acc_init(Oc, integral_constant<int, 2 * (m - M_ACC) + 1>{});
            }
            __builtin_amdgcn_sched_barrier(0);
            // ---- the piece
            if constexpr (do_fin && !(MPPI_BX3_ABL & 8)) {
                if constexpr (m >= M_L3 + 4 && m < M_L3 + 4 + 16 && ((m - M_L3) & 3) != 3) // behind the Q MFMA, in front of the pair piece
                    l3_mfma(integral_constant<int, ((m - M_L3 - 4) / 4)>{}, integral_constant<int, ((m - M_L3) & 3)>{}, g);
                if constexpr (m >= M_L3 && m < M_L3 + 16) l3_pair(Oc, integral_constant<int, m - M_L3>{}, g);
            }
            if constexpr (do_fin && m == M_ST) l3_store(Oc, g);
            if constexpr (m == M_BAR1 || m == M_BAR2) {
                if constexpr (!(MPPI_BX3_ABL & 4)) __syncthreads(); // the partial sums / the image of set O are complete
            }
            if constexpr (do_fin && m >= M_FIN && m < M_FIN + S) fin_piece(Oc, integral_constant<int, m - M_FIN>{}, g);
            if constexpr (do_fin && m >= M_COST && m < M_COST + S) cost_piece(Oc, integral_constant<int, m - M_COST>{}, g);
            if constexpr (do_prep && m >= M_NOISE && m < M_NOISE + 2) noise_piece(Oc, integral_constant<int, m - M_NOISE>{}, g);
            if constexpr (do_prep && m >= M_IN && m < M_IN + 8) {
                if constexpr (((m - M_IN) & 1) == 0) input_piece(Oc, integral_constant<int, (m - M_IN)>{}, g);
                else insplit_piece(integral_constant<int, (m - M_IN) - 1>{}, g);
            }
            if constexpr (do_prep && m >= M_L1 && m < M_L1 + 6) l1_mfma(Oc, integral_constant<int, m - M_L1>{}, g);
            if constexpr (do_prep && m >= M_RELU && m < M_RELU + 16 * RELU_GAPS && !(MPPI_BX3_ABL & 16)) {
                if constexpr (RELU_GAPS == 2) {
                    relu_piece(Oc, integral_constant<int, (m - M_RELU) / 2>{}, integral_constant<int, ((m - M_RELU) & 1)>{}, g);
                } else {
                    relu_piece(Oc, integral_constant<int, m - M_RELU>{}, integral_constant<int, 0>{}, g);
                    relu_piece(Oc, integral_constant<int, m - M_RELU>{}, integral_constant<int, 1>{}, g);
                }
            }
            if constexpr (m == M_NG && Q == 1 && SRC == SRC_PHILOX && do_prep) {
                if ((t_prep & 3) == 1) {
                    const int gn = (t_prep >> 2) + 1;
                    if (gn < NG) noise_groups(gn);
                }
            }
            } // CUT
            if constexpr (m == M_NEXT) { // the next half-step streams set O
                b_request(O, integral_constant<int, 16>{});
                b_request(O, integral_constant<int, 17>{});
            }
        });
        __builtin_amdgcn_sched_barrier(0);
    };

    const int n_tiles = (K + kBx3R - 1) / kBx3R;
    MPPI_BX3_STAMP_BEGIN();
    for (int tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    k0 = tile * kBx3R;
    cA = 0.0f; cB = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) { xA[i] = x0[i]; xB[i] = x0[i]; }
    // ---- prologue: noise of group 0 of both sets; step 0 of set 0 up to its image
    if constexpr (SRC == SRC_PHILOX) noise_groups(0);
    __syncthreads();
    {
        integral_constant<int, 0> c0;
        StepRegs g;
        prep_request(c0, 0, g);
        noise_piece(c0, integral_constant<int, 0>{}, g);
        noise_piece(c0, integral_constant<int, 1>{}, g);
        static_for<0, 4>([&](auto ic) {
            input_piece(c0, integral_constant<int, 2 * decltype(ic)::value>{}, g);
            insplit_piece(integral_constant<int, 2 * decltype(ic)::value>{}, g);
        });
        static_for<0, 6>([&](auto ic) { l1_mfma(c0, ic, g); });
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(accA[0]), "+v"(accA[1])); // MFMA D -> vector reader
        static_for<0, 16>([&](auto ic) {
            relu_piece(c0, ic, integral_constant<int, 0>{}, g);
            relu_piece(c0, ic, integral_constant<int, 1>{}, g);
        });
        static_for<0, 8>([&](auto ic) { acc_init(c0, ic); });
    }
    __syncthreads();
    b_request(0, integral_constant<int, 0>{});
    b_request(0, integral_constant<int, 1>{});
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
    // ---- epilogue: set 1's last step
    {
        integral_constant<int, 1> c1;
        StepRegs g;
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(accB[0]), "+v"(accB[1]));
        static_for<0, 4>([&](auto fc) {
            static_for<0, 4>([&](auto ic) { l3_pair(c1, integral_constant<int, 4 * decltype(fc)::value + decltype(ic)::value>{}, g); });
            static_for<0, 3>([&](auto ic) { l3_mfma(fc, ic, g); });
        });
        asm volatile("s_nop 15\n\ts_nop 7" : "+v"(g.yacc));
        l3_store(c1, g);
        __syncthreads();
        static_for<0, S>([&](auto ic) { fin_request(c1, ic, g); });
        static_for<0, S>([&](auto nc) { fin_piece(c1, nc, g); });
        static_for<0, S>([&](auto nc) { cost_piece(c1, nc, g); });
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
                                partials + (size_t)record_slot(tile, rsc) * rsb, rsc);
    } // tiles
    MPPI_BX3_STAMP_END();
}

} // namespace mppi
