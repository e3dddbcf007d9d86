// mppi_step.hip.h — k_step_pc: the producer/consumer rollout (k_rollout_pc, mppi_kernels.hip.h) as a WHOLE control step in one
// launch and / or as an ARMED launch (r05). Same arithmetic, same Philox counters, same record algebra:
// sample costs and the update are bit-identical to the two-launch step (tests/test_step_gpu.py).
//
// STEP_FUSE  (nb <= 128 tiles, K <= 8192 — BASELINE configs[1] and the reference's own default K = 3000, H = 50): the finish rides in
//   the same grid. Workgroups [0, nb) are tiles; workgroups [nb, nb + ceil(H a / NW)) are COLUMN workgroups, one wavefront per
//   horizon-action column: it waits for every tile's record, combines them in k_finish_cols' order (same bits) and writes U', u and
//   the step counter. A tile's record travels as 8-byte {value, launch sequence} granules written by ONE write-through store each and
//   read with L1-bypassing loads (MI355X_MICROARCH.md, handoff-1to1): the tag IS the flag — no fence, no atomic, no counter. A column
//   wave polls ONE granule (one lane, with a sleep) until its sentinel tile has published and only then sweeps its 2 x 3 granules per
//   lane (tools/micro/chain_probe.hip: every thread sweeping from the start cost +3 us per step; with the sentinel the fused grid ties
//   two dependent launches on the GPU, 8.8 against 8.8 us at configs[1]'s shape, and halves the host's launch work per step).
// STEP_ARM  the launch happens BEFORE x is known (mppi_next arms step n+1 while step n's control goes back to the plant): producers
//   draw the whole horizon's noise and park the first two chunks of perturbed actions in LDS, the consumer wave watches an x slot in
//   fine-grained device memory that the host stores into directly (large BAR) — {x_i, seq} granules again — and starts the recurrence
//   the moment they arrive. What a synchronous step then pays is the granules' flight and the recurrence, not launch + dispatch + Philox
//   (chain_probe: host round trip 2.8 us armed against 8.2 us launch-after-write).
//   Who decides: ONLY tile 0's consumer. It accepts (x complete before its soft deadline) or aborts (deadline passed, or the host
//   stored the cancel tag) and says so in a decision granule (device) and in a pinned host word. Every other wave takes x from the host
//   slot as soon as it is complete but gives up only on tile 0's abort (or a hard deadline, which raises a sticky error): a late x that
//   races with the abort makes some tiles roll out for nothing, never a half-applied step — the column waves / the finish kernel apply
//   the update only under tile 0's accept. Every spin has a wall-clock bound: a silent host is a timeout, never a hang.
// STEP_PRE  (second session of r05) the PRE-LAUNCHED pipelined step: the launch of step n+1 goes to the handle's SECOND stream while step n
//   still runs on the first. Its workgroups take the CU slots step n's workgroups leave, the producers draw the whole horizon's noise —
//   half of a step's priced cycles, and the only part that does not depend on step n — and the consumer wave waits for the sequence U' of
//   step n, which k_finish_cols publishes as {value, tag} granules (FinishPre, mppi_kernels.hip.h); then the chunks are published from
//   registers + U and the recurrence runs as in k_rollout_pc. What it hides: the two dependent dispatches of a step (tools/two_controllers.py:
//   independent controllers fill them — 18.2 -> 14.2 us per step). The finish kernel fits beside four resident rollout workgroups on a CU
//   (104 x 4 + 32 of 512 VGPRs), so a waiting grid cannot starve it; the wait is bounded by the hard deadline (sticky error word) like
//   every other spin here. Records go out as plain floats for k_finish_cols; the Philox step index comes from the host's mirror.
#pragma once
#include "mppi_kernels.hip.h"

namespace mppi {

typedef unsigned long long u64;
enum { STEP_FUSE = 1, STEP_ARM = 2, STEP_PRE = 4 };
constexpr unsigned kArmAccept = 1u, kArmAbort = 2u, kArmCancelBit = 0x80000000u; // tags are 31-bit launch sequence numbers; bit 31 = the host's cancel

struct StepArgs {
    u64 *recs;               // FUSE: record granules, element (col, slot) at recs[col * nbp + slot], col 0 beta, 1 eta, 2 + c V[c]
    int nb, nbp;             // tiles; record slots (record_pad(nb) = 128 for FUSE)
    unsigned seq;            // launch sequence number: the tag of everything this launch publishes and accepts
    const u64 *xslot;        // ARM: x granules [s], stored by the host (fine-grained device memory)
    u64 *decision;           // ARM: tile 0's verdict {kArmAccept | kArmAbort, seq} (device)
    u64 *host_state;         // ARM: the same word in pinned host memory
    unsigned *err;           // sticky error word (pinned host memory): a hard deadline passed
    long long soft_ticks, hard_ticks; // 100 MHz ticks from the wave's start
    // the update (FUSE)
    const float *U_in;
    float *U_out, *u_out;
    unsigned long long *step_ctr;
    float *dbg;
    const float *clip;
    float neg_inv_lambda;
    int a, HA;
    // PRE: the sequence of THIS step as granules [HA] (written by the previous step's finish, tag utag); the Philox step index
    const u64 *ugr;
    unsigned utag;
    unsigned long long step_index;
    int *cu_ctr;             // PRE: arrivals per CU [4096] (key: XCC_ID << 8 | HW_ID's se, sh, cu), never reset
    u64 *ugr_out;            // PRE + FUSE: where the column waves leave the SHIFTED U' as granules tagged seq (the next step's ugr)
};

__device__ __forceinline__ void gr_store(u64 *p, float v, unsigned seq)
{
    __hip_atomic_store(p, ((u64)seq << 32) | (u64)__float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // global_store_dwordx2 sc1
}
__device__ __forceinline__ u64 gr_load(const u64 *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ bool wave_all(bool p) { return __builtin_amdgcn_ballot_w64(p) == __builtin_amdgcn_ballot_w64(true); }
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }

// The consumer wave of an armed tile waits for x. -> true: x[] holds the state (wave-uniform); false: the step is off (abort, error).
// decider = tile 0: the only wave that looks at the clock's soft deadline and at the host's cancel tag, and the only writer of the verdict.
template <int S>
__device__ __forceinline__ bool arm_wait(const StepArgs &sa, bool decider, int lane, float (&x)[S])
{
    const long long t0 = wall_clock64();
    const u64 *px = sa.xslot + (lane < S ? lane : S - 1);
    bool go = false;
    unsigned xbits = 0u;
    for (unsigned it = 0;; ++it) {
        const u64 g = __hip_atomic_load(px, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); // the host's stores arrive over the BAR: bypass every cache
        const unsigned tag = (unsigned)(g >> 32);
        if (wave_all(tag == sa.seq)) { xbits = (unsigned)g; go = true; break; }
        if (decider) {
            if (wave_any(tag == (sa.seq | kArmCancelBit))) break;
            if (wall_clock64() - t0 > sa.soft_ticks) break;
        } else {
            const u64 d = gr_load(sa.decision);
            if ((unsigned)(d >> 32) == sa.seq && (unsigned)d == kArmAbort) break;
            if ((it & 15u) == 15u && wall_clock64() - t0 > sa.hard_ticks) {
                if (lane == 0) __hip_atomic_store(sa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                break;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
    if (decider && lane == 0) {
        const u64 d = ((u64)sa.seq << 32) | (u64)(go ? kArmAccept : kArmAbort);
        __hip_atomic_store(sa.decision, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sa.host_state, d, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = __uint_as_float((unsigned)__builtin_amdgcn_readlane((int)xbits, i));
    return go;
}

#ifndef MPPI_PRE_NOWAIT
#define MPPI_PRE_NOWAIT 0 // (timing study, tools/build_unit_variant.py: 1 = take whatever the granules hold — what the wait itself costs)
#endif
// The consumer wave of a pre-launched tile waits for the sequence: lane 0 watches ONE granule (the last row's: any would do — every tile
// then sweeps them all) with a sleep, then the wave sweeps the HA granules (lane l: l, l + 64, ...) until every tag is this step's and parks
// the values in LDS. -> false: the hard deadline passed (sticky error word; the tile exits without a record).
__device__ __forceinline__ bool pre_wait(const StepArgs &sa, int lane, float *U_s)
{
    const long long t0 = wall_clock64();
    const int HA = sa.HA;
    for (unsigned it = 0;; ++it) {
        u64 g = 0ull;
        if (lane == 0) g = gr_load(sa.ugr + (HA - 1));
        if ((unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32)) == sa.utag || MPPI_PRE_NOWAIT) break;
        if ((it & 7u) == 7u && wall_clock64() - t0 > sa.hard_ticks) {
            if (lane == 0) __hip_atomic_store(sa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(8);
    }
    for (unsigned it = 0;; ++it) {
        bool ok = true;
        for (int i = lane; i < HA; i += 64) {
            const u64 g = gr_load(sa.ugr + i);
            ok = ok && (unsigned)(g >> 32) == sa.utag;
            U_s[i] = __uint_as_float((unsigned)g);
        }
        if (wave_all(ok) || MPPI_PRE_NOWAIT) return true;
        if ((it & 7u) == 7u && wall_clock64() - t0 > sa.hard_ticks) {
            if (lane == 0) __hip_atomic_store(sa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            return false;
        }
        __builtin_amdgcn_s_sleep(2);
    }
}

// One wavefront = one horizon-action column c of the fused step: (beta, eta, V_c) over the nbp = 128 record slots in k_finish_cols'
// order — column_combine with 256 threads gives thread t < 128 slot t, sums wave 0 (slots 0..63) and wave 1 (slots 64..127) on the DPP
// ladder and adds the four wave totals in order; here lane l holds slots l and l + 64 and the two ladders run in the one wave: the
// same double additions in the same association, so the same bits. Slots no tile owns are the neutral record (kPadBeta, 0, 0).
__device__ __forceinline__ void step_column(const StepArgs &sa, int c, int lane)
{
    const int nbp = sa.nbp, nb = sa.nb;
    // (pre-launched: the sequence and the step counter as k_finish_cols takes them under FinishPre — nothing here was written by the other stream's grid
    // as plain stores; the granule is read BELOW, once this grid's tiles have published: they could only after every granule of the sequence had come —
    // a column wave is resident long before the previous step's column waves have written them)
    float u_old = sa.ugr != nullptr ? 0.0f : sa.U_in[c];
    const unsigned long long step_old = sa.ugr != nullptr ? sa.step_index : sa.step_ctr[0];
    float lo = -INFINITY, hi = INFINITY;
    if (sa.clip != nullptr) { lo = sa.clip[c % sa.a]; hi = sa.clip[sa.a + c % sa.a]; }
    const u64 *pb = sa.recs, *pe = sa.recs + nbp, *pv = sa.recs + (size_t)(2 + c) * nbp;
    const long long t0 = wall_clock64();
    MPPI_COL_STAMP(sa, c, lane, 0);
    // sentinel: lane 0 polls ONE granule and the verdict — the beta of tile c mod nb, which its consumer stores at the tile's soft-min,
    // a butterfly ahead of the V columns: the sweeps below then poll for well under a microsecond, and only once their tile is nearly done
    const int sent = record_slot(c % nb, nbp);
    bool off = false;
    for (unsigned it = 0;; ++it) {
        u64 g = 0ull, d = 0ull;
        if (lane == 0) { g = gr_load(pb + sent); if (sa.decision != nullptr) d = gr_load(sa.decision); }
        const unsigned gtag = (unsigned)__builtin_amdgcn_readfirstlane((int)(g >> 32));
        const unsigned dtag = (unsigned)__builtin_amdgcn_readfirstlane((int)(d >> 32)), dval = (unsigned)__builtin_amdgcn_readfirstlane((int)d);
        if (gtag == sa.seq) break;
        if (dtag == sa.seq && dval == kArmAbort) { off = true; break; }
        if ((it & 7u) == 7u && wall_clock64() - t0 > sa.hard_ticks) {
            if (lane == 0) __hip_atomic_store(sa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            off = true;
            break;
        }
        __builtin_amdgcn_s_sleep(4);
    }
    if (off) return;
    MPPI_COL_STAMP(sa, c, lane, 1);
    // sweep: the lane's two slots x (beta, eta, V), repeated until every tag of a real slot matches
    const int q = nbp >> 3;
    float bb[2], ee[2], vv[2];
    bool real[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) { const int s = lane + 64 * i; real[i] = ((s % q) * 8 + s / q) < nb; } // record_slot's inverse
    for (unsigned it = 0;; ++it) {
        u64 g[6];
#pragma unroll
        for (int i = 0; i < 2; ++i) { g[3 * i] = gr_load(pb + lane + 64 * i); g[3 * i + 1] = gr_load(pe + lane + 64 * i); g[3 * i + 2] = gr_load(pv + lane + 64 * i); }
        bool ok = true;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) ok = ok && (!real[i] || (unsigned)(g[3 * i + j] >> 32) == sa.seq);
        if (wave_all(ok)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                bb[i] = real[i] ? __uint_as_float((unsigned)g[3 * i]) : kPadBeta;
                ee[i] = real[i] ? __uint_as_float((unsigned)g[3 * i + 1]) : 0.0f;
                vv[i] = real[i] ? __uint_as_float((unsigned)g[3 * i + 2]) : 0.0f;
            }
            break;
        }
        if ((it & 7u) == 7u) {
            bool dead = wall_clock64() - t0 > sa.hard_ticks;
            if (sa.decision != nullptr) { const u64 d = gr_load(sa.decision); dead = dead || ((unsigned)(d >> 32) == sa.seq && (unsigned)d == kArmAbort); }
            if (wave_any(dead)) {
                if (lane == 0 && wall_clock64() - t0 > sa.hard_ticks) __hip_atomic_store(sa.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                return;
            }
        }
        __builtin_amdgcn_s_sleep(2);
    }
    MPPI_COL_STAMP(sa, c, lane, 2);
    if (sa.ugr != nullptr) u_old = __uint_as_float((unsigned)gr_load(sa.ugr + c));
    const float beta = wave_min(fminf(bb[0], bb[1]));
    double se[2], sv[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const float r = expf(sa.neg_inv_lambda * (bb[i] - beta));
        se[i] = 0.0; sv[i] = 0.0;
        se[i] += (double)r * (double)ee[i];
        sv[i] += (double)r * (double)vv[i];
        se[i] = wave_sum_d(se[i]);
        sv[i] = wave_sum_d(sv[i]);
    }
    double eta = se[0], V = sv[0];
    eta += se[1]; V += sv[1];
    eta += 0.0; V += 0.0; // (the two empty waves of the 256-thread layout)
    eta += 0.0; V += 0.0;
    if (lane == 0) {
        if (c == 0 && sa.dbg != nullptr) { sa.dbg[0] = beta; sa.dbg[1] = (float)eta; }
        const float un = fminf(fmaxf(u_old + (float)(V / eta), lo), hi);
        sa.U_out[c] = un;                // U' ; the next step reads U_out + a (the shifted sequence)
        if (c < sa.a) sa.u_out[c] = un;  // mGetNew
        if (c == 0) sa.step_ctr[0] = step_old + 1ull;
        if (sa.ugr_out != nullptr) { // column c is row c - a of the next step's sequence, its last a rows are zero (mShift / mInit0)
            const int dst = c >= sa.a ? c - sa.a : sa.HA - sa.a + c;
            gr_store(sa.ugr_out + dst, c >= sa.a ? un : 0.0f, sa.seq);
        }
    }
    MPPI_COL_STAMP(sa, c, lane, 3);
}

// The kernel. Grid: nb tile workgroups (+ ceil(HA / (NP + 1)) column workgroups under STEP_FUSE), 64 (NP + 1) threads, dynamic LDS
// pc_lds_floats(A, NP) * 4 (+ H A floats under STEP_ARM: the nominal sequence, staged before x arrives).
// Without STEP_FUSE the tile records go out as plain floats for k_finish_cols (partials, rsb, rsc: as k_rollout_pc).
template <int A, int NP, int NSLOT, bool DIAG, int MODE>
__global__ __launch_bounds__(64 * (NP + 1), ((MODE & STEP_FUSE) ? 2 : NP == 5 ? 3 : (NSLOT * 4 * A <= 80 ? NP + 1 : 2))) void k_step_pc( // (a fused grid has at most 128 tiles: one workgroup per CU; the 6-wave workgroup serves at most 512: two per CU)
    const DevConsts *__restrict__ C, const float *__restrict__ x_dev, const float *__restrict__ U_dev,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int rsb, const int rsc, const int balance, const StepArgs sa)
{
    constexpr bool FUSE = (MODE & STEP_FUSE) != 0, ARM = (MODE & STEP_ARM) != 0, PRE = (MODE & STEP_PRE) != 0;
    static_assert(!PRE || !ARM, "a pre-launched step is not an armed one");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    constexpr int S = 2 * A;
    constexpr int NW = NP + 1;
    constexpr int CS = 4 * NP;
    constexpr int SLOT = pc_slot_floats(A);
    constexpr bool PACKED = SLOT != A + 1 || A == 3;
    constexpr int CH = CS * SLOT * 64;
    typedef float slot_t __attribute__((ext_vector_type(SLOT == 2 ? 2 : 4)));
    constexpr int NREG = NSLOT * 4 * A;
    const int tid = threadIdx.x;
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lane = tid & 63;
    if constexpr (FUSE) {
        if ((int)blockIdx.x >= sa.nb) { // column workgroup: one wave per column
            const int c = ((int)blockIdx.x - sa.nb) * NW + wave_hw;
            if (c < sa.HA) step_column(sa, c, lane);
            return;
        }
    }
    const int H = C->H;
    const int K = C->K_local;
    const int NG = (H + 3) / 4;
    const int nch = (NG + NP - 1) / NP;
    float *buf = smem;
    float *w_s = smem;
    MPPI_TL_DECL(); // (timing-study builds only, mppi_ablate.hip.h: slots 0..9 the consumer, 10 + 8 p + {0 start, 1..4 chunk published, 5 weights, 6 stored} producer p; 100 MHz stamps)
    float *U_s = smem + 2 * CH; // ARM, PRE: the nominal sequence [H A]
    __shared__ int go_s;        // ARM: -1 no x yet, 1 x is here (set by the consumer the moment it sees it), 0 the step is off; PRE: 1 the sequence is here, 0 off

    // role placement: as k_rollout_pc (SIMD-true consumer when the 4 waves sit on 4 SIMDs; speed only)
    int gen = (int)(blockIdx.x >> 8);
    if constexpr (PRE) {
        // A pre-launched grid takes the slots the previous step's workgroups leave, in whatever order they leave them: blockIdx says nothing
        // about which workgroups share a CU (with gen = blockIdx >> 8 a CU got up to four consumers on ONE SIMD: 25.7 us per step). The CU
        // counts its arrivals instead: consecutive arrivals take consecutive roles, whichever step they belong to.
        __shared__ int slot_s;
        if (tid == 0) {
            const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
            slot_s = atomicAdd(sa.cu_ctr + ((((hw >> 8) & 0xffu) | ((xcc & 0xfu) << 8)) & 4095u), 1);
        }
        __syncthreads();
        gen = (int)((unsigned)slot_s % (unsigned)(NW == 4 ? 4 : 3));
    }
    int wave = (wave_hw + NW - gen % NW) % NW;
    if (NW == 4 && balance) {
        __shared__ int simd_s[4];
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_REG_HW_ID[5:4]
        if (lane == 0) simd_s[wave_hw] = simd;
        __syncthreads();
        const int s0 = simd_s[0], s1 = simd_s[1], s2 = simd_s[2], s3 = simd_s[3];
        if (((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15) wave = (simd + 4 - (gen & 3)) & 3;
        wave = __builtin_amdgcn_readfirstlane(wave);
    }
    const int k0 = blockIdx.x * 64;
    const bool valid = (k0 + lane) < K;
    const int slot_b = record_slot(blockIdx.x, FUSE ? sa.nbp : rsc);
    auto put = [&](int col, float v) { // element (tile, col) of the record
        if constexpr (FUSE) gr_store(sa.recs + (size_t)col * sa.nbp + slot_b, v, sa.seq);
        else partials[(size_t)slot_b * rsb + (size_t)col * rsc] = v;
    };
    if constexpr (PRE) {
        if (tid == 0) go_s = -1;
        __syncthreads();
    }
    if constexpr (ARM) { // the nominal sequence into LDS while nothing else can be done (scalar loads after x arrives would sit on the critical path)
        for (int i = tid; i < H * A; i += 64 * NW) U_s[i] = U_dev[i];
        if (tid == 0) go_s = -1;
        __syncthreads();
    }

    if (wave != 0) {
        // ------------------------------------------------------------------ producers
        const int p = wave - 1;
        const unsigned int gk = (unsigned int)C->k_offset + (unsigned int)(k0 + lane);
        const unsigned long long base = (PRE ? sa.step_index : step_ctr[0]) * (unsigned long long)NG; // (PRE: step n-1's finish has not run yet — the host's mirror)
        const unsigned long long seed = C->seed;
        float eps_r[NREG];
        PcProducerConsts<A> pcst;
        pcst.template load<DIAG>(C);
        const PcProducerConsts<A> *PC = &pcst;
        MPPI_STAMP_RT(10 + 8 * p);
        // the noise of horizon group g = NP i + p -> eps_r (zeros where the group does not exist)
        auto draw = [&](auto ic) {
            constexpr int i = decltype(ic)::value;
            const int g = NP * i + p;
            if (i < nch && g < NG) {
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
            } else {
#pragma unroll
                for (int r = 0; r < 4 * A; ++r) eps_r[i * 4 * A + r] = 0.0f;
            }
        };
        // the group's 4 (step, lane) slots — perturbed action u_t + eps and the action cost — into chunk buffer i & 1
        auto publish = [&](auto ic, const float (&ug)[4][A]) {
            constexpr int i = decltype(ic)::value;
            float *cb = buf + (i & 1) * CH + (size_t)(4 * p) * SLOT * 64;
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                float e[A], u[A], slot[SLOT];
#pragma unroll
                for (int j = 0; j < SLOT; ++j) slot[j] = 0.0f;
#pragma unroll
                for (int j = 0; j < A; ++j) {
                    u[j] = ug[tl][j];
                    e[j] = eps_r[(i * 4 + tl) * A + j];
                    slot[j] = u[j] + e[j]; // to_apply, controller_base.cpp:258
                }
                slot[A] = action_cost<A, DIAG>(PC, u, e);
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
        };
        if constexpr (PRE) {
            // what does not need the sequence: the horizon's noise into registers, group by group for as long as the sequence has not come
            // (this launch became resident when the previous step's workgroups retired: what it draws now, it draws while that step's finish
            // runs); then the consumer's word (B_go), and from there k_rollout_pc's order — the groups not drawn yet are drawn in front of
            // their chunk, while the consumer is already at work on the earlier ones. The consumer's barrier count is k_rollout_pc's.
            int drawn = 0; // slots [0, drawn) hold their noise
            static_for<0, NSLOT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (drawn == i && *static_cast<volatile int *>(&go_s) < 0) { draw(ic); drawn = i + 1; }
            });
            __syncthreads(); // B_go: the sequence is in U_s (or the step is off)
            if (*static_cast<volatile int *>(&go_s) <= 0) return;
            static_for<0, NSLOT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int g = NP * i + p;
                if (balance && !FUSE) pc_set_prio(i, nch, gen, balance);
                if (i >= drawn) draw(ic); // (also the zeros of a group beyond the horizon: phase C sums every register)
                if (i < nch) {
                    if (g < NG) {
                        float ug[4][A];
#pragma unroll
                        for (int tl = 0; tl < 4; ++tl) {
                            const int tt = min(4 * g + tl, H - 1);
#pragma unroll
                            for (int j = 0; j < A; ++j) ug[tl][j] = U_s[tt * A + j];
                        }
                        publish(ic, ug);
                    }
                    __syncthreads(); // chunk i published
                }
            });
        } else if constexpr (!ARM) {
            // k_rollout_pc's order: per horizon group the nominal actions (scalar loads, hidden behind the Philox rounds), the noise, the slots
            static_for<0, NSLOT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int g = NP * i + p;
                if (balance && !FUSE) pc_set_prio(i, nch, gen, balance);
                float ug[4][A];
                if (i < nch && g < NG) {
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        const int tt = min(4 * g + tl, H - 1);
#pragma unroll
                        for (int j = 0; j < A; ++j) ug[tl][j] = U_dev[tt * A + j];
                    }
                }
                draw(ic);
                if (i < nch) {
                    if (g < NG) publish(ic, ug);
                    MPPI_STAMP_RT(10 + 8 * p + 1 + (i < 4 ? i : 3));
                    __syncthreads(); // chunk i published
                }
            });
        } else {
            // armed: everything that does not need x — the first two chunks parked in the two LDS buffers, then the rest of the horizon's
            // noise for as long as x has not come (a host that answers at once finds the consumer started after two groups' worth of
            // Philox, not six; one that takes its time finds nothing left to draw)
            auto publish_lds = [&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const int g = NP * i + p;
                if (i < nch && g < NG) {
                    float ug[4][A];
#pragma unroll
                    for (int tl = 0; tl < 4; ++tl) {
                        const int tt = min(4 * g + tl, H - 1);
#pragma unroll
                        for (int j = 0; j < A; ++j) ug[tl][j] = U_s[tt * A + j];
                    }
                    publish(ic, ug);
                }
            };
            draw(std::integral_constant<int, 0>{});
            publish_lds(std::integral_constant<int, 0>{});
            if constexpr (NSLOT > 1) {
                draw(std::integral_constant<int, 1>{});
                publish_lds(std::integral_constant<int, 1>{});
            }
            int drawn = 2; // slots [0, drawn) hold their noise
            static_for<2, NSLOT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (drawn == i && *static_cast<volatile int *>(&go_s) < 0) { draw(ic); drawn = i + 1; }
            });
            __syncthreads(); // B_go: the consumer has x (or the step is off); chunks 0 and 1 are published
            if (go_s <= 0) return;
            // chunk i goes into the buffer chunk i - 2 leaves: one barrier per consumed chunk, nch - 1 in all (the consumer's count)
            static_for<2, NSLOT>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if (i >= drawn) draw(ic); // (also the zeros of a group beyond the horizon: phase C sums every register)
                if (i < nch) {
                    __syncthreads(); // chunk i - 2 consumed
                    publish_lds(ic);
                }
            });
            if (nch >= 2) __syncthreads(); // chunk nch - 2 consumed (the last chunk was published before it)
        }
        __syncthreads(); // weights published by the consumer
        MPPI_STAMP_RT(10 + 8 * p + 5);
        // phase C from registers: V_b[t,j] = sum_k e_k eps[k,t,j]  (mWeightedNoise, controller_base.cpp:188-192)
        const float w = w_s[lane];
#pragma unroll
        for (int r = 0; r < NREG; ++r) eps_r[r] = w * eps_r[r];
        float tot[(NREG + 63) / 64];
        MPPI_WAVE_TRANSPOSE_SUM(NREG, eps_r, tot, lane);
        const int colbase = lane_column(lane);
#pragma unroll
        for (int m = 0; m < (NREG + 63) / 64; ++m) {
            const int n = 64 * m + colbase;
            const int i = n / (4 * A), rem = n - i * (4 * A);
            const int tl = rem / A, j = rem - tl * A;
            const int t = 4 * (NP * i + p) + tl;
            if (n < NREG && t < H) put(2 + t * A + j, tot[m]);
        }
        MPPI_STAMP_RT(10 + 8 * p + 6);
    } else {
        // ------------------------------------------------------------------ consumer
        PcConsumerConsts<S> ccst;
        ccst.load(C);
        const PcConsumerConsts<S> *CC = &ccst;
        float x[S];
        if constexpr (ARM) {
            const bool go = arm_wait<S>(sa, blockIdx.x == 0, lane, x);
            if (lane == 0) *static_cast<volatile int *>(&go_s) = go ? 1 : 0; // (the producers look at it between two groups of noise)
            __syncthreads(); // B_go
            if (!go) return;
        } else if constexpr (PRE) {
            const bool go = pre_wait(sa, lane, U_s);
            if (lane == 0) *static_cast<volatile int *>(&go_s) = go ? 1 : 0;
#pragma unroll
            for (int i = 0; i < S; ++i) x[i] = x_dev[i];
            __syncthreads(); // B_go
            if (!go) return;
            __syncthreads(); // chunk 0 published
        } else {
            MPPI_STAMP_RT(0);
            MPPI_STAMP(60); // (shader-clock stamps 60 / 61 against the 100 MHz stamps 0 / 9: the clock the tile really ran at)
#pragma unroll
            for (int i = 0; i < S; ++i) x[i] = x_dev[i];
            __syncthreads(); // chunk 0 published
        }
        // With one workgroup per CU (a fused grid) the tile lasts as long as this wave's H-step chain, and a lone wave is bound by the
        // latency between dependent instructions (~8 cycles; tools/timeline.py: 158 cycles per step at a = 2 for ~20 instructions), not by
        // their number. So: the top priority for good; the steps in groups of four — the NEXT group's slots requested before this group
        // is computed (the LDS round trip leaves the chain), the four recurrence steps first, then their four state costs, which do not
        // depend on each other, then the running sum in the reference's order. The same operations on the same operands: the same bits.
        if constexpr (FUSE) __builtin_amdgcn_s_setprio(3);
        float c = 0.0f;
        MPPI_STAMP_RT(1);
        auto load_group = [&](const float *cb, int t0, float (&v)[4][A], float (&ac)[4]) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if constexpr (PACKED) {
                    const slot_t sv = *static_cast<const slot_t *>(__builtin_assume_aligned(cb + ((t0 + q) * 64 + lane) * SLOT, SLOT * 4));
#pragma unroll
                    for (int j = 0; j < A; ++j) v[q][j] = sv[j];
                    ac[q] = sv[A];
                } else {
#pragma unroll
                    for (int j = 0; j < A; ++j) v[q][j] = cb[((t0 + q) * (A + 1) + j) * 64 + lane];
                    ac[q] = cb[((t0 + q) * (A + 1) + A) * 64 + lane];
                }
            }
        };
        PmPack<A> pk;
        pk.load(CC);
        PmState<A> st;
        st.from(x);
        auto run_group = [&](const float (&v)[4][A], const float (&ac)[4]) {
            PmState<A> xs[4];
            float sc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                pm_step_packed<A>(pk, st, v[q]);
                xs[q] = st;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) sc[q] = state_cost_packed<A>(pk, xs[q]); // cost on the POST-step state
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float tmp = sc[q] + ac[q]; // Step_cost_result cost_base.cpp:49
                c = c + tmp;                     // path_cost        controller_base.cpp:268
            }
        };
        for (int ch = 0; ch < nch; ++ch) {
            if (balance && !FUSE) pc_set_prio(ch, nch, gen, balance, MPPI_PC_CONSUMER_BOOST);
            const float *cb = buf + (ch & 1) * CH;
            const int tend = min(CS, H - ch * CS);
            const int ng = tend >> 2;
            float va[4][A], aca[4], vb[4][A], acb[4];
            // (the prefetch is unconditional — past the last group it re-reads that group: a load under a runtime predicate would be
            // waited for on the spot)
            if (ng > 0) load_group(cb, 0, va, aca);
            for (int g = 0; g < ng;) {
                load_group(cb, 4 * min(g + 1, ng - 1), vb, acb);
                run_group(va, aca);
                if (++g >= ng) break;
                load_group(cb, 4 * min(g + 1, ng - 1), va, aca);
                run_group(vb, acb);
                ++g;
            }
            for (int tl = 4 * ng; tl < tend; ++tl) { // ragged tail of the last chunk (H not a multiple of 4)
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
                pm_step_packed<A>(pk, st, v);
                const float sc = state_cost_packed<A>(pk, st);
                const float tmp = sc + ac;
                c = c + tmp;
            }
            MPPI_STAMP_RT(2 + (ch < 4 ? ch : 3));
            if (ch + 1 < nch) __syncthreads(); // chunk ch consumed / chunk ch+1 published
        }
        c = c + state_cost_packed<A>(pk, st); // terminal: x_H counted a second time, :271-272
        MPPI_STORE_COST(valid, cost + k0 + lane, c);
        // tile-local mBeta / mExpArg / mExp / mNabla (controller_base.cpp:166-182)
        const float beta = wave_min(valid ? c : INFINITY);
        const float arg = CC->neg_inv_lambda * (c - beta);
        const float ek = valid ? expf(arg) : 0.0f;
        const float eta = wave_sum(ek);
        w_s[lane] = ek;
        if (lane == 0) { put(0, beta); put(1, eta); }
        MPPI_STAMP_RT(9);
        MPPI_STAMP(61);
        __syncthreads(); // weights published
        MPPI_TL_DUMP(valid, cost + k0 + lane);
    }
}

} // namespace mppi
