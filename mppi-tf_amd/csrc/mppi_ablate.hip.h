// mppi_ablate.hip.h — the timing-study layer of the rollout kernels, in ONE place (VERDICT r03 housekeeping).
// The shipped library defines NONE of the symbols below: every macro then expands to the plain code and the kernels read without
// a single #if. The study builds (mppi-tf_amd/build.py build_variant, driven by tools/ablate.py and tools/timeline.py) define one of
//   MPPI_ABLATE_PHILOX    a cheap arithmetic stand-in for Philox + Box-Muller
//   MPPI_PHILOX_BLOCK_MAJOR  the Philox blocks of a horizon group one after the other (r03's order; same numbers, A/B timing of r04's round-major order)
//   MPPI_ABLATE_ROLLOUT   one recurrence step instead of H
//   MPPI_ABLATE_WSUM      no weighted-noise sum / transposing butterfly
//   MPPI_FINISH_STAGE=n   k_finish_cols stops after stage n (0 entry, 1 record loads, 2 the min over the records)
//   MPPI_PC_TIMELINE      every wave of k_rollout_pc stamps s_memtime / s_memrealtime and HW_ID at its phase boundaries into LDS
//                         and the consumer dumps the 64 stamps in place of the tile's costs
// Results of such builds are meaningless; only their kernel times and stamps count (profiles/r03_pc_ablation.txt, r03_finish_stages.txt).
#pragma once

// ---- noise stand-in --------------------------------------------------------------------------------------------------------
#if defined(MPPI_ABLATE_PHILOX)
#define MPPI_ABL_STANDIN_(z, n, gk, grp) \
    _Pragma("unroll") for (int j_ = 0; j_ < (n); ++j_) (z)[j_] = (float)((int)(((gk) * 2654435761ull + (grp) * 40503ull + j_) & 1023) - 512) * (1.0f / 512.0f)
#define MPPI_NORMALS_GROUP(A, seed, gk, grp, z) MPPI_ABL_STANDIN_(z, 4 * (A), gk, grp)
#define MPPI_NORMALS_GROUP_UB(A, seed, gk, grp, z) MPPI_ABL_STANDIN_(z, 4 * (A), gk, grp)
#elif defined(MPPI_PHILOX_BLOCK_MAJOR) // A/B: the Philox blocks of a group one after the other (r03) instead of round-major (r04)
#define MPPI_NORMALS_GROUP(A, seed, gk, grp, z) normals_group<A>(seed, gk, grp, z)
#define MPPI_NORMALS_GROUP_UB(A, seed, gk, grp, z) normals_group_ub_block_major<A>(seed, gk, grp, z)
#else
#define MPPI_NORMALS_GROUP(A, seed, gk, grp, z) normals_group<A>(seed, gk, grp, z)
#define MPPI_NORMALS_GROUP_UB(A, seed, gk, grp, z) normals_group_ub<A>(seed, gk, grp, z)
#endif

// ---- recurrence length -----------------------------------------------------------------------------------------------------
#if defined(MPPI_ABLATE_ROLLOUT)
#define MPPI_ABL_STEPS(n) 1                         /* k_rollout_tile: one step */
#define MPPI_ABL_CHUNK_STEPS(n, ch) ((ch) == 0 ? 1 : 0) /* k_rollout_pc: one step of the first chunk */
#else
#define MPPI_ABL_STEPS(n) (n)
#define MPPI_ABL_CHUNK_STEPS(n, ch) (n)
#endif

// ---- weighted-noise sum ----------------------------------------------------------------------------------------------------
#if defined(MPPI_ABLATE_WSUM)
#define MPPI_ABL_WSUM_COLS(n) 1
#define MPPI_WAVE_TRANSPOSE_SUM(NREG, regs, tot, lane) \
    _Pragma("unroll") for (int m_ = 0; m_ < ((NREG) + 63) / 64; ++m_) (tot)[m_] = (regs)[m_]
#else
#define MPPI_ABL_WSUM_COLS(n) (n)
#define MPPI_WAVE_TRANSPOSE_SUM(NREG, regs, tot, lane) wave_transpose_sum<NREG>(regs, tot, lane)
#endif

// ---- k_finish_cols stages --------------------------------------------------------------------------------------------------
#if defined(MPPI_FINISH_STAGE)
#define MPPI_FINISH_STOP(stage, beta_expr) \
    do { if (MPPI_FINISH_STAGE == (stage)) { beta_out = (beta_expr); eta_out = 1.0; V_out = 0.0; return; } } while (0)
#else
#define MPPI_FINISH_STOP(stage, beta_expr) do { } while (0)
#endif

// ---- k_rollout_pc phase timeline ---------------------------------------------------------------------------------------------
#if defined(MPPI_PC_TIMELINE)
__device__ __forceinline__ unsigned long long pc_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MPPI_TL_DECL() __shared__ float tl_s[64]; if (threadIdx.x < 64) tl_s[threadIdx.x] = 0.0f; __syncthreads()
#define MPPI_STAMP_RT(slot) do { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); __builtin_amdgcn_s_waitcnt(0xC07F); if (lane == 0 && (slot) < 64) tl_s[(slot)] = (float)(t_ & 0xFFFFFFull); } while (0)
#define MPPI_STAMP(slot) do { const unsigned long long t_ = pc_stamp(); if (lane == 0) tl_s[(slot)] = (float)(t_ & 0xFFFFFFull); } while (0)
// where this wave runs: HW_ID[15:0] (wave, simd, pipe, cu, sh, se) and XCC_ID[3:0]; role in slot
#define MPPI_TL_WHERE(wave) do { if (lane == 0) tl_s[48 + (wave)] = (float)(__builtin_amdgcn_s_getreg((15 << 11) | 4) | (__builtin_amdgcn_s_getreg((3 << 11) | 20) << 16)); } while (0)
#define MPPI_STORE_COST(valid, ptr, c) do { } while (0)
// let the producers finish and stamp (a few hundred cycles), then dump the stamps in place of the costs
#define MPPI_TL_DUMP(valid, ptr) do { for (int q_ = 0; q_ < 8; ++q_) __builtin_amdgcn_s_sleep(127); if (valid) *(ptr) = tl_s[lane]; } while (0)
// (k_step_pc's column wave 0: 100 MHz stamps of its start, its sentinel, its sweep and its store into the handle's debug words 2..5)
#define MPPI_COL_STAMP(sa, c, lane, i) do { if ((c) == 0 && (lane) == 0 && (sa).dbg != nullptr) (sa).dbg[2 + (i)] = (float)(__builtin_amdgcn_s_memrealtime() & 0xFFFFFFull); } while (0)
#else
#define MPPI_COL_STAMP(sa, c, lane, i) do { } while (0)
#define MPPI_TL_DECL() do { } while (0)
#define MPPI_STAMP_RT(slot) do { } while (0)
#define MPPI_STAMP(slot) do { } while (0)
#define MPPI_TL_WHERE(wave) do { } while (0)
#define MPPI_STORE_COST(valid, ptr, c) do { if (valid) *(ptr) = (c); } while (0)
#define MPPI_TL_DUMP(valid, ptr) do { } while (0)
#endif

// ---- k_rollout_mlp_bx3 (mppi_bx3.hip.h): cumulative cuts, ablations, the shader clock of workgroup 0 (profiles/r03_bx3_pieces.txt) ----
#ifndef MPPI_BX3_CUT
#define MPPI_BX3_CUT 1000 // the pieces behind MFMA m >= CUT are left out (the cumulative cost of a half-step's pieces)
#endif
#ifndef MPPI_BX3_ABL
#define MPPI_BX3_ABL 0 // bit set (wrong results): 1 no layer 3 / finish, 2 no preparation, 4 no barriers, 8 no layer-3 pieces, 16 no relu/split pieces, 32 no B-fragment reads in the stream
#endif
#ifdef MPPI_BX3_STAMP
#define MPPI_BX3_STAMP_BEGIN() const unsigned long long st_c0 = __builtin_amdgcn_s_memtime(), st_r0 = __builtin_amdgcn_s_memrealtime()
// shader clock against the 100 MHz reference: the clock the MFMA stream really ran at
#define MPPI_BX3_STAMP_END() \
    do { if (tid == 0 && blockIdx.x == 0) \
        printf("bx3 workgroup 0: %llu shader cycles in %llu ticks of 100 MHz = %.0f MHz\n", __builtin_amdgcn_s_memtime() - st_c0, \
               __builtin_amdgcn_s_memrealtime() - st_r0, 100.0 * (double)(__builtin_amdgcn_s_memtime() - st_c0) / (double)(__builtin_amdgcn_s_memrealtime() - st_r0)); } while (0)
#else
#define MPPI_BX3_STAMP_BEGIN() do { } while (0)
#define MPPI_BX3_STAMP_END() do { } while (0)
#endif

// ---- k_rollout_mlp2 (mppi_mlp2.hip.h): the in-kernel clock and the per-k-pair timeline of tools/micro/mlp2_bench.hip ----
#ifdef MPPI_MLP2_STAMP
__device__ unsigned long long g_mlp2_stamp[4 * 4096]; // per workgroup: s_memtime begin/end, s_memrealtime begin/end
#define MPPI_MLP2_STAMP_AT(which) \
    do { if (tid == 0 && blockIdx.x < 4096 && tile_is_first) { g_mlp2_stamp[4 * blockIdx.x + (which)] = __builtin_amdgcn_s_memtime(); \
                                                                 g_mlp2_stamp[4 * blockIdx.x + 2 + (which)] = __builtin_amdgcn_s_memrealtime(); } \
         if (which) tile_is_first = false; } while (0)
#else
#define MPPI_MLP2_STAMP_AT(which) do { } while (0)
#endif
#ifdef MPPI_MLP2_TRACE
// s_memtime after the k pairs listed in kMlp2TraceKp, in one steady-state half-step of workgroup 0 (4 waves)
constexpr int kMlp2TraceKp[] = {0, 4, 5, 6, 8, 9, 10, 28, 29, 32, 33, 34, 36, 37, 38, 44, 45, 46, 47, 48, 49, 50, 51, 66, 68, 69, 70, 100, 128};
constexpr int kMlp2TraceN = sizeof(kMlp2TraceKp) / sizeof(int);
__device__ unsigned long long g_mlp2_trace[4 * 32];
__host__ __device__ constexpr int mlp2_trace_slot(int kp)
{
    for (int i = 0; i < kMlp2TraceN; ++i)
        if (kMlp2TraceKp[i] == kp) return i;
    return -1;
}
#define MPPI_MLP2_TRACE_DECL() unsigned long long tr[kMlp2TraceN]
#define MPPI_MLP2_TRACE_AT(Q, kp) do { if constexpr ((Q) == 0 && mlp2_trace_slot(kp) >= 0) tr[mlp2_trace_slot(kp)] = __builtin_amdgcn_s_memtime(); } while (0)
#define MPPI_MLP2_TRACE_FLUSH(cond) do { if (cond) for (int i_ = 0; i_ < kMlp2TraceN; ++i_) g_mlp2_trace[w * 32 + i_] = tr[i_]; } while (0)
#else
#define MPPI_MLP2_TRACE_DECL() do { } while (0)
#define MPPI_MLP2_TRACE_AT(Q, kp) do { } while (0)
#define MPPI_MLP2_TRACE_FLUSH(cond) do { } while (0)
#endif

// ---- the two-wave pipelines (mppi_gen.hip.h, mppi_mlp32.hip.h): which wave a step waits for. Bit set (wrong results): 1 the network wave skips
// its Dense stack and output layer, 2 the pose / cost wave skips its per-step work (pose update, Euler angles, state cost) ----
#ifndef MPPI_PC_ABL
#define MPPI_PC_ABL 0
#endif

// ---- phase timeline of a two-wave pipeline (k_rollout_nnspeed_pc; tools/timeline_pc.py): both waves of a tile stamp s_memtime at their
// phase boundaries in steps MPPI_PCT_STEP and MPPI_PCT_STEP + 1 (slot = 16 * (step - MPPI_PCT_STEP) + 8 * wave kind + phase), and the cost
// wave writes the 64 stamps where the tile's costs go ----
#if defined(MPPI_PC_GEN_TIMELINE)
#ifndef MPPI_PCT_STEP
#define MPPI_PCT_STEP 20
#endif
__device__ __forceinline__ unsigned long long pct_stamp()
{
    unsigned long long t;
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
#define MPPI_PCT_DECL() __shared__ float pct_s[2][64]
#define MPPI_PCT(t, kind, phase) do { if ((t) == MPPI_PCT_STEP || (t) == MPPI_PCT_STEP + 1) { const unsigned long long t_ = pct_stamp(); \
        if (lane == 0) pct_s[pair][16 * ((t) - MPPI_PCT_STEP) + 8 * (kind) + (phase)] = (float)(t_ & 0xFFFFFFull); } } while (0)
#define MPPI_PCT_DUMP(valid, ptr, c) do { if (valid) *(ptr) = lane < 32 ? pct_s[pair][lane] : (c); } while (0)
#else
#define MPPI_PCT_DECL() do { } while (0)
#define MPPI_PCT(t, kind, phase) do { } while (0)
#define MPPI_PCT_DUMP(valid, ptr, c) do { if (valid) *(ptr) = (c); } while (0)
#endif

// ---- k_rollout_pc: the action-cost form resolved once around the producers' horizon loop (r05) or per step as before (A/B: tools/ablate.py kind_per_step) ----
#if defined(MPPI_PC_KIND_PER_STEP)
#define MPPI_PC_KIND_ONCE(cond) true
#define MPPI_PC_KIND_OF(k) (-1)
#else
#define MPPI_PC_KIND_ONCE(cond) (cond)
#define MPPI_PC_KIND_OF(k) (k)
#endif

// ---- k_rollout_pc: priority levels added to the consumer wave's progress level (A/B: tools/ablate.py consumer_boost) ----
#ifndef MPPI_PC_CONSUMER_BOOST
#define MPPI_PC_CONSUMER_BOOST 0
#endif

