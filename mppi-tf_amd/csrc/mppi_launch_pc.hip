// mppi_launch_pc.hip — instantiates k_rollout_pc (the hot configuration) for ONE action dimension (-DMPPI_UNIT_A).
#include "mppi_handle.hip.h"
#ifndef MPPI_UNIT_A
#error "compile with -DMPPI_UNIT_A=<action dimension 1..4> (mppi-tf_amd/build.py)"
#endif

// the hot configuration: producer/consumer kernel (k_rollout_pc) when the horizon fits its register file
template <int A, int NP, int NSLOT, int COST, int PASS>
static hipError_t launch_pc_pass(mppi_handle *h, hipStream_t st, const float *x_dev)
{
    const size_t lds = std::max(pc_lds_floats(A, NP) * 4, (size_t)h->pc_lds_min);
    const int nb = (h->K_local + 63) / 64;
    const dim3 g(nb), b(64 * (NP + 1));
    // tile records go out column-major ([2+HA][nb]): the finish kernel reads one column per workgroup
    // one round of workgroups (<= 4 per CU, all resident from the start): SIMD-true roles + progress priorities
    // bit 0: on; bits 8..23: the generations' head starts (pc_set_prio). Four workgroups per CU (the small-NSLOT instances): 10, 3, 3, 0 quarter chunks —
    // r05's sweep on three boxes (tools/tune_prio.py, profiles/r05_tune_prio.txt): kernel 15.05 -> 14.68 us at configs[2], 14.1 -> 13.4 at K = 49152, against
    // r04's 9, 6, 3, 0; two per CU (H > 80 at a = 3) keep r04's: the new ones cost 10 % there
    const int bias = h->pc_bias >= 0 ? h->pc_bias : ((NSLOT * 4 * A <= 80) ? 0x033a : 0x0369);
    const int balance = (nb <= 4 * 256 && !h->pc_no_balance) ? (1 | (bias << 8)) : 0;
    const DevConsts *dC = h->dC;
    const float *U = h->U_cur();
    const unsigned long long *stp = h->d_step;
    float *tile_mm = (PASS == PC_PASS_WEIGHTS && h->pc_range_given) ? nullptr : h->d_tile_mm; // (sharded: the agreed range is already in d_mm)
    if (hipError_t e = mppi_raise_lds_ceiling(h->sigma_diag ? reinterpret_cast<const void *>(k_rollout_pc<A, NP, NSLOT, true, COST, PASS>) : reinterpret_cast<const void *>(k_rollout_pc<A, NP, NSLOT, false, COST, PASS>), h->device, lds); e != hipSuccess) return e;
    if (h->sigma_diag) hipExtLaunchKernelGGL((k_rollout_pc<A, NP, NSLOT, true, COST, PASS>), g, b, (uint32_t)lds, st, h->kev0, h->kev1, 0, dC, x_dev, U, stp, h->d_cost, h->d_part, 1, h->nbp, balance, tile_mm, h->d_mm);
    else hipExtLaunchKernelGGL((k_rollout_pc<A, NP, NSLOT, false, COST, PASS>), g, b, (uint32_t)lds, st, h->kev0, h->kev1, 0, dC, x_dev, U, stp, h->d_cost, h->d_part, 1, h->nbp, balance, tile_mm, h->d_mm);
    return hipGetLastError();
}

// h->pc_pass: PC_PASS_PLAIN, or the two passes of normalizeCost (mppi_capi.hip sets it around its launches). The weights-only pass evaluates no
// state cost: ONE instance (the diagonal-Q one) serves every cost form.
template <int A, int NP, int NSLOT, int COST>
static hipError_t launch_pc_cost(mppi_handle *h, hipStream_t st, const float *x_dev)
{
    if (h->pc_pass == PC_PASS_WEIGHTS) return launch_pc_pass<A, NP, NSLOT, PC_COST_DIAG, PC_PASS_WEIGHTS>(h, st, x_dev);
    if (h->pc_pass == PC_PASS_COSTS) return launch_pc_pass<A, NP, NSLOT, COST, PC_PASS_COSTS>(h, st, x_dev);
    return launch_pc_pass<A, NP, NSLOT, COST, PC_PASS_PLAIN>(h, st, x_dev);
}

// the consumer's cost form: diagonal Q (the hot configuration), ElipseCost (elipse_cost.py:9-85; s >= 4), dense Q (static_cost.py:23-63)
template <int A, int NP, int NSLOT>
static hipError_t launch_pc_inst(mppi_handle *h, hipStream_t st, const float *x_dev)
{
    if constexpr (A >= 2) {
        if (h->hc.state_cost_kind == MPPI_STATE_COST_ELLIPSE) return launch_pc_cost<A, NP, NSLOT, PC_COST_ELLIPSE>(h, st, x_dev);
    }
    if (h->hc.q_full) return launch_pc_cost<A, NP, NSLOT, PC_COST_DENSE>(h, st, x_dev);
    // MPPI_FLAG_FP_CONTRACT: the step's one pass with fused multiply-adds (the two passes of normalizeCost keep the plain instances)
    if (h->fp_contract && h->pc_pass == PC_PASS_PLAIN) return launch_pc_pass<A, NP, NSLOT, PC_COST_DIAG_FMA, PC_PASS_PLAIN>(h, st, x_dev);
    return launch_pc_cost<A, NP, NSLOT, PC_COST_DIAG>(h, st, x_dev);
}

hipError_t MPPI_CAT(mppi_launch_pc_a, MPPI_UNIT_A)(MPPI_PC_PARAMS)
{
    constexpr int AA = MPPI_UNIT_A;
    const int NG = (h->H + 3) / 4;
    if (h->pc_np == 3) { // MPPI_PC_PRODUCERS=3: the 4-wave variant, kept for A/B timing
        const bool small = NG <= 18;
        return small ? launch_pc_inst<AA, 3, 6>(h, st, x_dev) : launch_pc_inst<AA, 3, 11>(h, st, x_dev);
    }
    const bool small = NG <= 20;
    return small ? launch_pc_inst<AA, 5, 4>(h, st, x_dev) : launch_pc_inst<AA, 5, 8>(h, st, x_dev);
}
