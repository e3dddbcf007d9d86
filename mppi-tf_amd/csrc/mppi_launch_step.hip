// mppi_launch_step.hip — instantiates k_step_pc (the whole step in one launch / the armed launch, mppi_step.hip.h) for ONE action
// dimension (-DMPPI_UNIT_A). Diagonal-Q quadratic cost, the step's one pass: the shapes mppi_capi.hip's step_shape_ok admits.
#include "mppi_handle.hip.h"
#include "mppi_step.hip.h"
#ifndef MPPI_UNIT_A
#error "compile with -DMPPI_UNIT_A=<action dimension 1..4> (mppi-tf_amd/build.py)"
#endif

template <int A, int NP, int NSLOT, int MODE>
static hipError_t launch_step_inst(mppi_handle *h, hipStream_t st, const mppi_step_launch *L)
{
    constexpr bool FUSE = (MODE & STEP_FUSE) != 0, ARM = (MODE & STEP_ARM) != 0, PRE = (MODE & STEP_PRE) != 0;
    constexpr int NW = NP + 1;
    const size_t lds = std::max(pc_lds_floats(A, NP) * 4 + ((ARM || PRE) ? sizeof(float) * (size_t)h->HA : 0), (size_t)h->pc_lds_min);
    const int nb = (h->K_local + 63) / 64;
    const int ncw = FUSE ? (h->HA + NW - 1) / NW : 0;
    const dim3 g(nb + ncw), b(64 * NW);
    const int bias = h->pc_bias >= 0 ? h->pc_bias : ((NSLOT * 4 * A <= 80) ? 0x033a : 0x0369); // as mppi_launch_pc.hip
    const int balance = (nb <= 4 * 256 && !h->pc_no_balance) ? (1 | (bias << 8)) : 0;
    StepArgs sa{};
    sa.recs = h->d_step_recs; sa.nb = nb; sa.nbp = 128; sa.seq = L->seq;
    sa.xslot = h->d_xslot; sa.decision = ARM ? h->d_decision : nullptr;
    sa.host_state = h->d_arm; sa.err = reinterpret_cast<unsigned *>(h->d_arm + 1);
    sa.soft_ticks = (long long)h->arm_us * 100ll;           // s_memrealtime: 100 MHz
    sa.hard_ticks = sa.soft_ticks + 50ll * 100000ll;        // + 50 ms: nothing in this kernel ever spins longer
    sa.U_in = L->U_in; sa.U_out = L->U_out; sa.u_out = L->u_out; sa.step_ctr = h->d_step; sa.dbg = h->d_dbg; sa.clip = h->d_clip;
    sa.neg_inv_lambda = h->hc.neg_inv_lambda; sa.a = h->a; sa.HA = h->HA;
    sa.ugr = L->ugr; sa.utag = L->utag; sa.step_index = L->step_index; sa.cu_ctr = h->d_cu_ctr; sa.ugr_out = L->ugr_out;
    if (PRE) sa.hard_ticks = 20ll * 100000ll; // 20 ms: a sequence that has not come by then never will (sticky error word)
    const DevConsts *dC = h->dC;
    const unsigned long long *stp = h->d_step;
    const void *fn = h->sigma_diag ? reinterpret_cast<const void *>(k_step_pc<A, NP, NSLOT, true, MODE>) : reinterpret_cast<const void *>(k_step_pc<A, NP, NSLOT, false, MODE>);
    if (hipError_t e = mppi_raise_lds_ceiling(fn, h->device, lds); e != hipSuccess) return e;
    if (h->sigma_diag) hipExtLaunchKernelGGL((k_step_pc<A, NP, NSLOT, true, MODE>), g, b, (uint32_t)lds, st, h->kev0, h->kev1, 0, dC, L->x_dev, L->U_in, stp, h->d_cost, h->d_part, 1, h->nbp, balance, sa);
    else hipExtLaunchKernelGGL((k_step_pc<A, NP, NSLOT, false, MODE>), g, b, (uint32_t)lds, st, h->kev0, h->kev1, 0, dC, L->x_dev, L->U_in, stp, h->d_cost, h->d_part, 1, h->nbp, balance, sa);
    return hipGetLastError();
}

hipError_t MPPI_CAT(mppi_launch_step_a, MPPI_UNIT_A)(MPPI_STEP_PARAMS)
{
    constexpr int AA = MPPI_UNIT_A;
    const int NG = (h->H + 3) / 4;
    if (L->mode & STEP_FUSE) { // <= 128 tiles: always the 6-wave workgroup
        if (h->pc_np != 5) return hipErrorInvalidValue;
        const bool small = NG <= 20;
        if (L->mode & STEP_PRE) { // the one-launch step, pre-launched: two grids in flight (they fit side by side: at most 2 x (128 + 33) workgroups)
            if (h->fuse_step != 2 && NG <= 21) return launch_step_inst<AA, 7, 3, STEP_FUSE | STEP_PRE>(h, st, L);
            return small ? launch_step_inst<AA, 5, 4, STEP_FUSE | STEP_PRE>(h, st, L) : launch_step_inst<AA, 5, 8, STEP_FUSE | STEP_PRE>(h, st, L);
        }
        if (L->mode & STEP_ARM) return small ? launch_step_inst<AA, 5, 4, STEP_FUSE | STEP_ARM>(h, st, L) : launch_step_inst<AA, 5, 8, STEP_FUSE | STEP_ARM>(h, st, L);
        // One workgroup per CU and 3/4 of the chip empty: SEVEN producer waves (two waves on every SIMD, chunks of 28 steps) publish the
        // horizon 1.4x sooner than five and the consumer's chain no longer waits for its last chunk — 9.23 -> 8.98 us per step at
        // configs[1], 8.37 -> 8.18 at K = 3000 / H = 50 (r05). H <= 84; MPPI_TUNE_FUSED_STEP = 2 keeps the six-wave workgroup.
        if (h->fuse_step != 2 && NG <= 21) return launch_step_inst<AA, 7, 3, STEP_FUSE>(h, st, L);
        return small ? launch_step_inst<AA, 5, 4, STEP_FUSE>(h, st, L) : launch_step_inst<AA, 5, 8, STEP_FUSE>(h, st, L);
    }
    if (L->mode & STEP_PRE) { // the pre-launched pipelined step: more than 128 tiles (below, the fused step is one launch already)
        if (h->pc_np == 3) return NG <= 18 ? launch_step_inst<AA, 3, 6, STEP_PRE>(h, st, L) : launch_step_inst<AA, 3, 11, STEP_PRE>(h, st, L);
        return NG <= 20 ? launch_step_inst<AA, 5, 4, STEP_PRE>(h, st, L) : launch_step_inst<AA, 5, 8, STEP_PRE>(h, st, L);
    }
    if (!(L->mode & STEP_ARM)) return hipErrorInvalidValue; // (the plain rollout is k_rollout_pc)
    if (h->pc_np == 3) return NG <= 18 ? launch_step_inst<AA, 3, 6, STEP_ARM>(h, st, L) : launch_step_inst<AA, 3, 11, STEP_ARM>(h, st, L);
    return NG <= 20 ? launch_step_inst<AA, 5, 4, STEP_ARM>(h, st, L) : launch_step_inst<AA, 5, 8, STEP_ARM>(h, st, L);
}
