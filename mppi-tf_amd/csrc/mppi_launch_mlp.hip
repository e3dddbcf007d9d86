// mppi_launch_mlp.hip — instantiates the learned-model rollout kernels (k_rollout_mlp, _mlp2, _mlp_bx3, _mlp32, _mlp_small) for
// ONE action dimension (-DMPPI_UNIT_A).
#include "mppi_handle.hip.h"
#ifndef MPPI_UNIT_A
#error "compile with -DMPPI_UNIT_A=<action dimension 1..4> (mppi-tf_amd/build.py)"
#endif

// learned-model rollouts (k_rollout_mlp): 64 rollouts per workgroup of 8 waves
template <int A>
static hipError_t launch_mlp_a(mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev,
                               const float *eps, float *cost)
{
    const size_t lds = (h->mlp_v2 ? mlp2_lds_floats(2 * A, A, h->H) : mlp_lds_floats(2 * A, A)) * 4;
    const dim3 g(h->mlp_v2 ? std::min(h->nb_mlp, h->n_cu) : h->nb_mlp), b(h->mlp_v2 ? kMlp2Threads : kMlpThreads);
    if (mode != MODE_ROLLOUT && mode != MODE_COST_ONLY) return hipErrorInvalidValue;
    if (h->mlp_small == 32 && h->mlp_bx3) { // ... on the bf16 matrix cores, every operand split in two (MPPI_FLAG_MLP_BF16X3)
        hipExtLaunchKernelGGL((k_rollout_mlp32_bx3<A>), dim3(h->nb_mlp), dim3(kMlp32Threads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC,
                              (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp);
        return hipGetLastError();
    }
    if (h->mlp_small == 32 && h->mlp32_valu == 0) { // default (r04): the two-wave pipeline (network wave + cost wave per tile, two tiles per workgroup)
        const int wgs = (h->nb_mlp + 1) / 2;
        const int balance = (wgs <= 2 * h->n_cu && !h->pc_no_balance) ? 1 : 0; // SIMD-true roles while the whole grid is resident in one round
        if (h->sigma_diag)
            hipExtLaunchKernelGGL((k_rollout_mlp32_pc<A, true>), dim3(wgs), dim3(kMlp32PcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC,
                                  (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp, h->nb_mlp, balance);
        else
            hipExtLaunchKernelGGL((k_rollout_mlp32_pc<A, false>), dim3(wgs), dim3(kMlp32PcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC,
                                  (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp, h->nb_mlp, balance);
        return hipGetLastError();
    }
    if (h->mlp_small == 32 && h->mlp32_valu == 2) { // MPPI_TUNE_MLP32_VALU = 2: one wave per 32 rollouts, 2 waves per tile (A/B timing)
        hipExtLaunchKernelGGL((k_rollout_mlp32<A>), dim3(h->nb_mlp), dim3(kMlp32Threads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC,
                              (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp);
        return hipGetLastError();
    }
    if (h->mlp_small) { // one wave = one 64-rollout tile, weights through the scalar cache
        const dim3 gs(h->nb_mlp), bs(64);
        if (h->mlp_small == 16)
            hipExtLaunchKernelGGL((k_rollout_mlp_small<A, 16>), gs, bs, 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const MlpDev *)h->dM, h->small_args,
                                  x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp);
        else
            hipExtLaunchKernelGGL((k_rollout_mlp_small<A, 32>), gs, bs, 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const MlpDev *)h->dM, h->small_args,
                                  x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp);
        return hipGetLastError();
    }
#define MPPI_MLP_L(KERN, BIT)                                                                                           \
    do {                                                                                                                \
        auto kern = KERN;                                                                                               \
        if (hipError_t e_ = mppi_raise_lds_ceiling(reinterpret_cast<const void *>(kern), h->device, lds); e_ != hipSuccess) return e_; \
        hipExtLaunchKernelGGL(kern, g, b, (uint32_t)lds, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const MlpDev *)h->dM, x_dev, U_dev, eps, \
                              (const unsigned long long *)h->d_step, cost, h->d_part, src, mode, 1, h->nbp);                \
    } while (0)
    if (h->mlp_bx3) { // one tile-walking workgroup of 4 waves per CU
        const size_t ldsp = bx3_lds_floats(2 * A, A, h->H) * 4;
        const dim3 gp(std::min(h->nb_mlp, h->n_cu)), bp(kBx3Threads);
#define MPPI_BX3P_L(KERN)                                                                                               \
    do {                                                                                                                \
        auto kern = KERN;                                                                                               \
        if (hipError_t e_ = mppi_raise_lds_ceiling(reinterpret_cast<const void *>(kern), h->device, ldsp); e_ != hipSuccess) return e_; \
        hipExtLaunchKernelGGL(kern, gp, bp, (uint32_t)ldsp, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const MlpDev *)h->dM, x_dev, U_dev, eps, \
                              (const unsigned long long *)h->d_step, cost, h->d_part, mode, 1, h->nbp);                     \
    } while (0)
        if (src == SRC_PHILOX) {
            if (h->sigma_diag) MPPI_BX3P_L((k_rollout_mlp_bx3<A, true, SRC_PHILOX>));
            else MPPI_BX3P_L((k_rollout_mlp_bx3<A, false, SRC_PHILOX>));
        } else if (src == SRC_HBM) {
            MPPI_BX3P_L((k_rollout_mlp_bx3<A, false, SRC_HBM>));
        } else return hipErrorInvalidValue;
#undef MPPI_BX3P_L
    } else if (h->mlp_v2) {
        if constexpr (A <= 3) {
#define MPPI_MLP2_L(KERN, BIT)                                                                                          \
    do {                                                                                                                \
        auto kern = KERN;                                                                                               \
        if (hipError_t e_ = mppi_raise_lds_ceiling(reinterpret_cast<const void *>(kern), h->device, lds); e_ != hipSuccess) return e_; \
        hipExtLaunchKernelGGL(kern, g, b, (uint32_t)lds, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const MlpDev *)h->dM, x_dev, U_dev, eps, \
                              (const unsigned long long *)h->d_step, cost, h->d_part, mode, 1, h->nbp);                     \
    } while (0)
            if (src == SRC_PHILOX) {
                if (h->sigma_diag) MPPI_MLP2_L((k_rollout_mlp2<A, true, SRC_PHILOX>), 32);
                else MPPI_MLP2_L((k_rollout_mlp2<A, false, SRC_PHILOX>), 64);
            } else if (src == SRC_HBM) { // injected noise (API helpers, tests): one instance, the dense-Sigma arithmetic
                MPPI_MLP2_L((k_rollout_mlp2<A, false, SRC_HBM>), 256); // (exact for a diagonal Sigma too: it adds 0 * z terms)
            } else return hipErrorInvalidValue;
#undef MPPI_MLP2_L
        } else return hipErrorInvalidValue;
    } else if (h->sigma_diag) MPPI_MLP_L((k_rollout_mlp<A, true>), 2);
    else MPPI_MLP_L((k_rollout_mlp<A, false>), 4);
#undef MPPI_MLP_L
    return hipGetLastError();
}

hipError_t MPPI_CAT(mppi_launch_mlp_a, MPPI_UNIT_A)(MPPI_MLP_PARAMS)
{
    return launch_mlp_a<MPPI_UNIT_A>(h, st, src, mode, x_dev, U_dev, eps, cost);
}
