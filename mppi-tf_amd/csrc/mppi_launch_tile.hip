// mppi_launch_tile.hip — instantiates k_rollout_tile for ONE action dimension (-DMPPI_UNIT_A): 3 tile sizes x 2 Q forms x
// 7 (noise source, mode) pairs = 42 kernels per object.
#include "mppi_handle.hip.h"
#ifndef MPPI_UNIT_A
#error "compile with -DMPPI_UNIT_A=<action dimension 1..4> (mppi-tf_amd/build.py)"
#endif

// kernel dispatch
template <int A, int R, bool QFULL, int SRC, int MODE>
static hipError_t launch_tile_inst(mppi_handle *h, hipStream_t st, const float *x_dev, const float *U_dev,
                                   const float *eps, float *cost, float *part, float *noise_out)
{
    auto kern = k_rollout_tile<A, R, QFULL, SRC, MODE>;
    if (hipError_t e = mppi_raise_lds_ceiling(reinterpret_cast<const void *>(kern), h->device, h->tile_lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(h->nb), dim3(kThreads), h->tile_lds, st, h->dC, x_dev, U_dev, eps, h->d_step, cost, part, noise_out, 1, h->nbp);
    return hipGetLastError();
}

template <int A, int R, bool QFULL>
static hipError_t launch_tile_ar(mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev,
                                 const float *eps, float *cost, float *part, float *noise_out)
{
#define MPPI_TILE_CASE(SRC, MODE) \
    if (src == SRC && mode == MODE) return launch_tile_inst<A, R, QFULL, SRC, MODE>(h, st, x_dev, U_dev, eps, cost, part, noise_out);
    MPPI_TILE_CASE(SRC_PHILOX, MODE_ROLLOUT)
    MPPI_TILE_CASE(SRC_HBM, MODE_ROLLOUT)
    MPPI_TILE_CASE(SRC_PHILOX, MODE_COSTS_GIVEN)
    MPPI_TILE_CASE(SRC_HBM, MODE_COSTS_GIVEN)
    MPPI_TILE_CASE(SRC_PHILOX, MODE_COST_ONLY)
    MPPI_TILE_CASE(SRC_HBM, MODE_COST_ONLY)
    MPPI_TILE_CASE(SRC_PHILOX, MODE_NOISE_ONLY)
#undef MPPI_TILE_CASE
    return hipErrorInvalidValue;
}

template <int A>
static hipError_t launch_tile_a(mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev,
                                const float *eps, float *cost, float *part, float *noise_out)
{
    const bool qf = h->hc.q_full != 0;
#define MPPI_R_CASE(RR)                                                                                       \
    if (h->R == RR) return qf ? launch_tile_ar<A, RR, true>(h, st, src, mode, x_dev, U_dev, eps, cost, part, noise_out) \
                              : launch_tile_ar<A, RR, false>(h, st, src, mode, x_dev, U_dev, eps, cost, part, noise_out);
    MPPI_R_CASE(64)
    MPPI_R_CASE(32)
    MPPI_R_CASE(16)
#undef MPPI_R_CASE
    return hipErrorInvalidValue;
}

hipError_t MPPI_CAT(mppi_launch_tile_a, MPPI_UNIT_A)(MPPI_TILE_PARAMS)
{
    return launch_tile_a<MPPI_UNIT_A>(h, st, src, mode, x_dev, U_dev, eps, cost, part, noise_out);
}
