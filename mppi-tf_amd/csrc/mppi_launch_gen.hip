// mppi_launch_gen.hip — the 13-state AUV family (mppi_gen.hip.h): constants, launchers and helper kernels of handles whose
// model_base is AUVModel / NNAUVModel. One translation unit; mppi_capi.hip reaches it through the functions declared in
// mppi_handle.hip.h (mppi_gen_*), the handle keeps an opaque pointer to GenState.
#include "mppi_handle.hip.h"
#include "mppi_gen.hip.h"

#include <cmath>
#include <cstring>

struct GenState {
    GenConsts hg{};
    GenConsts *dG = nullptr;
};

static GenState *gs(const mppi_handle *h) { return static_cast<GenState *>(h->gen); }

// 6x6 inverse, Gauss-Jordan with partial pivoting in double, rounded once (tf.linalg.inv in fp64, auv_model.py:241)
static bool invert6(const double *A, double *out)
{
    double M[6][12];
    for (int i = 0; i < 6; ++i)
        for (int j = 0; j < 6; ++j) { M[i][j] = A[i * 6 + j]; M[i][6 + j] = i == j ? 1.0 : 0.0; }
    for (int c = 0; c < 6; ++c) {
        int p = c;
        for (int r = c + 1; r < 6; ++r) if (std::fabs(M[r][c]) > std::fabs(M[p][c])) p = r;
        if (std::fabs(M[p][c]) < 1e-300) return false;
        if (p != c) for (int j = 0; j < 12; ++j) std::swap(M[c][j], M[p][j]);
        const double piv = M[c][c];
        for (int j = 0; j < 12; ++j) M[c][j] /= piv;
        for (int r = 0; r < 6; ++r) {
            if (r == c) continue;
            const double f = M[r][c];
            if (f == 0.0) continue;
            for (int j = 0; j < 12; ++j) M[r][j] -= f * M[c][j];
        }
    }
    for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) out[i * 6 + j] = M[i][6 + j];
    return true;
}

// tensorflow_graphics quaternion.from_rotation_matrix on the host (fp32, the branch structure of the published algorithm;
// safe_unsigned_div(a, b) = a / (b + 10*FLT_MIN), eps_addition = 2*FLT_EPSILON) — elipse_cost.py:166
static void quat_from_rotation_matrix(const float *R, float *q)
{
    const float eps_add = 2.0f * 1.1920928955078125e-07f, eps_div = 10.0f * 1.1754943508222875e-38f;
    const float r00 = R[0], r01 = R[1], r02 = R[2], r10 = R[3], r11 = R[4], r12 = R[5], r20 = R[6], r21 = R[7], r22 = R[8];
    const float trace = (r00 + r11) + r22;
    if (trace > 0.0f) {
        const float sq = std::sqrt(trace + 1.0f) * 2.0f;
        q[3] = 0.25f * sq; q[0] = (r21 - r12) / (sq + eps_div); q[1] = (r02 - r20) / (sq + eps_div); q[2] = (r10 - r01) / (sq + eps_div);
    } else if (r00 > r11 && r00 > r22) {
        const float sq = std::sqrt((((1.0f + r00) - r11) - r22) + eps_add) * 2.0f;
        q[3] = (r21 - r12) / (sq + eps_div); q[0] = 0.25f * sq; q[1] = (r01 + r10) / (sq + eps_div); q[2] = (r02 + r20) / (sq + eps_div);
    } else if (r11 > r22) {
        const float sq = std::sqrt((((1.0f + r11) - r00) - r22) + eps_add) * 2.0f;
        q[3] = (r02 - r20) / (sq + eps_div); q[0] = (r01 + r10) / (sq + eps_div); q[1] = 0.25f * sq; q[2] = (r12 + r21) / (sq + eps_div);
    } else {
        const float sq = std::sqrt((((1.0f + r22) - r00) - r11) + eps_add) * 2.0f;
        q[3] = (r10 - r01) / (sq + eps_div); q[0] = (r02 + r20) / (sq + eps_div); q[1] = (r12 + r21) / (sq + eps_div); q[2] = 0.25f * sq;
    }
}

// Fill the 13-state constants from the config. Returns NULL on success, else the reason (MPPI_ERR_INVALID_ARG).
const char *mppi_gen_fill(mppi_handle *h, const mppi_config *cfg)
{
    GenState *g = new (std::nothrow) GenState();
    if (!g) return "out of host memory";
    h->gen = g;
    GenConsts &c = g->hg;
    c.rk = 1; c.dt = cfg->dt; c.damp_diag = 1;
    if (cfg->model_kind == MPPI_MODEL_AUV) {
        const mppi_auv_desc *d = cfg->auv;
        if (!d) return "the AUV model needs cfg.auv";
        if (!(d->mass > 0.0f) || !(d->volume > 0.0f) || !(d->density > 0.0f)) return "AUV: mass, volume and density have to be positive (auv_model.py:127-139)";
        if (d->rk != 1 && d->rk != 2 && d->rk != 4) return "AUV: rk must be 1, 2 or 4 (auv_model.py:282-306)";
        c.rk = d->rk;
        const float gravity = d->gravity > 0.0f ? d->gravity : 9.81f; // auv_model.py:236
        c.fng_z = (-d->mass) * gravity;
        c.fnb_z = (d->volume * d->density) * gravity;
        for (int i = 0; i < 3; ++i) { c.cog[i] = d->cog[i]; c.cob[i] = d->cob[i]; }
        // rigid-body mass [[m I, -m S(cog)], [m S(cog), I_b]] + added mass (auv_model.py:238-254), in double
        double M[36], Minv[36];
        const double m = d->mass, cg[3] = {d->cog[0], d->cog[1], d->cog[2]};
        const double S[9] = {0, -cg[2], cg[1], cg[2], 0, -cg[0], -cg[1], cg[0], 0};
        const double I[9] = {d->inertial[0], d->inertial[3], d->inertial[4], d->inertial[3], d->inertial[1], d->inertial[5],
                             d->inertial[4], d->inertial[5], d->inertial[2]}; // ixx iyy izz ixy ixz iyz
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                M[i * 6 + j] = i == j ? m : 0.0;
                M[i * 6 + 3 + j] = -(m * S[i * 3 + j]);
                M[(3 + i) * 6 + j] = m * S[i * 3 + j];
                M[(3 + i) * 6 + 3 + j] = I[i * 3 + j];
            }
        for (int i = 0; i < 36; ++i) M[i] += d->added_mass ? (double)d->added_mass[i] : 0.0;
        if (!invert6(M, Minv)) return "AUV: the total mass matrix is singular";
        for (int i = 0; i < 36; ++i) {
            c.mtot[i] = (float)M[i]; c.inv_mtot[i] = (float)Minv[i];
            c.lin_damp[i] = d->linear_damping ? d->linear_damping[i] : 0.0f;
            c.lin_damp_fwd[i] = d->linear_damping_forward_speed ? d->linear_damping_forward_speed[i] : 0.0f;
            if (i % 7 != 0 && (c.lin_damp[i] != 0.0f || c.lin_damp_fwd[i] != 0.0f)) c.damp_diag = 0;
        }
        for (int i = 0; i < 6; ++i) c.quad_damp[i] = d->quad_damping ? d->quad_damping[i] : 0.0f;
    }
    if (cfg->state_cost_kind == MPPI_STATE_COST_QUAT) {
        if (!cfg->quat_Q) return "StaticQuatCost needs cfg.quat_Q [10*10] (static_cost.py:92-100)";
        for (int i = 0; i < 100; ++i) c.q10[i] = cfg->quat_Q[i];
        for (int p = 0; p < 5; ++p)
            for (int j = 0; j < 10; ++j)
                for (int o = 0; o < 2; ++o) c.q10p[(p * 10 + j) * 2 + o] = c.q10[(2 * p + o) * 10 + j];
    }
    if (cfg->state_cost_kind == MPPI_STATE_COST_ELLIPSE3D) {
        const float *e = cfg->ellipse3d; // normal[3], aVec[3], axis[2], speed, mState, mVel
        if (!e) return "ElipseCost3D needs cfg.ellipse3d [11]";
        if (!(e[6] != 0.0f) || !(e[7] != 0.0f)) return "ElipseCost3D: the axes must be non-zero";
        // bVec = normal x aVec ; N = [aVec bVec normal] ; R = inv(N)^T (elipse_cost.py:141-167), inverse in double
        const float n[3] = {e[0], e[1], e[2]}, a[3] = {e[3], e[4], e[5]};
        const float b[3] = {n[1] * a[2] - n[2] * a[1], n[2] * a[0] - n[0] * a[2], n[0] * a[1] - n[1] * a[0]};
        double N[9], inv[9];
        for (int i = 0; i < 3; ++i) { N[i * 3] = a[i]; N[i * 3 + 1] = b[i]; N[i * 3 + 2] = n[i]; }
        const double det = N[0] * (N[4] * N[8] - N[5] * N[7]) - N[1] * (N[3] * N[8] - N[5] * N[6]) + N[2] * (N[3] * N[7] - N[4] * N[6]);
        if (std::fabs(det) < 1e-300) return "ElipseCost3D: normal and aVec must span a plane";
        inv[0] = (N[4] * N[8] - N[5] * N[7]) / det; inv[1] = (N[2] * N[7] - N[1] * N[8]) / det; inv[2] = (N[1] * N[5] - N[2] * N[4]) / det;
        inv[3] = (N[5] * N[6] - N[3] * N[8]) / det; inv[4] = (N[0] * N[8] - N[2] * N[6]) / det; inv[5] = (N[2] * N[3] - N[0] * N[5]) / det;
        inv[6] = (N[3] * N[7] - N[4] * N[6]) / det; inv[7] = (N[1] * N[6] - N[0] * N[7]) / det; inv[8] = (N[0] * N[4] - N[1] * N[3]) / det;
        float R[9];
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) R[i * 3 + j] = (float)inv[j * 3 + i];
        quat_from_rotation_matrix(R, c.e3_q);
        c.e3_axis[0] = e[6]; c.e3_axis[1] = e[7]; c.e3_axis[2] = 1.0f;
        c.e3_map[0] = -e[6] / e[7]; c.e3_map[1] = e[7] / e[6]; c.e3_map[2] = 0.0f;
        c.e3_gv = e[8]; c.e3_mS = e[9]; c.e3_mV = e[10];
    }
    return nullptr;
}

hipError_t mppi_gen_upload(mppi_handle *h)
{
    GenState *g = gs(h);
    if (!g->dG) { if (hipError_t e = hipMalloc((void **)&g->dG, sizeof(GenConsts)); e != hipSuccess) return e; }
    if (hipError_t e = hipMemcpyAsync(g->dG, &g->hg, sizeof(GenConsts), hipMemcpyHostToDevice, h->stream); e != hipSuccess) return e;
    return hipStreamSynchronize(h->stream);
}

void mppi_gen_destroy(mppi_handle *h)
{
    GenState *g = gs(h);
    if (!g) return;
    if (g->dG) (void)hipFree(g->dG);
    delete g;
    h->gen = nullptr;
}

// rollouts of a 13-state handle: one wave per 64-rollout tile
hipError_t mppi_launch_gen(mppi_handle *h, hipStream_t st, int src, int mode, const float *x_dev, const float *U_dev, const float *eps,
                           float *cost, float *part, float *noise_out)
{
    const GenState *g = gs(h);
    const dim3 grid(h->nb), block(64);
#define MPPI_GEN_L(MODEL, HID)                                                                                                     \
    do {                                                                                                                           \
        if (h->sigma_diag)                                                                                                         \
            hipExtLaunchKernelGGL((k_rollout_gen<MODEL, HID, true>), grid, block, 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG, \
                                  (const MlpDev *)h->dM, h->small_args, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, noise_out, \
                                  src, mode, 1, h->nbp);                                                                            \
        else                                                                                                                       \
            hipExtLaunchKernelGGL((k_rollout_gen<MODEL, HID, false>), grid, block, 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG, \
                                  (const MlpDev *)h->dM, h->small_args, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, noise_out, \
                                  src, mode, 1, h->nbp);                                                                            \
    } while (0)
#define MPPI_NNAUV32_L(KERN)                                                                                                       \
    hipExtLaunchKernelGGL(KERN, grid, dim3(kNnauv32Threads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG,        \
                          (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, src, mode, 1, h->nbp)
    // Dense(32) NNAUVModel: the matrix-core kernel (rollouts and cost-only passes; the record-from-given-costs / noise-export modes
    // and MPPI_TUNE_MLP32_VALU stay on the vector-ALU kernel)
    if (h->hc.model_kind == MPPI_MODEL_NN_AUV && h->mlp_small == 32 && h->mlp32_valu != 1 && (mode == MODE_ROLLOUT || mode == MODE_COST_ONLY) && noise_out == nullptr) {
        if (h->mlp_bx3) { // MPPI_FLAG_MLP_BF16X3: the bf16 matrix cores, every operand split in two
            if (h->sigma_diag) MPPI_NNAUV32_L(k_rollout_nnauv32_bx3<true>);
            else MPPI_NNAUV32_L(k_rollout_nnauv32_bx3<false>);
        } else if (h->mlp32_valu == 0) { // default (r04): the two-wave pipeline (network wave + cost wave per tile, two tiles per workgroup)
            const int wgs = (h->nb + 1) / 2;
            const int balance = (wgs <= 2 * h->n_cu && !h->pc_no_balance) ? 1 : 0; // (MPPI_TUNE_PC_BALANCE = 0: roles by wave index, A/B)
            if (h->sigma_diag)
                hipExtLaunchKernelGGL(k_rollout_nnauv_pc<true>, dim3(wgs), dim3(kNnauvPcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG,
                                      (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, src, mode, 1, h->nbp, h->nb, balance);
            else
                hipExtLaunchKernelGGL(k_rollout_nnauv_pc<false>, dim3(wgs), dim3(kNnauvPcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG,
                                      (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, src, mode, 1, h->nbp, h->nb, balance);
        } else { // MPPI_TUNE_MLP32_VALU = 2: one wave per 32 rollouts (A/B timing)
            if (h->sigma_diag) MPPI_NNAUV32_L(k_rollout_nnauv32<true>);
            else MPPI_NNAUV32_L(k_rollout_nnauv32<false>);
        }
        return hipGetLastError();
    }
    // NNAUVModelSpeed: the matrix-core kernel for rollouts and cost-only passes (r04); the other modes and MPPI_TUNE_MLP32_VALU stay
    // on the lane-per-rollout kernel
    if (h->hc.model_kind == MPPI_MODEL_NN_AUV_SPEED && h->mlp32_valu != 1 && (mode == MODE_ROLLOUT || mode == MODE_COST_ONLY) && noise_out == nullptr) {
        if (h->mlp32_valu == 2) { // MPPI_TUNE_MLP32_VALU = 2: the one-wave-per-32-rollouts matrix-core kernel (A/B timing)
            if (h->mlp_small == 16) { if (h->sigma_diag) MPPI_NNAUV32_L((k_rollout_nnspeed32<16, true>)); else MPPI_NNAUV32_L((k_rollout_nnspeed32<16, false>)); }
            else { if (h->sigma_diag) MPPI_NNAUV32_L((k_rollout_nnspeed32<32, true>)); else MPPI_NNAUV32_L((k_rollout_nnspeed32<32, false>)); }
            return hipGetLastError();
        }
        // default: the two-wave pipeline (network wave + pose wave per tile, two tiles per workgroup)
        const int wgs = (h->nb + 1) / 2;
        const int balance = (wgs <= 2 * h->n_cu && !h->pc_no_balance) ? 1 : 0; // (MPPI_TUNE_PC_BALANCE = 0: roles by wave index, A/B) // SIMD-true roles only while the whole grid is resident in one round
#define MPPI_NNSPEED_PC_L(KERN)                                                                                                    \
    hipExtLaunchKernelGGL(KERN, dim3(wgs), dim3(kNnspeedPcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG, \
                          (const MlpDev *)h->dM, x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, src, mode, 1, h->nbp, h->nb, balance)
        if (h->mlp_small == 16) { if (h->sigma_diag) MPPI_NNSPEED_PC_L((k_rollout_nnspeed_pc<16, true>)); else MPPI_NNSPEED_PC_L((k_rollout_nnspeed_pc<16, false>)); }
        else { if (h->sigma_diag) MPPI_NNSPEED_PC_L((k_rollout_nnspeed_pc<32, true>)); else MPPI_NNSPEED_PC_L((k_rollout_nnspeed_pc<32, false>)); }
#undef MPPI_NNSPEED_PC_L
        return hipGetLastError();
    }
#undef MPPI_NNAUV32_L
    // Fossen AUVModel: the two-wave pipeline (pose wave + velocity wave per tile, r04) for rollouts and cost-only passes; the other modes
    // and MPPI_TUNE_GEN_ONE_WAVE stay on the one-wave-per-tile kernel
    if (h->hc.model_kind == MPPI_MODEL_AUV && !h->gen_one_wave && (mode == MODE_ROLLOUT || mode == MODE_COST_ONLY) && noise_out == nullptr) {
        const int wgs = (h->nb + 1) / 2;
        const int balance = (wgs <= 2 * h->n_cu && !h->pc_no_balance) ? 1 : 0; // (MPPI_TUNE_PC_BALANCE = 0: roles by wave index, A/B)
        if (h->sigma_diag)
            hipExtLaunchKernelGGL(k_rollout_auv_pc<true>, dim3(wgs), dim3(kAuvPcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG,
                                  x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, src, mode, 1, h->nbp, h->nb, balance);
        else
            hipExtLaunchKernelGGL(k_rollout_auv_pc<false>, dim3(wgs), dim3(kAuvPcThreads), 0, st, h->kev0, h->kev1, 0, (const DevConsts *)h->dC, (const GenConsts *)g->dG,
                                  x_dev, U_dev, eps, (const unsigned long long *)h->d_step, cost, part, src, mode, 1, h->nbp, h->nb, balance);
        return hipGetLastError();
    }
    if (h->hc.model_kind == MPPI_MODEL_AUV) MPPI_GEN_L(GEN_MODEL_AUV, 32);
    else if (h->hc.model_kind == MPPI_MODEL_NN_AUV_SPEED) {
        if (h->mlp_small == 16) MPPI_GEN_L(GEN_MODEL_NNAUV_SPEED, 16);
        else MPPI_GEN_L(GEN_MODEL_NNAUV_SPEED, 32);
    } else if (h->mlp_small == 16) MPPI_GEN_L(GEN_MODEL_NNAUV, 16);
    else MPPI_GEN_L(GEN_MODEL_NNAUV, 32);
#undef MPPI_GEN_L
    return hipGetLastError();
}

const char *mppi_gen_kernel_name(const mppi_handle *h)
{
    const bool d = h->sigma_diag != 0; // the last template argument: exactly diagonal Sigma (as the profiler spells the instance)
    if (h->hc.model_kind == MPPI_MODEL_AUV) {
        if (!h->gen_one_wave) return d ? "mppi::k_rollout_auv_pc<true>" : "mppi::k_rollout_auv_pc<false>";
        return d ? "mppi::k_rollout_gen<0, 32, true>" : "mppi::k_rollout_gen<0, 32, false>";
    }
    if (h->hc.model_kind == MPPI_MODEL_NN_AUV_SPEED) {
        if (h->mlp32_valu == 0)
            return h->mlp_small == 16 ? (d ? "mppi::k_rollout_nnspeed_pc<16, true>" : "mppi::k_rollout_nnspeed_pc<16, false>")
                                      : (d ? "mppi::k_rollout_nnspeed_pc<32, true>" : "mppi::k_rollout_nnspeed_pc<32, false>");
        if (h->mlp32_valu == 2)
            return h->mlp_small == 16 ? (d ? "mppi::k_rollout_nnspeed32<16, true>" : "mppi::k_rollout_nnspeed32<16, false>")
                                      : (d ? "mppi::k_rollout_nnspeed32<32, true>" : "mppi::k_rollout_nnspeed32<32, false>");
        return h->mlp_small == 16 ? (d ? "mppi::k_rollout_gen<2, 16, true>" : "mppi::k_rollout_gen<2, 16, false>")
                                  : (d ? "mppi::k_rollout_gen<2, 32, true>" : "mppi::k_rollout_gen<2, 32, false>");
    }
    if (h->mlp_small == 32 && h->mlp32_valu != 1) {
        if (h->mlp_bx3) return d ? "mppi::k_rollout_nnauv32_bx3<true>" : "mppi::k_rollout_nnauv32_bx3<false>";
        if (h->mlp32_valu == 0) return d ? "mppi::k_rollout_nnauv_pc<true>" : "mppi::k_rollout_nnauv_pc<false>";
        return d ? "mppi::k_rollout_nnauv32<true>" : "mppi::k_rollout_nnauv32<false>";
    }
    return h->mlp_small == 16 ? (d ? "mppi::k_rollout_gen<1, 16, true>" : "mppi::k_rollout_gen<1, 16, false>")
                              : (d ? "mppi::k_rollout_gen<1, 32, true>" : "mppi::k_rollout_gen<1, 32, false>");
}

// NNAUVModel.build_step_graph in the reference's plain order (mul and add rounded separately, input index ascending, division by
// Xstd): the slow evaluation behind mppi_model_step, like k_mlp_step_ref. Reads the UNPADDED weights through MlpDev.
static __global__ void k_nnauv_step_ref(const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *__restrict__ x, int kx,
                                        const float *__restrict__ v, int k, float *__restrict__ scratch, float *__restrict__ out_next, int speed)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const int s = kGenS, a = kGenA, nin = speed ? kGenSpeedNin : kGenNin;
    float *cur = scratch + (size_t)i * 2 * 64, *nxt = cur + 64;
    const float *xi = x + (size_t)(kx == 1 ? 0 : i) * s;
    if (speed) { // NNAUVModelSpeed.prepare_data (nn_model.py:438-461): Euler angles, velocities, forces
        const float q[4] = {xi[3], xi[4], xi[5], xi[6]};
        float e[3];
        euler_from_quat(q, e);
        for (int j = 0; j < 3; ++j) cur[j] = (e[j] - M->xmean[j]) / M->xstd[j];
        for (int j = 0; j < 6; ++j) cur[3 + j] = (xi[7 + j] - M->xmean[3 + j]) / M->xstd[3 + j];
        for (int j = 0; j < a; ++j) cur[9 + j] = (v[(size_t)i * a + j] - M->xmean[9 + j]) / M->xstd[9 + j];
    } else {
        for (int j = 0; j < s - 3; ++j) cur[j] = (xi[3 + j] - M->xmean[j]) / M->xstd[j];
        for (int j = 0; j < a; ++j) cur[s - 3 + j] = (v[(size_t)i * a + j] - M->xmean[s - 3 + j]) / M->xstd[s - 3 + j];
    }
    int width = nin;
    for (int l = 0; l < M->n_layers; ++l) {
        const int out_w = M->widths[l];
        const float *W = M->Wl[l], *b = M->bl[l];
        const int ld = M->ld[l]; // row stride of W (the output layer is padded to an even width)
        for (int o = 0; o < out_w; ++o) {
            float acc = 0.0f;
            for (int j = 0; j < width; ++j) acc = acc + cur[j] * W[j * ld + o];
            acc = acc + b[o];
            nxt[o] = (l + 1 < M->n_layers && acc < 0.0f) ? 0.0f : acc;
        }
        float *t = cur; cur = nxt; nxt = t;
        width = out_w;
    }
    if (speed) { // next_state (:463-472)
        float xs[kGenS], delta[6];
        for (int o = 0; o < s; ++o) xs[o] = xi[o];
        for (int o = 0; o < 6; ++o) delta[o] = cur[o] * M->ystd[o] + M->ymean[o];
        nnauv_speed_next_state(C->dt, xs, delta);
        for (int o = 0; o < s; ++o) out_next[(size_t)i * s + o] = xs[o];
        return;
    }
    for (int o = 0; o < s; ++o) out_next[(size_t)i * s + o] = xi[o] + (cur[o] * M->ystd[o] + M->ymean[o]);
}

static __global__ void k_auv_step(const GenConsts *__restrict__ G, const float *__restrict__ x, int kx, const float *__restrict__ v, int k,
                                  float *__restrict__ out_next)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    float xs[kGenS], vs[kGenA];
    const float *xi = x + (size_t)(kx == 1 ? 0 : i) * kGenS;
#pragma unroll
    for (int j = 0; j < kGenS; ++j) xs[j] = xi[j];
#pragma unroll
    for (int j = 0; j < kGenA; ++j) vs[j] = v[(size_t)i * kGenA + j];
    auv_step(G, xs, vs);
#pragma unroll
    for (int j = 0; j < kGenS; ++j) out_next[(size_t)i * kGenS + j] = xs[j];
}

// x [kx,13], v [k,6] (device) -> next [k,13] (device); scratch: k*128 floats for the NNAUV model
hipError_t mppi_gen_model_step(mppi_handle *h, hipStream_t st, const float *x, int kx, const float *v, int k, float *scratch, float *out_next)
{
    const GenState *g = gs(h);
    const dim3 grid((k + 63) / 64), block(64);
    if (h->hc.model_kind == MPPI_MODEL_AUV) hipLaunchKernelGGL(k_auv_step, grid, block, 0, st, (const GenConsts *)g->dG, x, kx, v, k, out_next);
    else hipLaunchKernelGGL(k_nnauv_step_ref, grid, block, 0, st, (const DevConsts *)h->dC, (const MlpDev *)h->dM, x, kx, v, k, scratch, out_next,
                            h->hc.model_kind == MPPI_MODEL_NN_AUV_SPEED ? 1 : 0);
    return hipGetLastError();
}

hipError_t mppi_gen_costs(mppi_handle *h, hipStream_t st, const float *x, const float *u, const float *eps, int k, float *os, float *oa, float *ot)
{
    const GenState *g = gs(h);
    hipLaunchKernelGGL(k_gen_costs, dim3((k + 255) / 256), dim3(256), 0, st, (const DevConsts *)h->dC, (const GenConsts *)g->dG, x, u, eps, k, os, oa, ot);
    return hipGetLastError();
}

// AUVModel's pieces for k (state, action) pairs -> out [k, 124] (device pointers); see k_auv_pieces
hipError_t mppi_gen_auv_pieces(mppi_handle *h, hipStream_t st, const float *x, const float *u, int k, float *out)
{
    const GenState *g = gs(h);
    hipLaunchKernelGGL(k_auv_pieces, dim3((k + 63) / 64), dim3(64), 0, st, (const GenConsts *)g->dG, x, u, k, out);
    return hipGetLastError();
}

// ElipseCost3D's position / orientation / velocity error of k states (device pointers), out [k, 3]
hipError_t mppi_gen_e3_terms(mppi_handle *h, hipStream_t st, const float *x, int k, int in_plane, float *out)
{
    const GenState *g = gs(h);
    hipLaunchKernelGGL(k_e3_terms, dim3((k + 63) / 64), dim3(64), 0, st, (const GenConsts *)g->dG, x, k, in_plane, out);
    return hipGetLastError();
}
