// mppi_gen.hip.h — the 13-state AUV family in the model_base slot and the richer costs in the cost_base slot
// (SURVEY §8f row 4), as lane-per-rollout device code for gfx950:
//   models  AUVModel   (Fossen 6-DOF rigid body, quaternion attitude, rk1/rk2/"rk4")   models/auv_model.py:282-562
//           NNAUVModel (x' = x + denorm(nn(norm(concat(x[3:], u)))), Dense(16|32)x1..3)  models/nn_model.py:215-304
//           NNAUVModelSpeed (the network predicts the velocity delta from Euler angles, velocities, forces;
//                       the pose is integrated with the quaternion kinematics)           models/nn_model.py:307-588
//   costs   StaticCost     (x-g)^T Q (x-g), s = 13                                     costs/static_cost.py:40-63
//           StaticQuatCost d = (dp, 2 acos<q, g_q>, dv), d^T Q10 d                     costs/static_cost.py:73-159
//           ElipseCost3D   plane-frame pose, |sum (p/axis)^2 - 1| + tangent angle + | |v|^2 - speed^2 |   costs/elipse_cost.py:101-246
// State x = (pos[3], quat[x,y,z,w], lin vel[3], ang vel[3]), action = 6 generalised forces; s_dim = 13 and a_dim = 6 are
// NOT tied by s = 2a as in the point-mass kernels. Arithmetic in the reference's order, every operation rounded on its
// own (-ffp-contract=off), products with the structural zeros of the reference's dense matrices skipped (exact up to the
// sign of a zero) — the same contract as mppi_device.hip.h, checked by the parity tests against the CPU restatement.
//
// k_rollout_gen: one rollout per LANE, one wave = one 64-rollout tile = one (beta, eta, V) record, everything in
// registers; the wave-uniform model constants arrive through scalar loads (GenConsts in HBM / the weights through the
// scalar cache as in k_rollout_mlp_small). The tile record regenerates the noise from its Philox counters
// (mlp_tile_record): at 400-1500 vector instructions per model step that costs under 10 % and keeps the LDS free.
#pragma once

namespace mppi {

constexpr int kGenS = 13, kGenA = 6;
constexpr int kGenNin = kGenS + kGenA - 3; // NNAUVModel's input: the state without the position, then the action

enum GenModel { GEN_MODEL_AUV = 0, GEN_MODEL_NNAUV = 1, GEN_MODEL_NNAUV_SPEED = 2 };
constexpr int kGenSpeedNin = 15; // NNAUVModelSpeed: 3 Euler angles, 6 velocities, 6 forces

// Device-resident constants of the 13-state family (beside DevConsts, which keeps lambda, Sigma, goal, Q, seed ...).
struct GenConsts {
    // AUVModel (auv_model.py:85-245), matrices row-major [6x6]
    int rk;                 // 1, 2, 4
    int damp_diag;          // linear_damping and linear_damping_forward_speed are both diagonal (the config files' form)
    float dt;
    float fng_z, fnb_z;     // (-mass)*gravity ; (volume*density)*gravity              :452-456
    float cog[3], cob[3];
    float mtot[36];         // rigid body + added mass                                   :247-254
    float inv_mtot[36];     // its inverse (host, double Gauss-Jordan, rounded once)     :241
    float lin_damp[36], lin_damp_fwd[36], quad_damp[6];
    // StaticQuatCost: Q [10x10] (goal comes from DevConsts.goal[13])                    static_cost.py:92-100
    float q10[100];
    float q10p[100];        // the same, rows paired: q10p[(p*10 + j)*2 + o] = q10[(2p + o)*10 + j] (one 8-byte scalar load feeds v_pk_mul_f32)
    // ElipseCost3D after prepare_consts                                                  elipse_cost.py:141-167
    float e3_q[4], e3_axis[3], e3_map[3], e3_gv, e3_mS, e3_mV;
    __device__ __forceinline__ float lin_diag(int i) const { return lin_damp[i * 7]; }
    __device__ __forceinline__ float fwd_diag(int i) const { return lin_damp_fwd[i * 7]; }
    // a = M (v1, v2) as the reference splits it (M11 v1 + M12 v2 ; M21 v1 + M22 v2), and y = inv(M) r as a dense row sum
    __device__ __forceinline__ void mass_times(const float (&vel)[6], float (&a)[6]) const
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float s1 = (mtot[i * 6] * vel[0] + mtot[i * 6 + 1] * vel[1]) + mtot[i * 6 + 2] * vel[2];
            const float s2 = (mtot[i * 6 + 3] * vel[3] + mtot[i * 6 + 4] * vel[4]) + mtot[i * 6 + 5] * vel[5];
            a[i] = s1 + s2;
        }
    }
    __device__ __forceinline__ void inv_mass_times(const float (&r)[6], float (&y)[6]) const
    {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float acc = inv_mtot[i * 6] * r[0];
#pragma unroll
            for (int j = 1; j < 6; ++j) acc = acc + inv_mtot[i * 6 + j] * r[j];
            y[i] = acc;
        }
    }
};

// The Fossen model's constants as a kernel-local copy (same accessors as GenConsts, so auv_state_dot takes either): the two
// 6x6 matrices every state_dot multiplies with live in VECTOR registers (72 of the 256 a one-wave-per-SIMD kernel owns), as
// row pairs. Left to the scalar file — 280 constants for ~100 SGPRs — hipcc spills them into VGPR lanes and pays a
// v_readlane per use.
typedef float f32x2g __attribute__((ext_vector_type(2)));
struct AuvLocal {
    int rk, damp_diag;
    float dt, fng_z, fnb_z, cog[3], cob[3];
    f32x2g mt2[18], iv2[18];              // rows (2p, 2p+1) of column j side by side: [p*6 + j] = (M[2p][j], M[2p+1][j])
    const float *lin_damp, *lin_damp_fwd; // the dense forms stay behind scalar loads (damp_diag = 0 only)
    float ldd[6], lfd[6], quad_damp[6];   // diagonals of the linear damping matrices, the quadratic coefficients
    __device__ __forceinline__ void load(const GenConsts *__restrict__ G)
    {
        rk = G->rk; damp_diag = G->damp_diag; dt = G->dt; fng_z = G->fng_z; fnb_z = G->fnb_z;
#pragma unroll
        for (int i = 0; i < 3; ++i) { cog[i] = G->cog[i]; cob[i] = G->cob[i]; }
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                mt2[p * 6 + j] = f32x2g{G->mtot[(2 * p) * 6 + j], G->mtot[(2 * p + 1) * 6 + j]};
                iv2[p * 6 + j] = f32x2g{G->inv_mtot[(2 * p) * 6 + j], G->inv_mtot[(2 * p + 1) * 6 + j]};
                asm volatile("" : "+v"(mt2[p * 6 + j]), "+v"(iv2[p * 6 + j]));
            }
        lin_damp = G->lin_damp; lin_damp_fwd = G->lin_damp_fwd;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            ldd[i] = G->lin_damp[i * 7]; lfd[i] = G->lin_damp_fwd[i * 7]; quad_damp[i] = G->quad_damp[i];
            asm volatile("" : "+v"(ldd[i]), "+v"(lfd[i]), "+v"(quad_damp[i]));
        }
    }
    __device__ __forceinline__ float lin_diag(int i) const { return ldd[i]; }
    __device__ __forceinline__ float fwd_diag(int i) const { return lfd[i]; }
    // two rows per v_pk_mul_f32 / v_pk_add_f32, every row's operations and their order as in GenConsts' scalar forms
    __device__ __forceinline__ void mass_times(const float (&vel)[6], float (&a)[6]) const
    {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            const f32x2g s1 = (mt2[p * 6] * f32x2g{vel[0], vel[0]} + mt2[p * 6 + 1] * f32x2g{vel[1], vel[1]}) + mt2[p * 6 + 2] * f32x2g{vel[2], vel[2]};
            const f32x2g s2 = (mt2[p * 6 + 3] * f32x2g{vel[3], vel[3]} + mt2[p * 6 + 4] * f32x2g{vel[4], vel[4]}) + mt2[p * 6 + 5] * f32x2g{vel[5], vel[5]};
            const f32x2g t = s1 + s2;
            a[2 * p] = t.x; a[2 * p + 1] = t.y;
        }
    }
    __device__ __forceinline__ void inv_mass_times(const float (&r)[6], float (&y)[6]) const
    {
#pragma unroll
        for (int p = 0; p < 3; ++p) {
            f32x2g acc = iv2[p * 6] * f32x2g{r[0], r[0]};
#pragma unroll
            for (int j = 1; j < 6; ++j) acc = acc + iv2[p * 6 + j] * f32x2g{r[j], r[j]};
            y[2 * p] = acc.x; y[2 * p + 1] = acc.y;
        }
    }
};

// ---------------------------------------------------------------------------------------- AUVModel
// auv_model.py:353-398 body2inertial_transform
__device__ __forceinline__ void auv_b2i(const float (&q)[4], float (&rot)[9], float (&T)[12])
{
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    rot[0] = 1.0f - 2.0f * (y * y + z * z);
    rot[1] = 2.0f * (x * y - z * w);
    rot[2] = 2.0f * (x * z + y * w);
    rot[3] = 2.0f * (x * y + z * w);
    rot[4] = 1.0f - 2.0f * (x * x + z * z);
    rot[5] = 2.0f * (y * z - x * w);
    rot[6] = 2.0f * (x * z - y * w);
    rot[7] = 2.0f * (y * z + x * w);
    rot[8] = 1.0f - 2.0f * (x * x + y * y);
    const float t[12] = {w, -z, y, z, w, -x, -y, x, w, -x, -y, -z}; // rows rxt, ryt, rzt, rwt (:388-396)
#pragma unroll
    for (int i = 0; i < 12; ++i) T[i] = 0.5f * t[i];
}

__device__ __forceinline__ void cross3(const float (&a)[3], const float (&b)[3], float (&o)[3])
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

// dense row sum  y_i = sum_j M[i][j] x_j  in index order, first add dropped (0 + p = p up to the sign of a zero)
template <int N>
__device__ __forceinline__ void matvec_n(const float *__restrict__ M, const float (&x)[N], float (&y)[N])
{
#pragma unroll
    for (int i = 0; i < N; ++i) {
        float acc = M[i * N] * x[0];
#pragma unroll
        for (int j = 1; j < N; ++j) acc = acc + M[i * N + j] * x[j];
        y[i] = acc;
    }
}

// auv_model.py:308-351 state_dot = (J(eta) nu, invM (tau - C nu - D nu - g)); the zero blocks of the reference's dense products are skipped.
// In two halves, so that k_rollout_auv_pc can give them to two waves (same expressions, same order: what either kernel computes per
// component is what auv_state_dot computes):
//   auv_pose_rates  xd[0..6] = J(eta) nu: position rates rot . v_lin, quaternion rates T . v_ang          (needs q, all 6 velocities)
//   auv_vel_rates   xd[7..12] = invM (tau - C nu - D nu - g(eta))                                        (needs q — rot's third row —, vel, tau)
// pieces: NULL, or 18 floats that receive C nu [6], D nu [6], g [6] (what the reference's tests look at one by one)
__device__ __forceinline__ void auv_pose_rates(const float (&q)[4], const float (&vel)[6], float (&xd)[7])
{
    float rot[9], T[12];
    auv_b2i(q, rot, T);
    // J = [[rot, 0], [0, T]] (:335-351); the zero blocks contribute exact zeros
#pragma unroll
    for (int i = 0; i < 3; ++i) xd[i] = (rot[i * 3] * vel[0] + rot[i * 3 + 1] * vel[1]) + rot[i * 3 + 2] * vel[2];
#pragma unroll
    for (int i = 0; i < 4; ++i) xd[3 + i] = (T[i * 3] * vel[3] + T[i * 3 + 1] * vel[4]) + T[i * 3 + 2] * vel[5];
}

// the restoring forces and moments g(eta) (:450-480): f_g = R^T (0, 0, fng_z) = R[2][:]*fng_z ; f_b likewise ; moments cog x f_g, cob x f_b.
// A function of the quaternion alone: k_rollout_auv_pc lets the POSE wave evaluate it (it holds q and has time to spare) and hand the six
// values to the velocity wave instead of the quaternion
template <class GT>
__device__ __forceinline__ void auv_restoring(const GT *__restrict__ G, const float (&q)[4], float (&g6)[6])
{
    float rot[9], T[12];
    auv_b2i(q, rot, T); // only rot's third row is used here (the rest is dead code to the compiler)
    float fbg[3], fbb[3], mbg[3], mbb[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) { fbg[i] = rot[6 + i] * G->fng_z; fbb[i] = rot[6 + i] * G->fnb_z; }
    const float cog[3] = {G->cog[0], G->cog[1], G->cog[2]}, cob[3] = {G->cob[0], G->cob[1], G->cob[2]};
    cross3(cog, fbg, mbg);
    cross3(cob, fbb, mbb);
#pragma unroll
    for (int i = 0; i < 3; ++i) { g6[i] = -(fbg[i] + fbb[i]); g6[3 + i] = -(mbg[i] + mbb[i]); }
}

template <class GT>
__device__ __forceinline__ void auv_vel_rates_g(const GT *__restrict__ G, const float (&g6)[6], const float (&vel)[6], const float (&u)[kGenA],
                                                float (&xdv)[6], float *pieces = nullptr)
{
    // damping (:482-510): D = (-lin - v0*fwd) + (-(diag(quad) |diag(v)|)); D v as the dense row sum
    float Dv[6];
    const float v0 = vel[0];
    if (G->damp_diag) { // off-diagonal entries: (-0 - v0*0) + (-0) = +-0, times v_j = +-0: dropped
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            const float d1 = (-1.0f * G->lin_diag(i)) - (v0 * G->fwd_diag(i));
            const float dii = d1 + (-1.0f * (G->quad_damp[i] * fabsf(vel[i])));
            Dv[i] = dii * vel[i];
        }
    } else {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            float acc = 0.0f;
#pragma unroll
            for (int j = 0; j < 6; ++j) {
                const float d1 = (-1.0f * G->lin_damp[i * 6 + j]) - (v0 * G->lin_damp_fwd[i * 6 + j]);
                const float dij = i == j ? d1 + (-1.0f * (G->quad_damp[i] * fabsf(vel[i]))) : d1 + (-0.0f);
                acc = acc + dij * vel[j];
            }
            Dv[i] = acc;
        }
    }
    // Coriolis (:512-545): a1 = M11 v1 + M12 v2, a2 = M21 v1 + M22 v2; C = [[0, -S(a1)], [-S(a1), -S(a2)]]
    float a6[6];
    G->mass_times(vel, a6);
    const float a1[3] = {a6[0], a6[1], a6[2]}, a2[3] = {a6[3], a6[4], a6[5]};
    // -S(a) = [[-0, a2, -a1], [-a2, -0, a0], [a1, -a0, -0]]; rows of C v as running sums over j = 0..5 without the zero terms
    float Cv[6];
    Cv[0] = a1[2] * vel[4] + (-a1[1]) * vel[5];
    Cv[1] = (-a1[2]) * vel[3] + a1[0] * vel[5];
    Cv[2] = a1[1] * vel[3] + (-a1[0]) * vel[4];
    Cv[3] = ((a1[2] * vel[1] + (-a1[1]) * vel[2]) + a2[2] * vel[4]) + (-a2[1]) * vel[5];
    Cv[4] = (((-a1[2]) * vel[0] + a1[0] * vel[2]) + (-a2[2]) * vel[3]) + a2[0] * vel[5];
    Cv[5] = ((a1[1] * vel[0] + (-a1[0]) * vel[1]) + a2[1] * vel[3]) + (-a2[0]) * vel[4];
    float rhs[6];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float g_f = g6[i], g_m = g6[3 + i];
        rhs[i] = ((u[i] - Cv[i]) - Dv[i]) - g_f;
        rhs[3 + i] = ((u[3 + i] - Cv[3 + i]) - Dv[3 + i]) - g_m;
        if (pieces != nullptr) { pieces[12 + i] = g_f; pieces[15 + i] = g_m; }
    }
    if (pieces != nullptr) {
#pragma unroll
        for (int i = 0; i < 6; ++i) { pieces[i] = Cv[i]; pieces[6 + i] = Dv[i]; }
    }
    G->inv_mass_times(rhs, xdv);
}

template <class GT>
__device__ __forceinline__ void auv_vel_rates(const GT *__restrict__ G, const float (&q)[4], const float (&vel)[6], const float (&u)[kGenA],
                                              float (&xdv)[6], float *pieces = nullptr)
{
    float g6[6];
    auv_restoring(G, q, g6);
    auv_vel_rates_g(G, g6, vel, u, xdv, pieces);
}

template <class GT>
__device__ __forceinline__ void auv_state_dot(const GT *__restrict__ G, const float (&x)[kGenS], const float (&u)[kGenA],
                                              float (&xd)[kGenS], float *pieces = nullptr)
{
    const float q[4] = {x[3], x[4], x[5], x[6]};
    float vel[6], xp[7], xv[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) vel[i] = x[7 + i];
    auv_pose_rates(q, vel, xp);
    auv_vel_rates(G, q, vel, u, xv, pieces);
#pragma unroll
    for (int i = 0; i < 7; ++i) xd[i] = xp[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) xd[7 + i] = xv[i];
}

// tf.math.l2_normalize on the quaternion (auv_model.py:422-448): q * (1 / sqrt(max(sum q^2, 1e-12)))
__device__ __forceinline__ void normalize_quat(float (&x)[kGenS])
{
    float ss = x[3] * x[3];
#pragma unroll
    for (int i = 1; i < 4; ++i) ss = ss + x[3 + i] * x[3 + i];
    ss = ss < 1e-12f ? 1e-12f : ss;
    const float inv = 1.0f / sqrtf(ss);
#pragma unroll
    for (int i = 0; i < 4; ++i) x[3 + i] = x[3 + i] * inv;
}

// auv_model.py:279-306 step(rk) + normalize_quat. rk is wave-uniform.
template <class GT>
__device__ __forceinline__ void auv_step(const GT *__restrict__ G, float (&x)[kGenS], const float (&u)[kGenA])
{
    const float dt = G->dt;
    float k1[kGenS], tmp[kGenS];
    auv_state_dot(G, x, u, k1);
    if (G->rk == 2) {
        float xs[kGenS], k2[kGenS];
#pragma unroll
        for (int i = 0; i < kGenS; ++i) xs[i] = x[i] + dt * k1[i];
        auv_state_dot(G, xs, u, k2);
#pragma unroll
        for (int i = 0; i < kGenS; ++i) tmp[i] = (dt / 2.0f) * (k1[i] + k2[i]);
    } else if (G->rk == 4) { // the reference's formula, k4*dt inside the sum (:299-300)
        float xs[kGenS], k2[kGenS], k3[kGenS], k4[kGenS];
#pragma unroll
        for (int i = 0; i < kGenS; ++i) xs[i] = x[i] + (dt * k1[i]) / 2.0f;
        auv_state_dot(G, xs, u, k2);
#pragma unroll
        for (int i = 0; i < kGenS; ++i) xs[i] = x[i] + (dt * k2[i]) / 2.0f;
        auv_state_dot(G, xs, u, k3);
#pragma unroll
        for (int i = 0; i < kGenS; ++i) xs[i] = x[i] + dt * k3[i];
        auv_state_dot(G, xs, u, k4);
        const float sixth = (float)(1.0 / 6.0);
#pragma unroll
        for (int i = 0; i < kGenS; ++i) tmp[i] = (sixth * ((k1[i] + 2.0f * k2[i]) + (2.0f * k3[i] + k4[i] * dt))) * dt;
    } else {
#pragma unroll
        for (int i = 0; i < kGenS; ++i) tmp[i] = k1[i] * dt;
    }
#pragma unroll
    for (int i = 0; i < kGenS; ++i) x[i] = x[i] + tmp[i];
    normalize_quat(x);
}

// ---------------------------------------------------------------------------------------- costs
// static_cost.py:141-159 dist + :114-139 state_cost
__device__ __forceinline__ float state_cost_quat(const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const float (&x)[kGenS])
{
    float d[10], left[10];
    float dot = x[3] * C->goal[3];
#pragma unroll
    for (int i = 1; i < 4; ++i) dot = dot + x[3 + i] * C->goal[3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i) d[i] = x[i] - C->goal[i];
    // <q, g> of two unit quaternions can round to 1 + 1 ulp in fp32 when the vehicle sits AT a goal attitude that is not axis-aligned:
    // acosf is NaN there, one NaN cost makes eta, U' and the warm start NaN for good (the reference has the same hazard in fp64,
    // static_cost.py:151). Clamped to [-1, 1]: the reference's value wherever the reference's is finite.
    d[3] = 2.0f * acosf(fminf(fmaxf(dot, -1.0f), 1.0f));
#pragma unroll
    for (int i = 0; i < 6; ++i) d[4 + i] = x[7 + i] - C->goal[7 + i];
#pragma unroll
    for (int p = 0; p < 5; ++p) { // rows 2p, 2p+1 of Q d: the row sums of matvec_n<10>, two at a time
        const f32x2g *qp = reinterpret_cast<const f32x2g *>(G->q10p) + p * 10;
        f32x2g a2 = qp[0] * f32x2g{d[0], d[0]};
#pragma unroll
        for (int jj = 1; jj < 10; ++jj) a2 = a2 + qp[jj] * f32x2g{d[jj], d[jj]};
        left[2 * p] = a2.x; left[2 * p + 1] = a2.y;
    }
    float acc = d[0] * left[0];
#pragma unroll
    for (int i = 1; i < 10; ++i) acc = acc + d[i] * left[i];
    return acc;
}

// tensorflow_graphics quaternion.multiply (x, y, z, w)
__device__ __forceinline__ void quat_mul(const float (&p)[4], const float (&q)[4], float (&o)[4])
{
    const float x1 = p[0], y1 = p[1], z1 = p[2], w1 = p[3], x2 = q[0], y2 = q[1], z2 = q[2], w2 = q[3];
    o[0] = ((x1 * w2 + y1 * z2) - z1 * y2) + w1 * x2;
    o[1] = ((-x1 * z2 + y1 * w2) + z1 * x2) + w1 * y2;
    o[2] = ((x1 * y2 - y1 * x2) + z1 * w2) + w1 * z2;
    o[3] = ((-x1 * x2 - y1 * y2) - z1 * z2) + w1 * w2;
}

__device__ __forceinline__ void l2_normalize3(float (&v)[3])
{
    float ss = (v[0] * v[0] + v[1] * v[1]) + v[2] * v[2];
    ss = ss < 1e-12f ? 1e-12f : ss;
    const float inv = 1.0f / sqrtf(ss);
#pragma unroll
    for (int i = 0; i < 3; ++i) v[i] = v[i] * inv;
}

// elipse_cost.py:124-246 ElipseCost3D: the three terms position_error, orientation_error, velocity_error of a state.
// in_plane: the pose is ALREADY expressed in the plane frame (what the reference's unit tests feed the three methods with).
__device__ __forceinline__ void e3_terms(const GenConsts *__restrict__ G, const float (&x)[kGenS], const bool in_plane,
                                         float &pc, float &oc, float &vc)
{
    const float q[4] = {G->e3_q[0], G->e3_q[1], G->e3_q[2], G->e3_q[3]};
    const float qc[4] = {-q[0], -q[1], -q[2], q[3]};
    // the pose in the plane frame: rotate(position, q) = (q (p, 0)) conj(q) ; multiply(q, quat)
    const float pq[4] = {x[0], x[1], x[2], 0.0f};
    float t4[4], r4[4], qpf[4];
    quat_mul(q, pq, t4);
    quat_mul(t4, qc, r4);
    const float xq[4] = {x[3], x[4], x[5], x[6]};
    quat_mul(q, xq, qpf);
    if (in_plane) {
#pragma unroll
        for (int i = 0; i < 3; ++i) r4[i] = x[i];
#pragma unroll
        for (int i = 0; i < 4; ++i) qpf[i] = x[3 + i];
    }
    // position error: |sum_i (p_i/axis_i)^2 - 1|
    float pd = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) { const float r = r4[i] / G->e3_axis[i]; pd = i == 0 ? r * r : pd + r * r; }
    pc = fabsf(pd - 1.0f);
    // orientation error: tangent (p1*(-a/b), p0*(b/a), p2*0) normalised; q_t = between_two_vectors_3d(e_x, tangent); relative_angle
    float tg[3] = {r4[1] * G->e3_map[0], r4[0] * G->e3_map[1], r4[2] * G->e3_map[2]};
    const float nrm = sqrtf((tg[0] * tg[0] + tg[1] * tg[1]) + tg[2] * tg[2]);
#pragma unroll
    for (int i = 0; i < 3; ++i) tg[i] = tg[i] / nrm;
    float v1[3] = {1.0f, 0.0f, 0.0f};
    l2_normalize3(v1);
    l2_normalize3(tg);
    const float cos_theta = (v1[0] * tg[0] + v1[1] * tg[1]) + v1[2] * tg[2];
    float real_part = 1.0f + cos_theta;
    float axis[3];
    cross3(v1, tg, axis);
    if (real_part < 1e-6f) { // antiparallel: an arbitrary axis orthogonal to v1 = e_x: |x| largest -> rotate around z: (-y, x, 0)
        axis[0] = -v1[1]; axis[1] = v1[0]; axis[2] = 0.0f;
        real_part = 0.0f;
    }
    float qt[4] = {axis[0], axis[1], axis[2], real_part};
    {
        float ss = ((qt[0] * qt[0] + qt[1] * qt[1]) + qt[2] * qt[2]) + qt[3] * qt[3];
        ss = ss < 1e-12f ? 1e-12f : ss;
        const float inv = 1.0f / sqrtf(ss);
#pragma unroll
        for (int i = 0; i < 4; ++i) qt[i] = qt[i] * inv;
    }
    float dot = ((qt[0] * qpf[0] + qt[1] * qpf[1]) + qt[2] * qpf[2]) + qt[3] * qpf[3];
    dot = dot * (1.0f - 4.0f * (2.0f * 1.1920928955078125e-07f)); // safe_shrink by 4 eps_addition(float32)
    oc = 2.0f * acosf(fabsf(dot));
    // velocity error: | |v|^2 - speed^2 |, |v| = sqrt of the sum of squares, squared again
    const float v = sqrtf((x[7] * x[7] + x[8] * x[8]) + x[9] * x[9]);
    vc = fabsf(v * v - G->e3_gv * G->e3_gv);
}

// elipse_cost.py:124-139 state_cost = mS*position + mS*orientation + mV*velocity
__device__ __forceinline__ float state_cost_e3(const GenConsts *__restrict__ G, const float (&x)[kGenS])
{
    float pc, oc, vc;
    e3_terms(G, x, false, pc, oc, vc);
    return (G->e3_mS * pc + G->e3_mS * oc) + G->e3_mV * vc;
}

// the cost_base slot of the 13-state kernels: wave-uniform branch on the cost the controller was built with
__device__ __forceinline__ float gen_state_cost(const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const float (&x)[kGenS])
{
    if (C->state_cost_kind == MPPI_STATE_COST_QUAT) return state_cost_quat(C, G, x);
    if (C->state_cost_kind == MPPI_STATE_COST_ELLIPSE3D) return state_cost_e3(G, x);
    if (C->q_full) return state_cost<kGenS, true>(C, x);
    return state_cost<kGenS, false>(C, x);
}

// ---------------------------------------------------------------------------------------- NNAUVModel on the vector ALU
// (nn_model.py:215-239, 289-304) the Dense stack through the scalar cache, as k_rollout_mlp_small. The output layer is
// padded to 14 columns (W3 [HID x 14], b3 [14], column 13 zero) on the host so that outputs stay pairs.
template <int HID>
__device__ __forceinline__ void nnauv_step(const MlpDev *__restrict__ M, const float *const (&Wp)[kMlpSmallMaxLayers],
                                           const float *const (&bp)[kMlpSmallMaxLayers], int n_hidden,
                                           const float (&xm)[kGenNin], const float (&xr)[kGenNin], float (&x)[kGenS], const float (&v)[kGenA])
{
    constexpr int H2 = HID / 2, S2 = (kGenS + 1) / 2;
    float in[kGenNin];
#pragma unroll
    for (int i = 0; i < kGenS - 3; ++i) in[i] = (x[3 + i] - xm[i]) * xr[i];
#pragma unroll
    for (int i = 0; i < kGenA; ++i) in[kGenS - 3 + i] = (v[i] - xm[kGenS - 3 + i]) * xr[kGenS - 3 + i];
    f32x2s ha[H2], hb[H2];
    float hin[HID];
    dense_pairs<kGenNin, H2, true>(Wp[0], bp[0], in, ha);
    for (int l = 1; l < n_hidden; ++l) {
#pragma unroll
        for (int o2 = 0; o2 < H2; ++o2) { hin[2 * o2] = ha[o2].x; hin[2 * o2 + 1] = ha[o2].y; }
        const float *Wl = l == 1 ? Wp[1] : Wp[2], *bl = l == 1 ? bp[1] : bp[2];
        dense_pairs<HID, H2, true>(Wl, bl, hin, hb);
#pragma unroll
        for (int o2 = 0; o2 < H2; ++o2) ha[o2] = hb[o2];
    }
#pragma unroll
    for (int o2 = 0; o2 < H2; ++o2) { hin[2 * o2] = ha[o2].x; hin[2 * o2 + 1] = ha[o2].y; }
    f32x2s y[S2];
    const float *Wo = n_hidden == 1 ? Wp[1] : (n_hidden == 2 ? Wp[2] : Wp[3]);
    const float *bo = n_hidden == 1 ? bp[1] : (n_hidden == 2 ? bp[2] : bp[3]);
    dense_pairs<HID, S2, false>(Wo, bo, hin, y);
#pragma unroll
    for (int i = 0; i < kGenS; ++i) {
        const float yi = (i & 1) ? y[i / 2].y : y[i / 2].x;
        x[i] = x[i] + (yi * M->ystd[i] + M->ymean[i]); // next_state = state + delta (nn_model.py:303-304)
    }
}

// ---------------------------------------------------------------------------------------- NNAUVModelSpeed
// tensorflow_graphics euler.from_quaternion as NNAUVModelSpeed.to_euler calls it (nn_model.py:566-588): (x, y, z, w) -> (theta_x, theta_y,
// theta_z), R = Rz Ry Rx; entries "safe-shrunk" by (1 - 4 eps), the gimbal-lock branch for | |r20| - 1 | < 1e-6 (the tests pin the
// CPU restatement of the same algorithm against scipy's as_euler('xyz')). asinf / atan2f are the device library's.
__device__ __forceinline__ float nonzero_sign(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
// atan2 for the pose wave of k_rollout_nnspeed_pc (r05, second session). The device library's atan2f is 43 vector instructions — an IEEE
// division (12) and the zero / infinity / NaN cases — twice per step and rollout, on the wave that shares its SIMD's vector pipe with the
// network wave. Here x is never zero (euler_from_quat adds +-eps to it) and nothing is infinite: min/max, v_rcp_f32 (1 ulp), an odd
// polynomial in explicit fused multiply-adds (fitted on [0, 1], 1.07 ulp over 2e6 fp32 arguments against fp64: the recipe is in DESIGN_HISTORY),
// the two quadrant folds and the sign: 21 instructions; 2.2 ulp with a correctly rounded reciprocal, 3.1 with v_rcp_f32's 1 ulp always against it,
// < 3.5e-7 rad (tests/test_pose_atan2.py restates it from these coefficients). NOT used by the kernels whose costs are held bit for bit to the
// CPU restatement's (the Fossen model has no Euler angles). Measured: 142.8 -> 141.4 us, 1 % (the step is the network wave's chain).
__device__ __forceinline__ float atan2_pose(float y, float x)
{
    const float ax = fabsf(x), ay = fabsf(y);
    const float mx = fmaxf(ax, ay), mn = fminf(ax, ay);
    const float t = mn * __builtin_amdgcn_rcpf(mx);
    const float s = t * t;
    float p = 0x1.7ed2fap-9f;
    p = __builtin_fmaf(p, s, -0x1.0c2c96p-6f);
    p = __builtin_fmaf(p, s, 0x1.61fe5ep-5f);
    p = __builtin_fmaf(p, s, -0x1.3556fap-4f);
    p = __builtin_fmaf(p, s, 0x1.b4e15p-4f);
    p = __builtin_fmaf(p, s, -0x1.230aep-3f);
    p = __builtin_fmaf(p, s, 0x1.9978f4p-3f);
    p = __builtin_fmaf(p, s, -0x1.5554dcp-2f);
    float a = __builtin_fmaf(t * s, p, t);
    a = ay > ax ? (float)(3.14159265358979323846 / 2.0) - a : a;
    a = x < 0.0f ? (float)3.14159265358979323846 - a : a;
    return __builtin_copysignf(a, y);
}
template <bool FAST = false>
__device__ __forceinline__ void euler_from_quat(const float (&q)[4], float (&e)[3])
{
    constexpr float eps = 2.0f * 1.1920928955078125e-07f, shr = 1.0f - 4.0f * eps;
    const float x = q[0], y = q[1], z = q[2], w = q[3];
    const float tx = (2.0f * x) * shr, ty = (2.0f * y) * shr, tz = (2.0f * z) * shr;
    const float twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y, tyz = tz * y, tzz = tz * z;
    float r00 = (1.0f - (tyy + tzz)) * shr;
    const float r10 = (txy + twz) * shr, r21 = (tyz + twx) * shr;
    float r22 = (1.0f - (txx + tyy)) * shr;
    const float r20 = (txz - twy) * shr, r01 = (txy - twz) * shr;
    float r02 = (txz + twy) * shr;
    if (fabsf(fabsf(r20) - 1.0f) < 1.0e-6f) { // gimbal lock
        const float sg = nonzero_sign(r20);
        r02 = nonzero_sign(r02) * eps + r02;
        e[0] = atan2f(-sg * r01, -sg * r02);
        e[1] = -sg * (float)(3.14159265358979323846 / 2.0);
        e[2] = 0.0f;
        return;
    }
    const float th_y = -asinf(r20);
    // sign_cos_theta_y of the published algorithm = sign(cos(th_y)): outside the gimbal-lock branch |r20| <= 1 - 1e-6, so
    // |th_y| <= pi/2 - 1.4e-3 and cos(th_y) >= 1.4e-3 — the sign is +1 for every input that reaches this line, and the products
    // with it are exact. The cosf it replaces was 117 of this function's 324 vector instructions (full-range argument reduction).
    r00 = nonzero_sign(r00) * eps + r00;
    r22 = nonzero_sign(r22) * eps + r22;
    e[2] = FAST ? atan2_pose(r10, r00) : atan2f(r10, r00);
    e[0] = FAST ? atan2_pose(r21, r22) : atan2f(r21, r22);
    e[1] = th_y;
}

// nn_model.py:463-472 next_state: pose' = normalize_quat(pose + J(x) vel dt), vel' = vel + delta; J = [[rot, 0], [0, T]] with THIS class's
// T rows (:545-555: (-x,-y,-z), (w,-z,y), (z,w,-x), (-y,x,w) — AUVModel's rows rotated by one; the model is the specification)
template <bool FAST = false> // FAST (the pose wave of k_rollout_nnspeed_pc): the quaternion's 1 / sqrt as ONE v_rsq_f32 (1 ulp) — tf.math.l2_normalize is x * rsqrt(max(sum x^2, eps)) — instead of an IEEE square root and an IEEE division (27 instructions)
__device__ __forceinline__ void nnauv_speed_next_state(float dt, float (&x)[kGenS], const float (&delta)[6])
{
    const float q[4] = {x[3], x[4], x[5], x[6]};
    float rot[9], T[12], xn[7];
    auv_b2i(q, rot, T);
#pragma unroll
    for (int i = 0; i < 3; ++i) xn[i] = x[i] + ((rot[i * 3] * x[7] + rot[i * 3 + 1] * x[8]) + rot[i * 3 + 2] * x[9]) * dt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = (i + 3) & 3;
        xn[3 + i] = x[3 + i] + ((T[r * 3] * x[10] + T[r * 3 + 1] * x[11]) + T[r * 3 + 2] * x[12]) * dt;
    }
#pragma unroll
    for (int i = 0; i < 7; ++i) x[i] = xn[i];
    if constexpr (FAST) {
        float ss = x[3] * x[3];
#pragma unroll
        for (int i = 1; i < 4; ++i) ss = ss + x[3 + i] * x[3 + i];
        const float inv = __builtin_amdgcn_rsqf(ss < 1e-12f ? 1e-12f : ss);
#pragma unroll
        for (int i = 0; i < 4; ++i) x[3 + i] = x[3 + i] * inv;
    } else {
        normalize_quat(x);
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) x[7 + i] = x[7 + i] + delta[i];
}

// build_step_graph (:358-380): the Dense stack through the scalar cache (dense_pairs, as nnauv_step); 15 inputs, 6 outputs
template <int HID>
__device__ __forceinline__ void nnauv_speed_step(const DevConsts *__restrict__ C, const MlpDev *__restrict__ M, const float *const (&Wp)[kMlpSmallMaxLayers],
                                                 const float *const (&bp)[kMlpSmallMaxLayers], int n_hidden,
                                                 const float (&xm)[kGenNin], const float (&xr)[kGenNin], float (&x)[kGenS], const float (&v)[kGenA])
{
    constexpr int H2 = HID / 2, NI = kGenSpeedNin;
    float in[NI], e[3];
    const float q[4] = {x[3], x[4], x[5], x[6]};
    euler_from_quat(q, e);
#pragma unroll
    for (int i = 0; i < 3; ++i) in[i] = (e[i] - xm[i]) * xr[i];
#pragma unroll
    for (int i = 0; i < 6; ++i) in[3 + i] = (x[7 + i] - xm[3 + i]) * xr[3 + i];
#pragma unroll
    for (int i = 0; i < 6; ++i) in[9 + i] = (v[i] - xm[9 + i]) * xr[9 + i];
    f32x2s ha[H2], hb[H2];
    float hin[HID];
    dense_pairs<NI, H2, true>(Wp[0], bp[0], in, ha);
    for (int l = 1; l < n_hidden; ++l) {
#pragma unroll
        for (int o2 = 0; o2 < H2; ++o2) { hin[2 * o2] = ha[o2].x; hin[2 * o2 + 1] = ha[o2].y; }
        const float *Wl = l == 1 ? Wp[1] : Wp[2], *bl = l == 1 ? bp[1] : bp[2];
        dense_pairs<HID, H2, true>(Wl, bl, hin, hb);
#pragma unroll
        for (int o2 = 0; o2 < H2; ++o2) ha[o2] = hb[o2];
    }
#pragma unroll
    for (int o2 = 0; o2 < H2; ++o2) { hin[2 * o2] = ha[o2].x; hin[2 * o2 + 1] = ha[o2].y; }
    f32x2s y[3];
    const float *Wo = n_hidden == 1 ? Wp[1] : (n_hidden == 2 ? Wp[2] : Wp[3]);
    const float *bo = n_hidden == 1 ? bp[1] : (n_hidden == 2 ? bp[2] : bp[3]);
    dense_pairs<HID, 3, false>(Wo, bo, hin, y);
    float delta[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) delta[i] = ((i & 1) ? y[i / 2].y : y[i / 2].x) * M->ystd[i] + M->ymean[i];
    nnauv_speed_next_state(C->dt, x, delta);
}

// ---------------------------------------------------------------------------------------- the rollout kernel
// MODEL: GEN_MODEL_AUV | GEN_MODEL_NNAUV ; HID: hidden width of the NNAUV network (16 | 32; ignored for the AUV model).
// mode: MODE_ROLLOUT (costs + record) | MODE_COST_ONLY | MODE_COSTS_GIVEN (record from given costs) ; noise_out != NULL
// additionally exports the noise the step used ([K, H, A], MPPI_DBG_NOISE) — and with MODE_NOISE_ONLY does nothing else.
template <int MODEL, int HID, bool DIAG = false>
__global__ __launch_bounds__(64) void k_rollout_gen(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const MlpDev *__restrict__ M, const MlpSmallArgs P,
    const float *__restrict__ x_dev, const float *__restrict__ U_dev, const float *__restrict__ eps_hbm,
    const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    float *__restrict__ noise_out, const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr int S = kGenS, A = kGenA; // DIAG: Sigma and its inverse are exactly diagonal (the off-diagonal products are exact zeros: skipped)
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int lane = threadIdx.x;
    const int k0 = blockIdx.x * 64;
    const bool valid = (k0 + lane) < K;
    const int kk = valid ? k0 + lane : K - 1; // lanes past K recompute the last sample, outside every sum
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
    float c = 0.0f;
    if (MODE == MODE_COSTS_GIVEN) {
        c = cost[kk];
    } else {
        float xm[kGenNin], xr[kGenNin];
        if (MODEL == GEN_MODEL_NNAUV || MODEL == GEN_MODEL_NNAUV_SPEED) {
#pragma unroll
            for (int i = 0; i < kGenNin; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; } // (the speed model uses 15 of the 16)
        }
        float x[S];
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = x_dev[i];
        float z[4 * A];
        AuvLocal al;
        if (MODEL == GEN_MODEL_AUV) al.load(G);
        for (int t = 0; t < H; ++t) {
            if (SRC == SRC_PHILOX && (t & 3) == 0) normals_group<A>(seed, gk, base + (unsigned long long)(t >> 2), z);
            float u[A], e[A], v[A];
            if (SRC == SRC_PHILOX) {
                float z1[A];
#pragma unroll
                for (int i = 0; i < A; ++i) { // z[(t & 3) * A + i] without a dynamically indexed register array
                    float zi = z[i];
#pragma unroll
                    for (int tl = 1; tl < 4; ++tl) zi = (t & 3) == tl ? z[tl * A + i] : zi;
                    z1[i] = zi;
                }
                scale_noise<A, DIAG>(C, z1, e);
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
            }
            if (noise_out != nullptr && valid) {
#pragma unroll
                for (int i = 0; i < A; ++i) noise_out[(size_t)kk * HA + t * A + i] = e[i];
            }
            if (MODE == MODE_NOISE_ONLY) continue;
#pragma unroll
            for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; } // to_apply, controller_base.cpp:258
            const float ac = action_cost<A, DIAG>(C, u, e);
            if (MODEL == GEN_MODEL_AUV) {
                auv_step(&al, x, v);
            } else {
                int zoff; // opaque zero offset: keeps the weight loads inside the horizon loop (see k_rollout_mlp_small)
                asm("s_mov_b32 %0, 0" : "=s"(zoff) : "s"(t));
                const float *Wp[kMlpSmallMaxLayers], *bp[kMlpSmallMaxLayers];
#pragma unroll
                for (int l = 0; l < kMlpSmallMaxLayers; ++l) { Wp[l] = P.W[l] + zoff; bp[l] = P.b[l] + zoff; }
                if (MODEL == GEN_MODEL_NNAUV_SPEED) nnauv_speed_step<HID>(C, M, Wp, bp, P.n_layers - 1, xm, xr, x, v);
                else nnauv_step<HID>(M, Wp, bp, P.n_layers - 1, xm, xr, x, v);
            }
            const float sc = gen_state_cost(C, G, x); // cost on the POST-step state
            const float tmp = sc + ac;                // Step_cost_result cost_base.cpp:49
            c = c + tmp;                              // path_cost        controller_base.cpp:268
        }
        if (MODE == MODE_NOISE_ONLY) return;
        c = c + gen_state_cost(C, G, x); // terminal cost, controller_base.cpp:271-272
        if (valid) cost[k0 + lane] = c;
        if (MODE == MODE_COST_ONLY) return;
    }
    mlp_tile_record<A, DIAG, 1>(C, c, valid, 0, lane, kk, H, NG, SRC, eps_hbm, seed, gk, base,
                                partials + (size_t)record_slot(blockIdx.x, rsc) * rsb, rsc);
}

// state / action / step cost of k samples on a 13-state handle (mppi_state_cost, mppi_action_cost, mppi_step_cost)
__global__ void k_gen_costs(const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const float *__restrict__ x,
                            const float *__restrict__ u, const float *__restrict__ eps, int k,
                            float *__restrict__ out_state, float *__restrict__ out_action, float *__restrict__ out_step)
{
    const int a = C->a; // any action dimension (zero-padded instance: adding exact zeros changes no sum)
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    float sc = 0.0f, ac = 0.0f;
    if (x != nullptr) {
        float xs[kGenS];
#pragma unroll
        for (int j = 0; j < kGenS; ++j) xs[j] = x[(size_t)i * kGenS + j];
        sc = gen_state_cost(C, G, xs);
    }
    if (u != nullptr) {
        float us[kMaxA], es[kMaxA];
#pragma unroll
        for (int j = 0; j < kMaxA; ++j) { us[j] = j < a ? u[j] : 0.0f; es[j] = j < a ? eps[(size_t)i * a + j] : 0.0f; }
        ac = action_cost<kMaxA>(C, us, es);
    }
    if (out_state) out_state[i] = sc;
    if (out_action) out_action[i] = ac;
    if (out_step) out_step[i] = sc + ac;
}

// ---------------------------------------------------------------------------------------- NNAUVModel on the matrix cores
// k_rollout_nnauv32: NNAUVModel with the reference's Dense(32, relu) x 1..3 + Dense(13) network (nn_model.py:54-60, 215-304) as
// k_rollout_mlp32 runs the point-mass networks (mppi_mlp32.hip.h): a 32-wide layer is one v_mfma_f32_32x32x2_f32 tile, the
// accumulator layout of one layer IS the B-operand layout of the next (units renumbered u(r, hh) = 8 (r >> 2) + 4 hh + (r & 3)),
// weights and biases stationary in registers, no LDS and no barrier between layers. What changes for the 13-state family:
// 16 inputs = concat(x[3:], v) (8 k pairs in layer 1 instead of 5), 13 outputs (the output layer on the vector ALU as 7 pairs,
// W3 rows padded to 16 in LDS), 6 Philox blocks per horizon group, and the cost_base slot is gen_state_cost (quadratic with any Q,
// StaticQuatCost, ElipseCost3D). One wave = 32 rollouts (both lane halves carry rollout j's state), a workgroup = 2 waves = one
// 64-rollout tile record.
constexpr int kNnauv32Threads = 128;

template <bool DIAG>
__global__ __launch_bounds__(kNnauv32Threads) void k_rollout_nnauv32(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr,
    float *__restrict__ cost, float *__restrict__ partials, const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr int S = kGenS, A = kGenA, NIN = kGenNin, XOFF = 3, SP = (S + 1) / 2, HID = 32, K1H = NIN / 2, W3LD = 16;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float w3_s[HID * W3LD];   // output-layer rows, padded to 16
    __shared__ float z_s[2][4 * A][32];                                // per wave: the normals of one horizon group
    __shared__ float cost_s[64];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
    const int k0 = blockIdx.x * 64;
    const int kk = min(k0 + 32 * w + j, K - 1); // rollouts past K recompute the last sample, outside every sum
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned int gk = (unsigned int)C->k_offset + (unsigned int)kk;
    const int n_hidden = M->n_layers - 1;
    auto unit_of = [](int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); };

    // ---- stationary operands
    float a1[K1H];
#pragma unroll
    for (int s1 = 0; s1 < K1H; ++s1) a1[s1] = M->Wl[0][(2 * s1 + hh) * HID + j];
    f32x16 b1t, bht[2];
    float ah[2][16];
#pragma unroll
    for (int r = 0; r < 16; ++r) b1t[r] = M->bl[0][unit_of(r, hh)];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const bool have = l + 2 <= n_hidden; // hidden-to-hidden layer l exists
        const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
        for (int s = 0; s < 16; ++s) ah[l][s] = have ? Wl[unit_of(s, hh) * HID + j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) bht[l][r] = have ? bl[unit_of(r, hh)] : 0.0f;
    }
    const float *W3g = M->Wl[n_hidden], *b3g = M->bl[n_hidden];
    const int ld3 = M->ld[n_hidden]; // 14: the host pads the 13-wide output layer to an even width
    for (int i = tid; i < HID * W3LD; i += kNnauv32Threads) w3_s[i] = (i & 15) < S ? W3g[(i >> 4) * ld3 + (i & 15)] : 0.0f;
    float xm[NIN], xr[NIN], b3v[S], ysd[S], ymn[S];
#pragma unroll
    for (int i = 0; i < NIN; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; }
#pragma unroll
    for (int i = 0; i < S; ++i) { b3v[i] = b3g[i]; ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    float x[S], c = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];
    __syncthreads();

    // the layers are single asm statements on fixed accumulator registers (mppi_mfma32.hip.h): MFMAs, wait states and relu together
    // output layer + next_state (state + de-normalised delta) + step cost, from the (relu'd) accumulators of the last hidden layer
    auto finish = [&](const f32x16 &hacc, float ac) {
        f32x2 py[SP];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 8 * (r >> 2) + (r & 3); // + 4 hh through the lane's base
            const float *wp = w3_s + (4 * hh + row) * W3LD;
            float wv[W3LD];
#pragma unroll
            for (int q4 = 0; q4 < W3LD / 4; ++q4) {
                const f32x4 t4 = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp + 4 * q4, 16));
                wv[4 * q4] = t4.x; wv[4 * q4 + 1] = t4.y; wv[4 * q4 + 2] = t4.z; wv[4 * q4 + 3] = t4.w;
            }
            const f32x2 h2 = {hacc[r], hacc[r]};
#pragma unroll
            for (int p2 = 0; p2 < SP; ++p2) {
                const f32x2 w2 = {wv[2 * p2], wv[2 * p2 + 1]};
                py[p2] = r == 0 ? h2 * w2 : __builtin_elementwise_fma(h2, w2, py[p2]);
            }
        }
#pragma unroll
        for (int n = 0; n < S; ++n) { // the other half's rows: lower + upper, in every lane
            float a = (n & 1) ? py[n / 2].y : py[n / 2].x, b = a;
            permlane32_swap(a, b); // a = the lower half's partial in all lanes, b = the upper half's
            const float y = (a + b) + b3v[n];
            x[n] = x[n] + (y * ysd[n] + ymn[n]);
        }
        const float sc = gen_state_cost(C, G, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    };

    for (int t = 0; t < H; ++t) {
        if (SRC == SRC_PHILOX && (t & 3) == 0) { // this wave's normals of the group: block q by the half with q & 1 == hh
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < A; ++q) {
                if ((q & 1) == hh) {
                    const float4 n = normals_of_block(seed, (unsigned long long)gk, (base + (unsigned long long)(t >> 2)) * A + q);
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
            scale_noise<A, DIAG>(C, z1, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(C, u, e);
        // layer 1: the B operand of lane (j, hh), k pair s1, is input 2 s1 + hh of rollout j; input i = x[3 + i] for i < 10, else v[i - 10]
        float bv[K1H];
#pragma unroll
        for (int s1 = 0; s1 < K1H; ++s1) {
            const int i0 = 2 * s1, i1 = 2 * s1 + 1;
            const float r0 = i0 < S - XOFF ? x[XOFF + i0] : v[i0 - (S - XOFF)], r1 = i1 < S - XOFF ? x[XOFF + i1] : v[i1 - (S - XOFF)];
            bv[s1] = hh ? (r1 - xm[i1]) * xr[i1] : (r0 - xm[i0]) * xr[i0];
        }
        f32x16 acc0;
        mfma32_layer1<K1H>(acc0, a1, bv, b1t);
        if (n_hidden >= 2) {
            f32x16 acc1;
            mfma32_hidden_64_80(acc1, acc0, ah[0], bht[0]);
            if (n_hidden >= 3) {
                mfma32_hidden_80_64(acc0, acc1, ah[1], bht[1]);
                finish(acc0, ac);
            } else {
                finish(acc1, ac);
            }
        } else {
            finish(acc0, ac);
        }
    }
    c = c + gen_state_cost(C, G, x); // terminal cost, controller_base.cpp:271-272
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

// k_rollout_nnspeed32<HID>: NNAUVModelSpeed (nn_model.py:307-588) with its Dense(HID, relu) x 1..3 + Dense(6) network on the matrix cores,
// exact fp32 (r04). On k_rollout_gen<2, HID> the 848 weights of a step come through the scalar cache and, at one wave
// per SIMD, every s_load wait is exposed; its counters read 738 `other` vector instructions per wave-step (moves and lane reads of spilled
// scalars) beside 860 of arithmetic. Here, as in k_rollout_nnauv32: the weights are stationary A operands, a layer is one asm statement of
// v_mfma_f32_32x32x2_f32 (mppi_mfma32.hip.h), the accumulators of one layer are the B operands of the next. What this model adds:
//   * 15 inputs = (3 Euler angles of the attitude, 6 body velocities, 6 forces), padded to 8 k pairs (input 15 has zero weights);
//   * HID = 16 (the reference's shape, nn_model.py:340-346) fills half of the 32-row tile: units 0..15 are accumulator registers 0..7 of
//     the two lane halves, a hidden layer is 8 k pairs (mfma32_hidden8_*), the tile's other rows carry zero weights and biases;
//   * the 6 outputs (the velocity delta) on the vector ALU from the accumulators (W3 rows padded to 8 in LDS, lane halves combined with
//     v_permlane32_swap), then next_state: the pose integrated with the class's own quaternion kinematics, quaternion renormalised;
//   * euler_from_quat (asinf, two atan2f) per step and rollout on the vector ALU.
// One wave = 32 rollouts (both lane halves carry rollout j's state), a workgroup = 2 waves = one 64-rollout tile record.
template <int HID, bool DIAG>
__global__ __launch_bounds__(kNnauv32Threads) void k_rollout_nnspeed32(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr,
    float *__restrict__ cost, float *__restrict__ partials, const int SRC, const int MODE, const int rsb, const int rsc)
{
    static_assert(HID == 16 || HID == 32, "Dense(16) or Dense(32) hidden layers");
    constexpr int S = kGenS, A = kGenA, NIN = kGenSpeedNin, NOUT = 6, K1H = 8, NP = HID / 2, W3LD = 8;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float w3_s[HID * W3LD];   // output-layer rows [unit][6 outputs + 2 zeros]
    __shared__ float z_s[2][4 * A][32];                                // per wave: the normals of one horizon group
    __shared__ float cost_s[64];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
    const int k0 = blockIdx.x * 64;
    const int kk = min(k0 + 32 * w + j, K - 1); // rollouts past K recompute the last sample, outside every sum
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned int gk = (unsigned int)C->k_offset + (unsigned int)kk;
    const int n_hidden = M->n_layers - 1;
    auto unit_of = [](int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); };
    const bool row_live = j < HID; // lane (j, hh) supplies row j of the A operand: output unit j of the layer

    // ---- stationary operands
    float a1[K1H];
#pragma unroll
    for (int s1 = 0; s1 < K1H; ++s1) a1[s1] = (row_live && 2 * s1 + hh < NIN) ? M->Wl[0][(2 * s1 + hh) * HID + j] : 0.0f;
    f32x16 b1t, bht[2];
    float ah[2][NP];
#pragma unroll
    for (int r = 0; r < 16; ++r) b1t[r] = unit_of(r, hh) < HID ? M->bl[0][unit_of(r, hh)] : 0.0f;
#pragma unroll
    for (int l = 0; l < 2; ++l) {
        const bool have = l + 2 <= n_hidden; // hidden-to-hidden layer l exists
        const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
        for (int s = 0; s < NP; ++s) ah[l][s] = (have && row_live) ? Wl[unit_of(s, hh) * HID + j] : 0.0f; // k pair s = input units u(s, 0), u(s, 1)
#pragma unroll
        for (int r = 0; r < 16; ++r) bht[l][r] = (have && unit_of(r, hh) < HID) ? bl[unit_of(r, hh)] : 0.0f;
    }
    const float *W3g = M->Wl[n_hidden], *b3g = M->bl[n_hidden];
    const int ld3 = M->ld[n_hidden];
    for (int i = tid; i < HID * W3LD; i += kNnauv32Threads) w3_s[i] = (i & 7) < NOUT ? W3g[(i >> 3) * ld3 + (i & 7)] : 0.0f;
    float xm[NIN + 1], xr[NIN + 1], b3v[NOUT], ysd[NOUT], ymn[NOUT];
#pragma unroll
    for (int i = 0; i < NIN; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; }
    xm[NIN] = 0.0f; xr[NIN] = 0.0f; // the padding input of the eighth k pair
#pragma unroll
    for (int i = 0; i < NOUT; ++i) { b3v[i] = b3g[i]; ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    const float dt = C->dt;
    float x[S], c = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];
    __syncthreads();

    // output layer + next_state (nn_model.py:463-472) + step cost, from the (relu'd) accumulators of the last hidden layer
    auto finish = [&](const f32x16 &hacc, float ac) {
        f32x2 py[NOUT / 2];
#pragma unroll
        for (int r = 0; r < NP; ++r) { // the lane half's units u(r, hh)
            const float *wp = w3_s + (8 * (r >> 2) + 4 * hh + (r & 3)) * W3LD;
            const f32x4 t0 = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp, 16));
            const f32x4 t1 = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp + 4, 16));
            const f32x2 h2 = {hacc[r], hacc[r]};
            const f32x2 w01 = {t0.x, t0.y}, w23 = {t0.z, t0.w}, w45 = {t1.x, t1.y};
            py[0] = r == 0 ? h2 * w01 : __builtin_elementwise_fma(h2, w01, py[0]);
            py[1] = r == 0 ? h2 * w23 : __builtin_elementwise_fma(h2, w23, py[1]);
            py[2] = r == 0 ? h2 * w45 : __builtin_elementwise_fma(h2, w45, py[2]);
        }
        float delta[NOUT];
#pragma unroll
        for (int n = 0; n < NOUT; ++n) { // the other half's units: lower + upper, in every lane
            float a = (n & 1) ? py[n / 2].y : py[n / 2].x, b = a;
            permlane32_swap(a, b); // a = the lower half's partial in all lanes, b = the upper half's
            const float y = (a + b) + b3v[n];
            delta[n] = y * ysd[n] + ymn[n];
        }
        nnauv_speed_next_state(dt, x, delta);
        const float sc = gen_state_cost(C, G, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    };

    for (int t = 0; t < H; ++t) {
        if (SRC == SRC_PHILOX && (t & 3) == 0) { // this wave's normals of the group: block q by the half with q & 1 == hh
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int q = 0; q < A; ++q) {
                if ((q & 1) == hh) {
                    const float4 n = normals_of_block(seed, (unsigned long long)gk, (base + (unsigned long long)(t >> 2)) * A + q);
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
            scale_noise<A, DIAG>(C, z1, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(C, u, e);
        // inputs (prepare_data, nn_model.py:438-461): Euler angles of the attitude, body velocities, forces; input 15 is the zero padding
        const float q4[4] = {x[3], x[4], x[5], x[6]};
        float eu[3];
        euler_from_quat(q4, eu);
        float in[NIN + 1];
#pragma unroll
        for (int i = 0; i < 3; ++i) in[i] = eu[i];
#pragma unroll
        for (int i = 0; i < 6; ++i) { in[3 + i] = x[7 + i]; in[9 + i] = v[i]; }
        in[NIN] = 0.0f;
        float bv[K1H]; // the B operand of lane (j, hh), k pair s1, is input 2 s1 + hh of rollout j
#pragma unroll
        for (int s1 = 0; s1 < K1H; ++s1) bv[s1] = hh ? (in[2 * s1 + 1] - xm[2 * s1 + 1]) * xr[2 * s1 + 1] : (in[2 * s1] - xm[2 * s1]) * xr[2 * s1];
        f32x16 acc0;
        mfma32_layer1<K1H>(acc0, a1, bv, b1t);
        if (n_hidden >= 2) {
            f32x16 acc1;
            if constexpr (HID == 16) mfma32_hidden8_64_80(acc1, acc0, ah[0], bht[0]);
            else mfma32_hidden_64_80(acc1, acc0, ah[0], bht[0]);
            if (n_hidden >= 3) {
                if constexpr (HID == 16) mfma32_hidden8_80_64(acc0, acc1, ah[1], bht[1]);
                else mfma32_hidden_80_64(acc0, acc1, ah[1], bht[1]);
                finish(acc0, ac);
            } else {
                finish(acc1, ac);
            }
        } else {
            finish(acc0, ac);
        }
    }
    c = c + gen_state_cost(C, G, x); // terminal cost, controller_base.cpp:271-272
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

// ---- hand-off between the two waves of a tile WITHOUT a workgroup barrier (r05). A slot is [N values + 1 tag][64 lanes]; lane l of the writer
// serves lane l of the reader (lane = rollout in both waves). Writer: the values, then the tag (the step the values belong to) — the LDS
// instructions of one wave execute in order, so a reader that sees the tag sees the values. Reader: the tag, then the values, in one batch
// (in order again: values read behind a fresh tag are fresh); the whole wave repeats the batch until every lane's tag is the expected one.
// Against s_barrier: the writer does not wait for its stores to land (no s_waitcnt before a barrier), the two tiles of a workgroup no longer
// wait for each other, and the wave that is ahead sleeps in 64-cycle naps instead of holding a barrier slot. Every spin is bounded
// (a wave that never sees its tag returns false; the caller poisons its cost with a NaN, which no test lets pass).
constexpr int kHandoffSpins = 1 << 18;
template <int N>
__device__ __forceinline__ void handoff_put(float *slot, const float (&v)[N], const int tag)
{
#pragma unroll
    for (int i = 0; i < N; ++i) slot[i * 64] = v[i];
    asm volatile("" ::: "memory"); // the tag's store stays behind the values' in the instruction stream
    slot[N * 64] = __int_as_float(tag);
    asm volatile("" ::: "memory");
}
template <int N>
__device__ __forceinline__ bool handoff_get(const float *slot, float (&v)[N], const int tag)
{
#pragma nounroll
    for (int spin = 0; spin < kHandoffSpins; ++spin) {
        asm volatile("" ::: "memory");
        const int got = __float_as_int(slot[N * 64]);
        asm volatile("" ::: "memory"); // the values' loads stay behind the tag's
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = slot[i * 64];
        if (__builtin_amdgcn_ballot_w64(got != tag) == 0ull) return true;
        __builtin_amdgcn_s_sleep(1);
    }
    return false;
}

// k_rollout_nnspeed_pc<HID>: NNAUVModelSpeed as a two-wave pipeline per 64-rollout tile (r04). In k_rollout_nnspeed32 a wave is 32
// rollouts whose per-rollout vector work — Euler angles, quaternion kinematics, the 13-term cost, the noise: ~650 instructions a step —
// runs on 64 lanes for 32 results, and the f32 MFMA does not overlap with it. The model itself offers the split: next_state integrates
// the POSE from the OLD velocities (nn_model.py:463-472), only the velocities take the network's output. So per tile
//   wave N (network): lane = rollout. Noise, v = u + eps, action cost; inputs (Euler angles from P, its velocities, v), the Dense stack as
//       TWO column blocks of 32 rollouts per weight register (mfma32x2_*: one v_permlane32_swap of the lane's (even, odd) inputs yields
//       the B operands of both blocks), the 6 outputs, vel' = vel + delta;
//   wave P (pose):    lane = rollout. cost of the state the previous step produced, pose' = normalize(pose + J(q) vel dt), Euler(q');
// exchange through double-buffered LDS — P -> N: 3 Euler angles; N -> P: 6 velocities + the action cost — one workgroup barrier per step:
// N(t) and P(t) both depend on step t-1 only and run side by side. Every per-rollout instruction now serves 64 rollouts.
// A workgroup is TWO tiles (4 waves, one per SIMD); the role comes from the SIMD a wave runs on, flipped for every second workgroup of a
// CU, so that each SIMD hosts one N and one P wave (as k_rollout_pc places its consumer; falls back to the wave index).
// Same arithmetic per rollout as k_rollout_nnspeed32 (the output layer's half sums in the same order): same bars.
// HID = 16 (the reference's NNAUVModelSpeed: Dense(16) x 3) takes v_mfma_f32_16x16x4_f32 tiles: on 32x32x2 tiles half of every layer's 32
// rows are empty — 48 MFMAs of 64 cycles per tile and step, 3072 of ~8200 cycles. 16 units x 16 rollouts x 4 inputs per instruction, four
// column blocks per wave, 32 cycles each: 48 MFMAs, 1536 cycles. Lane (n, g): D register i = unit 4 g + i of rollout n — again the next
// layer's B operand as it stands (k slot (s, g) = unit 4 g + s: the weights are loaded in that order). Inputs reach the four column blocks by
// a 4 x 4 transpose of 16-lane rows per k step (2 v_permlane32_swap + 2 v_permlane16_swap), the 6 outputs come back by the same butterfly
// with sums. Through builtins: with 4-register accumulators hipcc's own allocation and hazard padding are fine (145 VGPRs); the relu on an accumulator is ONE v_max_i32. 0.222 -> 0.188 ms.
#ifndef MPPI_NNSPEED_POSE_FAST
#define MPPI_NNSPEED_POSE_FAST 1 // (A/B: tools/build_unit_variant.py -DMPPI_NNSPEED_POSE_FAST=0 = the device library's atan2f and the IEEE 1 / sqrt in the pose wave)
#endif
constexpr int kNnspeedPcThreads = 256;
struct GenQuadConsts { // kernel-local copy of the diagonal quadratic cost (no constant re-fetch behind the per-step barrier)
    float goal[kGenS], qdiag[kGenS];
};

template <int HID, bool DIAG>
__global__ __launch_bounds__(kNnspeedPcThreads, 2) void k_rollout_nnspeed_pc(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr,
    float *__restrict__ cost, float *__restrict__ partials, const int SRC, const int MODE, const int rsb, const int rsc, const int n_tiles,
    const int balance)
{
    static_assert(HID == 16 || HID == 32, "Dense(16) or Dense(32) hidden layers");
    constexpr int S = kGenS, A = kGenA, NIN = kGenSpeedNin, NOUT = 6, K1H = 8, NP = HID / 2, W3LD = 8;
    constexpr bool MF16 = HID == 16; // Dense(16) layers on v_mfma_f32_16x16x4_f32 (no half-empty 32-row tiles)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float w3_s[HID * W3LD]; // output-layer rows [unit][6 outputs + 2 zeros]
    __shared__ float eu_s[2][2][4][64];   // [tile of the workgroup][step parity][3 angles, tag][rollout]            P -> N   (handoff_put / _get)
    __shared__ float vel_s[2][2][8][64];  // [tile][step parity][6 velocities of the NEXT state, action cost, tag][rollout]   N -> P
    __shared__ float cost_s[2][64];
    __shared__ int simd_s[4];
    MPPI_PCT_DECL(); // (timing study: mppi_ablate.hip.h)
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- which tile, which role
    int pair = wave_hw >> 1, role = wave_hw & 1; // role 0 = network, 1 = pose
    {
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_REG_HW_ID[5:4]
        if (lane == 0) simd_s[wave_hw] = simd;
        __syncthreads();
        const int s0 = simd_s[0], s1 = simd_s[1], s2 = simd_s[2], s3 = simd_s[3];
        if (balance && ((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15) {
            const int gen = (int)(blockIdx.x >> 8); // arrival order on the CU (256 CUs)
            pair = simd & 1;
            role = ((simd >> 1) ^ gen) & 1;
        }
        pair = __builtin_amdgcn_readfirstlane(pair);
        role = __builtin_amdgcn_readfirstlane(role);
    }
    const int tile = 2 * (int)blockIdx.x + pair;
    const bool tile_ok = tile < n_tiles; // the second tile of the last workgroup may not exist: its waves still keep every barrier
    const int k0 = tile * 64;
    const bool valid = tile_ok && (k0 + lane) < K;
    const int kk = min(k0 + lane, K - 1); // rollouts past K recompute the last sample, outside every sum
    const int n_hidden = M->n_layers - 1;
    const float *W3g = M->Wl[n_hidden];
    const int ld3 = M->ld[n_hidden];
    for (int i = tid; i < HID * W3LD; i += kNnspeedPcThreads) w3_s[i] = (i & 7) < NOUT ? W3g[(i >> 3) * ld3 + (i & 7)] : 0.0f;
    float c = 0.0f;

    if (role == 0) {
        // ================================================================================= wave N: noise, network, velocities
        if (balance) __builtin_amdgcn_s_setprio(3); // (one round of the grid only: beyond it the age order staggers the workgroups' phases, as in k_rollout_pc)
        // the wave a step waits for goes first on its SIMD (the other kind fills the gaps)
        const int j = lane & 31, hh = lane >> 5;
        const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
        const unsigned long long seed = C->seed;
        const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
        auto unit_of = [](int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); };
        const bool row_live = j < HID; // lane (j, hh) supplies row j of the A operand: output unit j of the layer
        // 32x32x2 tiles (HID = 32): weights as (even, odd) k pairs in the lane halves
        float a1[MF16 ? 1 : K1H];
        f32x16 b1t, bht[2];
        float ah[2][MF16 ? 1 : NP];
        // 16x16x4 tiles (HID = 16): lane = (m = lane & 15, g = lane >> 4). A operand of k step s: W[input of slot (s, g)][unit m]; layer 1 takes
        // input 4 s + g, a hidden layer unit 4 g + s — where register s of the previous layer's accumulators holds its units (D register i of
        // lane (n, g) = unit 4 g + i of rollout n): the accumulators ARE the next layer's B operands, as on the 32-wide tiles
        const int m16 = lane & 15, g16 = lane >> 4;
        float a1q[4], ahq[2][4];
        f32x2 w3q[4][NOUT / 2]; // output-layer weights of the lane's four units, outputs in pairs (packed fp32: a lone wave issues a v_pk_fma in the time of a v_fma)
        f32x4 b1q, bhq[2];
        if constexpr (MF16) {
#pragma unroll
            for (int s1 = 0; s1 < 4; ++s1) { // k slot 4 s1 + g16 of layer 1: the inputs in PAIRS that exist as pairs (slots 0-2 Euler angles, 3 empty, 4-9 velocities, 10-15 forces)
                const int slot = 4 * s1 + g16, inp = slot < 3 ? slot : slot - 1;
                a1q[s1] = slot != 3 ? M->Wl[0][inp * HID + m16] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) b1q[i] = M->bl[0][4 * g16 + i];
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                const bool have = l + 2 <= n_hidden;
                const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
                for (int s1 = 0; s1 < 4; ++s1) ahq[l][s1] = have ? Wl[(4 * g16 + s1) * HID + m16] : 0.0f;
#pragma unroll
                for (int i = 0; i < 4; ++i) bhq[l][i] = have ? bl[4 * g16 + i] : 0.0f;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int o = 0; o < NOUT; ++o) w3q[i][o >> 1][o & 1] = W3g[(4 * g16 + i) * ld3 + o];
        } else {
#pragma unroll
        for (int s1 = 0; s1 < K1H; ++s1) a1[s1] = (row_live && 2 * s1 + hh < NIN) ? M->Wl[0][(2 * s1 + hh) * HID + j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) b1t[r] = unit_of(r, hh) < HID ? M->bl[0][unit_of(r, hh)] : 0.0f;
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const bool have = l + 2 <= n_hidden;
            const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
            for (int s1 = 0; s1 < NP; ++s1) ah[l][s1] = (have && row_live) ? Wl[unit_of(s1, hh) * HID + j] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) bht[l][r] = (have && unit_of(r, hh) < HID) ? bl[unit_of(r, hh)] : 0.0f;
        }
        }
        const float *b3g = M->bl[n_hidden];
        float xm[NIN + 1], xr[NIN + 1], b3v[NOUT], ysd[NOUT], ymn[NOUT];
#pragma unroll
        for (int i = 0; i < NIN; ++i) { xm[i] = M->xmean[i]; xr[i] = 1.0f / M->xstd[i]; }
        xm[NIN] = 0.0f; xr[NIN] = 0.0f;
#pragma unroll
        for (int i = 0; i < NOUT; ++i) { b3v[i] = b3g[i]; ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
        f32x2 b3p[NOUT / 2], ysdp[NOUT / 2], ymnp[NOUT / 2]; // the same in pairs (the 16-wide path's packed output stage)
#pragma unroll
        for (int i = 0; i < NOUT; ++i) { b3p[i >> 1][i & 1] = b3v[i]; ysdp[i >> 1][i & 1] = ysd[i]; ymnp[i >> 1][i & 1] = ymn[i]; }
        PcProducerConsts<A> pcst; // Sigma, Sigma^-1, lambda: a kernel-local copy (no re-fetch behind the barriers)
        pcst.template load<DIAG>(C);
        f32x2 xm2[8], xr2[8]; // input mean / 1/std in the slot order of the 16-wide path
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int inp = k < 3 ? k : k - 1;
            xm2[k >> 1][k & 1] = k != 3 ? xm[inp] : 0.0f;
            xr2[k >> 1][k & 1] = k != 3 ? xr[inp] : 0.0f;
        }
        f32x2 sg2[A / 2], si2[A / 2]; // diagonal of Sigma and of its inverse, in pairs
#pragma unroll
        for (int i = 0; i < A; ++i) { sg2[i >> 1][i & 1] = pcst.sigma[i * kMaxA + i]; si2[i >> 1][i & 1] = pcst.sigma_inv[i * kMaxA + i]; }
        const bool ac_packed = DIAG && pcst.action_cost_kind == MPPI_ACTION_COST_CPP; // (wave-uniform) lambda u' Sigma^-1 eps of a diagonal Sigma: products in pairs, the sum in index order as action_cost has it
        f32x2 velp[3]; // the six velocities, in pairs
#pragma unroll
        for (int i = 0; i < 6; ++i) velp[i >> 1][i & 1] = x_dev[7 + i];
        vel_s[pair][0][7][lane] = 0.0f; vel_s[pair][1][7][lane] = 0.0f; // no step handed over yet (tags)
        __syncthreads(); // w3_s, tags
        bool fresh = true; // every hand-off arrived

        // output layer of both column blocks + vel' = vel + delta
        auto finish = [&](const f32x16 &hA, const f32x16 &hB) {
            f32x2 pA[NOUT / 2], pB[NOUT / 2];
#pragma unroll
            for (int r = 0; r < NP; ++r) { // the lane half's units u(r, hh)
                const float *wp = w3_s + (8 * (r >> 2) + 4 * hh + (r & 3)) * W3LD;
                const f32x4 t0 = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp, 16));
                const f32x4 t1 = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp + 4, 16));
                const f32x2 w01 = {t0.x, t0.y}, w23 = {t0.z, t0.w}, w45 = {t1.x, t1.y};
                const f32x2 a2 = {hA[r], hA[r]}, b2 = {hB[r], hB[r]};
                pA[0] = r == 0 ? a2 * w01 : __builtin_elementwise_fma(a2, w01, pA[0]);
                pA[1] = r == 0 ? a2 * w23 : __builtin_elementwise_fma(a2, w23, pA[1]);
                pA[2] = r == 0 ? a2 * w45 : __builtin_elementwise_fma(a2, w45, pA[2]);
                pB[0] = r == 0 ? b2 * w01 : __builtin_elementwise_fma(b2, w01, pB[0]);
                pB[1] = r == 0 ? b2 * w23 : __builtin_elementwise_fma(b2, w23, pB[1]);
                pB[2] = r == 0 ? b2 * w45 : __builtin_elementwise_fma(b2, w45, pB[2]);
            }
#pragma unroll
            for (int n = 0; n < NOUT; ++n) {
                float lo = (n & 1) ? pA[n / 2].y : pA[n / 2].x, up = (n & 1) ? pB[n / 2].y : pB[n / 2].x;
                permlane32_swap(lo, up); // lane l: lo = the lower half's partial of ITS rollout, up = the upper half's
                const float y = (lo + up) + b3v[n];
                velp[n >> 1][n & 1] = velp[n >> 1][n & 1] + (y * ysd[n] + ymn[n]);
            }
        };

        for (int g = 0; g < NG; ++g) {
            float z[4 * A];
            if (SRC == SRC_PHILOX) normals_group<A>(seed, gk, base + (unsigned long long)g, z);
#pragma unroll
            for (int tl = 0; tl < 4; ++tl) {
                const int t = 4 * g + tl;
                if (t < H) { // (wave-uniform)
                    MPPI_PCT(t, 0, 0); // step start
                    // noise, v = u + eps, action cost — on pairs (v_pk_*_f32: the wave a step waits for is alone on its SIMD's vector pipe for
                    // most of it, and a lone wave issues a packed instruction in the time of a plain one). Same operations per element, same order of the sums.
                    f32x2 u2[A / 2], e2[A / 2], v2[A / 2];
                    if (SRC == SRC_PHILOX) {
                        if constexpr (DIAG) {
#pragma unroll
                            for (int q = 0; q < A / 2; ++q) e2[q] = sg2[q] * f32x2{z[tl * A + 2 * q], z[tl * A + 2 * q + 1]};
                        } else {
                            float z1[A], e1[A];
#pragma unroll
                            for (int i = 0; i < A; ++i) z1[i] = z[tl * A + i];
                            scale_noise<A, DIAG>(&pcst, z1, e1);
#pragma unroll
                            for (int i = 0; i < A; ++i) e2[i >> 1][i & 1] = e1[i];
                        }
                    } else {
#pragma unroll
                        for (int i = 0; i < A; ++i) e2[i >> 1][i & 1] = eps_hbm[(size_t)kk * HA + t * A + i];
                    }
#pragma unroll
                    for (int i = 0; i < A; ++i) u2[i >> 1][i & 1] = U_dev[t * A + i];
#pragma unroll
                    for (int q = 0; q < A / 2; ++q) v2[q] = u2[q] + e2[q];
                    float ac;
                    if (ac_packed) {
                        f32x2 pr[A / 2];
#pragma unroll
                        for (int q = 0; q < A / 2; ++q) pr[q] = u2[q] * (si2[q] * e2[q]);
                        float mix = pr[0][0];
#pragma unroll
                        for (int i = 1; i < A; ++i) mix = mix + pr[i >> 1][i & 1];
                        ac = pcst.lambda * mix;
                    } else {
                        float u[A], e[A];
#pragma unroll
                        for (int i = 0; i < A; ++i) { u[i] = u2[i >> 1][i & 1]; e[i] = e2[i >> 1][i & 1]; }
                        ac = action_cost<A, DIAG>(&pcst, u, e);
                    }
                    MPPI_PCT(t, 0, 1); // noise, action cost done
                    // inputs (prepare_data, nn_model.py:438-461): Euler angles (from P), body velocities, forces; input 15 = zero padding
                    float in[NIN + 1], eu3[3];
                    fresh = handoff_get<3>(&eu_s[pair][t & 1][0][lane], eu3, t + 1) && fresh; // Euler angles of pose t (P's step t-1; x0's before the loop)
                    if constexpr (!MF16) {
#pragma unroll
                        for (int i = 0; i < 3; ++i) in[i] = eu3[i];
#pragma unroll
                        for (int i = 0; i < 6; ++i) { in[3 + i] = velp[i >> 1][i & 1]; in[9 + i] = v2[i >> 1][i & 1]; }
                        in[NIN] = 0.0f;
                    }
                    if constexpr ((MPPI_PC_ABL & 1) != 0) { // (timing study, tools/ablate.py pc_*: no network)
                    } else if constexpr (MF16) {
                        // B operands of the four 16-rollout column blocks: per k step a 4 x 4 transpose of 16-lane rows across four registers
                        // (inputs 4 s .. 4 s + 3 of the lane's rollout -> input 4 s + g of rollout 16 c + n in lane (n, g) of register c)
                        auto relu16 = [](float v) { return __int_as_float(max(__float_as_int(v), 0)); }; // ONE v_max_i32: a float below zero (and -0) is a negative int; fmaxf costs a canonicalising v_max_f32 first, and so does a v_med3_f32 hipcc recognises
                        f32x4 acc[4];
                        MPPI_PCT(t, 0, 2); // inputs read
                        f32x2 in2[8]; // the 16 input slots in pairs
                        in2[0] = f32x2{eu3[0], eu3[1]};
                        in2[1] = f32x2{eu3[2], 0.0f};
#pragma unroll
                        for (int q = 0; q < 3; ++q) { in2[2 + q] = velp[q]; in2[5 + q] = v2[q]; }
#pragma unroll
                        for (int s1 = 0; s1 < 4; ++s1) {
                            const f32x2 x01 = (in2[2 * s1] - xm2[2 * s1]) * xr2[2 * s1], x23 = (in2[2 * s1 + 1] - xm2[2 * s1 + 1]) * xr2[2 * s1 + 1];
                            float xq[4] = {x01[0], x01[1], x23[0], x23[1]};
                            permlane32_swap(xq[0], xq[2]);
                            permlane32_swap(xq[1], xq[3]);
                            permlane16_swap(xq[0], xq[1]);
                            permlane16_swap(xq[2], xq[3]);
#pragma unroll
                            for (int c4 = 0; c4 < 4; ++c4)
                                acc[c4] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1q[s1], xq[c4], s1 == 0 ? b1q : acc[c4], 0, 0, 0);
                        }
#pragma unroll
                        for (int l = 0; l < 2; ++l) {
                            if (l + 2 <= n_hidden) { // (wave-uniform)
                                f32x4 nxt[4];
#pragma unroll
                                for (int s1 = 0; s1 < 4; ++s1)
#pragma unroll
                                    for (int c4 = 0; c4 < 4; ++c4)
                                        nxt[c4] = __builtin_amdgcn_mfma_f32_16x16x4f32(ahq[l][s1], relu16(acc[c4][s1]), s1 == 0 ? bhq[l] : nxt[c4], 0, 0, 0);
#pragma unroll
                                for (int c4 = 0; c4 < 4; ++c4) acc[c4] = nxt[c4];
                            }
                        }
                        // output layer: the lane's four units of each column block, then the sum over the four lane groups, transposed back to
                        // lane = rollout (two swap levels: the butterfly of the input transpose run backwards)
                        MPPI_PCT(t, 0, 3); // the three layers issued
                        f32x2 part[4][NOUT / 2];
#pragma unroll
                        for (int c4 = 0; c4 < 4; ++c4)
#pragma unroll
                            for (int i = 0; i < 4; ++i) {
                                const float hv = relu16(acc[c4][i]);
                                const f32x2 h2 = {hv, hv};
#pragma unroll
                                for (int o = 0; o < NOUT / 2; ++o) part[c4][o] = i == 0 ? h2 * w3q[i][o] : __builtin_elementwise_fma(h2, w3q[i][o], part[c4][o]);
                            }
#pragma unroll
                        for (int o2 = 0; o2 < NOUT / 2; ++o2) {
                            f32x2 ys;
#pragma unroll
                            for (int e = 0; e < 2; ++e) {
                                float p0 = part[0][o2][e], p1 = part[1][o2][e], p2 = part[2][o2][e], p3 = part[3][o2][e];
                                permlane32_swap(p0, p2); // p0 = [c0 g0, c0 g1, c2 g0, c2 g1], p2 = [c0 g2, c0 g3, c2 g2, c2 g3]
                                permlane32_swap(p1, p3);
                                float s02 = p0 + p2, s13 = p1 + p3; // rows: (c0: g0+g2, c0: g1+g3, c2: .., c2: ..) and the same of c1 / c3
                                permlane16_swap(s02, s13);           // s02 = [c0, c1, c2, c3] (g0+g2), s13 = [c0, c1, c2, c3] (g1+g3)
                                ys[e] = s02 + s13;
                            }
                            const f32x2 y = ys + b3p[o2];
                            const f32x2 dl = y * ysdp[o2]; // (two roundings: the compile is -ffp-contract=off, and so was the unpacked form)
                            velp[o2] = velp[o2] + (dl + ymnp[o2]);
                        }
                    } else {
                    float ba[K1H], bb[K1H];
#pragma unroll
                    for (int s1 = 0; s1 < K1H; ++s1) {
                        float ev = (in[2 * s1] - xm[2 * s1]) * xr[2 * s1], od = (in[2 * s1 + 1] - xm[2 * s1 + 1]) * xr[2 * s1 + 1];
                        permlane32_swap(ev, od); // ev: (even, odd) input of rollouts 0..31 in the lane halves; od: of rollouts 32..63
                        ba[s1] = ev; bb[s1] = od;
                    }
                    f32x16 accA, accB;
                    mfma32x2_layer1_8<NP>(accA, accB, a1, ba, bb, b1t);
                    if (n_hidden >= 2) {
                        f32x16 hA, hB;
                        if constexpr (HID == 16) mfma32x2_hidden8_lo_hi(hA, hB, accA, accB, ah[0], bht[0]);
                        else mfma32x2_hidden_lo_hi(hA, hB, accA, accB, ah[0], bht[0]);
                        if (n_hidden >= 3) {
                            if constexpr (HID == 16) mfma32x2_hidden8_hi_lo(accA, accB, hA, hB, ah[1], bht[1]);
                            else mfma32x2_hidden_hi_lo(accA, accB, hA, hB, ah[1], bht[1]);
                            finish(accA, accB);
                        } else {
                            finish(hA, hB);
                        }
                    } else {
                        finish(accA, accB);
                    }
                    }
                    MPPI_PCT(t, 0, 4); // network, output layer, velocities done
                    {
                        const float out7[7] = {velp[0][0], velp[0][1], velp[1][0], velp[1][1], velp[2][0], velp[2][1], fresh ? ac : __builtin_nanf("")};
                        handoff_put<7>(&vel_s[pair][t & 1][0][lane], out7, t + 1); // step t handed over
                    }
                    MPPI_PCT(t, 0, 5); // handed over
                    MPPI_PCT(t, 0, 6);
                }
            }
        }
    } else {
        // ================================================================================= wave P: cost, pose, Euler angles
        GenQuadConsts qc;
        const bool quad_diag = C->state_cost_kind == MPPI_STATE_COST_QUADRATIC && !C->q_full;
#pragma unroll
        for (int i = 0; i < S; ++i) { qc.goal[i] = C->goal[i]; qc.qdiag[i] = C->qdiag[i]; }
        const float dt = C->dt;
        float x[S];
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = x_dev[i];
        auto cost_of = [&](const float (&xs)[S]) { return quad_diag ? state_cost<S, false>(&qc, xs) : gen_state_cost(C, G, xs); };
        {
            const float q4[4] = {x[3], x[4], x[5], x[6]};
            float eu[3];
            euler_from_quat(q4, eu);
            eu_s[pair][1][3][lane] = 0.0f; // (tag: nothing there yet)
            handoff_put<3>(&eu_s[pair][0][0][lane], eu, 1);
        }
        __syncthreads(); // w3_s, tags
        bool fresh = true;
        const float zero6[6] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        for (int t = 0; t < H; ++t) {
            MPPI_PCT(t, 1, 0); // step start
            if (t >= 1) { // the state step t-1 produced: its pose is here, its velocities and the step's action cost come from N
                float in7[7];
                fresh = handoff_get<7>(&vel_s[pair][(t - 1) & 1][0][lane], in7, t) && fresh; // N's step t-1
#pragma unroll
                for (int i = 0; i < 6; ++i) x[7 + i] = in7[i];
                const float ac = in7[6];
                const float sc = (MPPI_PC_ABL & 2) ? x[0] : cost_of(x); // cost on the POST-step state
                const float tmp = sc + ac;   // Step_cost_result cost_base.cpp:49
                c = c + tmp;                 // path_cost        controller_base.cpp:268
            }
            MPPI_PCT(t, 1, 1); // cost done
            if constexpr ((MPPI_PC_ABL & 2) == 0) // (timing study: no pose work)
            nnauv_speed_next_state<MPPI_NNSPEED_POSE_FAST != 0>(dt, x, zero6); // the pose from the OLD velocities (nn_model.py:463-472); x[7..12] + 0 is exact
            MPPI_PCT(t, 1, 2); // pose done
            if ((MPPI_PC_ABL & 2) == 0 && t + 1 < H) {
                const float q4[4] = {x[3], x[4], x[5], x[6]};
                float eu[3];
                euler_from_quat<MPPI_NNSPEED_POSE_FAST != 0>(q4, eu);
                handoff_put<3>(&eu_s[pair][(t + 1) & 1][0][lane], eu, t + 2);
            }
            MPPI_PCT(t, 1, 5); // Euler angles done, handed over
            MPPI_PCT(t, 1, 6);
        }
        float in7[7];
        fresh = handoff_get<7>(&vel_s[pair][(H - 1) & 1][0][lane], in7, H) && fresh;
#pragma unroll
        for (int i = 0; i < 6; ++i) x[7 + i] = in7[i];
        const float sc = cost_of(x);
        c = c + (sc + in7[6]);
        c = c + sc; // terminal cost: x_H counted a second time, controller_base.cpp:271-272
        if (!fresh) c = __builtin_nanf(""); // a hand-off that never arrived: visible
        cost_s[pair][lane] = c;
        MPPI_PCT_DUMP(valid, cost + k0 + lane, c);
    }
    __syncthreads();
    if (MODE == MODE_COST_ONLY || !tile_ok) return;
    const float ct = cost_s[pair][lane];
    mlp_tile_record<A, DIAG, 2>(C, ct, valid, role, lane, kk, H, NG, SRC, eps_hbm, C->seed, (unsigned long long)C->k_offset + (unsigned long long)kk,
                                step_ctr[0] * (unsigned long long)NG, partials + (size_t)record_slot(tile, rsc) * rsb, rsc);
}

// k_rollout_auv_pc: the Fossen AUVModel (auv_model.py:282-562) as a two-wave pipeline per 64-rollout tile (r04).
// k_rollout_gen<0> is one wave per tile = ONE wave per SIMD at K = 65536, and a lone wave issues a vector instruction every ~5.4 cycles
// where two issue one every ~2.7 (profiles/r02_valu_issue.json): the kernel ran at half the issue rate by construction (0.19 ms, 980
// vector instructions per wave and step). state_dot splits by ROWS without touching any row's arithmetic:
//   wave A (pose):     lane = rollout. xd[0..6] = J(eta) nu (auv_pose_rates), the pose's Runge-Kutta update, the quaternion
//                      normalisation, the cost of the state a step produced, the noise: v = u + eps and the action cost of the NEXT step,
//                      and g(eta) of every stage state (auv_restoring: the one piece of the velocity rates that needs the quaternion alone);
//   wave B (velocity): lane = rollout. xd[7..12] = invM (tau - C nu - D nu - g(eta)) (auv_vel_rates_g: the two 6x6 products, damping,
//                      Coriolis), the velocities' Runge-Kutta update.
// Each needs the other's half of every Runge-Kutta stage state: A the 6 velocities, B the 6 restoring forces and moments (until late r04 the
// quaternion: B is the wave a stage waits for, A had the time — 0.1396 -> 0.1355 ms, still bit-identical) — handed over through double-
// buffered LDS with ONE workgroup barrier per stage (2 per step at rk2); the perturbed action travels A -> B once per step, a step ahead.
// Every row keeps the reference's operations in the reference's order (the pieces are the ones auv_state_dot is made of; the cost is summed by
// wave A alone, in index order): sample costs stay BIT-IDENTICAL to the fp32 CPU restatement.
// A workgroup is two tiles (4 waves, one per SIMD); roles by the SIMD a wave runs on, as in k_rollout_nnspeed_pc.
constexpr int kAuvPcThreads = 256;

template <bool DIAG>
__global__ __launch_bounds__(kAuvPcThreads, 2) void k_rollout_auv_pc(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const float *__restrict__ x_dev, const float *__restrict__ U_dev,
    const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr, float *__restrict__ cost, float *__restrict__ partials,
    const int SRC, const int MODE, const int rsb, const int rsc, const int n_tiles, const int balance)
{
    constexpr int S = kGenS, A = kGenA;
    __shared__ float g_s[2][2][6][64];    // [tile of the workgroup][barrier parity][restoring forces g(eta) of the stage state][rollout]   A -> B
    __shared__ float vel_s[2][2][6][64];  // [tile][barrier parity][velocities of the stage state][rollout]                       B -> A
    __shared__ float act_s[2][2][7][64];  // [tile][step parity][perturbed action v, action cost][rollout]                        A -> B (v), A keeps the cost
    __shared__ float cost_s[2][64];
    __shared__ int simd_s[4];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pair = wave_hw >> 1, role = wave_hw & 1; // role 0 = velocity wave (the heavier one), 1 = pose wave
    {
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_REG_HW_ID[5:4]
        if (lane == 0) simd_s[wave_hw] = simd;
        __syncthreads();
        const int s0 = simd_s[0], s1 = simd_s[1], s2 = simd_s[2], s3 = simd_s[3];
        if (balance && ((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15) {
            const int gen = (int)(blockIdx.x >> 8);
            pair = simd & 1;
            role = ((simd >> 1) ^ gen) & 1;
        }
        pair = __builtin_amdgcn_readfirstlane(pair);
        role = __builtin_amdgcn_readfirstlane(role);
    }
    const int tile = 2 * (int)blockIdx.x + pair;
    const bool tile_ok = tile < n_tiles; // the second tile of the last workgroup may not exist: its waves still keep every barrier
    const int k0 = tile * 64;
    const bool valid = tile_ok && (k0 + lane) < K;
    const int kk = min(k0 + lane, K - 1);
    const int rk = G->rk;
    const float dt = G->dt;
    int nbar = 0; // barriers passed so far: the parity of the stage hand-off buffers (the same sequence in both waves)

    if (role == 0) {
        // ================================================================================= wave B: velocities
        AuvLocal al;
        al.load(G);
        float vel[6];
#pragma unroll
        for (int i = 0; i < 6; ++i) vel[i] = x_dev[7 + i];
        // stage hand-off: publish the stage state's velocities, barrier, fetch its restoring forces (the one piece of the velocity rates that
        // is a function of the quaternion alone: the pose wave evaluates it)
        auto swap_stage = [&](const float (&vs)[6], float (&gs)[6]) {
#pragma unroll
            for (int i = 0; i < 6; ++i) vel_s[pair][nbar & 1][i][lane] = vs[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 6; ++i) gs[i] = g_s[pair][nbar & 1][i][lane];
            ++nbar;
        };
        for (int t = 0; t < H; ++t) {
            float q[6], v[A], k1[6], tmp[6]; // (q: g(eta) of the stage state)
            swap_stage(vel, q);
#pragma unroll
            for (int i = 0; i < A; ++i) v[i] = act_s[pair][t & 1][i][lane]; // to_apply of step t (wave A prepared it a step ahead)
            auv_vel_rates_g(&al, q, vel, v, k1);
            if (rk == 2) {
                float vs[6], k2[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) vs[i] = vel[i] + dt * k1[i];
                swap_stage(vs, q);
                auv_vel_rates_g(&al, q, vs, v, k2);
#pragma unroll
                for (int i = 0; i < 6; ++i) tmp[i] = (dt / 2.0f) * (k1[i] + k2[i]);
            } else if (rk == 4) { // the reference's formula, k4*dt inside the sum (:299-300)
                float vs[6], k2[6], k3[6], k4[6];
#pragma unroll
                for (int i = 0; i < 6; ++i) vs[i] = vel[i] + (dt * k1[i]) / 2.0f;
                swap_stage(vs, q);
                auv_vel_rates_g(&al, q, vs, v, k2);
#pragma unroll
                for (int i = 0; i < 6; ++i) vs[i] = vel[i] + (dt * k2[i]) / 2.0f;
                swap_stage(vs, q);
                auv_vel_rates_g(&al, q, vs, v, k3);
#pragma unroll
                for (int i = 0; i < 6; ++i) vs[i] = vel[i] + dt * k3[i];
                swap_stage(vs, q);
                auv_vel_rates_g(&al, q, vs, v, k4);
                const float sixth = (float)(1.0 / 6.0);
#pragma unroll
                for (int i = 0; i < 6; ++i) tmp[i] = (sixth * ((k1[i] + 2.0f * k2[i]) + (2.0f * k3[i] + k4[i] * dt))) * dt;
            } else {
#pragma unroll
                for (int i = 0; i < 6; ++i) tmp[i] = k1[i] * dt;
            }
#pragma unroll
            for (int i = 0; i < 6; ++i) vel[i] = vel[i] + tmp[i];
        }
        float q_unused[6];
        swap_stage(vel, q_unused); // the velocities of x_H for wave A's last step cost and the terminal cost
    } else {
        // ================================================================================= wave A: pose, cost, noise
        if (balance) __builtin_amdgcn_s_setprio(3); // (one round of the grid only: beyond it the age order staggers the workgroups' phases, as in k_rollout_pc)
        // the wave a stage waits for goes first on its SIMD (measured: this one — 0.1366 ms without priorities, 0.1342 with the
                                       // velocity wave first, 0.1278 with this one first; the other kind fills the gaps)
        GenQuadConsts qc;
        const bool quad_diag = C->state_cost_kind == MPPI_STATE_COST_QUADRATIC && !C->q_full;
#pragma unroll
        for (int i = 0; i < S; ++i) { qc.goal[i] = C->goal[i]; qc.qdiag[i] = C->qdiag[i]; }
        PcProducerConsts<A> pcst; // Sigma, Sigma^-1, lambda: a kernel-local copy (no re-fetch behind the barriers)
        pcst.template load<DIAG>(C);
        const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
        const unsigned long long seed = C->seed;
        const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
        auto cost_of = [&](const float (&xs)[S]) { return quad_diag ? state_cost<S, false>(&qc, xs) : gen_state_cost(C, G, xs); };
        float x[S], c = 0.0f, z[4 * A];
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = x_dev[i];
        // v = u + eps and the action cost of step t -> act_s[t & 1] (mPrepareAction / mPrepareNoise, controller_base.cpp:205-213, :258)
        auto prepare = [&](int t) {
            float e[A], u[A];
            if (SRC == SRC_PHILOX) {
                if ((t & 3) == 0) normals_group<A>(seed, gk, base + (unsigned long long)(t >> 2), z);
                float z1[A];
                const int tl = t & 3; // wave-uniform: four statically indexed copies instead of a dynamically indexed register array
                if (tl == 0) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[i]; }
                else if (tl == 1) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[A + i]; }
                else if (tl == 2) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[2 * A + i]; }
                else { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[3 * A + i]; }
                scale_noise<A, DIAG>(&pcst, z1, e);
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
            }
#pragma unroll
            for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; act_s[pair][t & 1][i][lane] = u[i] + e[i]; }
            act_s[pair][t & 1][6][lane] = action_cost<A, DIAG>(&pcst, u, e);
        };
        // stage hand-off: publish the restoring forces g(eta) of the stage state (auv_restoring: a function of its quaternion alone, and this
        // wave has the time the velocity wave lacks), barrier, fetch its velocities
        struct { float fng_z, fnb_z, cog[3], cob[3]; } rl;
        rl.fng_z = G->fng_z; rl.fnb_z = G->fnb_z;
#pragma unroll
        for (int i = 0; i < 3; ++i) { rl.cog[i] = G->cog[i]; rl.cob[i] = G->cob[i]; }
        auto swap_stage = [&](const float (&ps)[7], float (&vs)[6]) {
            const float q4[4] = {ps[3], ps[4], ps[5], ps[6]};
            float g6[6];
            auv_restoring(&rl, q4, g6);
#pragma unroll
            for (int i = 0; i < 6; ++i) g_s[pair][nbar & 1][i][lane] = g6[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 6; ++i) vs[i] = vel_s[pair][nbar & 1][i][lane];
            ++nbar;
        };
        prepare(0);
        for (int t = 0; t < H; ++t) {
            float pose[7], vel[6], k1[7], tmp[7];
#pragma unroll
            for (int i = 0; i < 7; ++i) pose[i] = x[i];
            swap_stage(pose, vel);
            if (t >= 1) { // the state step t-1 produced is complete now: its cost, with that step's action cost
#pragma unroll
                for (int i = 0; i < 6; ++i) x[7 + i] = vel[i];
                const float sc = cost_of(x);                           // cost on the POST-step state
                const float step_c = sc + act_s[pair][(t - 1) & 1][6][lane]; // Step_cost_result cost_base.cpp:49
                c = c + step_c;                                        // path_cost        controller_base.cpp:268
            }
            const float q0[4] = {pose[3], pose[4], pose[5], pose[6]};
            auv_pose_rates(q0, vel, k1);
            if (rk == 2) {
                float ps[7], vs[6], k2[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) ps[i] = pose[i] + dt * k1[i];
                swap_stage(ps, vs);
                const float q1[4] = {ps[3], ps[4], ps[5], ps[6]};
                auv_pose_rates(q1, vs, k2);
#pragma unroll
                for (int i = 0; i < 7; ++i) tmp[i] = (dt / 2.0f) * (k1[i] + k2[i]);
            } else if (rk == 4) {
                float ps[7], vs[6], k2[7], k3[7], k4[7];
#pragma unroll
                for (int i = 0; i < 7; ++i) ps[i] = pose[i] + (dt * k1[i]) / 2.0f;
                swap_stage(ps, vs);
                { const float qq[4] = {ps[3], ps[4], ps[5], ps[6]}; auv_pose_rates(qq, vs, k2); }
#pragma unroll
                for (int i = 0; i < 7; ++i) ps[i] = pose[i] + (dt * k2[i]) / 2.0f;
                swap_stage(ps, vs);
                { const float qq[4] = {ps[3], ps[4], ps[5], ps[6]}; auv_pose_rates(qq, vs, k3); }
#pragma unroll
                for (int i = 0; i < 7; ++i) ps[i] = pose[i] + dt * k3[i];
                swap_stage(ps, vs);
                { const float qq[4] = {ps[3], ps[4], ps[5], ps[6]}; auv_pose_rates(qq, vs, k4); }
                const float sixth = (float)(1.0 / 6.0);
#pragma unroll
                for (int i = 0; i < 7; ++i) tmp[i] = (sixth * ((k1[i] + 2.0f * k2[i]) + (2.0f * k3[i] + k4[i] * dt))) * dt;
            } else {
#pragma unroll
                for (int i = 0; i < 7; ++i) tmp[i] = k1[i] * dt;
            }
#pragma unroll
            for (int i = 0; i < 7; ++i) x[i] = x[i] + tmp[i];
            normalize_quat(x);
            if (t + 1 < H) prepare(t + 1); // a step ahead: wave B reads it right behind the next step's first barrier
        }
        float pose[7], vel[6];
#pragma unroll
        for (int i = 0; i < 7; ++i) pose[i] = x[i];
        swap_stage(pose, vel);
#pragma unroll
        for (int i = 0; i < 6; ++i) x[7 + i] = vel[i];
        const float sc = cost_of(x);
        c = c + (sc + act_s[pair][(H - 1) & 1][6][lane]);
        c = c + sc; // terminal cost: x_H counted a second time, controller_base.cpp:271-272
        cost_s[pair][lane] = c;
        if (valid) cost[k0 + lane] = c;
    }
    __syncthreads();
    if (MODE == MODE_COST_ONLY || !tile_ok) return;
    const float ct = cost_s[pair][lane];
    mlp_tile_record<A, DIAG, 2>(C, ct, valid, role, lane, kk, H, NG, SRC, eps_hbm, C->seed, (unsigned long long)C->k_offset + (unsigned long long)kk,
                                step_ctr[0] * (unsigned long long)NG, partials + (size_t)record_slot(tile, rsc) * rsb, rsc);
}

// k_rollout_nnauv_pc: NNAUVModel (Dense(32) x 1..3 + Dense(13), nn_model.py:215-304) as a two-wave pipeline per 64-rollout tile (r04), the
// split of k_rollout_nnspeed_pc applied to this model: in k_rollout_nnauv32 a wave is 32 rollouts and everything per rollout — noise, the
// 13-term cost, the 7-pair output layer, the state update: 1267 vector instructions per tile and step — runs on 64 lanes for 32 results.
//   wave N (network): lane = rollout. Inputs (its state without the position, the perturbed action from C), the Dense stack on TWO column
//       blocks of 32 rollouts per weight register (mfma32x2_*), the 13 outputs from the accumulators, x' = x + denorm(y);
//   wave C (cost):    lane = rollout. The cost of the state the previous step produced (any cost_base kind), and the noise: v = u + eps
//       and the action cost of the NEXT step.
// N -> C: the 13-state; C -> N: the 6 perturbed actions, a step ahead; one workgroup barrier per step. Same arithmetic per rollout as
// k_rollout_nnauv32 (the output layer's half sums in the same order): same bars. Two tiles per workgroup, roles by SIMD.
constexpr int kNnauvPcThreads = 256;

template <bool DIAG>
__global__ __launch_bounds__(kNnauvPcThreads, 2) void k_rollout_nnauv_pc(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr,
    float *__restrict__ cost, float *__restrict__ partials, const int SRC, const int MODE, const int rsb, const int rsc, const int n_tiles,
    const int balance)
{
    constexpr int S = kGenS, A = kGenA, NIN = kGenNin, XOFF = 3, SP = (S + 1) / 2, HID = 32, K1H = NIN / 2, W3LD = 16;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float w3_s[HID * W3LD]; // output-layer rows, padded to 16
    __shared__ float xs_s[2][2][S][64];   // [tile of the workgroup][step parity][state the step produced][rollout]     N -> C
    __shared__ float act_s[2][2][7][64];  // [tile][step parity][perturbed action v, action cost][rollout]               C -> N (v), C keeps the cost
    __shared__ float cost_s[2][64];
    __shared__ __attribute__((aligned(16))) float cst_s[5][16]; // xmean, 1/xstd, b3, ystd, ymean: wave-uniform, read back by broadcast every step —
                                                                // kept in registers they cost the network wave 71 VGPRs, the kernel its second wave per SIMD
    __shared__ int simd_s[4];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_hw = __builtin_amdgcn_readfirstlane(tid >> 6);
    int pair = wave_hw >> 1, role = wave_hw & 1; // role 0 = network, 1 = cost
    {
        const int simd = (int)__builtin_amdgcn_s_getreg((1 << 11) | (4 << 6) | 4); // HW_REG_HW_ID[5:4]
        if (lane == 0) simd_s[wave_hw] = simd;
        __syncthreads();
        const int s0 = simd_s[0], s1 = simd_s[1], s2 = simd_s[2], s3 = simd_s[3];
        if (balance && ((1 << s0) | (1 << s1) | (1 << s2) | (1 << s3)) == 15) {
            const int gen = (int)(blockIdx.x >> 8);
            pair = simd & 1;
            role = ((simd >> 1) ^ gen) & 1;
        }
        pair = __builtin_amdgcn_readfirstlane(pair);
        role = __builtin_amdgcn_readfirstlane(role);
    }
    const int tile = 2 * (int)blockIdx.x + pair;
    const bool tile_ok = tile < n_tiles;
    const int k0 = tile * 64;
    const bool valid = tile_ok && (k0 + lane) < K;
    const int kk = min(k0 + lane, K - 1);
    const int n_hidden = M->n_layers - 1;
    const float *W3g = M->Wl[n_hidden];
    const int ld3 = M->ld[n_hidden]; // 14: the host pads the 13-wide output layer to an even width
    for (int i = tid; i < HID * W3LD; i += kNnauvPcThreads) w3_s[i] = (i & 15) < S ? W3g[(i >> 4) * ld3 + (i & 15)] : 0.0f;
    if (tid < 16) {
        const float *b3c = M->bl[n_hidden];
        cst_s[0][tid] = tid < NIN ? M->xmean[tid] : 0.0f;
        cst_s[1][tid] = tid < NIN ? 1.0f / M->xstd[tid] : 0.0f;
        cst_s[2][tid] = tid < S ? b3c[tid] : 0.0f;
        cst_s[3][tid] = tid < S ? M->ystd[tid] : 0.0f;
        cst_s[4][tid] = tid < S ? M->ymean[tid] : 0.0f;
    }

    if (role == 0) {
        // ================================================================================= wave N: network, state
        if (balance) __builtin_amdgcn_s_setprio(3); // (one round of the grid only: beyond it the age order staggers the workgroups' phases, as in k_rollout_pc)
        // the wave a step waits for goes first on its SIMD (the other kind fills the gaps)
        const int j = lane & 31, hh = lane >> 5;
        auto unit_of = [](int r, int half) { return 8 * (r >> 2) + 4 * half + (r & 3); };
        float a1[K1H];
#pragma unroll
        for (int s1 = 0; s1 < K1H; ++s1) a1[s1] = M->Wl[0][(2 * s1 + hh) * HID + j];
        f32x16 b1t, bht[2];
        float ah[2][16];
#pragma unroll
        for (int r = 0; r < 16; ++r) b1t[r] = M->bl[0][unit_of(r, hh)];
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            const bool have = l + 2 <= n_hidden;
            const float *Wl = have ? M->Wl[l + 1] : M->Wl[0], *bl = have ? M->bl[l + 1] : M->bl[0];
#pragma unroll
            for (int s1 = 0; s1 < 16; ++s1) ah[l][s1] = have ? Wl[unit_of(s1, hh) * HID + j] : 0.0f;
#pragma unroll
            for (int r = 0; r < 16; ++r) bht[l][r] = have ? bl[unit_of(r, hh)] : 0.0f;
        }
        float x[S];
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = x_dev[i];
        __syncthreads(); // w3_s, and the perturbed action of step 0

        // output layer of both column blocks + next_state (state + de-normalised delta, nn_model.py:303-304)
        auto finish = [&](const f32x16 &hA, const f32x16 &hB) {
            f32x2 pA[SP], pB[SP];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float *wp = w3_s + (8 * (r >> 2) + 4 * hh + (r & 3)) * W3LD;
                float wv[W3LD];
#pragma unroll
                for (int q4 = 0; q4 < W3LD / 4; ++q4) {
                    const f32x4 t4 = *static_cast<const f32x4 *>(__builtin_assume_aligned(wp + 4 * q4, 16));
                    wv[4 * q4] = t4.x; wv[4 * q4 + 1] = t4.y; wv[4 * q4 + 2] = t4.z; wv[4 * q4 + 3] = t4.w;
                }
                const f32x2 a2 = {hA[r], hA[r]}, b2 = {hB[r], hB[r]};
#pragma unroll
                for (int p2 = 0; p2 < SP; ++p2) {
                    const f32x2 w2 = {wv[2 * p2], wv[2 * p2 + 1]};
                    pA[p2] = r == 0 ? a2 * w2 : __builtin_elementwise_fma(a2, w2, pA[p2]);
                    pB[p2] = r == 0 ? b2 * w2 : __builtin_elementwise_fma(b2, w2, pB[p2]);
                }
            }
#pragma unroll
            for (int n = 0; n < S; ++n) {
                float lo = (n & 1) ? pA[n / 2].y : pA[n / 2].x, up = (n & 1) ? pB[n / 2].y : pB[n / 2].x;
                permlane32_swap(lo, up); // lane l: lo = the lower half's partial of ITS rollout, up = the upper half's
                const float y = (lo + up) + cst_s[2][n];
                x[n] = x[n] + (y * cst_s[3][n] + cst_s[4][n]);
            }
        };

        for (int t = 0; t < H; ++t) {
            float in[NIN];
#pragma unroll
            for (int i = 0; i < S - XOFF; ++i) in[i] = x[XOFF + i];
#pragma unroll
            for (int i = 0; i < A; ++i) in[S - XOFF + i] = act_s[pair][t & 1][i][lane]; // to_apply of step t (wave C prepared it a step ahead)
            float ba[K1H], bb[K1H];
#pragma unroll
            for (int s1 = 0; s1 < K1H; ++s1) {
                float ev = (in[2 * s1] - cst_s[0][2 * s1]) * cst_s[1][2 * s1], od = (in[2 * s1 + 1] - cst_s[0][2 * s1 + 1]) * cst_s[1][2 * s1 + 1];
                permlane32_swap(ev, od); // ev: (even, odd) input of rollouts 0..31 in the lane halves; od: of rollouts 32..63
                ba[s1] = ev; bb[s1] = od;
            }
            f32x16 accA, accB;
            mfma32x2_layer1_8<16>(accA, accB, a1, ba, bb, b1t);
            if (n_hidden >= 2) {
                f32x16 hA, hB;
                mfma32x2_hidden_lo_hi(hA, hB, accA, accB, ah[0], bht[0]);
                if (n_hidden >= 3) {
                    mfma32x2_hidden_hi_lo(accA, accB, hA, hB, ah[1], bht[1]);
                    finish(accA, accB);
                } else {
                    finish(hA, hB);
                }
            } else {
                finish(accA, accB);
            }
#pragma unroll
            for (int i = 0; i < S; ++i) xs_s[pair][t & 1][i][lane] = x[i];
            __syncthreads(); // step t handed over
        }
    } else {
        // ================================================================================= wave C: cost, noise
        GenQuadConsts qc;
        const bool quad_diag = C->state_cost_kind == MPPI_STATE_COST_QUADRATIC && !C->q_full;
#pragma unroll
        for (int i = 0; i < S; ++i) { qc.goal[i] = C->goal[i]; qc.qdiag[i] = C->qdiag[i]; }
        PcProducerConsts<A> pcst;
        pcst.template load<DIAG>(C);
        const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
        const unsigned long long seed = C->seed;
        const unsigned long long gk = (unsigned long long)C->k_offset + (unsigned long long)kk;
        auto cost_of = [&](const float (&xs)[S]) { return quad_diag ? state_cost<S, false>(&qc, xs) : gen_state_cost(C, G, xs); };
        float c = 0.0f, z[4 * A];
        auto prepare = [&](int t) { // v = u + eps and the action cost of step t -> act_s[t & 1]
            float e[A], u[A];
            if (SRC == SRC_PHILOX) {
                if ((t & 3) == 0) normals_group<A>(seed, gk, base + (unsigned long long)(t >> 2), z);
                float z1[A];
                const int tl = t & 3;
                if (tl == 0) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[i]; }
                else if (tl == 1) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[A + i]; }
                else if (tl == 2) { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[2 * A + i]; }
                else { _Pragma("unroll") for (int i = 0; i < A; ++i) z1[i] = z[3 * A + i]; }
                scale_noise<A, DIAG>(&pcst, z1, e);
            } else {
#pragma unroll
                for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
            }
#pragma unroll
            for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; act_s[pair][t & 1][i][lane] = u[i] + e[i]; }
            act_s[pair][t & 1][6][lane] = action_cost<A, DIAG>(&pcst, u, e);
        };
        prepare(0);
        __syncthreads(); // w3_s, and the perturbed action of step 0
        float x[S];
        for (int t = 0; t < H; ++t) {
            if (t >= 1) { // the state step t-1 produced
#pragma unroll
                for (int i = 0; i < S; ++i) x[i] = xs_s[pair][(t - 1) & 1][i][lane];
                const float sc = cost_of(x);                                   // cost on the POST-step state
                const float step_c = sc + act_s[pair][(t - 1) & 1][6][lane]; // Step_cost_result cost_base.cpp:49
                c = c + step_c;                                                // path_cost        controller_base.cpp:268
            }
            if (t + 1 < H) prepare(t + 1);
            __syncthreads(); // step t handed over
        }
#pragma unroll
        for (int i = 0; i < S; ++i) x[i] = xs_s[pair][(H - 1) & 1][i][lane];
        const float sc = cost_of(x);
        c = c + (sc + act_s[pair][(H - 1) & 1][6][lane]);
        c = c + sc; // terminal cost: x_H counted a second time, controller_base.cpp:271-272
        cost_s[pair][lane] = c;
        if (valid) cost[k0 + lane] = c;
    }
    __syncthreads();
    if (MODE == MODE_COST_ONLY || !tile_ok) return;
    const float ct = cost_s[pair][lane];
    mlp_tile_record<A, DIAG, 2>(C, ct, valid, role, lane, kk, H, NG, SRC, eps_hbm, C->seed, (unsigned long long)C->k_offset + (unsigned long long)kk,
                                step_ctr[0] * (unsigned long long)NG, partials + (size_t)record_slot(tile, rsc) * rsb, rsc);
}

// k_rollout_nnauv32_bx3: the same NNAUVModel network on the BF16 matrix cores at fp32-class accuracy (opt-in MPPI_FLAG_MLP_BF16X3), the
// design of k_rollout_mlp32_bx3 (mppi_mlp32b.hip.h): every operand split x = hi + lo (two bf16), three products per term, a 32-wide
// layer = 6 v_mfma_f32_32x32x16_bf16, registers 8 kb .. 8 kb + 7 of a lane (relu'd, split) are the next layer's B fragment of k-block
// kb, and the output layer rides the matrix core too: A row m carries output (m & 3) + 4 (m >> 3) — 13 of the 16 a lane half holds,
// each output once per half — so register n of EVERY lane is output n of its rollout. The 16 inputs fill layer 1's k-block: its bias
// is the C operand. Builtin MFMAs (hipcc sees their hazards), compiled in VGPR form.
template <bool DIAG>
__global__ __launch_bounds__(kNnauv32Threads) void k_rollout_nnauv32_bx3(
    const DevConsts *__restrict__ C, const GenConsts *__restrict__ G, const MlpDev *__restrict__ M, const float *__restrict__ x_dev,
    const float *__restrict__ U_dev, const float *__restrict__ eps_hbm, const unsigned long long *__restrict__ step_ctr,
    float *__restrict__ cost, float *__restrict__ partials, const int SRC, const int MODE, const int rsb, const int rsc)
{
    constexpr int S = kGenS, A = kGenA, NIN = kGenNin, XOFF = 3, HID = 32;
    static_assert(NIN == 16 && S <= 16, "the inputs are exactly one k-block; the outputs fit the 16 registers both lane halves share");
    typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
    __shared__ float z_s[2][4 * A][32]; // per wave: the normals of one horizon group
    __shared__ float cost_s[64];
    const int H = C->H, HA = H * A, K = C->K_local;
    const int NG = (H + 3) / 4;
    const int tid = threadIdx.x, w = tid >> 6, lane = tid & 63, j = lane & 31, hh = lane >> 5;
    const int k0 = blockIdx.x * 64;
    const int kk = min(k0 + 32 * w + j, K - 1); // rollouts past K recompute the last sample, outside every sum
    const unsigned long long base = step_ctr[0] * (unsigned long long)NG;
    const unsigned long long seed = C->seed;
    const unsigned int gk = (unsigned int)C->k_offset + (unsigned int)kk;
    const int n_hidden = M->n_layers - 1;
    auto frag = [](const int (&v)[4]) { return __builtin_bit_cast(bf16x8, i32x4{v[0], v[1], v[2], v[3]}); };
    auto k_unit = [&](int kb, int e) { return 16 * kb + 8 * (e >> 2) + 4 * hh + (e & 3); }; // hidden unit behind k slot 8 hh + e of k-block kb
    auto row_of = [&](int r) { return (r & 3) + 8 * (r >> 2) + 4 * hh; };                     // accumulator register r of this lane half

    // ---- stationary operands (hi, lo)
    bf16x8 a1h, a1l, ahh[2][2], ahl[2][2], aoh[2], aol[2];
    f32x16 b1t, bht[2], bot;
    {
        int hi[4], lo[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) // layer 1: natural k order 8 hh + e
            split_pair(M->Wl[0][(8 * hh + 2 * q) * HID + j], M->Wl[0][(8 * hh + 2 * q + 1) * HID + j], hi[q], lo[q]);
        a1h = frag(hi); a1l = frag(lo);
#pragma unroll
        for (int r = 0; r < 16; ++r) b1t[r] = M->bl[0][row_of(r)];
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
        const int ld3 = M->ld[n_hidden];               // 14: the host pads the 13-wide output layer to an even width
        const int out_j = (j & 3) + 4 * (j >> 3);      // the output this lane's A row carries (>= S: none)
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                split_pair(out_j < S ? W3g[k_unit(kb, 2 * q) * ld3 + (out_j < S ? out_j : 0)] : 0.0f,
                           out_j < S ? W3g[k_unit(kb, 2 * q + 1) * ld3 + (out_j < S ? out_j : 0)] : 0.0f, hi[q], lo[q]);
            aoh[kb] = frag(hi); aol[kb] = frag(lo);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) bot[r] = r < S ? b3g[r < S ? r : 0] : 0.0f; // register r = row (r & 3) + 8 (r >> 2) + 4 hh = output r
    }
    float xms[8], xrs[8], ysd[S], ymn[S]; // the lane's 8 inputs are k = 8 hh + e
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        xms[e] = hh ? M->xmean[8 + e] : M->xmean[e];
        xrs[e] = 1.0f / (hh ? M->xstd[8 + e] : M->xstd[e]);
    }
#pragma unroll
    for (int i = 0; i < S; ++i) { ysd[i] = M->ystd[i]; ymn[i] = M->ymean[i]; }
    float x[S], c = 0.0f;
#pragma unroll
    for (int i = 0; i < S; ++i) x[i] = x_dev[i];
    __syncthreads();

    auto next_b = [&](const f32x16 &acc, bf16x8 (&bh)[2], bf16x8 (&bl)[2]) { // relu + split: the next layer's two B fragments
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
            int hi[4], lo[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) { // v_med3_f32(x, 0, +inf): a relu hipcc can see behind its own MFMAs (mppi_mlp32b.hip.h)
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
                    const float4 n = normals_of_block(seed, (unsigned long long)gk, (base + (unsigned long long)(t >> 2)) * A + q);
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
            scale_noise<A, DIAG>(C, z1, e);
        } else {
#pragma unroll
            for (int i = 0; i < A; ++i) e[i] = eps_hbm[(size_t)kk * HA + t * A + i];
        }
#pragma unroll
        for (int i = 0; i < A; ++i) { u[i] = U_dev[t * A + i]; v[i] = u[i] + e[i]; }
        const float ac = action_cost<A, DIAG>(C, u, e);
        // layer 1: the B fragment of lane (j, hh) is inputs 8 hh .. 8 hh + 7 of rollout j; input i = x[3 + i] for i < 10, else v[i - 10]
        auto raw = [&](int i) { return i < S - XOFF ? x[i < S - XOFF ? XOFF + i : 0] : v[i >= S - XOFF ? i - (S - XOFF) : 0]; };
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
        f32x16 acc = b1t;
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
        acc = layer(aoh, aol, bh, bl, bot); // register n (< 13) of every lane: output n of rollout j, bias included
#pragma unroll
        for (int n = 0; n < S; ++n) x[n] = x[n] + (acc[n] * ysd[n] + ymn[n]); // next_state = state + delta (nn_model.py:303-304)
        const float sc = gen_state_cost(C, G, x); // cost on the POST-step state
        const float tmp = sc + ac;
        c = c + tmp;
    }
    c = c + gen_state_cost(C, G, x); // terminal cost, controller_base.cpp:271-272
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

// ElipseCost3D's three terms for k states -> out [k, 3] (position, orientation, velocity error)
__global__ void k_e3_terms(const GenConsts *__restrict__ G, const float *__restrict__ x, int k, int in_plane, float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    float xs[kGenS], pc, oc, vc;
#pragma unroll
    for (int j = 0; j < kGenS; ++j) xs[j] = x[(size_t)i * kGenS + j];
    e3_terms(G, xs, in_plane != 0, pc, oc, vc);
    out[(size_t)i * 3] = pc; out[(size_t)i * 3 + 1] = oc; out[(size_t)i * 3 + 2] = vc;
}

// D(nu) and C(nu) as the 6x6 MATRICES the reference builds (auv_model.py:482-545) — the rollout never materialises them
// (auv_state_dot forms D nu and C nu directly); used by k_auv_pieces so that the reference's matrix expectations can be
// checked on the device and the fast forms against the matrices.
__device__ __forceinline__ void auv_damping_matrix(const GenConsts *__restrict__ G, const float (&vel)[6], float (&D)[36])
{
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j < 6; ++j) {
            const float d1 = (-1.0f * G->lin_damp[i * 6 + j]) - (vel[0] * G->lin_damp_fwd[i * 6 + j]);
            D[i * 6 + j] = i == j ? d1 + (-1.0f * (G->quad_damp[i] * fabsf(vel[i]))) : d1 + (-0.0f);
        }
}

__device__ __forceinline__ void auv_coriolis_matrix(const GenConsts *__restrict__ G, const float (&vel)[6], float (&Cm)[36])
{
    float a[2][3];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const float *m = G->mtot + (3 * blk + i) * 6;
            a[blk][i] = ((m[0] * vel[0] + m[1] * vel[1]) + m[2] * vel[2]) + ((m[3] * vel[3] + m[4] * vel[4]) + m[5] * vel[5]);
        }
#pragma unroll
    for (int i = 0; i < 36; ++i) Cm[i] = 0.0f;
#pragma unroll
    for (int blk = 0; blk < 2; ++blk) {
        const float *v = a[blk];
        const float ns[9] = {-0.0f, v[2], -v[1], -v[2], -0.0f, v[0], v[1], -v[0], -0.0f}; // -skew(v)
#pragma unroll
        for (int i = 0; i < 3; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                if (blk == 0) { Cm[i * 6 + 3 + j] = ns[i * 3 + j]; Cm[(3 + i) * 6 + j] = ns[i * 3 + j]; }
                else Cm[(3 + i) * 6 + 3 + j] = ns[i * 3 + j];
            }
    }
}

// the pieces of AUVModel the reference's tests call one by one (scripts/test.py:264-539), per (state, action) pair:
// out [k, 124] = rotBtoI [9] | TBtoIquat [12] | C nu [6] | D nu [6] | g [6] | state_dot [13] | D [36] | C [36]
constexpr int kAuvPieces = 124;
__global__ void k_auv_pieces(const GenConsts *__restrict__ G, const float *__restrict__ x, const float *__restrict__ u, int k,
                             float *__restrict__ out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    float xs[kGenS], us[kGenA], xd[kGenS], rot[9], T[12], pc[18], D[36], Cm[36];
#pragma unroll
    for (int j = 0; j < kGenS; ++j) xs[j] = x[(size_t)i * kGenS + j];
#pragma unroll
    for (int j = 0; j < kGenA; ++j) us[j] = u[(size_t)i * kGenA + j];
    const float q[4] = {xs[3], xs[4], xs[5], xs[6]};
    const float vel[6] = {xs[7], xs[8], xs[9], xs[10], xs[11], xs[12]};
    auv_b2i(q, rot, T);
    auv_state_dot(G, xs, us, xd, pc);
    auv_damping_matrix(G, vel, D);
    auv_coriolis_matrix(G, vel, Cm);
    float *o = out + (size_t)i * kAuvPieces;
#pragma unroll
    for (int j = 0; j < 9; ++j) o[j] = rot[j];
#pragma unroll
    for (int j = 0; j < 12; ++j) o[9 + j] = T[j];
#pragma unroll
    for (int j = 0; j < 18; ++j) o[21 + j] = pc[j];
#pragma unroll
    for (int j = 0; j < kGenS; ++j) o[39 + j] = xd[j];
#pragma unroll
    for (int j = 0; j < 36; ++j) { o[52 + j] = D[j]; o[88 + j] = Cm[j]; }
}

} // namespace mppi
