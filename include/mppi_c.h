/*
 * mppi_c.h — C-ABI of the MI355X-native MPPI control step (libmppi_hip.so).
 *
 * This is the drop-in boundary for ONE path of NicolayP/mppi-tf: the per-control-step MPPI
 * update (perturb K control sequences, roll the model H steps, cost every step, soft-min
 * weight, reduce to the new nominal sequence).  The reference has no FFI of its own; its
 * boundary is the public C++ API of `ControllerBase` (include/controller_base.hpp:60-109 in
 * the reference) as driven by its host loop (src/main.cpp:30-45).  Every entry point below
 * names the reference interface it replaces.  include/mppi/controller_base.hpp re-creates
 * the reference's C++ classes on top of these calls; INTEGRATION.md shows the binding.
 *
 * Conventions
 *  - plain C, no torch / HIP types in signatures; `void *stream` is a hipStream_t (NULL = the
 *    handle's own, non-blocking stream; to enqueue on the legacy default stream — torch's default — pass
 *    hipStreamLegacy, (hipStream_t)1: a step must run on a stream that is ordered against whatever produces
 *    x_dev and consumes u_dev / the records);
 *  - all host buffers are caller-owned, fp32, row-major, with the reference's trailing
 *    singleton dropped:  x[s]  U[tau,a]  eps[K,tau,a]  cost[K];
 *  - pointers named *_dev are DEVICE pointers on the handle's GPU;
 *  - every call returns an mppi_status (0 = OK) and never aborts (the reference LOG(FATAL)s
 *    on a TF error, controller_base.cpp:141); mppi_last_error() gives the text;
 *  - one handle = one controller, NOT thread-safe per handle (same as the reference: one
 *    session, mutable m_U); distinct handles are independent;
 *  - host-pointer calls are synchronous; *_device calls only enqueue on the stream.
 */
#ifndef MPPI_C_H_
#define MPPI_C_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPPI_ABI_VERSION 5 /* 5: MPPI_TUNE_FUSED_STEP / _ARMED_US / _ARMED_ALWAYS / _PRELAUNCH (no entry point changed); 4: mppi_shard_step (one call per sharded step, the caller's collectives as function pointers), mppi_learner_save /
                              _load / _peek, MPPI_TUNE_TRACE (roctx ranges); mppi_set_mlp orders against the last step's stream;
                              3: the 13-state AUV family (MPPI_MODEL_AUV / _NN_AUV, mppi_auv_desc), StaticQuatCost / ElipseCost3D state costs,
                              mppi_auv_pieces, the learner (mppi_learner_*); 2: state_cost_kind / ellipse, transition log, tuning */
#define MPPI_MAX_S 32  /* largest state dimension  */
#define MPPI_MAX_A 16  /* largest action dimension */

typedef struct mppi_handle mppi_handle;

typedef enum {
    MPPI_OK = 0,
    MPPI_ERR_INVALID_ARG = 1,    /* bad size / NULL (ref: setGoal -> false, controller_base.cpp:126-130) */
    MPPI_ERR_NO_DEVICE = 2,      /* no HIP device / device ordinal out of range */
    MPPI_ERR_HIP = 3,            /* a HIP runtime call failed; see mppi_last_error */
    MPPI_ERR_UNSUPPORTED = 4,    /* shape / option outside what the kernels implement */
    MPPI_ERR_SINGULAR_SIGMA = 5, /* Σ not invertible (ref: MatrixInverse fails, cost_base.cpp:39) */
    MPPI_ERR_ALLOC = 6,
    MPPI_ERR_IO = 7,
    MPPI_ERR_EXCHANGE = 8        /* direct record exchange: a packet missed its deadline (mppi_shard_p2p_step) */
} mppi_status;

enum { MPPI_MODEL_POINT_MASS = 0, /* x' = A x + (B/m) v, src/model_base.cpp:53-82 */
       MPPI_MODEL_MLP = 1,        /* x' = x + denorm(MLP(norm([x;v]))), nn_model.py:215-304 convention, point-mass state (s = 2a) */
       MPPI_MODEL_AUV = 2,        /* Fossen 6-DOF AUVModel, s = 13 (pos, quat xyzw, lin vel, ang vel), a = 6: models/auv_model.py:282-562 */
       MPPI_MODEL_NN_AUV = 3,     /* NNAUVModel: x' = x + denorm(nn(norm(concat(x[3:], u)))), s = 13, a = 6, nn input s+a-3 = 16,
                                     Dense(16|32, relu) x 1..3 + Dense(13): models/nn_model.py:179-304 */
       MPPI_MODEL_NN_AUV_SPEED = 4 }; /* NNAUVModelSpeed: the network predicts the velocity delta from (Euler angles, velocities,
                                     forces) = 15 inputs, Dense(16|32, relu) x 1..3 + Dense(6); the pose is integrated with the
                                     quaternion kinematics over cfg.dt and renormalised: models/nn_model.py:307-588 */

enum { MPPI_ACTION_COST_CPP = 0,  /* λ uᵀΣ⁻¹ε                          src/cost_base.cpp:63-68   */
       MPPI_ACTION_COST_PY = 1 }; /* ½[γ(uᵀΣ⁻¹u+2uᵀΣ⁻¹ε)+λ(1-1/υ)εᵀΣ⁻¹ε]  costs/cost_base.py:114-170 */

enum { MPPI_STATE_COST_QUADRATIC = 0, /* (x-g)ᵀQ(x-g)   src/cost_base.cpp:56-61, costs/static_cost.py:40-63          */
       MPPI_STATE_COST_ELLIPSE = 1,   /* 2D elliptic track, state (x, vx, y, vy, ..): m_state·|((x-cx)/a)²+((y-cy)/b)²-1|
                                         + m_vel·(sqrt(vx²+vy²)-speed)²                 costs/elipse_cost.py:9-85       */
       MPPI_STATE_COST_QUAT = 2,      /* StaticQuatCost (s = 13): d = (pos-g, 2·acos<quat, g_quat>, vel-g) [10], dᵀ Q d with
                                         Q = cfg.quat_Q [10*10], goal = cfg.goal [13]      costs/static_cost.py:73-159  */
       MPPI_STATE_COST_ELLIPSE3D = 3 };/* ElipseCost3D (s = 13): the pose taken into the ellipse's plane frame, then
                                         mS·|Σ(p_i/axis_i)²-1| + mS·angle(pose, tangent) + mV·||v|²-speed²|
                                         with cfg.ellipse3d                                costs/elipse_cost.py:101-246 */

/* mppi_config.flags */
enum { MPPI_FLAG_UPSILON_SCALES_NOISE = 1, /* Py build_noise: eps = (υΣ)·z while the cost keeps Σ⁻¹ of the
                                              un-augmented Σ (controller_base.py:362-368, cost_base.py:35-41) */
       MPPI_FLAG_MLP_BF16X3 = 2, /* learned models (point mass: 256- or 32-wide hidden layers; NNAUVModel: 32-wide): the layers run on the BF16
                                   matrix cores with every fp32 operand split into two bf16 values and three products per
                                   term (fp32 accumulate): ~2x the rounding error of the exact-fp32 path (sample costs
                                   within 1e-6 relative of fp64 on the synthetic network), 1.7-3.3x its speed. Off by
                                   default: the default MLP path is exact fp32 (v_mfma_f32_32x32x2_f32). */
       MPPI_FLAG_FP_CONTRACT = 4 /* point mass, diagonal quadratic cost, Philox noise: the rollout kernel evaluates the model step, the state cost
                                   and the action cost with FUSED multiply-adds (one rounding per pair instead of two) — ~15 % fewer vector
                                   instructions in an issue-bound kernel. Sample costs are then NOT bit-identical to the reference's op-by-op
                                   fp32 evaluation: they agree with an fp64 evaluation to 2e-6 relative (as the unfused evaluation does), the update to 1e-5 (tests). Off by
                                   default; the fused one-launch step and the armed launches are not taken by such a handle. */ };

/* what mppi_debug_get returns (observer_base.py:101-187 logs the same intermediates) */
enum { MPPI_DBG_COSTS = 0,    /* c[K_local]         sample costs of the last step            */
       MPPI_DBG_BETA = 1,     /* [1]                min cost (over all shards after finish)  */
       MPPI_DBG_ETA = 2,      /* [1]                Σ exp(-(c-β)/λ)                          */
       MPPI_DBG_WEIGHTS = 3,  /* w[K_local]         exp(-(c-β)/λ)/η                          */
       MPPI_DBG_NOISE = 4,    /* eps[K_local,tau,a] the noise the last step used             */
       MPPI_DBG_U_UPDATED = 5,/* U'[tau,a]          updated sequence before the shift        */
       MPPI_DBG_AUX = 6       /* [8]                beta, eta, and the words timing-study builds stamp (tools/timeline_step.py) */ };

/* Learned model_base: Dense(relu) x (n_layers-1) + Dense(linear) -> s_dim outputs. */
typedef struct {
    int32_t n_layers;
    const int32_t *widths;   /* [n_layers] output width of every layer; widths[n_layers-1] == s_dim */
    const float *const *W;   /* W[l]: [in x out] row-major (Keras kernel layout)                    */
    const float *const *b;   /* b[l]: [out]                                                         */
    const float *xmean, *xstd; /* [s+a] input normalisation, NULL -> 0 / 1 (nn_model.py:289-293)     */
    const float *ymean, *ystd; /* [s]   output de-normalisation, NULL -> 0 / 1 (nn_model.py:295-297) */
} mppi_mlp_desc;

/* AUVModel's `parameters` (models/auv_model.py:85-245; config/models/rexrov2.default.yaml). Matrices [6*6] row-major; a
 * diagonal given in the reference's 6-vector form is expanded by the caller. NULL matrix = zeros. */
typedef struct {
    float mass, volume, density;
    float gravity;             /* 0 -> 9.81 (auv_model.py:236) */
    float cog[3], cob[3];      /* centre of gravity / buoyancy in the body frame */
    float inertial[6];         /* ixx, iyy, izz, ixy, ixz, iyz */
    const float *added_mass;                    /* Ma [36] */
    const float *linear_damping;                /* [36] */
    const float *linear_damping_forward_speed;  /* [36] */
    const float *quad_damping;                  /* [6]  */
    int32_t rk;                /* 1 Euler, 2 Heun, 4 the reference's rk4 expression (auv_model.py:282-306) */
} mppi_auv_desc;

/* Everything the reference hard-codes in ControllerBase's constructor
 * (controller_base.cpp:23-71) or reads from its config YAML files, as one POD block. */
typedef struct {
    uint32_t struct_size;     /* = sizeof(mppi_config); set by mppi_config_init */
    int32_t k;                /* samples K (GLOBAL count when sharded)          */
    int32_t tau;              /* horizon H                                      */
    int32_t s_dim, a_dim;
    float dt, mass;
    float lambda, gamma, upsilon;
    int32_t action_cost_kind; /* MPPI_ACTION_COST_*                             */
    int32_t normalize_cost;   /* Py normalizeCost, controller_base.py:468-474   */
    const float *sigma;       /* [a*a] Σ multiplies z AND Σ⁻¹ enters the cost; NULL -> I  */
    const float *goal;        /* [s]; NULL -> (1,0) per axis (controller_base.cpp:43-46)  */
    const float *Q;           /* NULL -> ones (controller_base.cpp:58-60)                 */
    int32_t q_is_full;        /* 0: Q is the [s] diagonal (C++ Diag(in_Q)); 1: Q is [s*s] (Py) */
    uint64_t seed;            /* Philox key; default 1 (controller_base.cpp:199)          */
    int32_t model_kind;       /* MPPI_MODEL_*                                             */
    const mppi_mlp_desc *mlp; /* only for MPPI_MODEL_MLP                                  */
    int32_t device;           /* HIP device ordinal                                       */
    int32_t shard_rank;       /* this handle owns samples [rank*k/count, (rank+1)*k/count) */
    int32_t shard_count;      /* 1 = unsharded                                            */
    int32_t flags;            /* MPPI_FLAG_* bits, 0 = the C++ reference's behaviour      */
    int32_t state_cost_kind;  /* MPPI_STATE_COST_*; 0 = the quadratic cost (goal, Q)       */
    const float *ellipse;     /* MPPI_STATE_COST_ELLIPSE: [7] a, b, cx, cy, speed, m_state, m_vel (ElipseCost's
                                 constructor arguments, elipse_cost.py:10-46); needs s_dim >= 4 */
    const mppi_auv_desc *auv; /* MPPI_MODEL_AUV only */
    const float *quat_Q;      /* MPPI_STATE_COST_QUAT: Q [10*10] row-major (static_cost.py:92-100) */
    const float *ellipse3d;   /* MPPI_STATE_COST_ELLIPSE3D: [11] normal[3], aVec[3], axis a, b, speed, mState, mVel
                                 (ElipseCost3D's constructor arguments, elipse_cost.py:102-140; `center` is stored by the
                                 reference and never used) */
} mppi_config;

/* ---- library ------------------------------------------------------------------------- */
int mppi_abi_version(void);
const char *mppi_version(void);
const char *mppi_status_string(mppi_status st);
/* number of visible HIP devices (0 when there is none; never fails) */
int mppi_device_count(void);

/* ---- construction (replaces ControllerBase::ControllerBase, controller_base.cpp:23-71) -- */
/* Fill cfg with the reference constructor's defaults: λ=1, γ=υ=1, Σ=I, Q=1, goal=(1,0)…,
 * seed=1, point-mass model, C++ action cost, device 0, unsharded. */
mppi_status mppi_config_init(mppi_config *cfg, int k, int tau, float dt, float mass, int s_dim, int a_dim);
/* Build the controller: copies every array in cfg, allocates device state (U = 0). */
mppi_status mppi_create(const mppi_config *cfg, mppi_handle **out);
/* replaces ControllerBase::~ControllerBase (controller_base.cpp:124) */
void mppi_destroy(mppi_handle *h);
const char *mppi_last_error(const mppi_handle *h);

/* ---- the host loop's calls -------------------------------------------------------------- */
/* replaces ControllerBase::setGoal (controller_base.cpp:126-133). Takes effect on the next
 * step (the reference bakes the goal into the graph and ignores later calls — a bug we do
 * not reproduce). n must equal s_dim. */
mppi_status mppi_set_goal(mppi_handle *h, const float *goal, int n);
/* Replaces the learned model's weights and normalisation on an existing handle (same layer widths as at creation): what the
 * reference's learner does to the variables the controller's graph reads (learners/learner_base.py:469-496 apply_gradients;
 * models/nn_model.py set_Xmean_Xstd / set_Ymean_Ystd / update_weights). Waits for the handle's stream AND for the stream the last
 * device-resident step was enqueued on (mppi_next_device / mppi_shard_*: a rollout in flight must not read half-replaced weights;
 * ADVICE r03), copies, takes effect with the next step. MPPI_MODEL_MLP / _NN_AUV / _NN_AUV_SPEED handles. */
mppi_status mppi_set_mlp(mppi_handle *h, const mppi_mlp_desc *mlp);
/* replaces ControllerBase::next (controller_base.cpp:135-153): one control step with noise
 * drawn on the device (Philox4x32-10 via rocRAND's engine); returns u = U'[0] in u_out[a],
 * keeps shift(U') as the warm start, logs (x,u) like m_db.addX/addU. */
mppi_status mppi_next(mppi_handle *h, const float *x, int n_x, float *u_out, int n_u);
/* Same step with the noise INJECTED (the reference's tests inject noise,
 * test_controller.cpp:24-35): eps is [K_local, tau, a] for this handle's shard. */
mppi_status mppi_next_with_noise(mppi_handle *h, const float *x, int n_x, const float *eps, size_t n_eps,
                                 float *u_out, int n_u);
/* The transition log (m_db, data_base.cpp:14-71). The reference appends to it on every next(); here it is a
 * preallocated ring of max_rows rows that must be switched on (the C++ / Python ControllerBase mirrors do so in their
 * constructors): 0 = off (default, nothing is recorded and mppi_next allocates nothing). Clears the log. When full
 * the oldest row is overwritten. */
mppi_status mppi_set_transition_log(mppi_handle *h, int max_rows);
/* What mppi_to_csv will NOT write, so that a truncated file never goes unnoticed (the reference's m_db grows without bound and
 * writes every transition, data_base.cpp:52-71): rows_overwritten = transitions lost because the ring was full since the log
 * was sized; rows_without_successor = rows held whose saveNext was skipped (left out of the CSV). Any output may be NULL. */
mppi_status mppi_transition_log_stats(mppi_handle *h, uint64_t *rows_held, uint64_t *rows_overwritten, uint64_t *rows_without_successor);
/* replaces ControllerBase::saveNext (controller_base.cpp:155-163): x_next is the successor of the LAST logged (x, u);
 * a row whose saveNext was skipped is left out of the CSV instead of shifting the later rows. */
mppi_status mppi_save_next(mppi_handle *h, const float *x_next, int n);
/* replaces ControllerBase::toCSV -> DataBase::toCSV (controller_base.cpp:164, data_base.cpp:52-71), byte for byte:
 * header cells and values each followed by ',' (lines end with a comma), values as std::to_string(float) = "%f". */
mppi_status mppi_to_csv(mppi_handle *h, const char *filename);
enum { MPPI_CSV_REFERENCE = 0,  /* the reference's bytes (see mppi_to_csv)                     */
       MPPI_CSV_ROUNDTRIP = 1 };/* "%.9g" values (fp32 round-trips), no trailing commas        */
mppi_status mppi_to_csv_format(mppi_handle *h, const char *filename, int format);

/* ---- options of the Python reference's update ------------------------- */
/* clip_act (controller_base.py:500-504; the call at :453 is commented out in the reference, so off
 * by default): clamp every row of the updated sequence U' = U + Σ w·eps to [a_min[j], a_max[j]]
 * (the model's min_act()/max_act(), model_base.py:121-125) before u = U'[0] is taken.
 * n must equal a_dim; a_min = a_max = NULL switches it off. */
mppi_status mppi_set_action_limits(mppi_handle *h, const float *a_min, const float *a_max, int n);
/* filterSeq (controller_base.py:277-291): scipy.signal.savgol_filter(actionSeq, window, polyorder,
 * deriv=0, axis=0) [mode 'interp'] on the stored (shifted) sequence after every step; the smoothed
 * sequence is the next step's warm start (the reference computes it into an attribute it never
 * reads). window odd, 1 <= window <= tau, 0 <= polyorder < window; window = 0 switches it off
 * (default). The reference's literals (10, 9) have an even window, which 'interp' mode does not
 * define uniquely; they are rejected. */
mppi_status mppi_set_sequence_filter(mppi_handle *h, int window, int polyorder);

/* ---- controller state (deterministic replay; SURVEY §5 checkpoint row) ------------------ */
mppi_status mppi_get_action_sequence(mppi_handle *h, float *U, int n);        /* n = tau*a */
mppi_status mppi_set_action_sequence(mppi_handle *h, const float *U, int n);
mppi_status mppi_get_step_counter(mppi_handle *h, uint64_t *step);            /* RNG offset */
mppi_status mppi_set_step_counter(mppi_handle *h, uint64_t step);
mppi_status mppi_debug_get(mppi_handle *h, int what, float *out, size_t n);

/* ---- the reference's public graph helpers, as plain functions --------------------------- */
/* ModelBase::mBuildFreeStepGraph / mBuildActionStepGraph / mBuildModelStepGraph
 * (model_base.cpp:53-82). x is [kx,s] with kx == k or kx == 1 (broadcast initial state,
 * model_base.hpp:74-75); v is [k,a]. Any output may be NULL. out_free is [kx,s]. */
mppi_status mppi_model_step(mppi_handle *h, const float *x, int kx, const float *v, int k,
                            float *out_free, float *out_action, float *out_next);
/* AUVModel's intermediate quantities, as the reference's tests look at them one by one (scripts/test.py:264-539): for k
 * (state x[13], action u[6]) pairs -> out[k*124] = rotBtoI[9] | TBtoIquat[12] | C(nu)nu[6] | D(nu)nu[6] | g(eta)[6] | state_dot[13] |
 * D(nu)[36] | C(nu)[36]
 * (auv_model.py:353-398 body2inertial_transform, :512-545, :482-510, :450-480, :308-351). MPPI_MODEL_AUV handles only. */
mppi_status mppi_auv_pieces(mppi_handle *h, const float *x, const float *u, int k, float *out);
/* ElipseCost3D.position_error / orientation_error / velocity_error (costs/elipse_cost.py:169-246) of k states x[k*13] ->
 * out[k*3]. in_plane_frame = 1: the pose in x is ALREADY in the ellipse's plane frame (what the reference's unit tests pass to
 * the three methods); 0: the state is taken into the plane frame first, as state_cost does (:124-139).
 * MPPI_STATE_COST_ELLIPSE3D handles only. */
mppi_status mppi_ellipse3d_terms(mppi_handle *h, const float *x, int k, int in_plane_frame, float *out);
/* CostBase::mStateCost / mBuildFinalStepCostGraph (cost_base.cpp:52-61): x[k,s] -> out[k] */
mppi_status mppi_state_cost(mppi_handle *h, const float *x, int k, float *out);
/* CostBase::mActionCost (cost_base.cpp:63-68 or the Py form): u[a], eps[k,a] -> out[k] */
mppi_status mppi_action_cost(mppi_handle *h, const float *u, const float *eps, int k, float *out);
/* CostBase::mBuildStepCostGraph (cost_base.cpp:43-50): state + action cost -> out[k] */
mppi_status mppi_step_cost(mppi_handle *h, const float *x, const float *u, const float *eps, int k, float *out);
/* ControllerBase::mBuildModelGraph (controller_base.cpp:226-273): the full H-step rollout
 * cost from x[s], U[tau,a], eps[K_local,tau,a] -> cost_out[K_local]. Does not touch state. */
mppi_status mppi_rollout_cost(mppi_handle *h, const float *x, const float *U, const float *eps, float *cost_out);
/* ControllerBase::mBeta…mWeightedNoise + mBuildUpdateGraph (controller_base.cpp:166-192,
 * 215-224) on given costs: cost[K_local], eps[K_local,tau,a], U[tau,a]. Outputs (any may be
 * NULL): beta[1], exp_arg[K], exp[K], nabla[1], weights[K], weighted_noise[tau,a], U_new[tau,a]. */
mppi_status mppi_update(mppi_handle *h, const float *cost, const float *eps, const float *U,
                        float *beta, float *exp_arg, float *exp_out, float *nabla, float *weights,
                        float *weighted_noise, float *U_new);
/* ControllerBase::mGetNew / mShift(+mInit0) (controller_base.cpp:310-329), host-side slices */
mppi_status mppi_get_new(const float *U, int tau, int a, int nb, float *out);
mppi_status mppi_shift(const float *U, int tau, int a, const float *init, int nb_init, int nb, float *out);

/* ---- device-resident / asynchronous step (streams, K-sharding across GPUs) -------------- */
/* floats in one shard record: (beta_g, eta_g, V_g[tau*a]) */
int mppi_record_size(const mppi_handle *h);
/* samples this handle owns, and the global index of its first sample */
int mppi_local_samples(const mppi_handle *h);
int mppi_sample_offset(const mppi_handle *h);
/* One whole step, enqueue only: x_dev[s] -> u_dev[a]; U and the step counter advance on the
 * device. Unsharded handles only. The nominal sequence alternates between two device buffers from
 * one call to the next (the shift is a pointer offset), so a captured hipGraph must hold an EVEN
 * number of consecutive steps to be replayable (any number with a sequence filter set: the filter
 * writes back into the first buffer). */
mppi_status mppi_next_device(mppi_handle *h, const float *x_dev, float *u_dev, void *stream);
/* Sharded step, phase 1: rollouts + local soft-min of this shard -> record_dev[record_size]. */
mppi_status mppi_shard_partial(mppi_handle *h, const float *x_dev, float *record_dev, void *stream);
/* Sharded step, phase 2: combine n_records records (all shards, rank order — the result of an
 * all-gather) with r_g = exp(-(beta_g-beta)/λ), apply U' = U + V/η, emit u_dev[a], shift U. */
mppi_status mppi_shard_finish(mppi_handle *h, const float *records_dev, int n_records, float *u_dev, void *stream);
/* normalize_cost on a sharded handle (controller_base.py:468-474: c' = (c - min c)/(max c - min c) over ALL K samples). Phase 1
 * splits in two around a second, 2-float exchange:
 *   mppi_shard_cost_range          rollouts of this shard; range_dev[2] = {-min, max} of ITS sample costs
 *   (the ranks reduce the pairs with ONE all-reduce(MAX): max of the -minima = -(global min), max of the maxima)
 *   mppi_shard_partial_normalized  range_dev[2] = the agreed {-min, max}: this shard's record from the costs normalised with it
 * then mppi_shard_finish as always. Same x_dev in both calls of a step; mppi_shard_partial and the direct exchange refuse such
 * a handle (MPPI_ERR_UNSUPPORTED). The result is the unsharded normalised step's up to the float rounding of the records (2e-6). */
mppi_status mppi_shard_cost_range(mppi_handle *h, const float *x_dev, float *range_dev, void *stream);
mppi_status mppi_shard_partial_normalized(mppi_handle *h, const float *x_dev, const float *range_dev, float *record_dev, void *stream);
/* ONE call per sharded step on the collective path: rollouts of this shard -> its record -> the CALLER's all-gather
 * -> combine, update, shift; enqueue only, x_dev[s] -> u_dev[a]. The library links no collective library: `coll` carries the two
 * functions with EXACTLY the signatures of ncclAllGather / ncclAllReduce (RCCL exports them under these names), so a native host
 * passes them as they are (examples/host_loop_sharded.cpp) and a Python host their addresses (mppi-tf_amd/distributed.py resolves
 * them from the librccl.so its process already holds). The library calls
 *     all_gather(own record (device, record_size floats), records (device, shard_count * record_size floats), record_size,
 *                MPPI_COLL_FLOAT32, comm, stream)          — in place: the own record IS records + shard_rank * record_size
 * and, for a normalize_cost handle only, first
 *     all_reduce(range, range, 2, MPPI_COLL_FLOAT32, MPPI_COLL_MAX, comm, stream)   — {-min, max} of the sample costs, in place
 * on the step's stream, between its own kernels. A non-zero return of either aborts the step with MPPI_ERR_EXCHANGE (nothing of
 * the update has been enqueued then; mppi_last_error carries the code). The send / receive buffers belong to the handle. With
 * shard_count == 1 and coll == NULL the collective is skipped (record -> finish on one device: the sharded sequence of kernels
 * without a communicator). Same result, bit for bit, as mppi_shard_partial -> all-gather -> mppi_shard_finish. */
enum { MPPI_COLL_FLOAT32 = 7 /* = ncclFloat32 */, MPPI_COLL_MAX = 2 /* = ncclMax */ };
typedef struct {
    int (*all_gather)(const void *sendbuff, void *recvbuff, size_t sendcount, int datatype, void *comm, void *stream);
    int (*all_reduce)(const void *sendbuff, void *recvbuff, size_t count, int datatype, int op, void *comm, void *stream); /* normalize_cost only; else may be NULL */
    void *comm; /* ncclComm_t of this rank */
} mppi_collectives;
mppi_status mppi_shard_step(mppi_handle *h, const float *x_dev, float *u_dev, const mppi_collectives *coll, void *stream);
/* Wait for the handle's own stream. */
mppi_status mppi_synchronize(mppi_handle *h);

/* ---- K-sharding, direct exchange: the records travel as peer stores inside the finish kernel --
 * A control step of the analytic model is ~25 us, of the same order as one small collective launched
 * on its own, so the exchange can also be fused into the finish kernel: every rank owns an INBOX
 * (uncached device memory) that all peers map (hipIpc across processes, peer access inside one
 * process); workgroup c of the finish kernel stores this shard's (beta_g, eta_g, V_g[c]) as 8-byte
 * {value, sequence} packets into every rank's inbox (xGMI peer stores), spins until the packets of
 * all ranks have arrived in its own, then combines them in rank order exactly as mppi_shard_finish
 * does — same bits as the all-gather path, no collective library on the step's critical path.
 * Bring-up (all ranks, same order):
 *   mppi_shard_p2p_export   allocate the inbox, get its device pointer and a 64-byte hipIpcMemHandle_t
 *   (exchange the handles out of band, e.g. over the job's bootstrap channel)
 *   mppi_shard_p2p_open     map a peer's inbox into this process (other processes' inboxes only)
 *   mppi_shard_p2p_attach   the shard_count inbox pointers in rank order (own entry = own inbox)
 *   mppi_shard_p2p_probe    one self-test round (known payload, same stores/spins); run it a few
 *                           times on every rank and agree on the outcome before trusting the path
 * then per control step mppi_shard_p2p_step on every rank (same number of calls everywhere: packets
 * carry a call sequence number). Every spin has a deadline (timeout_ms at attach); nothing hangs and nothing is
 * filled with garbage: a finish-kernel workgroup whose packets did not all arrive applies a ZERO update to its column
 * of U (the sequence still shifts, the step counter still advances) and raises a flag; launches already queued behind
 * it skip the exchange (zero updates); the next mppi_shard_p2p_step call returns MPPI_ERR_EXCHANGE and so does every
 * later one (the direct path is closed for the handle). Controls returned since the failing step are zero-update
 * controls, and U / the step counter may then differ between ranks: re-synchronise them (mppi_get/set_action_sequence,
 * mppi_get/set_step_counter from one rank) and continue with mppi_shard_partial / mppi_shard_finish, which is what
 * ShardedController.resync does. */
#define MPPI_MAX_SHARD_PEERS 16
#define MPPI_IPC_HANDLE_BYTES 64
mppi_status mppi_shard_p2p_export(mppi_handle *h, void *ipc_handle_out /* 64 B, may be NULL */, void **inbox_dev_out);
mppi_status mppi_shard_p2p_open(mppi_handle *h, const void *ipc_handle /* 64 B */, void **peer_inbox_out);
mppi_status mppi_shard_p2p_attach(mppi_handle *h, void *const *inboxes, int n, int timeout_ms);
/* ok_out = 1 if the packets of all ranks arrived with the expected payload before the deadline.
 * Synchronises the stream. */
mppi_status mppi_shard_p2p_probe(mppi_handle *h, void *stream, int *ok_out);
/* One whole sharded step, enqueue only: rollouts of this shard, exchange, update; x_dev[s] -> u_dev[a]. */
mppi_status mppi_shard_p2p_step(mppi_handle *h, const float *x_dev, float *u_dev, void *stream);
/* timed_out = 1 if any spin of any step since attach hit its deadline (complete for a step once the stream it ran
 * on has been synchronised). */
mppi_status mppi_shard_p2p_status(mppi_handle *h, int *timed_out);

/* ---- diagnostic switches (A/B timing, fault injection for the fallback tests) ----------------
 * Never needed for correct operation. The library reads NO environment variable: what a process inherits cannot
 * change kernels or inject faults; these calls are the only way. */
enum { MPPI_TUNE_FORCE_TILE_KERNEL = 0, /* 1: the LDS-tile rollout kernel instead of the producer/consumer one          */
       MPPI_TUNE_PC_PRODUCERS = 1,      /* 3 or 5 producer waves per workgroup (default: by tiles per CU)               */
       MPPI_TUNE_PC_BALANCE = 2,        /* 0: no SIMD-true role placement / progress priorities                         */
       MPPI_TUNE_PC_LDS_MIN = 3,        /* pad the rollout kernel's LDS to this many bytes (caps workgroups per CU)     */
       MPPI_TUNE_SYNC_SPIN = 4,         /* 0: mppi_next waits for the stream instead of watching the pinned u slot      */
       MPPI_TUNE_P2P_FAULT = 5,         /* 1: inbox export fails, 2: probe reports failure (exercise the RCCL fallback) */
       MPPI_TUNE_MLP_V1 = 6,            /* 1: exact-fp32 MLP rollouts on the first kernel (8 waves per workgroup) instead of k_rollout_mlp2 */
       MPPI_TUNE_MLP32_VALU = 7,        /* a Dense(32) point-mass network: 1 = k_rollout_mlp_small (vector ALU, scalar-cache weights), 2 = k_rollout_mlp32 (matrix cores, one wave
                                           per 32 rollouts) instead of k_rollout_mlp32_pc (matrix cores, network wave + cost wave per tile);
                                           NNAUVModelSpeed (Dense(16|32)): 1 = k_rollout_gen<2, HID> (vector ALU), 2 = k_rollout_nnspeed32<HID> (matrix cores, one wave per
                                           32 rollouts) instead of k_rollout_nnspeed_pc<HID> (matrix cores, network wave + pose wave per tile); NNAUVModel Dense(32):
                                           1 = k_rollout_gen<1, 32>, 2 = k_rollout_nnauv32 instead of k_rollout_nnauv_pc (network wave + cost wave per tile) */
       /* 1: roctx ranges "mppi:step" > "mppi:rollout" / "mppi:exchange" / "mppi:finish" around what every step enqueues (the reference brackets
        * its step with tf.profiler.experimental.start/stop, controller_base.py:241-248, 587-595). libroctx64.so is dlopen'ed on first use —
        * the library has no link dependency on it; the call fails with MPPI_ERR_UNSUPPORTED when it cannot be found. Shows in
        * `rocprofv3 --marker-trace`. */
       MPPI_TUNE_TRACE = 8,
       MPPI_TUNE_GEN_ONE_WAVE = 9,      /* 1: the Fossen AUVModel on k_rollout_gen<0> (one wave per 64-rollout tile) instead of k_rollout_auv_pc (pose wave + velocity wave per tile) */
       /* r05 (ABI 5), the point-mass producer/consumer path with the diagonal quadratic cost (mppi_step.hip.h): */
       MPPI_TUNE_FUSED_STEP = 10,       /* default 1: a handle of <= 128 tiles (K <= 8192) runs its Philox step as ONE launch — tiles and the column waves that
                                         * finish them in one grid, records handed over as {value, sequence} granules (seven producer waves per tile where H <= 84);
                                         * 0: rollout launch + finish launch; 2: one launch with the six-wave workgroup (five producers) at every horizon */
       MPPI_TUNE_ARMED_US = 11,         /* default 0 (off). N > 0: once two mppi_next calls have followed each other within N microseconds, a call leaves the NEXT
                                         * step's launch behind it, armed: resident on the GPU, its noise drawn, waiting up to N us for x. The next mppi_next then
                                         * only stores x into device memory (large BAR) and watches the pinned u slot — no launch, no dispatch between x and u.
                                         * While armed the launch occupies the GPU; a launch whose x does not come in time aborts by itself and changes nothing.
                                         * Controls are bit-identical to the unarmed path. MPPI_ERR_UNSUPPORTED without a large-BAR device. */
       MPPI_TUNE_ARMED_ALWAYS = 12,     /* 1: arm behind every mppi_next whatever the gap between the last two calls (tests of the deadline path) */
       MPPI_TUNE_PRELAUNCH = 13 };      /* default 0. 1: mppi_next_device on the handle's OWN stream (stream = NULL) runs PRE-LAUNCHED: steps alternate
                                         * between two streams of the handle, the rollout of step n+1 is resident and draws its noise while step n finishes, and takes
                                         * U' of step n from the finish kernel (<= 128 tiles: from the column waves of step n's one launch) as {value, tag} granules. x_dev must be complete when the call is made and stay
                                         * unchanged until the step has run — in particular it cannot wait for u of the step before (a throughput mode, not a closed loop);
                                         * results are bit-identical to the plain path. Any other entry point drains both streams. MPPI_ERR_UNSUPPORTED for a grid of more than
                                         * one round of the chip. */
mppi_status mppi_set_tuning(mppi_handle *h, int what, int value);

/* ---- measurement (the reference only has a commented-out chrono loop, main.cpp:55-64) ------ */
/* Bracket the rollout kernel and the finish kernel of the next <= max_steps steps with HIP
 * events ON THE STREAM THEY ARE LAUNCHED ON. */
mppi_status mppi_profile_begin(mppi_handle *h, int max_steps);
/* Synchronise, stop recording, return average kernel durations in milliseconds over the
 * n_steps recorded steps (any output may be NULL). Both are the dispatches' own begin/end (what rocprofv3 reports). */
mppi_status mppi_profile_end(mppi_handle *h, float *rollout_ms_avg, float *finish_ms_avg, int *n_steps);
/* Name of the rollout kernel instance a fused step of this handle launches, as rocprofv3 prints it
 * (e.g. "mppi::k_rollout_pc<3, 3, 6, true, 0, 0>"): what a roofline figure of this handle refers to. */
mppi_status mppi_rollout_kernel_name(const mppi_handle *h, char *buf, size_t n);

/* ---- the learner of the learned model_base (replaces LearnerBase.train / _train_step, learners/learner_base.py:324-358,
 * 469-496; train_all :146-153) --------------------------------------------------------------------------------------------
 * Full-batch Adam on the mean squared error of the network's prediction against the (normalised) targets, for the
 * reference's Dense networks (nn_model.py:54-60; widths <= 32, 1-4 layers, relu on the hidden layers). The data preparation
 * (prepare_training_data, normalisation statistics, augmentation) is host bookkeeping in the caller; what runs on the
 * device is forward, loss, backward (the weight-gradient GEMMs over the batch on the matrix cores, exact fp32) and the Adam
 * update — a train loop of many steps needs no host round trip. Its weights are what mppi_config.mlp of an
 * MPPI_MODEL_NN_AUV / MPPI_MODEL_MLP controller takes. Deterministic (no float atomics). Not thread-safe per learner. */
typedef struct mppi_learner mppi_learner;
/* widths[n_layers + 1] = inputs, hidden widths..., outputs; W[l] is [widths[l] x widths[l+1]] row-major (Keras kernel), b[l] [widths[l+1]] */
mppi_status mppi_learner_create(int n_layers, const int32_t *widths, const float *const *W, const float *const *b, int device,
                                mppi_learner **out);
void mppi_learner_destroy(mppi_learner *l);
const char *mppi_learner_last_error(const mppi_learner *l);
/* the training set: X [n, inputs], Y [n, outputs] (copied to the device) */
mppi_status mppi_learner_set_data(mppi_learner *l, const float *X, const float *Y, int n);
/* `steps` Adam steps (tf.optimizers.Adam's update: m, v, lr*sqrt(1-b2^t)/(1-b1^t), eps outside the root; Keras defaults
 * beta1 0.9, beta2 0.999, eps 1e-7). loss_first / loss_last (may be NULL): the loss of the first / last step's forward pass. */
mppi_status mppi_learner_train(mppi_learner *l, int steps, float lr, float beta1, float beta2, float eps, float *loss_first, float *loss_last);
/* loss of the CURRENT weights on the training set, no update; grads_out (NULL or [n_layers*33*32]: per layer rows 0..31 = dLoss/dW
 * padded to 32 x 32, row 32 = dLoss/db) ; pred_out (NULL or [n, outputs]) */
mppi_status mppi_learner_evaluate(mppi_learner *l, float *loss_out, float *grads_out, float *pred_out);
mppi_status mppi_learner_get_weights(mppi_learner *l, float *const *W, float *const *b);
mppi_status mppi_learner_set_weights(mppi_learner *l, const float *const *W, const float *const *b);
/* forget the Adam moments and the step count (a new tf.optimizers.Adam, learner_base.py:149) */
mppi_status mppi_learner_reset_optimizer(mppi_learner *l);
mppi_status mppi_learner_get_step(mppi_learner *l, int *step);
/* Persistence (replaces NNModel.save_params / load_params, models/nn_model.py:137-142 — a Keras SavedModel there — and
 * LearnerBase.save_params, learners/learner_base.py:66-68; the reference does not keep its optimizer state, this does, so that a resumed
 * run continues bit for bit). ONE flat little-endian file:
 *     char    magic[8] = "MPPILRN1"
 *     int32   n_layers, widths[5] (n_layers + 1 used, rest 0), adam_step, has_norm
 *     per layer l (in = widths[l], out = widths[l+1]), fp32, compact:  W[in][out] b[out]  mW[in][out] mb[out]  vW[in][out] vb[out]
 *     if has_norm, fp64:  xmean[widths[0]] xstd[widths[0]] ymean[widths[n_layers]] ystd[widths[n_layers]]
 * (the normalisation belongs to the model, not to the learner: the caller passes it in and gets it back; all four NULL = none).
 * mppi_learner_save writes beside the target and renames over it. mppi_learner_load: the learner's layer widths must match (mppi_learner_peek
 * reads them from a file, to create the learner with); weights, both Adam moments and the step count are replaced; *has_norm says whether the
 * file held a normalisation (the four outputs are written only then; they may be NULL). */
mppi_status mppi_learner_save(mppi_learner *l, const char *filename, const double *xmean, const double *xstd, const double *ymean, const double *ystd);
mppi_status mppi_learner_load(mppi_learner *l, const char *filename, double *xmean, double *xstd, double *ymean, double *ystd, int *has_norm);
mppi_status mppi_learner_peek(const char *filename, int *n_layers, int32_t *widths /* [5] */);

#ifdef __cplusplus
}
#endif
#endif /* MPPI_C_H_ */
