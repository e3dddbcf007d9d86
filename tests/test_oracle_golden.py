"""Pins the CPU oracle (oracle/) to the reference's own known-answer vectors.

Fixtures: tests/golden/*.json, transcribed from test/test_*.cpp and scripts/test.py
(see tests/golden/transcribe_reference_vectors.py). Tolerances are the reference's:
EXPECT_FLOAT_EQ = 4 ulp fp32 for the C++ vectors, assertAllClose(rtol=atol=1e-6) for the
fp64 Python vectors. CPU-only.
"""
import numpy as np
import pytest

from conftest import assert_float_eq, load_golden
from oracle import oracle as orc

F32, F64 = np.float32, np.float64


# ---------------------------------------------------------------- A4 blockDiag / model
def test_blockdiag_cpp():
    g = load_golden("blockdiag")
    for case in g["cases"]:
        assert_float_eq(orc.block_diag(g["A"], case["n"]), case["exp_a"], what="A n=%d" % case["n"])
        assert_float_eq(orc.block_diag(g["B"], case["n"]), case["exp_b"], what="B n=%d" % case["n"])
        assert orc.block_diag(g["A"], case["n"]).shape == (2 * case["n"], 2 * case["n"])
        assert orc.block_diag(g["B"], case["n"]).shape == (2 * case["n"], case["n"])


def test_pm_matrices_match_blockdiag_fixture():
    g = load_golden("blockdiag")
    for case in g["cases"]:
        n = case["n"]
        A, B = orc.pm_matrices(0.01, 1.5, 2 * n, n)
        assert_float_eq(A, case["exp_a"], what="A")
        assert_float_eq(B, case["exp_b"], what="B")


@pytest.mark.parametrize("idx", range(4))
def test_model_step_cpp(idx):
    sc = load_golden("model_cpp")["scenarios"][idx]
    A, B = orc.pm_matrices(sc["dt"], sc["mass"], sc["s"], sc["a"])
    free = orc.model_free_step(A, sc["state"])
    act = orc.model_action_step(B, sc["action"])
    res = orc.model_step(A, B, sc["state"], sc["action"])
    assert free.shape == np.asarray(sc["exp_free"]).shape  # [1,s] for the broadcast-init case
    assert_float_eq(free, sc["exp_free"], what=sc["name"] + " free")
    assert_float_eq(act, sc["exp_action"], what=sc["name"] + " action")
    assert_float_eq(res, sc["exp_result"], what=sc["name"] + " result")


@pytest.mark.parametrize("idx", range(4))
def test_model_step_py_fp64(idx):
    sc = load_golden("model_py")["scenarios"][idx]
    A, B = orc.pm_matrices(sc["dt"], sc["mass"], sc["s"], sc["a"], F64)
    np.testing.assert_allclose(orc.model_free_step(A, sc["state"], F64), sc["exp_free"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(orc.model_action_step(B, sc["action"], F64), sc["exp_action"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(orc.model_step(A, B, sc["state"], sc["action"], F64), sc["exp_result"],
                               rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("dtype", [F32, F64])
def test_model_three_step_recurrence(dtype):
    sc = load_golden("model_py")["step3"]
    A, B = orc.pm_matrices(sc["dt"], sc["mass"], sc["s"], sc["a"], dtype)
    x = np.asarray(sc["state"])
    for _ in range(sc["n_steps"]):
        x = orc.model_step(A, B, x, sc["action"], dtype)
    np.testing.assert_allclose(x, sc["exp_result"], rtol=1e-6, atol=1e-6)


# ---------------------------------------------------------------- A5-A7 cost (C++ form)
@pytest.mark.parametrize("idx", range(3))
def test_cost_cpp(idx):
    sc = load_golden("cost_cpp")["scenarios"][idx]
    p = orc.Problem(tau=1, s=sc["s"], a=sc["a"], lam=sc["lam"], sigma=sc["sigma"], goal=sc["goal"],
                    Q=sc["q_diag"], mlp={"W": [np.zeros((sc["s"] + sc["a"], sc["s"]))], "b": [np.zeros(sc["s"])]})
    assert_float_eq(p.state_cost(sc["state"]), sc["exp_state"], what=sc["name"] + " state")
    assert_float_eq(p.step_cost(sc["state"], sc["action"], sc["eps"]), sc["exp_step"], what=sc["name"] + " step")


# ---------------------------------------------------------------- A5-A7 cost (Python γ/υ form)
def _py_problem(sc, dtype, s=None):
    s = s or sc.get("s", 2 * sc["a"])
    return orc.Problem(tau=1, s=s, a=sc["a"], lam=sc["lam"], gamma=sc["gamma"], upsilon=sc["upsilon"],
                       sigma=sc["sigma"], goal=sc.get("goal", np.zeros(s)), Q=sc.get("Q", np.eye(s)),
                       action_cost=orc.ACTION_COST_PY, dtype=dtype,
                       mlp={"W": [np.zeros((s + sc["a"], s))], "b": [np.zeros(s)]})


@pytest.mark.parametrize("idx", range(5))
@pytest.mark.parametrize("dtype", [F32, F64])
def test_action_cost_py(idx, dtype):
    sc = load_golden("cost_py")["action_cost"][idx]
    p = _py_problem(sc, dtype)
    tol = 1e-6 if dtype is F64 else 2e-6
    np.testing.assert_allclose(p.action_cost(sc["action"], sc["noise"]), sc["exp_action"], rtol=tol, atol=tol)


@pytest.mark.parametrize("idx", range(4))
@pytest.mark.parametrize("dtype", [F32, F64])
def test_static_cost_py(idx, dtype):
    sc = load_golden("cost_py")["static_cost"][idx]
    p = _py_problem(sc, dtype)
    tol = 1e-6 if dtype is F64 else 2e-6
    np.testing.assert_allclose(p.action_cost(sc["action"], sc["noise"]), sc["exp_action"], rtol=tol, atol=tol)
    np.testing.assert_allclose(p.state_cost(sc["state"]), sc["exp_state"], rtol=tol, atol=tol)
    np.testing.assert_allclose(p.step_cost(sc["state"], sc["action"], sc["noise"]), sc["exp_step"], rtol=tol, atol=tol)


@pytest.mark.parametrize("idx", range(2))
@pytest.mark.parametrize("dtype", [F32, F64])
def test_elipse_cost_py(idx, dtype):
    """ElipseCost.state_cost (costs/elipse_cost.py:48-85) against TestElipseCost's literals (scripts/test.py:1098-1161)."""
    g = load_golden("cost_elipse")
    sc = g["scenarios"][idx]
    p = orc.Problem(tau=1, s=4, a=2, ellipse=g["ellipse"], dtype=dtype)
    tol = 1e-6 if dtype is F64 else 2e-6
    np.testing.assert_allclose(p.state_cost(sc["state"]), sc["exp_state_cost"], rtol=tol, atol=tol)


# ---------------------------------------------------------------- A3 data prep
def test_dataprep_slicing_is_the_khta_layout():
    """mPrepareAction/mPrepareNoise (controller_base.cpp:205-213) are U[t] and eps[:,t]; the oracle
    and the HIP path index the [K,H,a] / [H,a] row-major layout directly."""
    g = load_golden("controller_dataprep")
    n, a = np.asarray(g["noise"]), np.asarray(g["action"])
    for t in range(3):
        np.testing.assert_array_equal(a[t], g["a"][t])
        np.testing.assert_array_equal(n[:, t], g["n"][t])


# ---------------------------------------------------------------- A8-A9 update chain
@pytest.mark.parametrize("acc_double", [False, True])
def test_update_chain_cpp(acc_double):
    g = load_golden("controller_update_k5_tau3_a2")
    r = orc.update(g["cost"], g["noise"], g["action"], g["lam"], acc_double=acc_double)
    assert_float_eq(r["beta"], g["beta"], what="beta")
    assert_float_eq(r["arg"], g["exp_arg"], what="exp_arg")
    assert_float_eq(r["exp"], g["exp"], what="exp")
    assert_float_eq(r["nabla"], g["nabla"], what="nabla")
    assert_float_eq(r["w"], g["weights"], what="weights")
    assert_float_eq(r["wn"], g["weighted_noise"], what="weighted noise")
    assert_float_eq(np.sum(r["w"], dtype=np.float32), g["sum_weights"], what="sum w")
    assert_float_eq(r["Unew"], np.asarray(g["action"], F32) + np.asarray(g["weighted_noise"], F32), what="U'")


def test_update_chain_py_fp64():
    g = load_golden("controller_update_k5_tau3_a2")
    r = orc.update(g["cost"], g["noise"], g["action"], g["lam"], dtype=F64)
    for k_o, k_g in (("beta", "beta"), ("arg", "exp_arg"), ("exp", "exp"), ("nabla", "nabla"),
                     ("w", "weights"), ("wn", "weighted_noise")):
        np.testing.assert_allclose(r[k_o], g[k_g], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(r["w"].sum(), 1.0, rtol=1e-6)


# ---------------------------------------------------------------- A10 next / shift
def test_getnew_and_shift():
    g = load_golden("controller_getnew_shift")
    for nb, exp in g["getnew"].items():
        out = orc.get_new(g["action"], int(nb))
        assert out.shape == (int(nb), 2)
        assert_float_eq(out, np.asarray(exp, F32).reshape(int(nb), 2), ulps=0)
    for case in g["shift"]:
        out = orc.shift(g["action"], case["init"], case["nb"])
        assert out.shape == (3, 2)
        assert_float_eq(out, case["expected"], ulps=0)


# ---------------------------------------------------------------- self-consistency of the unpinned rows
def test_rollout_matches_stepwise_composition():
    """A7 end-to-end has no reference test: check the fused recurrence against the composition of the
    separately pinned step functions (model step, step cost, final cost) on random inputs."""
    rng = np.random.default_rng(0)
    K, H, s, a = 17, 9, 6, 3
    p = orc.Problem(tau=H, s=s, a=a, dt=0.1, mass=1.5, lam=0.7, sigma=0.25 * np.eye(a),
                    goal=[1, 0, .5, 0, .75, 0], Q=[1, 2, 3, 4, 5, 6])
    x0, U = rng.standard_normal(s), rng.standard_normal((H, a))
    eps = rng.standard_normal((K, H, a)).astype(F32)
    x = np.tile(np.asarray(x0, F32), (K, 1))
    c = np.zeros(K, F32)
    for t in range(H):
        v = (np.asarray(U[t], F32)[None, :] + eps[:, t]).astype(F32)
        x = orc.model_step(p.A, p.B, x, v)
        c = (c + p.step_cost(x, U[t], eps[:, t])).astype(F32)
    c = (c + p.state_cost(x)).astype(F32)
    got, traj = p.rollout_cost(x0, U, eps, traj=True)
    np.testing.assert_array_equal(got, c)
    np.testing.assert_array_equal(traj[:, -1], x)


def test_next_with_noise_is_update_get_shift():
    rng = np.random.default_rng(1)
    K, H, s, a = 64, 8, 4, 2
    p = orc.Problem(tau=H, s=s, a=a)
    x0, U = rng.standard_normal(s), rng.standard_normal((H, a)).astype(F32)
    eps = rng.standard_normal((K, H, a)).astype(F32)
    u, Unext, c = p.next_with_noise(x0, U, eps)
    r = orc.update(p.rollout_cost(x0, U, eps), eps, U, p.lam)
    np.testing.assert_array_equal(c, p.rollout_cost(x0, U, eps))
    np.testing.assert_array_equal(u, r["Unew"][0])
    np.testing.assert_array_equal(Unext[:-1], r["Unew"][1:])
    np.testing.assert_array_equal(Unext[-1], 0)


def test_combine_records_equals_global_update():
    """§8e exchange: per-shard (beta_g, eta_g, V_g) records recombine to the unsharded update."""
    rng = np.random.default_rng(2)
    K, H, a, lam, G = 96, 5, 3, 0.8, 4
    cost = (rng.standard_normal(K) * 3 + 10).astype(F32)
    eps = rng.standard_normal((K, H, a)).astype(F32)
    U = rng.standard_normal((H, a)).astype(F32)
    full = orc.update(cost, eps, U, lam)["Unew"]
    recs = []
    for g in range(G):
        sl = slice(g * K // G, (g + 1) * K // G)
        c64, e64 = cost[sl].astype(F64), eps[sl].astype(F64)
        b = c64.min()
        e = np.exp(-(c64 - b) / lam)
        recs.append(np.concatenate([[b, e.sum()], (e[:, None, None] * e64).sum(0).ravel()]))
    got = orc.combine_records(np.asarray(recs, F32), U, lam)
    np.testing.assert_allclose(got, full, rtol=0, atol=2e-6)


# ---------------------------------------------------------------- noise stream (A2, unpinned by the reference)
def test_philox_known_answer_vectors():
    """Random123's published known-answer tests for philox4x32-10 (kat_vectors)."""
    assert orc.philox4x32_10([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert orc.philox4x32_10([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert orc.philox4x32_10([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_noise_is_shard_invariant_and_standard_normal():
    z_all = orc.normals(seed=1, step=3, k_offset=0, k=4096, tau=16, a=3)
    z_lo = orc.normals(seed=1, step=3, k_offset=0, k=2048, tau=16, a=3)
    z_hi = orc.normals(seed=1, step=3, k_offset=2048, k=2048, tau=16, a=3)
    np.testing.assert_array_equal(z_all, np.concatenate([z_lo, z_hi]))
    assert abs(z_all.mean()) < 0.01 and abs(z_all.std() - 1) < 0.01
    assert not np.array_equal(z_all, orc.normals(1, 4, 0, 4096, 16, 3))
    sig = np.array([[0.5, 0.1, 0], [0, 0.25, 0], [0.2, 0, 1.0]], F32)
    e = orc.noise(1, 3, 0, 4096, 16, 3, sig)
    np.testing.assert_allclose(e, z_all @ sig.T, rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("hid,n_hidden", [(32, 3), (16, 3), (256, 2)], ids=["32x3", "16x3", "256x2"])
def test_mlp_step_restatement_against_numpy(hid, n_hidden):
    """orc_mlp_step (the learned-step convention of nn_model.py:215-239,289-304 with the reference's layer shapes,
    nn_model.py:54-60: Dense(32, relu) x3 + Dense(s)) against a plain numpy fp64 evaluation of the same formulas."""
    s, a = 6, 3
    rng = np.random.default_rng(hid + n_hidden)
    dims = [s + a] + [hid] * n_hidden + [s]
    W = [rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(n_hidden + 1)]
    b = [rng.uniform(-1, 1, dims[i + 1]) for i in range(n_hidden + 1)]
    mlp = dict(W=W, b=b, xmean=rng.uniform(-.1, .1, s + a), xstd=rng.uniform(.8, 1.2, s + a),
               ymean=rng.uniform(-.01, .01, s), ystd=rng.uniform(.8, 1.2, s))
    p = orc.Problem(tau=4, s=s, a=a, sigma=np.eye(a), goal=[1, 0, 1, 0, 1, 0], mlp=mlp, dtype=np.float64)
    for _ in range(20):
        x, v = rng.standard_normal(s), rng.standard_normal(a)
        h = (np.concatenate([x, v]) - mlp["xmean"]) / mlp["xstd"]
        for l in range(n_hidden + 1):
            h = h @ W[l] + b[l]
            if l < n_hidden:
                h = np.maximum(h, 0)
        np.testing.assert_allclose(p.mlp_step(x, v), x + (h * mlp["ystd"] + mlp["ymean"]), rtol=1e-12, atol=1e-13)


# ======================================================================= SURVEY §8f-4: AUV family, quaternion / 3D-ellipse costs
# The oracle's restatement of AUVModel / NNAUVModel / StaticQuatCost / ElipseCost3D against the reference's own literals
# (scripts/test.py TestAUVModel :237-586, TestNNAUVModel :587-684, TestElipse3DCost :1164-1360), fp64 like the Python reference,
# at tf.test's assertAllClose tolerance (rtol = atol = 1e-6).
CLOSE = dict(rtol=1e-6, atol=1e-6)


@pytest.fixture(scope="module")
def auv_golden():
    return load_golden("model_auv")


def test_auv_body_to_inertial_rotation(auv_golden):
    mdl = orc.AuvModel(auv_golden["params"], dtype=np.float64)
    for q, R in zip(auv_golden["b2i"]["quat"], auv_golden["b2i"]["rot_from_lib"]):
        rot, T = mdl.b2i(q)
        np.testing.assert_allclose(rot, R, **CLOSE)
        x, y, z, w = q  # auv_model.py:388-396: rows rxt, ryt, rzt, rwt, times 0.5
        np.testing.assert_allclose(T, 0.5 * np.array([[w, -z, y], [z, w, -x], [-y, x, w], [-x, -y, -z]]), rtol=0, atol=0)
        # the Jacobian maps body rates to quaternion rates: q . qdot = 0 for any angular velocity
        assert abs(np.dot(np.array(q)[[0, 1, 2]], T[:3] @ [0.3, -0.2, 0.5]) + q[3] * (T[3] @ [0.3, -0.2, 0.5])) < 1e-15


def test_auv_restoring_forces(auv_golden):
    mdl = orc.AuvModel(auv_golden["params"], dtype=np.float64)
    g = auv_golden["restoring"]
    for q, R, exp in zip(g["quat"], g["exp_rot"], g["exp_restoring"]):
        np.testing.assert_allclose(mdl.b2i(q)[0], R, **CLOSE)
        np.testing.assert_allclose(mdl.restoring(q), exp, **CLOSE)


def test_auv_damping_and_coriolis(auv_golden):
    mdl = orc.AuvModel(auv_golden["params"], dtype=np.float64)
    for v, D in zip(auv_golden["damping"]["vel"], auv_golden["damping"]["exp"]):
        np.testing.assert_allclose(mdl.damping(v), D, **CLOSE)
    np.testing.assert_allclose(mdl.coriolis(auv_golden["coriolis"]["vel"]), auv_golden["coriolis"]["exp"], **CLOSE)


def test_auv_step_properties(auv_golden):
    """The reference's step tests only print (scripts/test.py:541-586): the step is pinned as the composition of the pinned
    pieces. Properties: unit quaternion out; at rest with neutral buoyancy and no force nothing moves; rk1 = x + dt f(x);
    fp32 instantiation within fp32 of fp64."""
    P = dict(auv_golden["params"])
    m64, m32 = orc.AuvModel(P, dtype=np.float64), orc.AuvModel(P, dtype=np.float32)
    dt = float(np.float32(0.1))  # the model parameters are fp32 numbers in both instantiations (what the C-ABI carries)
    for x, u in zip(auv_golden["step_inputs"]["state"], auv_golden["step_inputs"]["action"]):
        xn = m64.step(x, u)
        assert abs(np.linalg.norm(xn[3:7]) - 1) < 1e-12
        k1 = m64.state_dot(x, u)
        xs = np.asarray(x) + dt * k1
        heun = np.asarray(x) + (dt / 2) * (k1 + m64.state_dot(xs, u))
        heun[3:7] /= np.linalg.norm(heun[3:7])
        np.testing.assert_allclose(xn, heun, rtol=1e-12, atol=1e-12)
        np.testing.assert_allclose(m32.step(x, u), xn, rtol=2e-5, atol=2e-5)
    neutral = dict(P, volume=1.0, cob=[0., 0., 0.], rk=1)  # W = B, no righting moment
    x0 = [1., 2., 3., 0., 0., 0., 1.] + [0.] * 6
    np.testing.assert_allclose(orc.AuvModel(neutral).step(x0, [0.] * 6), x0, rtol=0, atol=1e-15)
    e1 = orc.AuvModel(dict(P, rk=1))
    x, u = auv_golden["step_inputs"]["state"][1], auv_golden["step_inputs"]["action"][1]
    eul = np.asarray(x) + dt * e1.state_dot(x, u)
    eul[3:7] /= np.linalg.norm(eul[3:7])
    np.testing.assert_allclose(e1.step(x, u), eul, rtol=1e-12, atol=1e-12)


def test_nnauv_data_preparation():
    g = load_golden("model_nnauv")
    t = g["training_n1"]
    X, Y = orc.nnauv_prepare_training_data(t["state_t"], t["state_t1"], t["action"])
    np.testing.assert_allclose(X, t["exp_x"], **CLOSE)
    np.testing.assert_allclose(Y, t["exp_y"], **CLOSE)
    for key in ("prepare_n1", "prepare_n6"):
        np.testing.assert_allclose(orc.nnauv_prepare_data(g[key]["state"], g[key]["action"]), g[key]["exp"], **CLOSE)


def test_nnauv_step_is_state_plus_denormalised_network_output():
    """NNAUVModel.build_step_graph (nn_model.py:215-239): x' = x + nn(norm(concat(x[3:], u)))*Ystd + Ymean, against numpy."""
    rng = np.random.default_rng(0)
    dims = [16, 32, 32, 32, 13]
    mlp = dict(W=[rng.standard_normal((dims[i], dims[i + 1])) / 4 for i in range(4)], b=[rng.standard_normal(dims[i + 1]) / 10 for i in range(4)],
               xmean=rng.standard_normal(16) / 10, xstd=1 + rng.random(16), ymean=rng.standard_normal(13) / 10, ystd=1 + rng.random(13))
    p = orc.Problem(tau=4, s=13, a=6, sigma=np.eye(6), goal=np.zeros(13), nnauv=mlp, dtype=np.float64)
    x, u = rng.standard_normal((5, 13)), rng.standard_normal((5, 6))
    h = (orc.nnauv_prepare_data(x, u, mlp["xmean"], mlp["xstd"]))
    for l in range(4):
        h = h @ mlp["W"][l] + mlp["b"][l]
        if l < 3:
            h = np.maximum(h, 0)
    np.testing.assert_allclose(p.model_next(x, u), x + h * mlp["ystd"] + mlp["ymean"], rtol=1e-12, atol=1e-12)


def test_euler_from_quaternion_against_scipy():
    """NNAUVModelSpeed.to_euler calls tensorflow_graphics' euler.from_quaternion (third party, absent; the reference holds no test of this
    model): the restatement is pinned by an independent implementation — scipy's Rotation.as_euler('xyz') (extrinsic x-y-z = R = Rz Ry Rx) —
    on random attitudes, and by its own gimbal-lock branch at pitch = +-90 degrees."""
    from scipy.spatial.transform import Rotation
    rng = np.random.default_rng(0)
    q = rng.standard_normal((4000, 4))
    q /= np.linalg.norm(q, axis=1, keepdims=True)
    e = orc.euler_from_quaternion(q)
    ref = Rotation.from_quat(q).as_euler("xyz")
    assert np.abs(np.angle(np.exp(1j * (e - ref)))).max() < 1e-11
    e32 = orc.euler_from_quaternion(q, np.float32).astype(np.float64)
    far = np.abs(np.abs(e[:, 1]) - np.pi / 2) > 0.2  # away from the lock the fp32 evaluation (of the fp32-rounded attitude) is well conditioned
    assert np.abs(np.angle(np.exp(1j * (e32 - e))))[far].max() < 1e-5
    for sign in (1.0, -1.0):  # pitch = +-90 degrees: theta_z = 0, theta_y = +-pi/2, the attitude is reproduced
        ql = Rotation.from_euler("xyz", [0.3, sign * np.pi / 2, 0.0]).as_quat()[None]
        el = orc.euler_from_quaternion(ql)[0]
        assert abs(el[1] - sign * np.pi / 2) < 1e-12 and el[2] == 0.0
        assert np.abs(Rotation.from_euler("xyz", el).as_matrix() - Rotation.from_quat(ql[0]).as_matrix()).max() < 1e-6


def test_nnauv_speed_step_against_numpy():
    """NNAUVModelSpeed.build_step_graph (nn_model.py:358-380, 438-472): velocity delta from the network on (Euler angles, velocities,
    forces); pose integrated with J(x) over dt with THIS class's T rows (:545-555), quaternion renormalised. Against numpy."""
    rng = np.random.default_rng(1)
    dims = [15, 16, 16, 16, 6]
    mlp = dict(W=[rng.standard_normal((dims[i], dims[i + 1])) / 4 for i in range(4)], b=[rng.standard_normal(dims[i + 1]) / 10 for i in range(4)],
               xmean=rng.standard_normal(15) / 10, xstd=1 + rng.random(15), ymean=rng.standard_normal(6) / 10, ystd=1 + rng.random(6))
    dt = 0.1
    p = orc.Problem(tau=4, s=13, a=6, dt=dt, sigma=np.eye(6), goal=np.zeros(13), nnauv_speed=mlp, dtype=np.float64)
    x, u = rng.standard_normal((6, 13)), rng.standard_normal((6, 6))
    x[:, 3:7] /= np.linalg.norm(x[:, 3:7], axis=1, keepdims=True)
    h = orc.nnauv_speed_prepare_data(x, u, mlp["xmean"], mlp["xstd"])
    assert h.shape == (6, 15)
    for l in range(4):
        h = h @ mlp["W"][l] + mlp["b"][l]
        if l < 3:
            h = np.maximum(h, 0)
    delta = h * mlp["ystd"] + mlp["ymean"]
    exp = x.copy()
    for i in range(6):
        qx, qy, qz, qw = x[i, 3:7]
        rot = np.array([[1 - 2 * (qy * qy + qz * qz), 2 * (qx * qy - qz * qw), 2 * (qx * qz + qy * qw)],
                        [2 * (qx * qy + qz * qw), 1 - 2 * (qx * qx + qz * qz), 2 * (qy * qz - qx * qw)],
                        [2 * (qx * qz - qy * qw), 2 * (qy * qz + qx * qw), 1 - 2 * (qx * qx + qy * qy)]])
        T = 0.5 * np.array([[-qx, -qy, -qz], [qw, -qz, qy], [qz, qw, -qx], [-qy, qx, qw]])
        pose = x[i, :7] + np.concatenate([rot @ x[i, 7:10], T @ x[i, 10:13]]) * dt
        pose[3:7] /= np.linalg.norm(pose[3:7])
        exp[i, :7], exp[i, 7:] = pose, x[i, 7:] + delta[i]
    got = p.model_next(x, u)
    np.testing.assert_allclose(got, exp, rtol=1e-11, atol=1e-11)
    assert np.abs(np.linalg.norm(got[:, 3:7], axis=1) - 1).max() < 1e-12
    X, Y = orc.nnauv_speed_prepare_training_data(x, got, u)
    np.testing.assert_allclose(Y, delta, rtol=1e-10, atol=1e-10)  # the training target IS the velocity delta


def test_elipse3d_cost_pieces():
    g = load_golden("cost_elipse3d")
    b = g["base"]
    for pc in g["prep_const"]:
        e = orc.Ellipse3D(pc["normal"], pc["aVec"], b["axis"], b["speed"], b["m_state"], b["m_vel"])
        np.testing.assert_allclose(e.R, pc["exp_R"], **CLOSE)
    pl = g["plane"]
    e = orc.Ellipse3D(pl["normal"], pl["aVec"], b["axis"], b["speed"], b["m_state"], b["m_vel"])
    np.testing.assert_allclose([e.position_error(p) for p in g["position_error"]["position"]], g["position_error"]["exp"], **CLOSE)
    np.testing.assert_allclose([e.orientation_error(p) for p in g["orientation_error"]["pose"]], g["orientation_error"]["exp"], **CLOSE)
    np.testing.assert_allclose([e.velocity_error(v) for v in g["velocity_error"]["velocity"]], g["velocity_error"]["exp"], **CLOSE)
    t = g["tf_rot"]
    np.testing.assert_allclose(orc.quat_rotate(t["position"], t["q"]), t["exp_pos"], **CLOSE)
    np.testing.assert_allclose(orc.quat_multiply(t["q"], t["quat"]), t["exp_quat"], **CLOSE)
    # state_cost = mS*position + mS*orientation + mV*velocity of the pose taken into the plane frame (no reference expectation)
    for x in g["state_cost_inputs"]:
        pose = np.concatenate([orc.quat_rotate(x[:3], e.q), orc.quat_multiply(e.q, x[3:7])])
        want = b["m_state"] * e.position_error(pose[:3]) + b["m_state"] * e.orientation_error(pose) + b["m_vel"] * e.velocity_error(x[7:])
        assert abs(e.state_cost(x) - want) < 1e-12
    # for an ORTHONORMAL plane basis (the reference's test plane has a normal of length sqrt(2): its R is no rotation and
    # its q no unit quaternion — restated as is) the plane quaternion is a unit quaternion reproducing R
    e = orc.Ellipse3D([0., np.sin(0.4), np.cos(0.4)], [1., 0., 0.], b["axis"], 1., 1., 1.)
    assert abs(np.linalg.norm(e.q) - 1) < 1e-12
    for v in np.eye(3):
        np.testing.assert_allclose(orc.quat_rotate(v, e.q), e.R @ v, rtol=0, atol=1e-12)


def test_static_quat_cost_dist_and_value():
    """StaticQuatCost (static_cost.py:73-159; no reference test): dist = (dp, 2 acos<q, g_q>, dv) [10], cost = d^T Q d."""
    goal = [1., 2., -10., 0., 0., 0., 1.] + [0.] * 6
    s = np.sin(0.3), np.cos(0.3)
    x = [0., 2.5, -9., 0., 0., s[0], s[1], 1., 0., -1., 0.1, 0.2, 0.3]
    d = orc.quat_dist(x, goal)
    np.testing.assert_allclose(d, [-1., .5, 1., 0.6, 1., 0., -1., .1, .2, .3], rtol=1e-12, atol=1e-12)
    Q = np.diag([1000.] * 3 + [100.] + [1.] * 6)
    p = orc.Problem(tau=2, s=13, a=6, sigma=np.eye(6), goal=goal, Q=Q, quat_cost=True, dtype=np.float64)
    np.testing.assert_allclose(p.state_cost([x]), [d @ Q @ d], rtol=1e-12)
    p32 = orc.Problem(tau=2, s=13, a=6, sigma=np.eye(6), goal=goal, Q=np.diag(Q), quat_cost=True)
    np.testing.assert_allclose(p32.state_cost([x]), [d @ Q @ d], rtol=2e-6)


def test_auv_rollout_through_the_oracle_problem(auv_golden):
    """The AUV model and the quaternion cost inside the full path (rollout -> update -> shift): finite, deterministic, and the
    fp32 instantiation tracks fp64."""
    H, K = 6, 64
    goal = [1., 2., -3., 0., 0., 0., 1.] + [0.] * 6
    kw = dict(tau=H, s=13, a=6, lam=1.0, sigma=200. * np.eye(6), goal=goal, Q=[100.] * 3 + [10.] * 4 + [1.] * 6, auv=auv_golden["params"])
    p64, p32 = orc.Problem(dtype=np.float64, **kw), orc.Problem(**kw)
    rng = np.random.default_rng(1)
    eps = (200. * rng.standard_normal((K, H, 6))).astype(np.float32)
    x0 = np.array([0., 0., 0., 0., 0., 0., 1.] + [0.] * 6, np.float32)
    U = np.zeros((H, 6), np.float32)
    u64, U64, c64 = p64.next_with_noise(x0, U, eps)
    u32, U32, c32 = p32.next_with_noise(x0, U, eps)
    assert np.isfinite(c64).all() and (np.abs(c32 - c64) / np.abs(c64)).max() < 1e-4
    assert np.abs(np.asarray(U32, np.float64) - U64).max() < 1e-2 * np.abs(U64).max() + 1e-3
