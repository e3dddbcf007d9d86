"""SURVEY §8f row 4 on the GPU: the 13-state AUV family in the model_base slot (AUVModel, NNAUVModel) and StaticQuatCost /
ElipseCost3D in the cost_base slot, through the C-ABI, against the CPU oracle and the reference's own literals
(scripts/test.py TestAUVModel :237-586, TestNNAUVModel :587-684, TestElipse3DCost :1164-1360).

Bars: AUVModel pieces, steps and rollout costs BIT-IDENTICAL to the fp32 oracle (same operation order on both sides, both
compiled without FMA contraction; the host constants come from the same double Gauss-Jordan inverse); the quaternion /
3D-ellipse costs within 2e-6 relative (device acosf vs libm); learned-model rollouts as the other MLP kernels (costs within
4x the fp32 CPU's own error against fp64). Control updates: |dU'| <= 1e-5 x the noise scale (north_star's 1e-5 for unit noise).
"""
import os

import numpy as np
import pytest

from conftest import load_golden
from oracle import oracle as orc

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
F32 = np.float32
CLOSE = dict(rtol=1e-6, atol=1e-6)


@pytest.fixture(scope="module")
def m():
    import __graft_entry__ as g
    g.build()
    import mppi_tf_amd as mod
    return mod


@pytest.fixture(scope="module")
def G():
    return load_golden("model_auv")


def rand_states(k, seed=0, vel=1.0):
    rng = np.random.default_rng(seed)
    x = rng.standard_normal((k, 13))
    x[:, 3:7] /= np.linalg.norm(x[:, 3:7], axis=1, keepdims=True)
    x[:, 7:] *= vel
    return x.astype(F32), (200.0 * rng.standard_normal((k, 6))).astype(F32)


REXROV = None


def rexrov(m):
    from mppi_tf_amd.auv import auv_task
    return auv_task(8)["auv"]


# ------------------------------------------------------------------------------------------------ AUVModel
def test_golden_auv_rotation_restoring_damping_coriolis_on_device(m, G):
    """The reference's TestAUVModel through the device code, with the reference's method names."""
    mdl = m.AUVModel(actionDim=6, dt=0.1, parameters=G["params"])
    k = 3
    mdl.set_k(k)
    quat = np.zeros((k, 13, 1))
    quat[:, 3:7, 0] = G["b2i"]["quat"]
    pose, _ = mdl.prepare_data(quat)
    mdl.body2inertial_transform(pose)
    np.testing.assert_allclose(mdl._rotBtoI, G["b2i"]["rot_from_lib"], **CLOSE)
    for i, (x, y, z, w) in enumerate(G["b2i"]["quat"]):
        np.testing.assert_allclose(mdl._TBtoIquat[i], 0.5 * np.array([[w, -z, y], [z, w, -x], [-y, x, w], [-x, -y, -z]]), rtol=1e-7, atol=0)
    jac = mdl.get_jacobian()
    assert jac.shape == (k, 7, 6) and not jac[:, :3, 3:].any() and not jac[:, 3:, :3].any()
    # test_restoring
    r = G["restoring"]
    pose = np.zeros((2, 7, 1))
    pose[:, 3:7, 0] = r["quat"]
    mdl.body2inertial_transform(pose)
    np.testing.assert_allclose(mdl._rotBtoI, r["exp_rot"], rtol=1e-6, atol=2e-6)  # the test's quaternions carry 7 digits
    np.testing.assert_allclose(mdl.restoring_forces("rest")[..., 0], r["exp_restoring"], rtol=2e-6, atol=5e-3)
    # test_damping / test_corrolis: the 6x6 matrices
    d = mdl.damping_matrix("damp", np.array(G["damping"]["vel"])[..., None])
    np.testing.assert_allclose(d, G["damping"]["exp"], **CLOSE)
    c = mdl.coriolis_matrix("coriolis", np.array([G["coriolis"]["vel"]])[..., None])
    np.testing.assert_allclose(c[0], G["coriolis"]["exp"], **CLOSE)


@pytest.mark.parametrize("params", ["test", "rexrov2"])
def test_auv_pieces_bit_exact_and_fast_forms_equal_matrix_forms(m, G, params):
    """Every intermediate of state_dot for 256 random (state, action) pairs: bit-identical to the fp32 oracle; and the rollout's
    direct forms of D nu / C nu equal the 6x6 matrices (the reference's formulation) applied to nu."""
    P = G["params"] if params == "test" else rexrov(m)
    h = m.Handle(k=1, tau=1, s_dim=13, a_dim=6, sigma=np.eye(6), goal=np.zeros(13), auv=P)
    o = orc.AuvModel(P, dtype=F32)
    x, u = rand_states(256, 1, vel=2.0)
    pc = h.auv_pieces(x, u)
    for i in range(x.shape[0]):
        rot, T = o.b2i(x[i, 3:7])
        np.testing.assert_array_equal(pc["rot"][i], rot)
        np.testing.assert_array_equal(pc["T"][i], T)
        np.testing.assert_array_equal(pc["g"][i], o.restoring(x[i, 3:7]))
        np.testing.assert_array_equal(pc["D"][i], o.damping(x[i, 7:]))
        np.testing.assert_array_equal(pc["C"][i] + 0.0, o.coriolis(x[i, 7:]) + 0.0)  # (+0.0: -0 and +0 compare equal anyway)
        np.testing.assert_array_equal(pc["xdot"][i], o.state_dot(x[i], u[i]))
        Dv = np.array([np.add.reduce((pc["D"][i][r] * x[i, 7:]).astype(F32), dtype=F32) for r in range(6)])  # row sums in index order
        np.testing.assert_allclose(pc["Dv"][i], Dv, rtol=3e-7, atol=0)
        np.testing.assert_allclose(pc["Cv"][i], (pc["C"][i].astype(np.float64) @ x[i, 7:]), rtol=2e-6, atol=2e-6 * np.abs(pc["C"][i]).max() * np.abs(x[i, 7:]).max() * 6)


@pytest.mark.parametrize("rk", [1, 2, 4])
def test_auv_step_bit_exact(m, G, rk):
    """AUVModel.build_step_graph (Euler / Heun / the reference's rk4 expression + quaternion normalisation): device == fp32 oracle,
    bit for bit, and within fp32 of the fp64 oracle. Includes the reference's own step inputs (scripts/test.py:541-586)."""
    P = dict(G["params"], rk=rk)
    mdl = m.AUVModel(actionDim=6, dt=0.1, parameters=P)
    o32, o64 = orc.AuvModel(P, dtype=F32), orc.AuvModel(P, dtype=np.float64)
    x, u = rand_states(200, 2)
    x = np.vstack([x, np.array(G["step_inputs"]["state"], F32)])
    u = np.vstack([u, np.array(G["step_inputs"]["action"], F32)])
    got = mdl.build_step_graph("step", x[..., None], u[..., None])[..., 0]
    ref = np.stack([o32.step(x[i], u[i]) for i in range(x.shape[0])])
    np.testing.assert_array_equal(got, ref)
    truth = np.stack([o64.step(x[i], u[i]) for i in range(x.shape[0])])
    np.testing.assert_allclose(got, truth, rtol=2e-5, atol=2e-5)
    np.testing.assert_allclose(np.linalg.norm(got[:, 3:7], axis=1), 1.0, atol=3e-7)


# ------------------------------------------------------------------------------------------------ NNAUVModel
def make_nnauv(seed=0, hid=32, n_hidden=3):
    rng = np.random.default_rng(seed)
    dims = [16] + [hid] * n_hidden + [13]
    W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(F32) for i in range(n_hidden + 1)]
    b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(F32) for i in range(n_hidden + 1)]
    W[-1] *= 0.1
    b[-1] *= 0.1
    return dict(W=W, b=b, xmean=rng.uniform(-0.1, 0.1, 16).astype(F32), xstd=rng.uniform(0.8, 1.2, 16).astype(F32),
                ymean=rng.uniform(-0.01, 0.01, 13).astype(F32), ystd=rng.uniform(0.8, 1.2, 13).astype(F32))


def test_golden_nnauv_data_preparation(m):
    """TestNNAUVModel's literals through the mirror class (host bookkeeping around the path)."""
    g = load_golden("model_nnauv")
    nn = m.NNAUVModel()
    t = g["training_n1"]
    X, Y = nn.prepare_training_data(np.array(t["state_t"])[..., None], np.array(t["state_t1"])[..., None], np.array(t["action"])[..., None])
    np.testing.assert_allclose(X, t["exp_x"], **CLOSE)
    np.testing.assert_allclose(Y, t["exp_y"], **CLOSE)
    for key in ("prepare_n1", "prepare_n6"):
        np.testing.assert_allclose(nn.prepare_data(np.array(g[key]["state"])[..., None], np.array(g[key]["action"])[..., None]), g[key]["exp"], **CLOSE)


@pytest.mark.parametrize("hid,n_hidden", [(32, 3), (16, 3), (32, 1), (16, 2)])
def test_nnauv_single_step_reference_order_is_bit_exact(m, hid, n_hidden):
    mlp = make_nnauv(3, hid, n_hidden)
    nn = m.NNAUVModel(weights=mlp)
    nn.set_Xmean_Xstd(mlp["xmean"], mlp["xstd"])
    nn.set_Ymean_Ystd(mlp["ymean"], mlp["ystd"])
    p32 = orc.Problem(tau=2, s=13, a=6, sigma=np.eye(6), goal=np.zeros(13), nnauv=mlp)
    x, u = rand_states(64, 4)
    u = (u / 200).astype(F32)
    got = nn.build_step_graph("nn", x[..., None], u[..., None])[..., 0]
    np.testing.assert_array_equal(got, p32.model_next(x, u))


def make_nnauv_speed(seed=0, hid=16, n_hidden=3):
    rng = np.random.default_rng(seed)
    dims = [15] + [hid] * n_hidden + [6]
    W = [(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(F32) for i in range(n_hidden + 1)]
    b = [(rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i])).astype(F32) for i in range(n_hidden + 1)]
    W[-1] *= 0.1
    b[-1] *= 0.1
    return dict(W=W, b=b, xmean=rng.uniform(-0.1, 0.1, 15).astype(F32), xstd=rng.uniform(0.8, 1.2, 15).astype(F32),
                ymean=rng.uniform(-0.01, 0.01, 6).astype(F32), ystd=rng.uniform(0.8, 1.2, 6).astype(F32))


@pytest.mark.parametrize("hid,n_hidden", [(16, 3), (32, 3), (16, 1)])
def test_nnauv_speed_single_step_and_data_preparation(m, hid, n_hidden):
    """NNAUVModelSpeed (nn_model.py:307-588) through the mirror class: one device step against the fp32 and fp64 oracle (same operation
    order in the reference-order helper kernel; the device's asinf / atan2f against libm's: a few ulp), unit quaternions out, and the
    host-side prepare_data / prepare_training_data against the oracle's restatement."""
    mlp = make_nnauv_speed(3, hid, n_hidden)
    nn = m.NNAUVModelSpeed(weights=mlp, dt=0.1)
    nn.set_Xmean_Xstd(mlp["xmean"], mlp["xstd"])
    nn.set_Ymean_Ystd(mlp["ymean"], mlp["ystd"])
    mk = lambda dt: orc.Problem(tau=2, s=13, a=6, dt=0.1, sigma=np.eye(6), goal=np.zeros(13), nnauv_speed=mlp, dtype=dt)
    p32, p64 = mk(F32), mk(np.float64)
    x, u = rand_states(256, 4)
    u = (u / 200).astype(F32)
    got = nn.build_step_graph("nn", x[..., None], u[..., None])[..., 0]
    ref32, ref64 = p32.model_next(x, u), p64.model_next(x, u)
    np.testing.assert_allclose(got, ref32, rtol=2e-6, atol=2e-6)
    assert np.abs(got - ref64).max() <= 4 * max(np.abs(ref32 - ref64).max(), 1e-6)
    assert np.abs(np.linalg.norm(got[:, 3:7], axis=1) - 1).max() < 1e-6
    np.testing.assert_allclose(nn.prepare_data(x[..., None], u[..., None]), orc.nnauv_speed_prepare_data(x, u, mlp["xmean"], mlp["xstd"]),
                               rtol=1e-9, atol=1e-9)
    X, Y = nn.prepare_training_data(x[..., None], ref64[..., None], u[..., None], norm=False)
    Xo, Yo = orc.nnauv_speed_prepare_training_data(x, ref64, u)
    np.testing.assert_allclose(X, Xo, rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(Y, Yo, rtol=1e-12, atol=1e-12)
    bad = make_nnauv_speed(0, 16, 2)  # the velocity delta has 6 components
    bad["W"][-1], bad["b"][-1] = np.zeros((16, 13), F32), np.zeros(13, F32)
    with pytest.raises(m.MppiError):
        m.Handle(k=64, tau=4, s_dim=13, a_dim=6, sigma=np.eye(6), goal=np.zeros(13), nnauv_speed=bad)


SPEED_KERNELS = {0: "mppi::k_rollout_nnspeed_pc<%d, true>", 1: "mppi::k_rollout_gen<2, %d, true>", 2: "mppi::k_rollout_nnspeed32<%d, true>"}  # (true: the diagonal-Sigma instance)


@pytest.mark.parametrize("hid,n_hidden,valu", [(16, 3, 0), (32, 2, 0), (16, 1, 0), (32, 3, 0), (16, 2, 0), (16, 3, 2), (32, 2, 2), (16, 1, 2), (32, 3, 2), (16, 3, 1), (32, 2, 1)],
                         ids=["16x3-pc", "32x2-pc", "16x1-pc", "32x3-pc", "16x2-pc", "16x3-mfma32", "32x2-mfma32", "16x1-mfma32", "32x3-mfma32", "16x3-valu", "32x2-valu"])
@pytest.mark.parametrize("cost", ["quadratic", "quat"])
def test_nnauv_speed_control_step_against_oracle(m, hid, n_hidden, valu, cost):
    """NNAUVModelSpeed in the full path — on the matrix cores as a two-wave pipeline per tile (k_rollout_nnspeed_pc<hid>, r04: a network
    wave runs the Dense stack as v_mfma_f32_32x32x2_f32 layers on two column blocks, a pose wave the cost, the quaternion kinematics and
    the Euler angles), on the one-wave-per-32-rollouts matrix-core kernel (k_rollout_nnspeed32<hid>, MPPI_TUNE_MLP32_VALU = 2) and on the
    lane-per-rollout kernel (k_rollout_gen<2, hid>, = 1): costs as close to fp64 as an fp32 CPU evaluation is (4x), U' at 1e-5 on unit
    noise; the fused Philox step on its own exported noise; 4-way sharding; ControllerBase(model=NNAUVModelSpeed, ...)."""
    import torch
    K, H = 2048, 10
    mlp = make_nnauv_speed(7, hid, n_hidden)
    sigma = 0.25 * np.eye(6)
    goal_q = GOAL13[:3] + [0.0, 0.0, np.sin(0.5), np.cos(0.5)] + [0.0] * 6
    ck = dict(goal=GOAL13, Q=np.array([10.0] * 3 + [5.0] * 4 + [1.0] * 6)) if cost == "quadratic" else dict(goal=goal_q, Q=Q10 / 10, quat_cost=True)
    cfg = dict(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, nnauv_speed=mlp, seed=9, tuning={"mlp32_valu": valu} if valu else None, **ck)
    h = m.Handle(**cfg)
    assert h.rollout_kernel_name() == SPEED_KERNELS[valu] % hid
    mk = lambda dt: orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, nnauv_speed=mlp, threads=0, dtype=dt, **ck)
    p32, p64 = mk(F32), mk(np.float64)
    rng = np.random.default_rng(1)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U = (0.1 * rng.standard_normal((H, 6))).astype(F32)
    eps = (0.25 * rng.standard_normal((K, H, 6))).astype(F32)
    h.set_action_sequence(U)
    h.next_with_noise(x0, eps)
    u64, U64, c64 = p64.next_with_noise(x0, U, eps)
    u32, U32, c32 = p32.next_with_noise(x0, U, eps)
    c = h.debug_get(m.DBG_COSTS).astype(np.float64)
    rel = lambda a: float((np.abs(a - c64) / np.abs(c64)).max())
    eu, eu_cpu = float(np.abs(h.get_action_sequence() - U64).max()), float(np.abs(np.asarray(U32, np.float64) - U64).max())
    print("NNAUVSpeed %dx%d %s: rel cost err GPU %.3g / CPU %.3g; max|dU'| GPU %.3g / CPU %.3g" % (hid, n_hidden, cost, rel(c), rel(c32.astype(np.float64)), eu, eu_cpu))
    assert rel(c) < 2e-5 and rel(c) < 4 * max(rel(c32.astype(np.float64)), 1e-6)
    assert eu <= max(1e-5, 4 * eu_cpu)
    U_in = h.get_action_sequence()
    h.next(x0)
    noise = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(noise, orc.noise(9, 1, 0, K, H, 6, sigma), rtol=0, atol=5e-6)
    _, U64b, c64b = p64.next_with_noise(x0, U_in, noise)
    _, U32b, _ = p32.next_with_noise(x0, U_in, noise)
    assert np.isfinite(c64b).all()
    assert np.abs(h.get_action_sequence() - U64b).max() <= max(1e-5, 4 * np.abs(np.asarray(U32b, np.float64) - U64b).max())
    shards = 4
    hs = [m.Handle(shard_rank=g, shard_count=shards, **cfg) for g in range(shards)]
    xd = torch.tensor(x0, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    ud = torch.zeros(6, device="cuda")
    one = torch.cuda.Stream()
    for g, hg in enumerate(hs):
        hg.set_action_sequence(U)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr(), one.cuda_stream)
    hs[0].shard_finish(recs.data_ptr(), shards, ud.data_ptr(), one.cuda_stream)
    one.synchronize()
    full = m.Handle(**cfg)
    full.set_action_sequence(U)
    np.testing.assert_allclose(ud.cpu().numpy(), full.next(x0), rtol=0, atol=2e-6)
    if cost == "quat":  # the mirror classes: ControllerBase(model=NNAUVModelSpeed, cost=StaticQuatCost) builds the same controller
        nn = m.NNAUVModelSpeed(weights=mlp, dt=0.1)
        nn.set_Xmean_Xstd(mlp["xmean"], mlp["xstd"])
        nn.set_Ymean_Ystd(mlp["ymean"], mlp["ystd"])
        ctl = m.ControllerBase(model=nn, cost=m.StaticQuatCost(1.0, 1.0, 1.0, sigma, np.array(goal_q)[:, None], Q10 / 10), k=K, tau=H, sDim=13, aDim=6,
                               lam=1.0, sigma=sigma, seed=9)
        assert ctl._h.rollout_kernel_name() == SPEED_KERNELS[0] % hid
        assert np.isfinite(ctl.next(x0[:, None])).all()


# ------------------------------------------------------------------------------------------------ costs
GOAL13 = [1.0, 2.0, -3.0, 0.0, 0.0, np.sin(0.2), np.cos(0.2)] + [0.0] * 6
Q10 = np.diag([100.0] * 3 + [10.0] + [1.0] * 6) + 0.01


def test_static_quat_cost_on_device(m):
    x, _ = rand_states(300, 5)
    cost = m.StaticQuatCost(1.0, 1.0, 1.0, np.eye(6), np.array(GOAL13)[:, None], Q10)
    got = cost.state_cost("c", x[..., None]).ravel()
    p64 = orc.Problem(tau=2, s=13, a=6, sigma=np.eye(6), goal=GOAL13, Q=Q10, quat_cost=True, dtype=np.float64)
    np.testing.assert_allclose(got, p64.state_cost(x), rtol=3e-6)
    d = np.stack([orc.quat_dist(x[i], GOAL13) for i in range(5)])
    np.testing.assert_allclose(cost.dist(x[:5, :, None])[..., 0], d, rtol=1e-6, atol=1e-6)
    with pytest.raises(AssertionError):
        m.StaticQuatCost(1.0, 1.0, 1.0, np.eye(6), np.array(GOAL13)[:, None], np.eye(13))
    # the step cost adds the Python action cost (cost_base.py:114-170)
    u, eps = np.linspace(0.1, 0.6, 6), np.random.default_rng(0).standard_normal((300, 6))
    step = cost.build_step_cost_graph("s", x[..., None], u[:, None], eps[..., None]).ravel()
    pq = orc.Problem(tau=2, s=13, a=6, sigma=np.eye(6), goal=GOAL13, Q=Q10, quat_cost=True, action_cost=orc.ACTION_COST_PY, dtype=np.float64)
    np.testing.assert_allclose(step, pq.step_cost(x, u, eps), rtol=3e-6)


def test_golden_elipse3d_cost_on_device(m):
    """TestElipse3DCost's literals through the device code with the reference's constructor and method names."""
    g = load_golden("cost_elipse3d")
    b, pl = g["base"], g["plane"]
    col = lambda v: np.array(v, np.float64)[:, None]
    mk = lambda normal, aVec, center: m.ElipseCost3D(1.0, 1.0, 1.0, np.eye(6), col(normal), col(aVec), col(b["axis"]), col(center),
                                                     b["speed"], 1.0, b["m_state"], b["m_vel"])
    for pc in g["prep_const"]:
        c = mk(pc["normal"], pc["aVec"], pc["center"])
        np.testing.assert_allclose(c.R, pc["exp_R"], **CLOSE)
        np.testing.assert_allclose(c.t, col(pc["center"]), **CLOSE)
    cost = mk(pl["normal"], pl["aVec"], pl["center"])
    np.testing.assert_allclose(cost.position_error(np.array(g["position_error"]["position"])[..., None]).ravel(), g["position_error"]["exp"], rtol=2e-6, atol=1e-6)
    np.testing.assert_allclose(cost.orientation_error(np.array(g["orientation_error"]["pose"])[..., None]), g["orientation_error"]["exp"], rtol=2e-6, atol=2e-6)
    np.testing.assert_allclose(cost.velocity_error(np.array(g["velocity_error"]["velocity"])[..., None]).ravel(), g["velocity_error"]["exp"], rtol=2e-6, atol=1e-6)
    # state_cost (no expectation in the reference): against the oracle, the reference's three states + random ones
    x = np.vstack([np.array(g["state_cost_inputs"], F32), rand_states(200, 6)[0]])
    e3 = dict(normal=pl["normal"], aVec=pl["aVec"], axis=b["axis"], speed=b["speed"], m_state=b["m_state"], m_vel=b["m_vel"])
    p64 = orc.Problem(tau=2, s=13, a=6, sigma=np.eye(6), goal=np.zeros(13), ellipse3d=e3, dtype=np.float64)
    got = cost.state_cost("c", x[..., None]).ravel()
    np.testing.assert_allclose(got, p64.state_cost(x), rtol=5e-6, atol=5e-6)


# ------------------------------------------------------------------------------------------------ rollouts and control steps
def auv_case(m, G, K, H, cost="quadratic", seed=0, **kw):
    P = dict(G["params"])
    sigma = 200.0 * np.eye(6)
    if cost == "quadratic":
        ck = dict(goal=GOAL13, Q=np.array([100.0] * 3 + [10.0] * 4 + [1.0] * 6))
        ok = dict(ck)
    elif cost == "dense":
        Qd = np.diag([100.0] * 3 + [10.0] * 4 + [1.0] * 6) + 0.05
        ck, ok = dict(goal=GOAL13, Q=Qd, q_is_full=True), dict(goal=GOAL13, Q=Qd)
    elif cost == "quat":
        ck = dict(goal=GOAL13, Q=Q10, quat_cost=True)
        ok = dict(ck)
    else:
        e3 = dict(normal=[0.0, np.sin(0.3), np.cos(0.3)], aVec=[1.0, 0.0, 0.0], axis=[2.0, 1.5], speed=1.0, m_state=50.0, m_vel=5.0)
        ck, ok = dict(ellipse3d=e3), dict(ellipse3d=e3, goal=np.zeros(13))
    h = m.Handle(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, auv=P, seed=seed + 1, **ck, **kw)
    mk = lambda dt: orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, auv=P, threads=0, dtype=dt, **ok)
    rng = np.random.default_rng(seed)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U = (50.0 * rng.standard_normal((H, 6))).astype(F32)
    eps = (200.0 * rng.standard_normal((K, H, 6))).astype(F32)
    return h, mk(F32), mk(np.float64), x0, U, eps


AUV_KERNELS = {False: "mppi::k_rollout_auv_pc<true>", True: "mppi::k_rollout_gen<0, 32, true>"}  # r04: pose wave + velocity wave per tile | one wave per tile
ONE_WAVE = pytest.mark.parametrize("one_wave", [False, True], ids=["pc", "onewave"])


def auv_tuning(one_wave):
    return dict(tuning={"gen_one_wave": 1}) if one_wave else {}


@ONE_WAVE
@pytest.mark.parametrize("cost", ["quadratic", "dense", "quat", "ellipse3d"])
@pytest.mark.parametrize("K,H", [(1024, 16), (100, 7), (129, 5)])
def test_auv_rollout_costs_against_oracle(m, G, K, H, cost, one_wave):
    """mBuildModelGraph with the Fossen model: sample costs bit-identical to the fp32 oracle for the quadratic costs (diagonal and
    dense Q), within 3e-6 relative for the quaternion / 3D-ellipse costs (acos); ragged K (an odd tile count: the two-tile workgroup of
    k_rollout_auv_pc runs half empty) and a horizon that is no multiple of 4. On both kernels: the two-wave pipeline and the one-wave-per-tile one."""
    h, p32, p64, x0, U, eps = auv_case(m, G, K, H, cost, **auv_tuning(one_wave))
    assert h.rollout_kernel_name() == AUV_KERNELS[one_wave]
    got = h.rollout_cost(x0, U, eps)
    ref = p32.rollout_cost(x0, U, eps)
    if cost in ("quadratic", "dense"):
        np.testing.assert_array_equal(got, ref)
    else:
        np.testing.assert_allclose(got, ref, rtol=3e-6)
    truth = p64.rollout_cost(x0, U, eps)
    assert (np.abs(got - truth) / np.abs(truth)).max() < 4 * max((np.abs(ref - truth) / np.abs(truth)).max(), 1e-6)


@ONE_WAVE
@pytest.mark.parametrize("cost", ["quadratic", "quat", "ellipse3d"])
def test_auv_control_step_against_oracle(m, G, cost, one_wave):
    """One control step with injected noise, then the fused Philox step on its own exported noise: U' within 1e-5 x the noise
    scale of the fp64 oracle; device noise = the oracle's Philox stream for a = 6; weights sum to 1."""
    K, H = 2048, 12
    h, p32, p64, x0, U, eps = auv_case(m, G, K, H, cost, seed=3, **auv_tuning(one_wave))
    h.set_action_sequence(U)
    u = h.next_with_noise(x0, eps)
    u64, U64, c64 = p64.next_with_noise(x0, U, eps)
    u32, U32, c32 = p32.next_with_noise(x0, U, eps)
    c = h.debug_get(m.DBG_COSTS)
    if cost == "quadratic":
        np.testing.assert_array_equal(c, c32)
    eu, eu_cpu = np.abs(h.get_action_sequence() - U64).max() / 200.0, np.abs(np.asarray(U32, np.float64) - U64).max() / 200.0
    print("AUV %s: max|dU'|/sigma GPU %.3g, fp32 CPU %.3g" % (cost, eu, eu_cpu))
    assert eu <= max(1e-5, 4 * eu_cpu) and np.abs(u - u64).max() / 200.0 <= max(1e-5, 4 * eu_cpu)
    assert abs(h.debug_get(m.DBG_WEIGHTS).astype(np.float64).sum() - 1) < 1e-5
    # fused path
    U_in = h.get_action_sequence()
    x1 = p64.model_next([x0], [u64])[0].astype(F32)
    u2 = h.next(x1)
    noise = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(noise / 200.0, orc.noise(h_seed(h), 1, 0, K, H, 6, 200.0 * np.eye(6)) / 200.0, rtol=0, atol=5e-6)
    u64b, U64b, _ = p64.next_with_noise(x1, U_in, noise)
    assert np.abs(h.get_action_sequence() - U64b).max() / 200.0 <= max(1e-5, 4 * eu_cpu)
    assert np.abs(u2 - u64b).max() / 200.0 <= max(1e-5, 4 * eu_cpu)


def h_seed(h):
    return 4  # auv_case(seed=3) builds the handle with seed + 1


def test_auv_sharded_equals_unsharded_and_normalize(m, G):
    import torch
    K, H, shards = 4096, 8, 4
    full, p32, p64, x0, U, eps = auv_case(m, G, K, H, "quat", seed=5)
    hs = [auv_case(m, G, K, H, "quat", seed=5, shard_rank=g, shard_count=shards)[0] for g in range(shards)]
    full.set_action_sequence(U)
    u_full = full.next(x0)
    xd = torch.tensor(x0, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    us = [torch.zeros(6, device="cuda") for _ in range(shards)]
    for g, hg in enumerate(hs):
        hg.set_action_sequence(U)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        hg.synchronize()
    for g, hg in enumerate(hs):
        hg.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        hg.synchronize()
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_allclose(us[g].cpu().numpy() / 200.0, u_full / 200.0, rtol=0, atol=2e-6)
        np.testing.assert_array_equal(hg.debug_get(m.DBG_COSTS), full.debug_get(m.DBG_COSTS)[hg.k_offset:hg.k_offset + hg.k_local])
    # Py normalizeCost on a 13-state handle: cost pass, min/max, record pass on the normalised costs
    hn = auv_case(m, G, 512, 6, "quadratic", seed=6, normalize_cost=True)[0]
    _, p32n, p64n, x0n, Un, epsn = auv_case(m, G, 512, 6, "quadratic", seed=6)
    hn.set_action_sequence(Un)
    un = hn.next_with_noise(x0n, epsn)
    u_ref, _, _ = p64n.next_with_noise(x0n, Un, epsn, normalize=True)
    assert np.abs(un - u_ref).max() / 200.0 < 1e-5


def test_auv_sharded_normalize_cost(m, G):
    """normalizeCost on a K-sharded 13-state controller (mppi_shard_cost_range / mppi_shard_partial_normalized): four shards agree on
    the global cost range and give the unsharded normalised step's control, replicated bit-identically."""
    import torch
    K, H, shards = 2048, 8, 4
    full, _, _, x0, U, _ = auv_case(m, G, K, H, "quat", seed=7, normalize_cost=True)
    hs = [auv_case(m, G, K, H, "quat", seed=7, normalize_cost=True, shard_rank=g, shard_count=shards)[0] for g in range(shards)]
    full.set_action_sequence(U)
    u_full = full.next(x0)
    c = full.debug_get(m.DBG_COSTS)
    xd = torch.tensor(x0, device="cuda")
    n = hs[0].record_size
    recs, rng = torch.zeros(shards * n, device="cuda"), torch.zeros(shards, 2, device="cuda")
    us = [torch.zeros(6, device="cuda") for _ in range(shards)]
    for g, hg in enumerate(hs):
        hg.set_action_sequence(U)
        hg.shard_cost_range(xd.data_ptr(), rng[g].data_ptr())
        hg.synchronize()
    agreed = rng.max(dim=0).values.contiguous()
    torch.cuda.synchronize()
    np.testing.assert_array_equal(agreed.cpu().numpy(), np.array([-c.min(), c.max()], F32))
    for g, hg in enumerate(hs):
        hg.shard_partial_normalized(xd.data_ptr(), agreed.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        hg.synchronize()
    for g, hg in enumerate(hs):
        hg.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        hg.synchronize()
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_allclose(us[g].cpu().numpy() / 200.0, u_full / 200.0, rtol=0, atol=2e-6)


@pytest.mark.parametrize("hid,n_hidden,tuning", [(32, 3, None), (32, 3, {"mlp32_valu": 1}), (16, 2, None), (32, 1, None), (32, 2, None),
                                                 (32, 3, {"mlp32_valu": 2}), (32, 1, {"mlp32_valu": 2}), (32, 2, {"mlp32_valu": 2}), (32, 3, "bf16x3"), (32, 1, "bf16x3")],
                         ids=["32x3-pc", "32x3-valu", "16x2", "32x1-pc", "32x2-pc", "32x3-mfma32", "32x1-mfma32", "32x2-mfma32", "32x3-bf16x3", "32x1-bf16x3"])
@pytest.mark.parametrize("cost", ["quadratic", "quat"])
def test_nnauv_control_step_against_oracle(m, hid, n_hidden, tuning, cost):
    """The learned 13-state model in the full path: NNAUVModel (input 16, output 13) with the quadratic or the quaternion cost —
    the combination the point-mass MLP kernels fence off (s = 2a, diagonal Q, quadratic cost). Costs as close to fp64 as an fp32
    CPU evaluation is (4x); U' at 1e-5 on unit noise; fused Philox step; 4-way sharding."""
    import torch
    K, H = 2048, 10
    mlp = make_nnauv(7, hid, n_hidden)
    sigma = 0.25 * np.eye(6)
    # NNAUVModel.next_state adds the predicted delta to the quaternion WITHOUT renormalising (nn_model.py:303-304) and StaticQuatCost
    # takes acos of the raw dot product (static_cost.py:149): a goal attitude one radian away from the start keeps <q, g_q> inside
    # (-1, 1) over the horizon (outside it the reference's cost is NaN, and so is this one)
    goal_q = GOAL13[:3] + [0.0, 0.0, np.sin(0.5), np.cos(0.5)] + [0.0] * 6
    if cost == "quat":
        mlp["W"][-1] = (0.2 * mlp["W"][-1]).astype(F32)
        mlp["b"][-1] = (0.2 * mlp["b"][-1]).astype(F32)
    ck = dict(goal=GOAL13, Q=np.array([10.0] * 3 + [5.0] * 4 + [1.0] * 6)) if cost == "quadratic" else dict(goal=goal_q, Q=Q10 / 10, quat_cost=True)
    bx3 = tuning == "bf16x3"  # MPPI_FLAG_MLP_BF16X3: k_rollout_nnauv32_bx3, held to the split-bf16 kernels' bars (8x the fp32 CPU's error, 2e-5)
    fac, bar = (8, 2e-5) if bx3 else (4, 1e-5)
    cfg = dict(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, nnauv=mlp, seed=9, tuning=None if bx3 else tuning, mlp_bf16x3=bx3, **ck)
    h = m.Handle(**cfg)
    # Dense(32): the matrix cores — the two-wave pipeline k_rollout_nnauv_pc (r04) by default, k_rollout_nnauv32 (one wave per 32 rollouts, the
    # accumulators of one layer are the next layer's B operands in both) with MPPI_TUNE_MLP32_VALU = 2 — unless tuned onto the vector ALU (= 1)
    tv = (tuning or {}).get("mlp32_valu", 0) if not bx3 else 0
    want = ("mppi::k_rollout_nnauv32_bx3<true>" if bx3 else "mppi::k_rollout_gen<1, %d, true>" % hid if (hid != 32 or tv == 1)
            else "mppi::k_rollout_nnauv32<true>" if tv == 2 else "mppi::k_rollout_nnauv_pc<true>")
    assert h.rollout_kernel_name() == want
    mk = lambda dt: orc.Problem(tau=H, s=13, a=6, lam=1.0, sigma=sigma, nnauv=mlp, threads=0, dtype=dt, **ck)
    p32, p64 = mk(F32), mk(np.float64)
    rng = np.random.default_rng(1)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U = (0.1 * rng.standard_normal((H, 6))).astype(F32)
    eps = (0.25 * rng.standard_normal((K, H, 6))).astype(F32)
    h.set_action_sequence(U)
    u = h.next_with_noise(x0, eps)
    u64, U64, c64 = p64.next_with_noise(x0, U, eps)
    u32, U32, c32 = p32.next_with_noise(x0, U, eps)
    c = h.debug_get(m.DBG_COSTS).astype(np.float64)
    rel = lambda a: float((np.abs(a - c64) / np.abs(c64)).max())
    eu, eu_cpu = float(np.abs(h.get_action_sequence() - U64).max()), float(np.abs(np.asarray(U32, np.float64) - U64).max())
    print("NNAUV %dx%d %s: rel cost err GPU %.3g / CPU %.3g; max|dU'| GPU %.3g / CPU %.3g" % (hid, n_hidden, cost, rel(c), rel(c32.astype(np.float64)), eu, eu_cpu))
    assert rel(c) < 2e-5 and rel(c) < fac * max(rel(c32.astype(np.float64)), 1e-6)
    assert eu <= max(bar, fac * eu_cpu)
    # fused Philox step on the exported noise + sharding
    U_in = h.get_action_sequence()
    u2 = h.next(x0)
    noise = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(noise, orc.noise(9, 1, 0, K, H, 6, sigma), rtol=0, atol=5e-6)
    _, U64b, c64b = p64.next_with_noise(x0, U_in, noise)
    _, U32b, _ = p32.next_with_noise(x0, U_in, noise)
    assert np.isfinite(c64b).all()
    assert np.abs(h.get_action_sequence() - U64b).max() <= max(bar, fac * np.abs(np.asarray(U32b, np.float64) - U64b).max())
    shards = 4
    hs = [m.Handle(shard_rank=g, shard_count=shards, **cfg) for g in range(shards)]
    xd = torch.tensor(x0, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    ud = torch.zeros(6, device="cuda")
    one = torch.cuda.Stream()  # ONE explicit stream for all shards (stream 0 would mean "each handle's own stream": a race)
    for g, hg in enumerate(hs):
        hg.set_action_sequence(U)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr(), one.cuda_stream)
    hs[0].shard_finish(recs.data_ptr(), shards, ud.data_ptr(), one.cuda_stream)
    one.synchronize()
    full = m.Handle(**cfg)
    full.set_action_sequence(U)
    np.testing.assert_allclose(ud.cpu().numpy(), full.next(x0), rtol=0, atol=2e-6)


def test_python_controller_with_the_auv_model_and_quaternion_cost(m, G):
    """ControllerBase(model=AUVModel, cost=StaticQuatCost) as scripts/main.py builds it for the AUV task: a few closed-loop steps
    against the oracle stepping the same plant."""
    K, H = 1024, 8
    model = m.AUVModel(actionDim=6, dt=0.1, parameters=G["params"])
    sigma = 200.0 * np.eye(6)
    cost = m.StaticQuatCost(1.0, 1.0, 1.0, sigma, np.array(GOAL13)[:, None], Q10)
    ctl = m.ControllerBase(model=model, cost=cost, k=K, tau=H, sDim=13, aDim=6, lam=1.0, sigma=sigma, seed=11)
    p64 = orc.Problem(tau=H, s=13, a=6, lam=1.0, sigma=sigma, auv=G["params"], goal=GOAL13, Q=Q10, quat_cost=True, action_cost=orc.ACTION_COST_PY,
                      threads=0, dtype=np.float64)
    x = np.array([0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0] + [0.0] * 6)
    U = np.zeros((H, 6))
    for step in range(3):
        u = ctl.next(x[:, None])
        noise = ctl._h.debug_get(m.DBG_NOISE)
        u64, U, _ = p64.next_with_noise(x, U, noise)
        assert np.abs(u - u64).max() / 200.0 < 1e-5, step
        x = p64.model_next([x], [u64])[0]
    assert np.isfinite(x).all()


def test_controller_predict_and_state_error(m, G):
    """ControllerBase.predict / state_error (controller_base.py:162-210) with log=True: save(x, u, xNext) compares the model's one-step
    prediction (on the device) with the observed next state; the observed state here comes from a plant with 3 % more mass, so the errors
    are the model mismatch — checked against the fp64 oracle stepping both parameter sets."""
    K, H = 256, 4
    model = m.AUVModel(actionDim=6, dt=0.1, parameters=G["params"])
    sigma = 200.0 * np.eye(6)
    cost = m.StaticQuatCost(1.0, 1.0, 1.0, sigma, np.array(GOAL13)[:, None], Q10)
    ctl = m.ControllerBase(model=model, cost=cost, k=K, tau=H, sDim=13, aDim=6, lam=1.0, sigma=sigma, seed=3, log=True)
    kw = dict(tau=H, s=13, a=6, lam=1.0, sigma=sigma, goal=GOAL13, Q=Q10, quat_cost=True, action_cost=orc.ACTION_COST_PY, threads=0, dtype=np.float64)
    p_model = orc.Problem(auv=G["params"], **kw)
    p_plant = orc.Problem(auv=dict(G["params"], mass=1.03 * G["params"]["mass"]), **kw)
    x = np.array([0.1, -0.2, 0.3, 0.0, 0.0, 0.0, 1.0, 0.2, 0.0, -0.1, 0.0, 0.05, 0.0])
    for step in range(3):
        u = ctl.next(x[:, None])
        xn = p_plant.model_next([x], [u])[0]
        ctl.save(x[:, None], u[:, None], xn[:, None])
        rec = ctl.predictions[-1]
        pred64 = p_model.model_next([x], [u])[0]
        e_pos, e_rot, e_vel, e_vel_dec = rec["error"]
        assert abs(e_pos - np.linalg.norm(xn[:3] - pred64[:3])) < 1e-5
        assert abs(e_rot - (1.0 - xn[3:7] @ pred64[3:7])) < 1e-5
        np.testing.assert_allclose(e_vel_dec, xn[-6:] - pred64[-6:], rtol=0, atol=2e-5)
        assert abs(e_vel - np.linalg.norm(xn[-6:] - pred64[-6:])) < 5e-5 and e_vel > 0
        np.testing.assert_allclose(rec["step_cost"], p_model.state_cost([x])[0], rtol=3e-6)
        assert rec["dist"] is not None
        x = xn
    assert len(ctl.predictions) == 3
    # the nominal rollout the reference left commented out in predict(): the updated sequence rolled from x on the device, against the
    # oracle's trajectory export on zero noise (its state costs + the terminal cost; the action cost of zero noise is the gamma u'S^-1 u term
    # of the Python form, which the reference's commented-out loop does not add either)
    traj, cost = ctl.predict_trajectory(x[:, None])
    U = ctl._h.get_action_sequence()
    c_o, tr_o = p_model.rollout_cost(x, U, np.zeros((1, H, 6)), traj=True)
    np.testing.assert_allclose(traj, tr_o[0], rtol=2e-5, atol=2e-5)
    sc = sum(p_model.state_cost([tr_o[0][t]])[0] for t in range(H)) + p_model.state_cost([tr_o[0][-1]])[0]
    np.testing.assert_allclose(cost, sc, rtol=1e-5)


def test_auv_family_argument_errors(m, G):
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=12, a_dim=6, sigma=np.eye(6), goal=np.zeros(12), auv=G["params"])
    assert e.value.status == 1 and "13" in str(e.value)
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=13, a_dim=6, sigma=np.eye(6), goal=np.zeros(13), auv=dict(G["params"], rk=3))
    assert e.value.status == 1 and "rk" in str(e.value)
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=13, a_dim=6, sigma=np.eye(6), goal=np.zeros(13), auv=dict(G["params"], mass=-1.0))
    assert e.value.status == 1
    bad = make_nnauv(0, 64, 2)
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=13, a_dim=6, sigma=np.eye(6), goal=np.zeros(13), nnauv=bad)
    assert e.value.status == 4 and "16 or 32" in str(e.value)
    with pytest.raises(m.MppiError) as e:  # the quaternion cost needs a 13-state model to roll out
        m.Handle(k=64, tau=4, s_dim=13, a_dim=6, sigma=np.eye(6), goal=GOAL13, Q=Q10, quat_cost=True)
    assert e.value.status == 4
    with pytest.raises(m.MppiError):
        m.Handle(k=1, tau=1, s_dim=13, a_dim=6, sigma=np.eye(6), goal=np.zeros(13)).auv_pieces(np.zeros((1, 13)), np.zeros((1, 6)))


@pytest.mark.parametrize("task", ["static_quat_task", "static_task_auv", "elipse3d_task"])
def test_python_entry_point_drives_the_auv(task):
    """examples/main.py = scripts/main.py's loop with the reference's get_model / get_cost switches (model.py:52-66, cost.py:51-64):
    --model rexrov2.yaml (type auv) with a static_quat / static / elipse3d task; the plant is the Fossen model stepped on the device.
    From 3.7 m away the vehicle is within half a metre of the goal position after 100 control steps."""
    import re
    import subprocess
    import sys
    cfgdir = os.path.join(ROOT, "examples", "config")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "main.py"), "--new", "--config", os.path.join(cfgdir, "uuv_sim.yaml"),
                        "--model", os.path.join(cfgdir, "rexrov2.yaml"), "--task", os.path.join(cfgdir, task + ".yaml"), "-s", "100"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if "control steps" in l][-1]
    if task == "elipse3d_task":
        assert "elipse3d state cost" in line and np.isfinite(float(re.search(r"state cost ([-0-9.e+]+)", line).group(1)))
    else:
        assert float(re.search(r"goal_p\| = ([0-9.]+) m", line).group(1)) < 0.5, line


def test_python_entry_point_learns_the_plant_online():
    """examples/main.py --model auv_nn_model.yaml --plant rexrov2.yaml -t N: the reference's learn-while-controlling loop (main.py:51-52,
    104-105): transitions of the plant fill the replay buffer, every N steps LearnerBase retrains NNAUVModel on the device and the
    weights travel into the LIVE controller (mppi_set_mlp). Checked here: the loop runs, every training round lowers its loss, the
    state stays finite (what a 13-state network learns from 120 transitions is not a controller yet)."""
    import re
    import subprocess
    import sys
    cfgdir = os.path.join(ROOT, "examples", "config")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "main.py"), "--new", "--config", os.path.join(cfgdir, "uuv_sim.yaml"),
                        "--model", os.path.join(cfgdir, "auv_nn_model.yaml"), "--plant", os.path.join(cfgdir, "rexrov2.yaml"),
                        "--task", os.path.join(cfgdir, "static_task_auv.yaml"), "-s", "120", "-t", "40"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    rounds = re.findall(r"normalised loss ([0-9.e+-]+) -> ([0-9.e+-]+)", r.stdout)
    assert len(rounds) == 3 and all(float(b) < 0.25 * float(a) for a, b in rounds), r.stdout
    assert np.isfinite(float(re.search(r"goal_p\| = ([0-9.]+) m", r.stdout).group(1)))


def test_python_entry_point_resumes_what_it_learned(tmp_path):
    """examples/main.py --train N --log_dir D (r04): every training round saves <D>/learner/weights_step<N> (LearnerBase.save_params,
    learner_base.py:66-68 — weights, normalisation, Adam moments, step count in one flat file) and a second run with the same --log_dir
    resumes from the newest one: its first training round starts where the first run's last one ended, not from scratch."""
    import re
    import subprocess
    import sys
    cfgdir = os.path.join(ROOT, "examples", "config")
    cmd = [sys.executable, os.path.join(ROOT, "examples", "main.py"), "--new", "--config", os.path.join(cfgdir, "uuv_sim.yaml"),
           "--model", os.path.join(cfgdir, "auv_nn_model.yaml"), "--plant", os.path.join(cfgdir, "rexrov2.yaml"),
           "--task", os.path.join(cfgdir, "static_task_auv.yaml"), "-s", "80", "-t", "40", "--log_dir", str(tmp_path)]
    first = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert first.returncode == 0, first.stderr[-2000:]
    saved = re.findall(r"saved (\S+weights_step(\d+))", first.stdout)
    assert [int(n) for _, n in saved] == [200, 400] and all(os.path.exists(f) for f, _ in saved), first.stdout
    assert "resumed" not in first.stdout
    second = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert second.returncode == 0, second.stderr[-2000:]
    assert re.search(r"resumed the learned model from \S+weights_step400 \(400 training epochs so far\)", second.stdout), second.stdout
    l1 = [float(a) for a, _ in re.findall(r"normalised loss ([0-9.e+-]+) -> ([0-9.e+-]+)", first.stdout)]
    l2 = [float(a) for a, _ in re.findall(r"normalised loss ([0-9.e+-]+) -> ([0-9.e+-]+)", second.stdout)]
    assert l2[0] < l1[0], (l1, l2)  # the resumed network starts from a trained state, not from scratch
    assert [int(n) for _, n in re.findall(r"saved (\S+weights_step(\d+))", second.stdout)] == [600, 800]


@pytest.mark.parametrize("kind,K,H", [("bf16x3", 300, 7), ("bf16x3", 65, 5), ("speed", 300, 7), ("speed", 1, 3), ("speed", 129, 6), ("speed", 64, 1), ("mfma", 65, 5),
                                      ("mfma", 129, 6), ("mfma", 1, 2)])
def test_learned_13_state_kernels_with_ragged_tiles(m, kind, K, H):
    """K that is no multiple of the 64-rollout tile (partial last tile, a single rollout) and a horizon that is no multiple of the 4-step
    Philox group, for the learned 13-state kernels: k_rollout_nnauv32(_bx3) and NNAUVModelSpeed's k_rollout_gen<2, .> — the fused step
    against the fp64 CPU restatement on the exported noise."""
    sigma = 0.25 * np.eye(6)
    ck = dict(goal=GOAL13, Q=np.array([10.0] * 3 + [5.0] * 4 + [1.0] * 6))
    if kind == "speed":
        mlp = make_nnauv_speed(11, 16, 3)
        mkw, okw = dict(nnauv_speed=mlp), dict(nnauv_speed=mlp)
    else:
        mlp = make_nnauv(11, 32, 3)
        mkw, okw = dict(nnauv=mlp, mlp_bf16x3=(kind == "bf16x3")), dict(nnauv=mlp)
    h = m.Handle(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, seed=21, **mkw, **ck)
    p64 = orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, threads=0, dtype=np.float64, **okw, **ck)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U_in = h.get_action_sequence()
    u = h.next(x0)
    noise = h.debug_get(m.DBG_NOISE)
    assert noise.shape == (K, H, 6)
    np.testing.assert_allclose(noise, orc.noise(21, 0, 0, K, H, 6, sigma), rtol=0, atol=5e-6)
    u64, U64, c64 = p64.next_with_noise(x0, U_in, noise)
    bar = 2e-5 if kind == "bf16x3" else 1e-5
    np.testing.assert_allclose(h.debug_get(m.DBG_COSTS), c64, rtol=2e-5)
    np.testing.assert_allclose(h.get_action_sequence(), U64, rtol=0, atol=bar)
    np.testing.assert_allclose(u, u64, rtol=0, atol=bar)


DENSE_SIGMA6 = (0.25 * np.eye(6) + 0.03 * np.add.outer(np.arange(6), np.arange(6)) / 5.0).astype(F32)


@pytest.mark.parametrize("kind", ["auv", "mfma", "bf16x3", "valu16", "speed"])
def test_13_state_kernels_with_a_dense_sigma(m, G, kind):
    """The 13-state kernels are instantiated for an exactly diagonal Sigma (off-diagonal products skipped: exact zeros) and for a dense
    one; every other test here uses a diagonal Sigma, this one the dense instances: noise = the CPU restatement's Philox stream scaled by
    the dense matrix, the fused step against the fp64 evaluation on it; the Fossen model's costs bit-identical to the fp32 one."""
    K, H = 1500, 9
    ck = dict(goal=GOAL13, Q=np.array([10.0] * 3 + [5.0] * 4 + [1.0] * 6))
    scale = 200.0 if kind == "auv" else 1.0
    sigma = (scale * DENSE_SIGMA6).astype(F32)
    if kind == "auv":
        mkw, okw = dict(auv=G["params"]), dict(auv=G["params"])
    elif kind == "speed":
        mlp = make_nnauv_speed(5, 16, 3)
        mkw, okw = dict(nnauv_speed=mlp), dict(nnauv_speed=mlp)
    else:
        mlp = make_nnauv(5, 16 if kind == "valu16" else 32, 3)
        mkw, okw = dict(nnauv=mlp, mlp_bf16x3=(kind == "bf16x3")), dict(nnauv=mlp)
    h = m.Handle(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, seed=8, **mkw, **ck)
    assert h.rollout_kernel_name().endswith("false>")
    mk = lambda dt: orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, threads=0, dtype=dt, **okw, **ck)
    p32, p64 = mk(F32), mk(np.float64)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U_in = h.get_action_sequence()
    u = h.next(x0)
    noise = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(noise, orc.noise(8, 0, 0, K, H, 6, sigma), rtol=0, atol=5e-6 * scale)
    u64, U64, c64 = p64.next_with_noise(x0, U_in, noise)
    c = h.debug_get(m.DBG_COSTS)
    if kind == "auv":
        np.testing.assert_array_equal(c, p32.next_with_noise(x0, U_in, noise)[2])
    np.testing.assert_allclose(c, c64, rtol=2e-5)
    bar = (2e-5 if kind == "bf16x3" else 1e-5) * scale
    np.testing.assert_allclose(h.get_action_sequence(), U64, rtol=0, atol=bar)
    np.testing.assert_allclose(u, u64, rtol=0, atol=bar)


# ------------------------------------------------------------------------------------------------ full size
# Every 13-state sub-record of bench.py runs K = 65536, H = 64 (1024 tiles = one wave per SIMD, a 1024-record finish, 16 horizon
# groups x 6 Philox blocks, the record pass regenerating the noise): the same (kernel, K, H) against the oracle here, on the bench's
# own task (auv_task: rexrov2, the static 13-state goal, 1500 N noise).
FULL_K, FULL_H = 65536, 64


def full_size_cfg(kind):
    """-> (Handle kwargs, oracle Problem kwargs, x0) of the bench's task with the model `kind`"""
    from mppi_tf_amd.auv import auv_task
    cfg = auv_task(FULL_H, learned=(kind != "auv"))
    x0 = np.asarray(cfg.pop("x0"), F32)
    ok = dict(tau=FULL_H, s=13, a=6, dt=cfg["dt"], lam=cfg["lam"], sigma=cfg["sigma"], goal=cfg["goal"], Q=cfg["Q"])
    if kind == "auv":
        ok["auv"] = cfg["auv"]
        return cfg, ok, x0
    # a learned model at the task's scales: forces of 1500 N enter the network through its input normalisation, the predicted
    # state delta leaves it through a small output scale — the recurrence stays bounded over 64 steps (synthetic weights)
    speed = kind == "speed"
    net = make_nnauv_speed(3, 16, 3) if speed else make_nnauv(3, 32, 3)
    n_state = 9 if speed else 10
    net["xstd"] = np.array([1.0] * n_state + [1500.0] * 6, F32)
    net["xmean"] = np.zeros(n_state + 6, F32)
    net["ystd"] = np.full(6 if speed else 13, 0.05, F32)
    net["ymean"] = np.zeros(6 if speed else 13, F32)
    key = "nnauv_speed" if speed else "nnauv"
    cfg[key], ok[key] = net, net
    if kind == "bf16x3":
        cfg["mlp_bf16x3"] = True
    return cfg, ok, x0


def shards_agree_with(m, cfg, x0, U_in, u, c, shards=8):
    """the K-sharded step (records exchanged by hand) against the unsharded one: same costs, replicated control within 2e-6 of sigma"""
    import torch
    hs = [m.Handle(k=FULL_K, shard_rank=g, shard_count=shards, **cfg) for g in range(shards)]
    xd = torch.tensor(x0, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    us = [torch.zeros(6, device="cuda") for _ in range(shards)]
    for g, hg in enumerate(hs):
        hg.set_action_sequence(U_in)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        hg.synchronize()
    for g, hg in enumerate(hs):
        hg.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        hg.synchronize()
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_allclose(us[g].cpu().numpy() / 1500.0, u / 1500.0, rtol=0, atol=2e-6)
        np.testing.assert_array_equal(hg.debug_get(m.DBG_COSTS), c[hg.k_offset:hg.k_offset + hg.k_local])


def update_is_the_recombination(m, h, U_in, c, eps, u, lam=1.0, scale=1500.0):
    """weights = the soft-min of the kernel's own costs, sum to 1; U' = U + sum_k w_k eps_k recomputed in fp64 from the exported noise"""
    c64 = c.astype(np.float64)
    e = np.exp(-(c64 - c64.min()) / lam)
    w = h.debug_get(m.DBG_WEIGHTS).astype(np.float64)
    assert abs(w.sum() - 1) < 1e-5 and (w >= 0).all()
    np.testing.assert_allclose(w, e / e.sum(), rtol=1e-5, atol=1e-12)
    Uupd = h.debug_get(m.DBG_U_UPDATED).astype(np.float64)
    want = U_in + np.tensordot(e / e.sum(), eps.astype(np.float64), axes=(0, 0))
    print("max|U' - fp64 recombination| / sigma = %.3g" % (np.abs(Uupd - want).max() / scale))
    np.testing.assert_allclose(Uupd / scale, want / scale, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(u, Uupd[0].astype(F32))


@ONE_WAVE
def test_full_size_fossen_auv_costs_bit_identical(m, one_wave):
    """k_rollout_auv_pc<true> / k_rollout_gen<0, 32, true> (AUVModel rk2, rexrov2) at the bench's size, fused Philox step: ALL 65536 sample
    costs bit-identical to the fp32 oracle on the exported noise (auv_model.py:282-333), the noise is the oracle's Philox stream for a = 6,
    U' is the fp64 recombination, and the 8-way sharded step is the unsharded one."""
    cfg, ok, x0 = full_size_cfg("auv")
    cfg.update(auv_tuning(one_wave))
    h = m.Handle(k=FULL_K, **cfg)
    assert h.rollout_kernel_name() == AUV_KERNELS[one_wave]
    p32 = orc.Problem(threads=0, **ok)
    U_in = (100.0 * np.random.default_rng(1).standard_normal((FULL_H, 6))).astype(F32)
    h.set_action_sequence(U_in)
    u = h.next(x0)
    c, eps = h.debug_get(m.DBG_COSTS), h.debug_get(m.DBG_NOISE)
    assert np.isfinite(c).all()
    np.testing.assert_array_equal(c, p32.rollout_cost(x0, U_in, eps))
    np.testing.assert_allclose(eps[:128] / 1500.0, orc.noise(1, 0, 0, 128, FULL_H, 6, cfg["sigma"]) / 1500.0, rtol=0, atol=5e-6)
    update_is_the_recombination(m, h, U_in, c, eps, u)
    shards_agree_with(m, cfg, x0, U_in, u, c)


@pytest.mark.parametrize("rk", [2, 4, 1])
def test_multi_round_fossen_auv_costs_bit_identical(m, rk):
    """K = 200001 (3126 tiles: several rounds per SIMD, a ragged last tile, the 16:1 record fold in front of the finish), rk2 / rk4 / rk1
    (1, 3 or no stage hand-offs inside a step of the two-wave pipeline): costs bit-identical to the fp32 oracle, U' the fp64 recombination."""
    cfg, ok, x0 = full_size_cfg("auv")
    cfg["auv"] = dict(cfg["auv"], rk=rk)
    ok["auv"] = cfg["auv"]
    K, H = 200001, 16
    cfg["tau"], ok["tau"] = H, H
    h = m.Handle(k=K, **cfg)
    p32 = orc.Problem(threads=0, **ok)
    U_in = (100.0 * np.random.default_rng(2).standard_normal((H, 6))).astype(F32)
    h.set_action_sequence(U_in)
    u = h.next(x0)
    c, eps = h.debug_get(m.DBG_COSTS), h.debug_get(m.DBG_NOISE)
    assert c.shape == (K,) and eps.shape == (K, H, 6)
    np.testing.assert_array_equal(c, p32.rollout_cost(x0, U_in, eps))
    update_is_the_recombination(m, h, U_in, c, eps, u)


@pytest.mark.parametrize("kind", ["mfma", "bf16x3", "speed"])
def test_full_size_learned_13_state_models(m, kind):
    """k_rollout_nnauv32<true>, k_rollout_nnauv32_bx3<true> (NNAUVModel, nn_model.py:215-304) and NNAUVModelSpeed's kernel (nn_model.py:307-588)
    at the bench's size: 1536 sampled costs as close to the fp64 oracle as the fp32 CPU evaluation is (4x; 8x for the split-bf16 kernel), U' = the
    fp64 recombination of the exported noise with the soft-min of the kernel's own costs, 8-way sharded == unsharded."""
    cfg, ok, x0 = full_size_cfg(kind)
    h = m.Handle(k=FULL_K, **cfg)
    name = h.rollout_kernel_name()
    assert name.startswith({"mfma": "mppi::k_rollout_nnauv_pc<true>", "bf16x3": "mppi::k_rollout_nnauv32_bx3<true>", "speed": "mppi::k_rollout_nnspeed_pc<16, true>"}[kind]), name
    p32, p64 = orc.Problem(threads=0, **ok), orc.Problem(threads=0, dtype=np.float64, **ok)
    U_in = (100.0 * np.random.default_rng(1).standard_normal((FULL_H, 6))).astype(F32)
    h.set_action_sequence(U_in)
    u = h.next(x0)
    c, eps = h.debug_get(m.DBG_COSTS), h.debug_get(m.DBG_NOISE)
    assert np.isfinite(c).all()
    idx = np.sort(np.random.default_rng(3).choice(FULL_K, 1536, replace=False))
    truth = p64.rollout_cost(x0, U_in, eps[idx])
    cpu = p32.rollout_cost(x0, U_in, eps[idx]).astype(np.float64)
    rel = lambda a: float((np.abs(a - truth) / np.abs(truth)).max())
    e_gpu, e_cpu = rel(c[idx].astype(np.float64)), rel(cpu)
    print("%s K=%d H=%d: max rel cost error on 1536 sampled rollouts GPU %.3g, fp32 CPU %.3g" % (name, FULL_K, FULL_H, e_gpu, e_cpu))
    fac = 8 if kind == "bf16x3" else 4
    assert e_gpu < fac * max(e_cpu, 1e-6)
    # the task's costs share a large constant part (1e4 x the squared distance to a goal 10 m away): measured against the SPREAD of the
    # costs over the samples — what the soft-min sees — the bar is the same
    spread = truth.std()
    s_gpu, s_cpu = float(np.abs(c[idx] - truth).max() / spread), float(np.abs(cpu - truth).max() / spread)
    print("   relative to the costs' spread over the samples (%.3g of their mean): GPU %.3g, fp32 CPU %.3g" % (spread / truth.mean(), s_gpu, s_cpu))
    assert s_gpu < fac * max(s_cpu, 1e-6)
    np.testing.assert_allclose(eps[:128] / 1500.0, orc.noise(1, 0, 0, 128, FULL_H, 6, cfg["sigma"]) / 1500.0, rtol=0, atol=5e-6)
    update_is_the_recombination(m, h, U_in, c, eps, u)
    shards_agree_with(m, cfg, x0, U_in, u, c)


def test_static_quat_cost_at_the_goal_attitude_is_finite(m):
    """<q, g_q> of two unit quaternions can round to 1 + 1 ulp when the vehicle sits AT a goal attitude that is not
    axis-aligned; acosf(1 + ulp) is NaN and one NaN cost poisons eta, U' and the warm start for good. The device clamps the dot
    product to [-1, 1]: the reference's value wherever the reference's is finite (elsewhere the reference's own cost is NaN,
    static_cost.py:141-159). Found by search: goal attitudes whose fp32 dot with themselves exceeds 1."""
    rng = np.random.default_rng(0)
    q = rng.standard_normal((20000, 4)).astype(F32)
    q = (q / np.linalg.norm(q, axis=1, keepdims=True).astype(F32)).astype(F32)
    dot = ((q[:, 0] * q[:, 0] + q[:, 1] * q[:, 1]) + q[:, 2] * q[:, 2]) + q[:, 3] * q[:, 3]  # the cost's own order, fp32
    over = np.nonzero(dot > 1.0)[0]
    assert over.size > 0
    g = q[over[0]]
    goal = [1.0, 2.0, -3.0] + [float(v) for v in g] + [0.0] * 6
    x = np.array(goal, F32)
    h = m.Handle(k=64, tau=4, s_dim=13, a_dim=6, sigma=np.eye(6), goal=goal, Q=Q10, quat_cost=True, auv=rexrov(m))
    c = h.state_cost(x[None])
    assert np.isfinite(c).all() and float(c[0]) == 0.0
    # the closed loop started at the goal stays finite
    hh = m.Handle(k=1024, tau=8, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=50.0 * np.eye(6), goal=goal, Q=Q10, quat_cost=True, auv=rexrov(m), seed=3)
    for _ in range(3):
        assert np.isfinite(hh.next(x)).all()
    assert np.isfinite(hh.get_action_sequence()).all()


# ------------------------------------------------------------------------------------------------ random shapes through the two-wave pipelines (r04)
def _random_shapes(n, seed):
    rng = np.random.default_rng(seed)
    return [(int(rng.integers(1, 700)), int(rng.integers(1, 23))) for _ in range(n)]


@pytest.mark.parametrize("K,H", _random_shapes(10, 41))
def test_auv_pipeline_random_shapes_bit_identical(m, G, K, H):
    """k_rollout_auv_pc at random (K, H) — partial tiles, odd tile counts (the second tile of the last workgroup missing), horizons that are no
    multiple of the Philox group, down to K = 1 or H = 1 — and a random rk: the fused Philox step's costs bit-identical to the fp32 oracle on the
    exported noise, its U' within 1e-5 of sigma of the fp64 update."""
    rk = (1, 2, 4)[(K + H) % 3]
    P = dict(G["params"], rk=rk)
    sigma = 200.0 * np.eye(6)
    ck = dict(goal=GOAL13, Q=np.array([100.0] * 3 + [10.0] * 4 + [1.0] * 6))
    h = m.Handle(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, auv=P, seed=K, **ck)
    assert h.rollout_kernel_name() == AUV_KERNELS[False]
    p32 = orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, auv=P, threads=0, **ck)
    p64 = orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, auv=P, threads=0, dtype=np.float64, **ck)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U = (50.0 * np.random.default_rng(H).standard_normal((H, 6))).astype(F32)
    h.set_action_sequence(U)
    u = h.next(x0)
    c, eps = h.debug_get(m.DBG_COSTS), h.debug_get(m.DBG_NOISE)
    np.testing.assert_array_equal(c, p32.rollout_cost(x0, U, eps))
    u64, U64, _ = p64.next_with_noise(x0, U, eps)
    u32, U32, _ = p32.next_with_noise(x0, U, eps)
    bar = max(1e-5, 4 * np.abs(np.asarray(U32, np.float64) - U64).max() / 200.0)
    assert np.abs(h.get_action_sequence() - U64).max() / 200.0 <= bar and np.abs(u - u64).max() / 200.0 <= bar


@pytest.mark.parametrize("kind", ["nnauv", "speed16", "speed32"])
@pytest.mark.parametrize("K,H", _random_shapes(6, 43))
def test_learned_pipelines_random_shapes(m, kind, K, H):
    """k_rollout_nnauv_pc / k_rollout_nnspeed_pc<16|32> at random (K, H) and a random number of hidden layers: the fused Philox step against the
    fp64 oracle on the exported noise (costs 2e-5 relative, U' 1e-5 or 4x the fp32 CPU evaluation's own error)."""
    n_hidden = 1 + (K + H) % 3
    sigma = 0.25 * np.eye(6)
    ck = dict(goal=GOAL13, Q=np.array([10.0] * 3 + [5.0] * 4 + [1.0] * 6))
    if kind == "nnauv":
        mlp = make_nnauv(K, 32, n_hidden)
        mkw, want = dict(nnauv=mlp), "mppi::k_rollout_nnauv_pc<true>"
    else:
        hid = 16 if kind == "speed16" else 32
        mlp = make_nnauv_speed(K, hid, n_hidden)
        mkw, want = dict(nnauv_speed=mlp), SPEED_KERNELS[0] % hid
    h = m.Handle(k=K, tau=H, s_dim=13, a_dim=6, dt=0.1, lam=1.0, sigma=sigma, seed=H, **mkw, **ck)
    assert h.rollout_kernel_name() == want
    p64 = orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, threads=0, dtype=np.float64, **mkw, **ck)
    x0 = np.array([0.5, -0.5, 0.2, 0.0, 0.0, 0.0, 1.0, 0.3, 0.0, -0.1, 0.0, 0.05, 0.0], F32)
    U_in = h.get_action_sequence()
    u = h.next(x0)
    noise = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(noise, orc.noise(H, 0, 0, K, H, 6, sigma), rtol=0, atol=5e-6)
    u64, U64, c64 = p64.next_with_noise(x0, U_in, noise)
    p32 = orc.Problem(tau=H, s=13, a=6, dt=0.1, lam=1.0, sigma=sigma, threads=0, **mkw, **ck)
    _, U32, _ = p32.next_with_noise(x0, U_in, noise)
    np.testing.assert_allclose(h.debug_get(m.DBG_COSTS), c64, rtol=2e-5)
    # few samples at lambda = 1: the fp32 CPU evaluation itself is the yardstick where it cannot hold 1e-5 (as for the other learned-model kernels)
    bar = max(1e-5, 4 * float(np.abs(np.asarray(U32, np.float64) - U64).max()))
    assert np.abs(h.get_action_sequence() - U64).max() <= bar and np.abs(u - u64).max() <= bar
