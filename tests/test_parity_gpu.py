"""Parity of the HIP path (through the C-ABI, libmppi_hip.so) against the CPU oracle and the
reference's golden vectors. Needs an MI355X: every test is marked `gpu`.

Bars (BASELINE.json north_star): control-update numerics within 1e-5 of the reference path on
identical noise; sample costs are additionally required to be BIT-IDENTICAL to the oracle's
unfused fp32 evaluation (both sides are compiled without FMA contraction).
"""
import numpy as np
import pytest

from conftest import assert_float_eq, load_golden
from oracle import oracle as orc

pytestmark = pytest.mark.gpu

F32 = np.float32
U_TOL = 1e-5  # stated fp32 tolerance on the control update (absolute, on U' and u)


@pytest.fixture(scope="module")
def m():
    import mppi_tf_amd
    assert mppi_tf_amd.load().mppi_device_count() >= 1, "no GPU visible to libmppi_hip.so"
    return mppi_tf_amd


GOAL3 = [1, 0, .5, 0, .75, 0]  # position = MuJoCo target site, velocity 0


def make_pair(m, K, H, a, dt=0.1, mass=1.0, lam=1.0, sigma=None, goal=None, Q=None, q_full=False,
              action_cost=0, gamma=1.0, upsilon=1.0, normalize=False, seed=1, **kw):
    s = 2 * a
    sigma = np.eye(a) * 0.25 if sigma is None else sigma
    goal = ((GOAL3 + [0.25, 0])[:s] if goal is None else goal)
    Q = np.ones(s) if Q is None else Q
    h = m.Handle(k=K, tau=H, s_dim=s, a_dim=a, dt=dt, mass=mass, lam=lam, sigma=sigma, goal=goal, Q=Q,
                 q_is_full=q_full, action_cost=action_cost, gamma=gamma, upsilon=upsilon,
                 normalize_cost=normalize, seed=seed, **kw)
    p = orc.Problem(tau=H, s=s, a=a, dt=dt, mass=mass, lam=lam, sigma=sigma, goal=goal, Q=Q,
                    action_cost=action_cost, gamma=gamma, upsilon=upsilon, threads=0)
    return h, p


# =============================================================== golden vectors through the C-ABI
@pytest.mark.parametrize("idx", range(4))
def test_golden_model_step_cpp(m, idx):
    """test/test_model.cpp:120-255 (StepTesting1/2, LargeTesting, InitTest) on the device."""
    sc = load_golden("model_cpp")["scenarios"][idx]
    mod = m.PointMassModel(sc["mass"], sc["dt"], sc["s"], sc["a"])
    st, ac = np.asarray(sc["state"])[..., None], np.asarray(sc["action"])[..., None]
    free = mod.build_free_step_graph("", st)
    assert free.shape == (len(sc["exp_free"]), sc["s"], 1)
    assert_float_eq(free, sc["exp_free"], what="free")
    assert_float_eq(mod.build_action_step_graph("", ac), sc["exp_action"], what="action")
    res = mod.build_step_graph("", st, ac)
    assert res.shape == (len(sc["action"]), sc["s"], 1)
    assert_float_eq(res, sc["exp_result"], what="result")


def test_golden_model_step_py(m):
    """scripts/test.py:43-218 incl. the 3-step recurrence; fp32 device vs fp64 expectations."""
    g = load_golden("model_py")
    for sc in g["scenarios"]:
        mod = m.PointMassModel(sc["mass"], sc["dt"], sc["s"], sc["a"])
        st, ac = np.asarray(sc["state"])[..., None], np.asarray(sc["action"])[..., None]
        np.testing.assert_allclose(mod.build_step_graph("", st, ac)[..., 0], sc["exp_result"], rtol=1e-6, atol=1e-6)
    sc = g["step3"]
    mod = m.PointMassModel(sc["mass"], sc["dt"], sc["s"], sc["a"])
    x, ac = np.asarray(sc["state"])[..., None], np.asarray(sc["action"])[..., None]
    for _ in range(sc["n_steps"]):
        x = mod.build_step_graph("", x, ac)
    np.testing.assert_allclose(x[..., 0], sc["exp_result"], rtol=1e-6, atol=1e-6)


@pytest.mark.parametrize("idx", range(3))
def test_golden_cost_cpp(m, idx):
    """test/test_cost.cpp:169-239 StateCost / StepCost."""
    sc = load_golden("cost_cpp")["scenarios"][idx]
    h = m.Handle(k=1, tau=1, s_dim=sc["s"], a_dim=sc["a"], lam=sc["lam"], sigma=sc["sigma"], goal=sc["goal"],
                 Q=sc["q_diag"], q_is_full=False)
    assert_float_eq(h.state_cost(sc["state"]), sc["exp_state"], what="state")
    assert_float_eq(h.step_cost(sc["state"], sc["action"], sc["eps"]), sc["exp_step"], what="step")


def test_golden_cost_py(m):
    """scripts/test.py:685-1096 TestCost / TestStaticCost through the mirrored classes."""
    g = load_golden("cost_py")
    for sc in g["action_cost"]:
        c = m.CostBase(sc["lam"], sc["gamma"], sc["upsilon"], np.asarray(sc["sigma"]))
        with pytest.raises(NotImplementedError):
            c.state_cost("", np.zeros((1, 2, 1)))
        got = c.action_cost("", np.asarray(sc["action"])[:, None], np.asarray(sc["noise"])[..., None])
        assert got.shape == (len(sc["noise"]), 1, 1)
        np.testing.assert_allclose(got.ravel(), sc["exp_action"], rtol=2e-6, atol=2e-6)
    for sc in g["static_cost"]:
        c = m.StaticCost(sc["lam"], sc["gamma"], sc["upsilon"], np.asarray(sc["sigma"]),
                         np.asarray(sc["goal"])[:, None], np.asarray(sc["Q"]))
        a, n, s = np.asarray(sc["action"])[:, None], np.asarray(sc["noise"])[..., None], np.asarray(sc["state"])[..., None]
        np.testing.assert_allclose(c.action_cost("", a, n).ravel(), sc["exp_action"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(c.state_cost("", s).ravel(), sc["exp_state"], rtol=2e-6, atol=2e-6)
        np.testing.assert_allclose(c.build_step_cost_graph("", s, a, n).ravel(), sc["exp_step"], rtol=2e-6, atol=2e-6)


def test_golden_update_chain(m):
    """test/test_controller.cpp:109-167 testUpdate: beta, exp_arg, exp, nabla, weights, Σwε, Σw=1."""
    g = load_golden("controller_update_k5_tau3_a2")
    h = m.Handle(k=g["k"], tau=g["tau"], s_dim=4, a_dim=g["a"], lam=g["lam"], dt=0.01)
    r = h.update(g["cost"], g["noise"], g["action"])
    assert_float_eq(r["beta"], g["beta"], what="beta")
    assert_float_eq(r["arg"], g["exp_arg"], what="exp_arg")
    assert_float_eq(r["exp"], g["exp"], what="exp")
    assert_float_eq(r["nabla"], g["nabla"], what="nabla")
    assert_float_eq(r["w"], g["weights"], what="weights")
    assert_float_eq(r["wn"], g["weighted_noise"], what="weighted noise")
    assert_float_eq(np.sum(r["w"], dtype=F32), 1.0, what="sum w")
    assert_float_eq(r["Unew"], np.asarray(g["action"], F32) + np.asarray(g["weighted_noise"], F32), what="U'")
    assert h.get_step_counter() == 0 and not h.get_action_sequence().any()  # stateless helper


def test_golden_getnew_shift(m):
    """test/test_controller.cpp:169-222 testNew / testShiftAndInit."""
    from mppi_tf_amd import _lib
    g = load_golden("controller_getnew_shift")
    for nb, exp in g["getnew"].items():
        out = _lib.get_new(g["action"], int(nb))
        assert out.shape == (int(nb), 2)
        assert_float_eq(out, np.asarray(exp, F32).reshape(int(nb), 2), ulps=0)
    for case in g["shift"]:
        assert_float_eq(_lib.shift(g["action"], case["init"], case["nb"]), case["expected"], ulps=0)


# =============================================================== A7: the H-step rollout cost
ROLLOUT_CASES = [
    dict(K=128, H=32, a=1),                                   # C1 point_mass1d
    dict(K=4096, H=64, a=2),                                  # C2 point_mass2d
    dict(K=65536, H=64, a=3),                                 # C3 point_mass3d, full size
    dict(K=1000, H=50, a=3, mass=5.0, lam=0.5),               # authors' default-ish, ragged last tile
    dict(K=1, H=1, a=1),                                      # degenerate
    dict(K=63, H=7, a=2, dt=0.01, mass=1.5),                  # less than one tile
    dict(K=65, H=3, a=4),                                     # one sample into the second tile
    dict(K=257, H=128, a=3),                                  # H=128: R=64 tile is 98 KB of LDS
    dict(K=300, H=256, a=3),                                  # forces the R=32 tile
    dict(K=100, H=500, a=4),                                  # forces the R=16 tile (136 KB of LDS)
    dict(K=513, H=20, a=3, q_full=True),                      # dense Q (Py StaticCost)
    dict(K=513, H=20, a=3, action_cost=1, gamma=2.0, upsilon=3.0, lam=10.0),  # Py γ/υ action cost
    dict(K=200, H=16, a=2, sigma=[[0.5, 0.1], [0.0, 0.25]]),  # non-diagonal Σ
]


@pytest.mark.parametrize("case", ROLLOUT_CASES, ids=lambda c: "K%d_H%d_a%d" % (c["K"], c["H"], c["a"]))
def test_rollout_cost_bit_exact(m, case):
    case = dict(case)
    K, H, a = case.pop("K"), case.pop("H"), case.pop("a")
    s = 2 * a
    rng = np.random.default_rng(K + H)
    if case.pop("q_full", False):
        B = rng.standard_normal((s, s))
        case["Q"], case["q_full"] = (B @ B.T / s + np.eye(s)).astype(F32), True
    h, p = make_pair(m, K, H, a, **case)
    x0 = rng.standard_normal(s).astype(F32)
    U = (0.3 * rng.standard_normal((H, a))).astype(F32)
    eps = (0.5 * rng.standard_normal((K, H, a))).astype(F32)
    got = h.rollout_cost(x0, U, eps)
    np.testing.assert_array_equal(got, p.rollout_cost(x0, U, eps))
    assert np.isfinite(got).all()


# =============================================================== A1: next() with injected noise
NEXT_CASES = [
    dict(K=128, H=32, a=1), dict(K=4096, H=64, a=2), dict(K=65536, H=64, a=3),
    dict(K=1000, H=50, a=3, mass=5.0), dict(K=70, H=5, a=2, lam=0.1), dict(K=3000, H=50, a=3, lam=10.0),
    dict(K=1, H=1, a=1), dict(K=2, H=3, a=2),              # degenerate: one sample gets weight 1
    dict(K=300, H=256, a=3), dict(K=100, H=500, a=4),      # long horizons: the R=32 / R=16 LDS-tile kernel
    dict(K=65537, H=16, a=3),                              # one sample past 1024 tiles: the 16:1 fold level
    dict(K=200001, H=16, a=3),                             # 3126 tiles: several rounds of workgroups, ragged last tile
]


@pytest.mark.parametrize("case", NEXT_CASES, ids=lambda c: "K%d_H%d_a%d" % (c["K"], c["H"], c["a"]))
def test_next_with_noise_matches_oracle(m, case):
    case = dict(case)
    K, H, a = case.pop("K"), case.pop("H"), case.pop("a")
    s = 2 * a
    h, p = make_pair(m, K, H, a, **case)
    rng = np.random.default_rng(7)
    x = np.zeros(s, F32)
    U_free = np.zeros((H, a), F32)  # the oracle running on its own warm start (drift check)
    n_steps = 3 if K * H <= 300000 else 2
    for step in range(n_steps):
        eps = (0.25 * rng.standard_normal((K, H, a))).astype(F32)
        U_in = h.get_action_sequence()          # identical state on both sides for the per-step bar
        u_gpu = h.next_with_noise(x, eps)
        u_ref, U_ref, c_ref = p.next_with_noise(x, U_in, eps)
        np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_ref)
        np.testing.assert_allclose(u_gpu, u_ref, rtol=0, atol=U_TOL)
        np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)
        assert not h.get_action_sequence()[-1].any()  # mInit0 zeros appended
        w = h.debug_get(m.DBG_WEIGHTS)
        assert abs(float(w.astype(np.float64).sum()) - 1.0) < 1e-5  # Σw = 1 (test_controller.cpp:166)
        assert h.debug_get(m.DBG_BETA) == c_ref.min()
        # free-running oracle (its own U) stays within a few U_TOL of the GPU over the closed loop
        u_free, U_free, _ = p.next_with_noise(x, U_free, eps)
        np.testing.assert_allclose(h.get_action_sequence(), U_free, rtol=0, atol=5 * U_TOL)
        # plant step on the host (same point mass), closed loop
        x = orc.model_step(p.A, p.B, x[None], u_gpu[None])[0]
    assert h.get_step_counter() == n_steps


def test_next_with_noise_normalize_cost(m):
    """Py normalizeCost=True (controller_base.py:468-474)."""
    K, H, a = 2000, 20, 3
    h, p = make_pair(m, K, H, a, normalize=True, lam=0.05)
    rng = np.random.default_rng(3)
    eps = (0.25 * rng.standard_normal((K, H, a))).astype(F32)
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    u_ref, U_ref, _ = p.next_with_noise(x, np.zeros((H, a), F32), eps, normalize=True)
    np.testing.assert_allclose(h.next_with_noise(x, eps), u_ref, rtol=0, atol=U_TOL)
    np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)


@pytest.mark.parametrize("K,H,a,lam", [(2000, 20, 3, 0.05), (65536, 64, 3, 1.0), (4096, 32, 2, 0.2), (300, 7, 1, 0.1), (512, 24, 4, 0.5),
                                       (140000, 8, 3, 0.3)])  # (2188 tiles: past 2048 the range comes from one k_cost_minmax launch, r04)
def test_normalize_cost_on_the_fused_path(m, K, H, a, lam):
    """normalizeCost=True on the fused Philox path (what scripts/main.py builds, controller_base.py:468-474). For the point-mass / diagonal-Q
    configuration this is TWO passes of the producer/consumer kernel: exp(-(c'-min c')/lambda) = exp(-(c-min c)/(lambda (max-min))), so the
    second pass makes the records of the RAW costs at the temperature the first pass's min / max define. Against the fp64 oracle's normalised
    update on the exported noise; the weights it reports are the normalised ones; a step without normalisation afterwards (another handle)
    is unaffected."""
    h, p = make_pair(m, K, H, a, normalize=True, lam=lam)
    p64 = orc.Problem(tau=H, s=2 * a, a=a, lam=lam, sigma=0.25 * np.eye(a), goal=(GOAL3 + [0.25, 0])[:2 * a], threads=0, dtype=np.float64)
    x = (np.array([0.1, 0, -0.2, 0, 0.3, 0, 0.05, 0])[:2 * a]).astype(F32)
    U_in = h.get_action_sequence()
    for step in range(2):
        u = h.next(x)
        eps = h.debug_get(m.DBG_NOISE)
        u_ref, U_ref, c_ref = p64.next_with_noise(x, U_in, eps, normalize=True)
        np.testing.assert_allclose(h.debug_get(m.DBG_COSTS), c_ref, rtol=2e-6)
        np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)
        np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)
        w = h.debug_get(m.DBG_WEIGHTS).astype(np.float64)
        cn = (c_ref - c_ref.min()) / (c_ref.max() - c_ref.min())
        w_ref = np.exp(-cn / lam)
        np.testing.assert_allclose(w, w_ref / w_ref.sum(), rtol=2e-4, atol=1e-9)
        U_in = h.get_action_sequence()


# =============================================================== A2: on-device noise
def test_device_noise_matches_oracle_restatement(m):
    """Philox4x32-10 counters are restated bit-exactly by the oracle; Box-Muller differs only through
    the device's fast sin/cos/log (rocRAND's __sincosf path) -> small absolute tolerance."""
    K, H, a = 1024, 16, 3
    sig = np.array([[0.5, 0.1, 0], [0, 0.25, 0], [0.2, 0, 1.0]], F32)
    h, _ = make_pair(m, K, H, a, sigma=sig, seed=1234567890123)
    x = np.zeros(6, F32)
    for step in range(2):
        h.next(x)
        got = h.debug_get(m.DBG_NOISE)
        ref = orc.noise(1234567890123, step, 0, K, H, a, sig)
        np.testing.assert_allclose(got, ref, rtol=0, atol=5e-6)
    z = got @ np.linalg.inv(sig).T
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02


def test_next_equals_next_with_its_own_noise(m):
    """The fused Philox step == the injected-noise step fed with the exported noise (regeneration
    consistency) == the oracle on that noise."""
    K, H, a = 4096, 64, 3
    h1, p = make_pair(m, K, H, a, seed=5)
    h2, _ = make_pair(m, K, H, a, seed=99)
    x = np.array([0.2, 0.1, -0.3, 0, 0.5, -0.1], F32)
    U = np.zeros((H, a), F32)
    for _ in range(3):
        u1 = h1.next(x)
        eps = h1.debug_get(m.DBG_NOISE)
        u2 = h2.next_with_noise(x, eps)
        np.testing.assert_array_equal(h1.debug_get(m.DBG_COSTS), h2.debug_get(m.DBG_COSTS))
        # the two steps run different kernels (producer/consumer vs LDS tile): same costs bit for bit,
        # weighted sums in a different (fixed) association
        np.testing.assert_allclose(u1, u2, rtol=0, atol=1e-6)
        np.testing.assert_allclose(h1.get_action_sequence(), h2.get_action_sequence(), rtol=0, atol=1e-6)
        h2.set_action_sequence(h1.get_action_sequence())
        u_ref, U, _ = p.next_with_noise(x, U, eps)
        np.testing.assert_allclose(u1, u_ref, rtol=0, atol=U_TOL)
        np.testing.assert_allclose(h1.get_action_sequence(), U, rtol=0, atol=U_TOL)


DENSE_SIGMA3 = np.array([[0.3, 0.05, 0.0], [0.02, 0.2, -0.04], [0.0, 0.03, 0.25]])


@pytest.mark.parametrize("K,H,a,sigma", [
    (65536, 64, 3, None), (1000, 50, 3, None), (4096, 64, 2, None), (128, 32, 1, None), (777, 100, 3, None),
    (300, 130, 4, None), (64, 3, 2, None),
    (40000, 20, 3, None),          # 513..1024 tiles: 3 producers, SIMD-true roles + progress priorities
    (70000, 16, 3, None),          # > 1024 tiles: several rounds, roles by wave index, no priorities
    (65600, 8, 3, None),           # 1025 tiles in 1152 record slots: every XCD class ends in a 16:1 fold group of neutral records only
    (90, 12, 2, None),             # 2 tiles in 128 record slots
    (5000, 24, 3, DENSE_SIGMA3),   # dense Σ: the non-diagonal instances
])
def test_producer_consumer_kernel_equals_tile_kernel(m, K, H, a, sigma):
    """k_rollout_pc (hot path) vs k_rollout_tile on the same Philox counters: costs bit-identical,
    update within rounding."""
    x = (0.1 * np.arange(2 * a)).astype(F32)
    ht, p = make_pair(m, K, H, a, seed=77, sigma=sigma, tuning={"force_tile_kernel": 1})
    hp, _ = make_pair(m, K, H, a, seed=77, sigma=sigma)
    for _ in range(2):
        ut, up = ht.next(x), hp.next(x)
        np.testing.assert_array_equal(hp.debug_get(m.DBG_COSTS), ht.debug_get(m.DBG_COSTS))
        np.testing.assert_allclose(up, ut, rtol=0, atol=1e-6)
        np.testing.assert_allclose(hp.get_action_sequence(), ht.get_action_sequence(), rtol=0, atol=1e-6)
        np.testing.assert_array_equal(hp.debug_get(m.DBG_NOISE), ht.debug_get(m.DBG_NOISE))
        ht.set_action_sequence(hp.get_action_sequence())
    # and against the oracle on the exported noise, from a fresh identical state
    h3, _ = make_pair(m, K, H, a, seed=77, sigma=sigma)
    u3 = h3.next(x)
    e3 = h3.debug_get(m.DBG_NOISE)
    u_ref, U_ref, c_ref = p.next_with_noise(x, np.zeros((H, a), F32), e3)
    np.testing.assert_array_equal(h3.debug_get(m.DBG_COSTS), c_ref)
    np.testing.assert_allclose(u3, u_ref, rtol=0, atol=U_TOL)
    np.testing.assert_allclose(h3.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)


def test_replay_is_deterministic(m):
    """Controller state = (U, Philox step): restoring both replays the same controls bit for bit."""
    K, H, a = 2048, 32, 2
    h, _ = make_pair(m, K, H, a, seed=11)
    x = np.array([0.0, 0.0, 0.3, 0.0], F32)
    h.next(x)
    U1, st = h.get_action_sequence(), h.get_step_counter()
    seq_a = [h.next(x) for _ in range(3)]
    h.set_action_sequence(U1)
    h.set_step_counter(st)
    seq_b = [h.next(x) for _ in range(3)]
    np.testing.assert_array_equal(seq_a, seq_b)
    h2, _ = make_pair(m, K, H, a, seed=12)
    assert not np.array_equal(h2.next(x), make_pair(m, K, H, a, seed=11)[0].next(x))


ELL = dict(a=1.5, b=0.8, cx=0.2, cy=-0.1, speed=0.7, m_state=2.0, m_vel=0.5)


@pytest.mark.parametrize("K,H,a,kind", [(65536, 64, 3, "dense"), (5000, 24, 3, "dense"), (4096, 64, 2, "dense"), (777, 9, 1, "dense"), (2048, 16, 4, "dense"),
                                        (65536, 64, 3, "ellipse"), (5000, 24, 2, "ellipse"), (2048, 16, 4, "ellipse")])
def test_producer_consumer_kernel_with_the_other_cost_forms(m, K, H, a, kind):
    """k_rollout_pc<..., COST>: the consumer's cost_base form is a template parameter (r03) — a dense Q (static_cost.py:23-63) and ElipseCost
    (elipse_cost.py:9-85) ride the hot kernel instead of the tile kernel (68 -> 25 us at C3). Costs bit-identical to the tile kernel's and
    to the fp32 CPU restatement's on the same Philox counters, the update within rounding; normalizeCost on top takes the two-pass form."""
    s = 2 * a
    rng = np.random.default_rng(K)
    x = (0.1 * np.arange(s)).astype(F32)
    if kind == "dense":
        B = rng.standard_normal((s, s))
        kw = dict(Q=(np.eye(s) + 0.05 * (B + B.T)).astype(F32), q_full=True)
        okw = dict(Q=kw["Q"])
    else:
        kw, okw = dict(ellipse=ELL), dict(ellipse=ELL)
    mk = lambda **t: m.Handle(k=K, tau=H, s_dim=s, a_dim=a, lam=1.0, sigma=0.25 * np.eye(a), goal=(GOAL3 + [0.25, 0])[:s], seed=5,
                              Q=kw.get("Q"), q_is_full=kw.get("q_full"), ellipse=kw.get("ellipse"), **t)
    hp, ht = mk(), mk(tuning={"force_tile_kernel": 1})
    assert hp.rollout_kernel_name().endswith(", 2, 0>" if kind == "dense" else ", 1, 0>") and mk(tuning=None).rollout_kernel_name() != "" and "k_rollout_tile" in ht.rollout_kernel_name()
    p32 = orc.Problem(tau=H, s=s, a=a, lam=1.0, sigma=0.25 * np.eye(a), goal=(GOAL3 + [0.25, 0])[:s], threads=0, **okw)
    for _ in range(2):
        U_in = hp.get_action_sequence()
        up, ut = hp.next(x), ht.next(x)
        cp = hp.debug_get(m.DBG_COSTS)
        np.testing.assert_array_equal(cp, ht.debug_get(m.DBG_COSTS))
        np.testing.assert_allclose(up, ut, rtol=0, atol=1e-6)
        np.testing.assert_allclose(hp.get_action_sequence(), ht.get_action_sequence(), rtol=0, atol=1e-6)
        eps = hp.debug_get(m.DBG_NOISE)
        u_ref, U_ref, c_ref = p32.next_with_noise(x, U_in, eps)
        np.testing.assert_array_equal(cp, c_ref)
        np.testing.assert_allclose(hp.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)
        ht.set_action_sequence(hp.get_action_sequence())
    hn = mk(normalize_cost=True)
    un = hn.next(x)
    _, Un_ref, _ = p32.next_with_noise(x, np.zeros((H, a), F32), hn.debug_get(m.DBG_NOISE), normalize=True)
    np.testing.assert_allclose(hn.get_action_sequence(), Un_ref, rtol=0, atol=2 * U_TOL)


# =============================================================== §8e: K-sharding records
@pytest.mark.parametrize("shards", [2, 8])
def test_sharded_records_combine_to_the_unsharded_step(m, shards):
    import torch
    K, H, a = 8192, 64, 3
    full, _ = make_pair(m, K, H, a, seed=21)
    x = np.array([0.2, 0.1, -0.3, 0, 0.5, -0.1], F32)
    xd = torch.tensor(x, device="cuda")
    hs = [make_pair(m, K, H, a, seed=21, shard_rank=g, shard_count=shards)[0] for g in range(shards)]
    assert sum(h.k_local for h in hs) == K and [h.k_offset for h in hs] == [g * K // shards for g in range(shards)]
    n = hs[0].record_size
    assert n == 2 + H * a
    recs = torch.zeros(shards * n, device="cuda")
    u = [torch.zeros(a, device="cuda") for _ in range(shards)]
    for step in range(2):
        u_full = full.next(x)
        for g, h in enumerate(hs):
            h.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
            h.synchronize()
        for g, h in enumerate(hs):
            h.shard_finish(recs.data_ptr(), shards, u[g].data_ptr())
            h.synchronize()
        for g, h in enumerate(hs):
            np.testing.assert_array_equal(u[g].cpu().numpy(), u[0].cpu().numpy())  # replicated, bit-identical
            np.testing.assert_allclose(u[g].cpu().numpy(), u_full, rtol=0, atol=2e-6)
            np.testing.assert_allclose(h.get_action_sequence(), full.get_action_sequence(), rtol=0, atol=2e-6)
        # shard costs are the corresponding slice of the unsharded costs (global-k Philox counters)
        if step == 0:  # later steps start from U's that differ by reduction-order rounding (<= 2e-6)
            c_full = full.debug_get(m.DBG_COSTS)
            for h in hs:
                np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_full[h.k_offset:h.k_offset + h.k_local])


def run_sharded_normalized(m, make, shards, x, steps=2):
    """The K-sharded normalizeCost step on ONE GPU: every shard reports its cost range, the ranges are reduced the way
    ShardedController reduces them (MAX over the {-min, max} pairs), every shard makes its record with the agreed range, the
    records are combined on every shard. Returns per step (u of every shard, the agreed range)."""
    import torch
    hs = [make(shard_rank=g, shard_count=shards) for g in range(shards)]
    n, a = hs[0].record_size, hs[0].a
    xd = torch.tensor(x, device="cuda")
    recs = torch.zeros(shards * n, device="cuda")
    rng = torch.zeros(shards, 2, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(shards)]
    out = []
    for step in range(steps):
        for g, h in enumerate(hs):
            h.shard_cost_range(xd.data_ptr(), rng[g].data_ptr())
            h.synchronize()
        agreed = rng.max(dim=0).values.contiguous()
        torch.cuda.synchronize()
        for g, h in enumerate(hs):
            h.shard_partial_normalized(xd.data_ptr(), agreed.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
            h.synchronize()
        for g, h in enumerate(hs):
            h.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
            h.synchronize()
        out.append(([u.cpu().numpy().copy() for u in us], agreed.cpu().numpy().copy()))
    return hs, out


@pytest.mark.parametrize("path", ["two_pass", "tile", "mlp32"])
@pytest.mark.parametrize("shards", [1, 3])
def test_sharded_normalize_cost_equals_the_unsharded_normalised_step(m, path, shards):
    """normalizeCost=True (controller_base.py:468-474) on a K-sharded controller: mppi_shard_cost_range -> the ranks agree on the
    global {min, max} -> mppi_shard_partial_normalized -> mppi_shard_finish. On the two-pass producer/consumer path, on the tile kernel
    (cost pass, normalise, record pass) and on a learned model. One shard holding everything and three shards (ragged: K is no
    multiple of 3) give mppi_next's controls to 2e-6 (the float rounding of the records), replicated bit-identically; the agreed range IS the range of the
    unsharded handle's costs."""
    K, H, a, lam = (5000, 24, 3, 0.3) if path != "mlp32" else (2048, 8, 3, 0.3)
    if path == "mlp32":
        mlp = make_mlp(6, 3, seed=4, hid=32, n_hidden=3)
        make = lambda **kw: make_mlp_pair(m, K, H, a, mlp, lam=lam, normalize_cost=True, seed=31, **kw)[0]
    else:
        tun = {"force_tile_kernel": 1} if path == "tile" else None
        make = lambda **kw: make_pair(m, K, H, a, lam=lam, normalize=True, seed=31, tuning=tun, **kw)[0]
    full = make()
    name = full.rollout_kernel_name()
    assert ("k_rollout_pc" in name) == (path == "two_pass"), name
    x = np.array([0.2, 0.1, -0.3, 0, 0.5, -0.1], F32)
    hs, out = run_sharded_normalized(m, make, shards, x)
    for step, (us, agreed) in enumerate(out):
        u_full = full.next(x)
        c = full.debug_get(m.DBG_COSTS)
        np.testing.assert_array_equal(agreed, np.array([-c.min(), c.max()], F32))
        for u in us[1:]:
            np.testing.assert_array_equal(u, us[0])
        np.testing.assert_allclose(us[0], u_full, rtol=0, atol=2e-6)
    for h in hs:
        np.testing.assert_allclose(h.get_action_sequence(), full.get_action_sequence(), rtol=0, atol=2e-6)
        assert h.get_step_counter() == 2
    # the plain phase 1 refuses a sharded normalising handle, and the normalising calls refuse a plain handle
    import torch
    if shards > 1:
        with pytest.raises(m.MppiError):
            hs[0].shard_partial(torch.zeros(6, device="cuda").data_ptr(), torch.zeros(hs[0].record_size, device="cuda").data_ptr())
    plain = make_pair(m, 256, 8, 3)[0]
    with pytest.raises(m.MppiError):
        plain.shard_cost_range(torch.zeros(6, device="cuda").data_ptr(), torch.zeros(2, device="cuda").data_ptr())


def test_sharded_normalize_cost_path_is_a_function_of_the_global_configuration(m):
    """with 132 < H <= 160 the producer/consumer kernel serves a handle only with 5 producer waves, and the producer
    count follows the SHARD's tile count (<= 512 tiles: 5, else 3). K = 65537 over two ranks gives shards of 512 and 513 tiles: before
    the fix rank 0 made raw-cost records at the range temperature (fast form) and rank 1 normalised-cost records at lambda. Both now
    take the cost pass + normalise + record pass form; the combined step is the unsharded normalised step."""
    K, H, a, lam = 65537, 140, 3, 0.4
    make = lambda **kw: make_pair(m, K, H, a, lam=lam, normalize=True, seed=33, **kw)[0]
    x = np.array([0.2, 0.1, -0.3, 0, 0.5, -0.1], F32)
    hs, out = run_sharded_normalized(m, make, 2, x)
    assert [(h.k_local + 63) // 64 for h in hs] == [512, 513]
    full = make()
    for us, agreed in out:
        u_full = full.next(x)
        c = full.debug_get(m.DBG_COSTS)
        np.testing.assert_array_equal(agreed, np.array([-c.min(), c.max()], F32))
        np.testing.assert_array_equal(us[1], us[0])
        np.testing.assert_allclose(us[0], u_full, rtol=0, atol=2e-6)
    for h in hs:
        np.testing.assert_allclose(h.get_action_sequence(), full.get_action_sequence(), rtol=0, atol=2e-6)


@pytest.mark.parametrize("normalize", [False, True])
@pytest.mark.parametrize("comm", ["none", "rccl"])
def test_shard_step_is_the_three_calls_in_one(m, normalize, comm):
    """mppi_shard_step: record -> the CALLER's all-gather -> finish in ONE C call, the collectives handed over as function
    pointers with ncclAllGather's / ncclAllReduce's signatures. Bit for bit the controls of mppi_shard_partial -> all-gather ->
    mppi_shard_finish — without a communicator (one shard, coll = NULL) and through RCCL's own entry points on a one-rank
    communicator created here (ncclCommInitRank via ctypes, the librccl.so torch holds); also for a normalize_cost handle, whose
    step has the second, 2-float all-reduce(MAX)."""
    import torch
    K, H, a, lam = 8192, 32, 3, 0.5
    make = lambda: make_pair(m, K, H, a, lam=lam, normalize=normalize, seed=17)[0]
    one, three = make(), make()
    coll = None
    if comm == "rccl":
        from mppi_tf_amd.rccl import RcclComm
        rc = RcclComm(0, 1)
        coll = rc.coll
    n = three.record_size
    rec, rng = torch.zeros(n, device="cuda"), torch.zeros(2, device="cuda")
    u1, u3 = torch.zeros(a, device="cuda"), torch.zeros(a, device="cuda")
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1], device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    for step in range(4):
        one.shard_step(x.data_ptr(), u1.data_ptr(), coll, st)
        if normalize:
            three.shard_cost_range(x.data_ptr(), rng.data_ptr(), st)
            three.shard_partial_normalized(x.data_ptr(), rng.data_ptr(), rec.data_ptr(), st)
        else:
            three.shard_partial(x.data_ptr(), rec.data_ptr(), st)
        three.shard_finish(rec.data_ptr(), 1, u3.data_ptr(), st)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(u1.cpu().numpy(), u3.cpu().numpy())
    np.testing.assert_array_equal(one.get_action_sequence(), three.get_action_sequence())
    assert one.get_step_counter() == three.get_step_counter() == 4
    # a sharded handle without an all-gather, and a failing collective: refused / reported, never a silent wrong step
    sh = make_pair(m, K, H, a, seed=17, shard_rank=0, shard_count=2)[0]
    with pytest.raises(m.MppiError):
        sh.shard_step(x.data_ptr(), u1.data_ptr(), None, st)
    import ctypes as C
    from mppi_tf_amd._lib import Collectives
    FAIL = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_int, C.c_void_p, C.c_void_p)(lambda *a_: 5)
    bad = Collectives(all_gather=C.cast(FAIL, C.c_void_p).value, all_reduce=None, comm=None)
    plain = make_pair(m, K, H, a, seed=17)[0]
    U0, s0 = plain.get_action_sequence(), plain.get_step_counter()
    with pytest.raises(m.MppiError) as ei:
        plain.shard_step(x.data_ptr(), u1.data_ptr(), bad, st)
    assert ei.value.status == 8 and "5" in str(ei.value)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(plain.get_action_sequence(), U0)  # nothing of the update was enqueued
    assert plain.get_step_counter() == s0


def test_sharded_controller_takes_the_one_call_path_with_its_own_communicator(m, monkeypatch):
    """ShardedController on the collective path = ONE C call per step (mppi_shard_step -> ncclAllGather on the controller's own
    communicator); MPPI_RCCL_CALL=torch keeps the three calls + torch.distributed. Same controls either way."""
    import torch
    from mppi_tf_amd.distributed import ShardedController
    monkeypatch.setenv("MPPI_FORCE_EXCHANGE", "1")
    monkeypatch.delenv("MPPI_EXCHANGE", raising=False)
    cfg = dict(k=4096, tau=32, s_dim=6, a_dim=3, sigma=0.25 * np.eye(3), goal=GOAL3, seed=11, lam=0.5)
    fast = ShardedController(exchange="rccl", **cfg)
    assert fast.rccl is not None and "one C call" in fast.rccl_note
    monkeypatch.setenv("MPPI_RCCL_CALL", "torch")
    slow = ShardedController(exchange="rccl", **cfg)
    assert slow.rccl is None
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1], device="cuda")
    for _ in range(3):
        ua, ub = fast.next(x), slow.next(x)
        torch.cuda.synchronize()
        np.testing.assert_array_equal(ua.cpu().numpy(), ub.cpu().numpy())


def test_sharded_controller_with_normalize_cost(m, monkeypatch):
    """ShardedController(normalize_cost=True) on the real backend, exchange forced on one rank: the range / record / finish sequence
    gives the unsharded handle's controls (2e-6); the direct exchange is not brought up for it (it carries the records only)."""
    import torch
    from mppi_tf_amd.distributed import ShardedController
    monkeypatch.setenv("MPPI_FORCE_EXCHANGE", "1")
    monkeypatch.delenv("MPPI_EXCHANGE", raising=False)
    cfg = dict(k=4096, tau=32, s_dim=6, a_dim=3, sigma=0.25 * np.eye(3), goal=GOAL3, seed=11, lam=0.5, normalize_cost=True)
    ctl = ShardedController(**cfg)
    assert ctl.exchange == "rccl" and ctl.normalize
    ref = m.Handle(**cfg)
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1], device="cuda")
    for _ in range(3):
        u = ctl.next(x)
        torch.cuda.synchronize()
        np.testing.assert_allclose(u.cpu().numpy(), ref.next(x.cpu().numpy()), rtol=0, atol=2e-6)
    with pytest.raises(RuntimeError):
        ShardedController(exchange="p2p", **cfg)


# =============================================================== full-size properties (C3)
def test_full_size_properties_point_mass3d(m):
    K, H, a = 65536, 64, 3
    h, p = make_pair(m, K, H, a, seed=1)
    x = np.zeros(6, F32)
    u = h.next(x)
    w = h.debug_get(m.DBG_WEIGHTS).astype(np.float64)
    assert abs(w.sum() - 1) < 1e-5 and (w >= 0).all()
    eps = h.debug_get(m.DBG_NOISE)
    Uupd = h.debug_get(m.DBG_U_UPDATED)
    # U' - U is a convex combination of the noise rows; recompute it in fp64 from the exported pieces
    wn = np.tensordot(w, eps.astype(np.float64), axes=(0, 0))
    np.testing.assert_allclose(Uupd, wn, rtol=0, atol=U_TOL)  # U was 0
    np.testing.assert_allclose(u, wn[0], rtol=0, atol=U_TOL)
    assert (np.abs(Uupd) <= np.abs(eps).max(0) + 1e-6).all()
    np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), p.rollout_cost(x, np.zeros((H, a), F32), eps))
    # closed loop drives the plant toward the goal
    goal = np.asarray(GOAL3, F32)
    d0 = np.linalg.norm(x - goal)
    for _ in range(60):
        x = orc.model_step(p.A, p.B, x[None], u[None])[0]
        u = h.next(x)
    assert np.linalg.norm(x - goal) < 0.5 * d0


# =============================================================== the C++ constructor / host loop
def test_cpp_constructor_defaults_and_host_loop(m, tmp_path):
    """ControllerBase(k,tau,dt,mass,s_dim,a_dim) defaults (controller_base.cpp:37-69) + the host loop of
    src/main.cpp:36-45 (next / saveNext / toCSV) + setGoal's size check (:126-133)."""
    c = m.ControllerBaseCpp(256, 16, 0.1, 1.0, 2, 1)
    p = orc.Problem(tau=16, s=2, a=1, dt=0.1, mass=1.0)  # oracle defaults are the same ctor defaults
    assert c.setGoal([1.0, 0.0, 3.0]) is False
    x = np.zeros(2, F32)
    U = np.zeros((16, 1), F32)
    for _ in range(4):
        u = c.next(x.tolist())
        assert isinstance(u, list) and len(u) == 1
        eps = c._h.debug_get(m.DBG_NOISE)
        u_ref, U, _ = p.next_with_noise(x, U, eps)
        np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)
        x = orc.model_step(p.A, p.B, x[None], np.asarray(u, F32)[None])[0]
        c.saveNext(x.tolist())
    f = tmp_path / "data.csv"
    c.toCSV(str(f))
    rows = f.read_text().strip().splitlines()
    # DataBase::toCSV's bytes (data_base.cpp:36-71): every cell followed by a comma, values as std::to_string(float)
    assert rows[0] == "x0,x1,u0,x_next0,x_next1," and len(rows) == 5
    import re
    assert all(re.fullmatch(r"(-?\d+\.\d{6},){5}", r) for r in rows[1:]), rows[1:]
    assert c.setGoal([0.5, 0.0]) is True
    c.next(x.tolist())
    assert abs(c._h.debug_get(m.DBG_COSTS).min()) >= 0


def test_error_behaviour(m):
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=0, tau=4, s_dim=2, a_dim=1)
    assert e.value.status == 1
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=8, tau=4, s_dim=2, a_dim=1, sigma=[[0.0]])
    assert e.value.status == 5
    h = m.Handle(k=8, tau=4, s_dim=2, a_dim=1)
    with pytest.raises(m.MppiError) as e:
        h.next([0.0, 0.0, 0.0])
    assert e.value.status == 1
    with pytest.raises(m.MppiError) as e:
        h.next_with_noise([0.0, 0.0], np.zeros(5))
    assert e.value.status == 1
    hc = m.Handle(k=8, tau=4, s_dim=4, a_dim=3)  # a cost-only shape (test_cost.cpp scenario 3)
    with pytest.raises(m.MppiError) as e:
        hc.next(np.zeros(4))
    assert e.value.status == 4 and "2*a_dim" in str(e.value)


# =============================================================== the C++ host side over the C-ABI
def test_cpp_facade_reference_vectors_and_host_loop(m):
    """include/mppi/*.hpp (ControllerBase/CostBase/ModelBase re-created over the C-ABI) against the
    reference's gtest vectors, and the reference-shaped host loop (src/main.cpp:30-64), as native binaries."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "tests", "cpp", "test_reference_vectors")
    loop = os.path.join(ROOT, "examples", "host_loop")
    assert os.path.exists(exe) and os.path.exists(loop), "run __graft_entry__.build() first"
    r = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all reference vectors pass" in r.stdout
    r = subprocess.run([loop, "4096", "32", "2", "80"], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "Execution time" in r.stdout
    # the same loop with armed launches (r05): same closed-loop result line — the controls are bit-identical — on the fused (<= 128 tiles) and the
    # two-kernel (rollout + finish) armed form; a device without a large BAR answers MPPI_ERR_UNSUPPORTED (exit code 4), which is not a failure of the loop
    for shape in (["4096", "32", "2", "80"], ["16384", "32", "3", "60"]):
        plain = subprocess.run([loop] + shape, capture_output=True, text=True, timeout=120)
        armed = subprocess.run([loop] + shape + ["-", "armed=2000"], capture_output=True, text=True, timeout=120)
        if armed.returncode == 4 and "large-BAR" in armed.stderr:
            continue
        assert plain.returncode == 0 and armed.returncode == 0, armed.stdout + armed.stderr
        assert plain.stdout.splitlines()[0] == armed.stdout.splitlines()[0], (plain.stdout, armed.stdout)


# =============================================================== §8f row 4 (first slice): the elliptic cost in the cost_base slot
def test_golden_elipse_cost_on_device(m):
    """ElipseCost.state_cost (costs/elipse_cost.py:48-85) through the device cost slot: TestElipseCost's literals
    (scripts/test.py:1098-1161) at the reference's tolerance, and bit-identical to the fp32 oracle."""
    g = load_golden("cost_elipse")
    e = g["ellipse"]
    cost = m.ElipseCost(1.0, 1.0, 1.0, np.eye(2), e["a"], e["b"], e["cx"], e["cy"], e["speed"], e["m_state"], e["m_vel"])
    p = orc.Problem(tau=1, s=4, a=2, ellipse=e)
    for sc in g["scenarios"]:
        st = np.asarray(sc["state"], F32)[..., None]
        got = cost.state_cost("", st)
        assert got.shape == (len(sc["state"]), 1, 1)
        np.testing.assert_allclose(got.ravel(), sc["exp_state_cost"], rtol=1e-6, atol=1e-6)
        np.testing.assert_array_equal(got.ravel(), p.state_cost(sc["state"]))
    rng = np.random.default_rng(0)
    X = (3 * rng.standard_normal((4096, 4))).astype(F32)
    p2 = orc.Problem(tau=1, s=4, a=2, ellipse=dict(a=4., b=2., cx=0.3, cy=-0.2, speed=5., m_state=1., m_vel=0.1))
    c2 = m.ElipseCost(1.0, 1.0, 1.0, np.eye(2), 4., 2., 0.3, -0.2, 5., 1., 0.1)
    np.testing.assert_array_equal(c2.state_cost("", X[..., None]).ravel(), p2.state_cost(X))  # correctly rounded / and sqrt
    with pytest.raises(AssertionError):
        cost.state_cost("", np.zeros((3, 6, 1)))
    with pytest.raises(m.MppiError):
        m.Handle(k=8, tau=2, s_dim=2, a_dim=1, ellipse=e)  # the cost reads (x, vx, y, vy)


@pytest.mark.parametrize("K,H", [(4096, 32), (1000, 17)])
def test_control_step_with_the_elipse_cost_matches_oracle(m, K, H):
    """The whole control step of a point_mass2d controller with the elliptic cost (general tile kernel): sample costs
    bit-identical to the oracle, update within 1e-5, injected and Philox noise; then the reference-shaped entry point
    (examples/main.py --task elipse_task.yaml) drives the plant onto the ellipse at the target speed."""
    ell = dict(a=2., b=1., cx=0., cy=0., speed=1., m_state=1., m_vel=0.1)
    cfg = dict(tau=H, dt=0.1, mass=1.0, lam=1.0, sigma=np.eye(2))
    h = m.Handle(k=K, s_dim=4, a_dim=2, ellipse=ell, seed=2, **cfg)
    p = orc.Problem(s=4, a=2, ellipse=ell, threads=0, **cfg)
    x = np.array([1.5, 0.0, 0.2, 0.4], F32)
    rng = np.random.default_rng(1)
    for step in range(2):
        U_in = h.get_action_sequence()  # the oracle starts every step from the handle's own sequence
        eps = rng.standard_normal((K, H, 2)).astype(F32)
        u = h.next_with_noise(x, eps)
        u_ref, U_ref, c_ref = p.next_with_noise(x, U_in, eps)
        np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_ref)
        np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)
        np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)
    U_in = h.get_action_sequence()
    u = h.next(x)
    eps = h.debug_get(m.DBG_NOISE)
    u_ref, U2, c_ref = p.next_with_noise(x, U_in, eps)
    np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_ref)
    np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)


def test_entry_point_follows_the_elipse(m):
    import os
    import subprocess
    import sys
    from conftest import ROOT
    r = subprocess.run([sys.executable, os.path.join(ROOT, "examples", "main.py"), "--new",
                        "--config", os.path.join(ROOT, "examples", "config", "point_mass2d.yaml"),
                        "--task", os.path.join(ROOT, "examples", "config", "elipse_task.yaml"), "-s", "150"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    xd = float(r.stdout.split("x_dist =")[1].split()[0])
    vd = float(r.stdout.split("v_dist =")[1].split()[0])
    assert abs(xd) < 0.3 and vd < 0.7, r.stdout  # on the ellipse (x_dist ~ 0) and moving along it (speed within 0.7 of the target 1)


# =============================================================== M2: learned 2x256 MLP model_base (parity unpinned by the reference)
def make_mlp(s, a, seed=0, hid=256, n_hidden=2):
    """SURVEY §8d synthetic weights: U(-1/sqrt(fan_in), 1/sqrt(fan_in)), last layer x0.1. hid=32, n_hidden=3 is the
    reference's own network shape (nn_model.py:54-60)."""
    rng = np.random.default_rng(seed)
    dims = [s + a] + [hid] * n_hidden + [s]
    W = [rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i]) for i in range(n_hidden + 1)]
    b = [rng.uniform(-1, 1, dims[i + 1]) / np.sqrt(dims[i]) for i in range(n_hidden + 1)]
    W[-1] *= 0.1
    b[-1] *= 0.1
    return dict(W=[w.astype(F32) for w in W], b=[v.astype(F32) for v in b],
                xmean=rng.uniform(-0.1, 0.1, s + a).astype(F32), xstd=rng.uniform(0.8, 1.2, s + a).astype(F32),
                ymean=rng.uniform(-0.01, 0.01, s).astype(F32), ystd=rng.uniform(0.8, 1.2, s).astype(F32))


def make_mlp_pair(m, K, H, a, mlp, lam=1.0, **kw):
    s = 2 * a
    sigma, goal = 0.25 * np.eye(a), (GOAL3 + [0.25, 0])[:s]
    h = m.Handle(k=K, tau=H, s_dim=s, a_dim=a, lam=lam, sigma=sigma, goal=goal, mlp=mlp, **kw)
    p32 = orc.Problem(tau=H, s=s, a=a, lam=lam, sigma=sigma, goal=goal, mlp=mlp, threads=0)
    p64 = orc.Problem(tau=H, s=s, a=a, lam=lam, sigma=sigma, goal=goal, mlp=mlp, threads=0, dtype=np.float64)
    return h, p32, p64


def test_mlp_single_step_reference_order_is_bit_exact(m):
    """mppi_model_step on an MLP handle evaluates the layers in the oracle's order (mul, add rounded
    separately, ascending input index): bit-identical."""
    a, s = 3, 6
    mlp = make_mlp(s, a, seed=3)
    h, p32, _ = make_mlp_pair(m, 64, 4, a, mlp)
    rng = np.random.default_rng(0)
    X, V = rng.standard_normal((50, s)).astype(F32), rng.standard_normal((50, a)).astype(F32)
    got = h.model_next(X, V)
    ref = np.stack([p32.mlp_step(X[i], V[i]) for i in range(50)])
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("bx3", [False, True], ids=["fp32mfma", "bf16x3"])
@pytest.mark.parametrize("K,H,a", [(2048, 64, 3), (100, 16, 3), (33, 5, 2), (4096, 32, 1), (512, 24, 4)])
def test_mlp_rollout_costs_within_fp32_of_truth(m, K, H, a, bx3):
    """MFMA rollouts (fp32: fmaf chains, k-ordered; bf16x3: MPPI_FLAG_MLP_BF16X3, three split-bf16 products per term)
    vs the fp64 oracle ('truth') and the fp32 oracle (unfused): the GPU must be as close to the truth as the fp32
    CPU evaluation is, up to a small factor (4x for the exact-fp32 kernel, 8x for the split-bf16 one)."""
    s = 2 * a
    mlp = make_mlp(s, a, seed=K)
    h, p32, p64 = make_mlp_pair(m, K, H, a, mlp, mlp_bf16x3=bx3)
    rng = np.random.default_rng(5)
    x0 = (0.2 * rng.standard_normal(s)).astype(F32)
    U = (0.1 * rng.standard_normal((H, a))).astype(F32)
    eps = (0.25 * rng.standard_normal((K, H, a))).astype(F32)
    got = h.rollout_cost(x0, U, eps).astype(np.float64)
    truth = p64.rollout_cost(x0, U, eps)
    cpu32 = p32.rollout_cost(x0, U, eps).astype(np.float64)
    err_gpu = np.abs(got - truth) / np.abs(truth)
    err_cpu = np.abs(cpu32 - truth) / np.abs(truth)
    print("max relative cost error: GPU %.3g, fp32 CPU %.3g" % (err_gpu.max(), err_cpu.max()))
    assert err_gpu.max() < 2e-5, err_gpu.max()
    assert err_gpu.max() < (8 if bx3 else 4) * max(err_cpu.max(), 1e-6), (err_gpu.max(), err_cpu.max())


# The two MLP rollout kernels: k_rollout_mlp (exact-fp32 MFMA, the default) and k_rollout_mlp_bx3
# (MPPI_FLAG_MLP_BF16X3). Bars on the control update U' (absolute, against the fp64
# oracle on identical noise): north_star's 1e-5 for the exact-fp32 kernel; 2e-5 for the split-bf16 kernel (its
# operands carry 16 mantissa bits: measured 2-5x the fp32 kernel's error). Both are ALWAYS held to a small multiple of
# the error the fp32 CPU oracle itself makes against fp64 (the reference computes in fp32 too): no worse than 4x / 8x.
# The absolute bar applies where the problem is conditioned for it, i.e. where the fp32 CPU evaluation itself lands
# within a quarter of the bar of fp64: the soft-min turns a cost error dc into a relative weight error dc/lambda, so
# with few samples, a long horizon (large costs) and lambda = 1 ANY two fp32 evaluations of the reference's graph
# differ by more than 1e-5 (the K=512, H=128 case below: fp32 CPU 1.3e-5, this kernel 1.3e-5) — "within 1e-5 of the
# reference" is then not defined by the reference either.
MLP_VARIANTS = [("fp32mfma", {}, 1e-5, 4.0), ("bf16x3", dict(mlp_bf16x3=True), 2e-5, 8.0)]
MLP_IDS = [v[0] for v in MLP_VARIANTS]
_oracle_cache = {}


def mlp_step_case(K, H, a, seed, lam=1.0, hid=256, n_hidden=2):
    """Inputs and the CPU oracles' answers (fp64 = truth, fp32 = what an fp32 CPU evaluation gives) of ONE control
    step with injected noise; cached so the three kernel variants share one oracle evaluation."""
    key = (K, H, a, seed, lam, hid, n_hidden)
    if key not in _oracle_cache:
        s = 2 * a
        mlp = make_mlp(s, a, seed=seed, hid=hid, n_hidden=n_hidden)
        sigma, goal = 0.25 * np.eye(a), (GOAL3 + [0.25, 0])[:s]
        kw = dict(tau=H, s=s, a=a, lam=lam, sigma=sigma, goal=goal, mlp=mlp, threads=0)
        p32, p64 = orc.Problem(**kw), orc.Problem(dtype=np.float64, **kw)
        rng = np.random.default_rng(seed + 1)
        x = (0.2 * rng.standard_normal(s)).astype(F32)
        U = (0.1 * rng.standard_normal((H, a))).astype(F32)
        eps = (0.25 * rng.standard_normal((K, H, a))).astype(F32)
        u64, U64, c64 = p64.next_with_noise(x, U, eps)
        u32, U32, c32 = p32.next_with_noise(x, U, eps)
        _oracle_cache.clear()  # one case at a time: eps is 50 MB at K=65536
        _oracle_cache[key] = dict(mlp=mlp, sigma=sigma, goal=goal, x=x, U=U, eps=eps, U64=np.asarray(U64, np.float64),
                                  c64=np.asarray(c64, np.float64), U32=np.asarray(U32, np.float64), c32=np.asarray(c32, np.float64))
    return _oracle_cache[key]


def check_mlp_step(m, K, H, a, seed, kw, u_bar, factor, lam=1.0, well_conditioned=True, hid=256, n_hidden=2):
    cs = mlp_step_case(K, H, a, seed, lam, hid, n_hidden)
    h = m.Handle(k=K, tau=H, s_dim=2 * a, a_dim=a, lam=lam, sigma=cs["sigma"], goal=cs["goal"], mlp=cs["mlp"], **kw)
    h.set_action_sequence(cs["U"])
    u = h.next_with_noise(cs["x"], cs["eps"])
    Uupd = h.debug_get(m.DBG_U_UPDATED).astype(np.float64)
    # the oracle returns the SHIFTED sequence: compare the shifted one the handle keeps (and u = U'[0] below)
    Unext = h.get_action_sequence().astype(np.float64)
    c = h.debug_get(m.DBG_COSTS).astype(np.float64)
    rel = lambda got: float((np.abs(got - cs["c64"]) / np.abs(cs["c64"])).max())
    ec_gpu, ec_cpu = rel(c), rel(cs["c32"])
    eu_gpu = float(np.abs(Unext - cs["U64"]).max())
    eu_cpu = float(np.abs(cs["U32"] - cs["U64"]).max())
    print("K=%d H=%d a=%d: max rel cost error GPU %.3g / fp32 CPU %.3g;  max|dU'| GPU %.3g / fp32 CPU %.3g (bar %.0e)"
          % (K, H, a, ec_gpu, ec_cpu, eu_gpu, eu_cpu, u_bar))
    assert ec_gpu < 2e-5 and ec_gpu < factor * max(ec_cpu, 1e-6), (ec_gpu, ec_cpu)
    assert eu_gpu <= factor * max(eu_cpu, 5e-7), (eu_gpu, eu_cpu)
    if eu_cpu <= u_bar / 4 or well_conditioned:
        assert eu_gpu <= u_bar, eu_gpu
    assert np.abs(u.astype(np.float64) - Uupd[0]).max() == 0  # u is U'[0]
    w = h.debug_get(m.DBG_WEIGHTS).astype(np.float64)
    assert abs(w.sum() - 1) < 1e-5
    return h


@pytest.mark.parametrize("name,kw,u_bar,factor", MLP_VARIANTS, ids=MLP_IDS)
def test_mlp_next_matches_oracle(m, name, kw, u_bar, factor):
    """One control step of the MLP controller on injected noise: U' within north_star's 1e-5 of the fp64 oracle for
    the exact-fp32 kernel (measured value printed), then the fused Philox path against the oracle on its own
    exported noise."""
    K, H, a = 4096, 32, 3
    h = check_mlp_step(m, K, H, a, 9, kw, u_bar, factor)
    cs = mlp_step_case(K, H, a, 9)
    p64 = orc.Problem(tau=H, s=2 * a, a=a, lam=1.0, sigma=cs["sigma"], goal=cs["goal"], mlp=cs["mlp"], threads=0, dtype=np.float64)
    U_in = h.get_action_sequence()
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    u = h.next(x)
    eps = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(eps, orc.noise(1, 1, 0, K, H, a, 0.25 * np.eye(a)), rtol=0, atol=5e-6)
    u_ref, U_ref, c_ref = p64.next_with_noise(x, U_in, eps)
    np.testing.assert_allclose(h.debug_get(m.DBG_COSTS), c_ref, rtol=2e-5)
    print("fused path: max|dU'| = %.3g" % np.abs(h.get_action_sequence() - U_ref).max())
    np.testing.assert_allclose(u, u_ref, rtol=0, atol=u_bar)
    np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=u_bar)


@pytest.mark.parametrize("a,dense", [(1, False), (2, False), (4, False), (3, True), (2, True)])
@pytest.mark.parametrize("hid,n_hidden", [(256, 2), (32, 3)], ids=["2x256", "32x3"])
def test_split_bf16_fused_philox_step_every_action_dim(m, a, dense, hid, n_hidden):
    """The split-bf16 kernels (k_rollout_mlp_bx3, k_rollout_mlp32_bx3) on the fused Philox path for the action dimensions and the dense
    Sigma that the a = 3 / diagonal tests above do not reach: the exported noise is the CPU restatement's Philox stream, costs within
    2e-5 and U' within the split-bf16 bar of the fp64 evaluation on that noise; ragged K."""
    K, H, s = 2000, 13, 2 * a
    mlp = make_mlp(s, a, seed=40 + a, hid=hid, n_hidden=n_hidden)
    rng = np.random.default_rng(a)
    sigma = 0.25 * np.eye(a)
    if dense:
        L = 0.25 * np.eye(a) + 0.05 * np.tril(rng.standard_normal((a, a)), -1)
        sigma = (L + L.T) / 2 + 0.1 * np.eye(a)
    goal = (GOAL3 + [0.25, 0])[:s]
    h = m.Handle(k=K, tau=H, s_dim=s, a_dim=a, lam=1.0, sigma=sigma, goal=goal, mlp=mlp, mlp_bf16x3=True, seed=3)
    assert ("k_rollout_mlp_bx3" if hid == 256 else "k_rollout_mlp32_bx3") in h.rollout_kernel_name()
    p64 = orc.Problem(tau=H, s=s, a=a, lam=1.0, sigma=sigma, goal=goal, mlp=mlp, threads=0, dtype=np.float64)
    U = (0.1 * rng.standard_normal((H, a))).astype(F32)
    h.set_action_sequence(U)
    x = (0.2 * rng.standard_normal(s)).astype(F32)
    u = h.next(x)
    eps = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(eps, orc.noise(3, 0, 0, K, H, a, sigma), rtol=0, atol=5e-6)
    u_ref, U_ref, c_ref = p64.next_with_noise(x, U, eps)
    np.testing.assert_allclose(h.debug_get(m.DBG_COSTS), c_ref, rtol=2e-5)
    np.testing.assert_allclose(u, u_ref, rtol=0, atol=2e-5)
    np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=2e-5)


@pytest.mark.parametrize("name,kw,u_bar,factor", MLP_VARIANTS, ids=MLP_IDS)
@pytest.mark.parametrize("K,H,lam,cond", [(65536, 8, 1.0, True), (8192, 128, 1.0, True), (512, 128, 1.0, False), (1000, 130, 8.0, True)],
                         ids=["K65536_H8", "K8192_H128", "K512_H128_illconditioned", "K1000_H130_lam8"])
def test_mlp_baseline_shapes_against_oracle(m, K, H, lam, cond, name, kw, u_bar, factor):
    """The BASELINE shapes' two long axes, each against the CPU oracle: K = 65536 (configs[3]: 1024 workgroups = 4
    rounds per CU, a 1024-record finish) at a horizon the oracle finishes in
    seconds, and H = 128 (configs[4]) / a ragged 130. The absolute 1e-5 bar holds wherever fp32 itself can hold it; the
    K=512, H=128, lambda=1 case documents where it cannot (see MLP_VARIANTS) and is held to the fp32 CPU's own error."""
    check_mlp_step(m, K, H, 3, 100 + H, kw, u_bar, factor, lam=lam, well_conditioned=cond)


# ---- the reference's own network shapes (nn_model.py:54-60: Dense(32, relu) x3 + Dense(s)): served by
# k_rollout_mlp_small (one rollout per lane, weights through the scalar cache into v_pk_fma_f32)
SMALL_NETS = [(32, 3), (16, 3), (32, 1), (16, 2)]
SMALL_IDS = ["32x3", "16x3", "32x1", "16x2"]


@pytest.mark.parametrize("hid,n_hidden", SMALL_NETS, ids=SMALL_IDS)
def test_small_mlp_single_step_reference_order_is_bit_exact(m, hid, n_hidden):
    """mppi_model_step with the reference's widths: the oracle's order, bit-identical."""
    a, s = 3, 6
    mlp = make_mlp(s, a, seed=11, hid=hid, n_hidden=n_hidden)
    h, p32, _ = make_mlp_pair(m, 64, 4, a, mlp)
    assert h.rollout_kernel_name().startswith("mppi::k_rollout_mlp32_pc<3, " if hid == 32 else "mppi::k_rollout_mlp_small<3, %d>" % hid)
    rng = np.random.default_rng(0)
    X, V = rng.standard_normal((50, s)).astype(F32), rng.standard_normal((50, a)).astype(F32)
    np.testing.assert_array_equal(h.model_next(X, V), np.stack([p32.mlp_step(X[i], V[i]) for i in range(50)]))


# (hidden width, hidden layers, Handle tuning): Dense(32) runs on the matrix cores — the two-wave pipeline k_rollout_mlp32_pc by default (r04),
# the one-wave-per-32-rollouts kernel k_rollout_mlp32 with mlp32_valu = 2 — and on the vector ALU (k_rollout_mlp_small) with mlp32_valu = 1;
# Dense(16) on the vector ALU
SMALL_KERNELS = [(32, 3, None), (32, 3, {"mlp32_valu": 1}), (16, 3, None), (32, 1, None), (32, 2, None), (16, 2, None), (32, 1, {"mlp32_valu": 1}),
                 (32, 3, "bf16x3"), (32, 2, "bf16x3"), (32, 1, "bf16x3"),  # "bf16x3": k_rollout_mlp32_bx3 (MPPI_FLAG_MLP_BF16X3)
                 (32, 3, {"mlp32_valu": 2}), (32, 2, {"mlp32_valu": 2}), (32, 1, {"mlp32_valu": 2})]
SMALL_KERNEL_IDS = ["32x3-mfma", "32x3-valu", "16x3", "32x1-mfma", "32x2-mfma", "16x2", "32x1-valu", "32x3-bf16x3", "32x2-bf16x3", "32x1-bf16x3",
                    "32x3-mfma1w", "32x2-mfma1w", "32x1-mfma1w"]


@pytest.mark.parametrize("hid,n_hidden,tuning", SMALL_KERNELS, ids=SMALL_KERNEL_IDS)
@pytest.mark.parametrize("K,H,a,cond", [(4096, 32, 3, True), (100, 16, 3, True), (33, 5, 2, True), (1000, 130, 1, False), (512, 24, 4, True)])
def test_small_mlp_control_step_against_oracle(m, K, H, a, cond, hid, n_hidden, tuning):
    """One control step on injected noise: sample costs as close to the fp64 oracle as an fp32 CPU evaluation is (4x),
    U' within north_star's 1e-5 wherever the fp32 CPU evaluation itself is (see MLP_VARIANTS: K=1000, H=130, a=1 at
    lambda=1 is not — fp32 CPU 2.2e-5, this kernel 1.6e-5); ragged K (33, 100, 1000: partial last tile) and a horizon
    that is no multiple of 4. `cond`: the shapes that ARE well conditioned carry the absolute 1e-5 bar unconditionally."""
    if tuning == "bf16x3":  # the split-bf16 kernel carries the 2x256 split-bf16 kernel's bars (MLP_VARIANTS)
        h = check_mlp_step(m, K, H, a, 300 + H, dict(mlp_bf16x3=True), 2e-5, 8.0, hid=hid, n_hidden=n_hidden, well_conditioned=cond)
        assert "k_rollout_mlp32_bx3" in h.rollout_kernel_name()
        return
    check_mlp_step(m, K, H, a, 300 + H, dict(tuning=tuning) if tuning else {}, 1e-5, 4.0, hid=hid, n_hidden=n_hidden, well_conditioned=cond)


@pytest.mark.parametrize("bx3", [False, True], ids=["fp32mfma", "bf16x3"])
def test_small_mlp_fused_philox_step_and_sharding(m, bx3):
    """The reference's {32,32,32,s} network on the fused Philox path: U' against the fp64 oracle on the noise the step
    exported; the device noise is the oracle's Philox stream; and the 4-way K-sharded step reproduces the unsharded one."""
    import torch
    K, H, a = 8192, 20, 3
    mlp = make_mlp(6, a, seed=5, hid=32, n_hidden=3)
    sigma = 0.25 * np.eye(a)
    cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, seed=7, mlp_bf16x3=bx3)
    h = m.Handle(**cfg)
    p64 = orc.Problem(tau=H, s=6, a=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, threads=0, dtype=np.float64)
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    U_in = h.get_action_sequence()
    u = h.next(x)
    eps = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(eps, orc.noise(7, 0, 0, K, H, a, sigma), rtol=0, atol=5e-6)
    u64, U64, c64 = p64.next_with_noise(x, U_in, eps)
    bar = 2e-5 if bx3 else 1e-5  # MLP_VARIANTS' bars
    assert np.abs(h.get_action_sequence().astype(np.float64) - np.asarray(U64)).max() <= bar
    assert np.abs(u.astype(np.float64) - np.asarray(u64)).max() <= bar
    c = h.debug_get(m.DBG_COSTS).astype(np.float64)
    assert (np.abs(c - np.asarray(c64)) / np.abs(np.asarray(c64))).max() < 2e-5
    # 4 shards on one device: partial records -> combine == the unsharded update
    shards = [m.Handle(shard_rank=r, shard_count=4, **cfg) for r in range(4)]
    xd = torch.tensor(x, device="cuda")
    recs = torch.empty((4, 2 + H * a), dtype=torch.float32, device="cuda")
    one = torch.cuda.Stream()  # ONE explicit stream for all shards (stream 0 means "the handle's own stream": those would race)
    torch.cuda.synchronize()
    for r, sh in enumerate(shards):
        sh.shard_partial(xd.data_ptr(), recs[r].data_ptr(), one.cuda_stream)
    ud = torch.empty(a, dtype=torch.float32, device="cuda")
    shards[0].shard_finish(recs.data_ptr(), 4, ud.data_ptr(), one.cuda_stream)
    one.synchronize()
    assert np.abs(ud.cpu().numpy() - u).max() <= 2e-6


def test_mlp_shapes_without_a_kernel_are_refused(m):
    """Widths that neither kernel family serves answer MPPI_ERR_UNSUPPORTED with the reason; the split-bf16 flag is the
    256-wide and the 32-wide networks'."""
    a, s = 3, 6
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=s, a_dim=a, sigma=0.25 * np.eye(a), mlp=make_mlp(s, a, hid=64, n_hidden=2))
    assert e.value.status == 4 and "16 or 32" in str(e.value)  # MPPI_ERR_UNSUPPORTED
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=s, a_dim=a, sigma=0.25 * np.eye(a), mlp=make_mlp(s, a, hid=16, n_hidden=3), mlp_bf16x3=True)
    assert e.value.status == 1  # MPPI_ERR_INVALID_ARG (the 32-wide network has a split-bf16 kernel, the 16-wide one does not)


def mlp_full_size_properties(m, h, p64, x, U_in, n_check=1536):
    """Size-independent properties of one fused (Philox) MLP step at a BASELINE size, from the exported pieces."""
    u = h.next(x) if h.k_local == h.k else None
    c = h.debug_get(m.DBG_COSTS).astype(np.float64)
    eps = h.debug_get(m.DBG_NOISE)
    lam = 1.0
    # (1) a random subset of the rollouts against the fp64 oracle on the noise the step really used
    idx = np.sort(np.random.default_rng(3).choice(h.k_local, n_check, replace=False))
    truth = p64.rollout_cost(x, U_in, eps[idx])
    err = float((np.abs(c[idx] - truth) / np.abs(truth)).max())
    print("K_local=%d H=%d: max rel cost error on %d sampled rollouts %.3g" % (h.k_local, h.tau, n_check, err))
    assert err < 2e-5
    # (2) the weights are the soft-min of the kernel's own costs and sum to 1
    e = np.exp(-(c - c.min()) / lam)
    return u, c, eps, e


@pytest.mark.parametrize("name,kw,u_bar,factor", MLP_VARIANTS, ids=MLP_IDS)
def test_full_size_properties_mlp_c4(m, name, kw, u_bar, factor):
    """BASELINE configs[3] at full size (point_mass3d + 2x256 MLP, K=65536, H=64), fused Philox path: sampled costs
    against the fp64 oracle, sum(w) = 1, U' = fp64 recombination of the exported noise with the soft-min of the
    kernel's own costs, and the 8-way K-sharded step equals the unsharded one."""
    import torch
    K, H, a = 65536, 64, 3
    mlp = make_mlp(6, a, seed=0)
    sigma = 0.25 * np.eye(a)
    cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, seed=1, **kw)
    h = m.Handle(**cfg)
    p64 = orc.Problem(tau=H, s=6, a=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, threads=0, dtype=np.float64)
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    U_in = (0.05 * np.random.default_rng(1).standard_normal((H, a))).astype(F32)
    h.set_action_sequence(U_in)
    u, c, eps, e = mlp_full_size_properties(m, h, p64, x, U_in)
    w = h.debug_get(m.DBG_WEIGHTS).astype(np.float64)
    assert abs(w.sum() - 1) < 1e-5 and (w >= 0).all()
    np.testing.assert_allclose(w, e / e.sum(), rtol=1e-5, atol=1e-12)
    Uupd = h.debug_get(m.DBG_U_UPDATED).astype(np.float64)
    want = U_in + np.tensordot(e / e.sum(), eps.astype(np.float64), axes=(0, 0))
    print("max|U' - fp64 recombination| = %.3g" % np.abs(Uupd - want).max())
    np.testing.assert_allclose(Uupd, want, rtol=0, atol=2e-6)
    np.testing.assert_array_equal(u, Uupd[0].astype(F32))
    np.testing.assert_allclose(eps[:64], orc.noise(1, 0, 0, 64, H, a, sigma), rtol=0, atol=5e-6)
    # 8 shards of 8192 (the record exchange done by hand): same costs, same control
    shards = 8
    hs = [m.Handle(shard_rank=g, shard_count=shards, **cfg) for g in range(shards)]
    xd = torch.tensor(x, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(shards)]
    for g, hg in enumerate(hs):
        hg.set_action_sequence(U_in)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        hg.synchronize()
    for g, hg in enumerate(hs):
        hg.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        hg.synchronize()
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_allclose(us[g].cpu().numpy(), u, rtol=0, atol=2e-6)
        np.testing.assert_array_equal(hg.debug_get(m.DBG_COSTS), c[hg.k_offset:hg.k_offset + hg.k_local].astype(F32))


@pytest.mark.parametrize("name,kw,u_bar,factor", MLP_VARIANTS, ids=MLP_IDS)
def test_full_size_properties_mlp_c5_per_gpu_shape(m, name, kw, u_bar, factor):
    """BASELINE configs[4] as ONE rank sees it: shard 5 of 8 of a K=524288, H=128 MLP controller (65536 rollouts per
    GPU). Global-k Philox counters (noise = the oracle's stream at k_offset), sampled costs against the fp64 oracle,
    and the shard record (beta_g, eta_g, V_g) against the fp64 recombination of the exported noise."""
    import torch
    K, H, a, rank, shards = 524288, 128, 3, 5, 8
    mlp = make_mlp(6, a, seed=0)
    sigma = 0.25 * np.eye(a)
    h = m.Handle(k=K, tau=H, s_dim=6, a_dim=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, seed=1,
                 shard_rank=rank, shard_count=shards, **kw)
    assert h.k_local == 65536 and h.k_offset == rank * 65536
    p64 = orc.Problem(tau=H, s=6, a=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, threads=0, dtype=np.float64)
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    U_in = (0.05 * np.random.default_rng(1).standard_normal((H, a))).astype(F32)
    h.set_action_sequence(U_in)
    xd = torch.tensor(x, device="cuda")
    rec = torch.zeros(h.record_size, device="cuda")
    ud = torch.zeros(a, device="cuda")
    h.shard_partial(xd.data_ptr(), rec.data_ptr())
    h.shard_finish(rec.data_ptr(), 1, ud.data_ptr())  # a one-record "gather": also advances the step counter
    h.synchronize()
    _, c, eps, e = mlp_full_size_properties(m, h, p64, x, U_in)
    np.testing.assert_allclose(eps[:32], orc.noise(1, 0, h.k_offset, 32, H, a, sigma), rtol=0, atol=5e-6)
    r = rec.cpu().numpy().astype(np.float64)
    assert r[0] == c.min()
    np.testing.assert_allclose(r[1], e.sum(), rtol=1e-6)
    V = np.tensordot(e, eps.astype(np.float64), axes=(0, 0)).ravel()
    print("max|V_g - fp64| / eta = %.3g" % (np.abs(r[2:] - V).max() / e.sum()))
    np.testing.assert_allclose(r[2:] / r[1], V / e.sum(), rtol=0, atol=2e-6)
    np.testing.assert_allclose(ud.cpu().numpy(), (U_in + (V / e.sum()).reshape(H, a))[0], rtol=0, atol=2e-6)


def test_mlp_sharded_equals_unsharded(m):
    import torch
    K, H, a, shards = 2048, 16, 3, 4
    mlp = make_mlp(6, a, seed=2)
    full = make_mlp_pair(m, K, H, a, mlp, seed=8)[0]
    hs = [make_mlp_pair(m, K, H, a, mlp, seed=8, shard_rank=g, shard_count=shards)[0] for g in range(shards)]
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    xd = torch.tensor(x, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(shards)]
    u_full = full.next(x)
    for g, h in enumerate(hs):
        h.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        h.synchronize()
    for g, h in enumerate(hs):
        h.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        h.synchronize()
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_allclose(us[g].cpu().numpy(), u_full, rtol=0, atol=2e-6)
    c_full = full.debug_get(m.DBG_COSTS)
    for h in hs:
        np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_full[h.k_offset:h.k_offset + h.k_local])


def _fused_step_vs_oracle_and_shards(m, K, H, a, sigma, seed, shards=4, **kw):
    """One FUSED (on-device Philox) control step of a 2x256 MLP handle: the device noise is the oracle's Philox stream through
    this Sigma, U' and u against the fp64 oracle on the noise the step exported (north_star's 1e-5), sum(w) = 1; then the
    `shards`-way K-sharded step (records combined by mppi_shard_finish) equals the unsharded one and keeps the costs bitwise."""
    import torch
    s = 2 * a
    mlp = make_mlp(s, a, seed=seed)
    goal = (GOAL3 + [0.25, 0])[:s]
    cfg = dict(k=K, tau=H, s_dim=s, a_dim=a, lam=1.0, sigma=sigma, goal=goal, mlp=mlp, seed=seed, **kw)
    h = m.Handle(**cfg)
    p64 = orc.Problem(tau=H, s=s, a=a, lam=1.0, sigma=sigma, goal=goal, mlp=mlp, threads=0, dtype=np.float64)
    x = (0.2 * np.random.default_rng(seed).standard_normal(s)).astype(F32)
    U_in = (0.05 * np.random.default_rng(seed + 1).standard_normal((H, a))).astype(F32)
    h.set_action_sequence(U_in)
    u = h.next(x)
    eps = h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(eps, orc.noise(seed, 0, 0, K, H, a, sigma), rtol=0, atol=5e-6)
    u64, U64, c64 = p64.next_with_noise(x, U_in, eps)
    eu = float(np.abs(h.get_action_sequence().astype(np.float64) - np.asarray(U64)).max())
    print("%s: max|dU'| = %.3g" % (h.rollout_kernel_name(), eu))
    assert eu <= 1e-5 and np.abs(u.astype(np.float64) - np.asarray(u64)).max() <= 1e-5
    c = h.debug_get(m.DBG_COSTS)
    assert (np.abs(c.astype(np.float64) - np.asarray(c64)) / np.abs(np.asarray(c64))).max() < 2e-5
    assert abs(h.debug_get(m.DBG_WEIGHTS).astype(np.float64).sum() - 1) < 1e-5
    hs = [m.Handle(shard_rank=g, shard_count=shards, **cfg) for g in range(shards)]
    xd = torch.tensor(x, device="cuda")
    n = hs[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(shards)]
    for g, hg in enumerate(hs):
        assert hg.rollout_kernel_name() == h.rollout_kernel_name()
        hg.set_action_sequence(U_in)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        hg.synchronize()
    for g, hg in enumerate(hs):
        hg.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        hg.synchronize()
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_allclose(us[g].cpu().numpy(), u, rtol=0, atol=2e-6)
        np.testing.assert_array_equal(hg.debug_get(m.DBG_COSTS), c[hg.k_offset:hg.k_offset + hg.k_local])
    return h


def test_mlp_first_kernel_fused_step_and_sharding(m):
    """k_rollout_mlp (round 1's 8-wave kernel) still ships: it is what tuning mlp_v1 selects and what serves a_dim = 4 (two h1
    images of k_rollout_mlp2 would exceed the LDS). Both get the full treatment: Philox path, U' at 1e-5, record path."""
    h = _fused_step_vs_oracle_and_shards(m, 4096, 16, 3, 0.25 * np.eye(3), seed=21, tuning={"mlp_v1": 1})
    assert h.rollout_kernel_name() == "mppi::k_rollout_mlp<3, true>"
    h = _fused_step_vs_oracle_and_shards(m, 2048, 12, 4, 0.25 * np.eye(4), seed=22)
    assert h.rollout_kernel_name() == "mppi::k_rollout_mlp<4, true>"


def test_mlp_dense_sigma_fused_step(m):
    """A learned-model control step with a DENSE Sigma on the fused Philox path: k_rollout_mlp2<A, false, SRC_PHILOX> (eps = Sigma z with
    every product, Sigma^-1 dense in the action cost), never launched by a test before."""
    sigma = np.array([[0.30, 0.05, 0.0], [0.02, 0.25, 0.03], [0.0, 0.04, 0.20]], F32)
    h = _fused_step_vs_oracle_and_shards(m, 4096, 16, 3, sigma, seed=23)
    assert h.rollout_kernel_name() == "mppi::k_rollout_mlp2<3, false, 0>"
    sigma2 = np.array([[0.3, 0.1], [0.05, 0.2]], F32)
    h = _fused_step_vs_oracle_and_shards(m, 1000, 9, 2, sigma2, seed=24, shards=3)
    assert h.rollout_kernel_name() == "mppi::k_rollout_mlp2<2, false, 0>"


def test_lds_ceiling_is_per_kernel_not_per_handle(m):
    """two MLP handles on one device whose rollout kernel instance needs different dynamic-LDS sizes (the h1
    images + H*a floats of nominal actions): large, small, large again — the second handle must not lower the ceiling
    under the first."""
    a = 3
    mlp = make_mlp(6, a, seed=1)
    mk = lambda H: m.Handle(k=512, tau=H, s_dim=6, a_dim=a, lam=1.0, sigma=0.25 * np.eye(a), goal=GOAL3, mlp=mlp, seed=3)
    big, small = mk(128), mk(8)
    x = np.zeros(6, F32)
    u1 = big.next(x)
    small.next(x)
    u2 = big.next(x)
    big2 = mk(128)
    np.testing.assert_array_equal(big2.next(x), u1)
    np.testing.assert_array_equal(big2.next(x), u2)


@pytest.mark.parametrize("name,kw,u_bar,factor", MLP_VARIANTS, ids=MLP_IDS)
def test_configs4_all_eight_shards_combine(m, name, kw, u_bar, factor):
    """BASELINE configs[4] IN FULL on the one GPU a test box has: all 8 shards of the K=524288, H=128 learned-model controller
    (65536 rollouts and one 386-float record each) run one after the other, mppi_shard_finish combines the 8 records on every
    shard. Checked: global-k Philox counters on every shard; every record's beta_g = min of its shard's costs; U' (all 384
    entries) and u against the fp64 recombination of the 8 exported noise blocks with the soft-min of the kernels' own 524288
    costs; the replicated results bit-identical on all 8 shards; sampled costs of one shard against the fp64 oracle."""
    import torch
    K, H, a, shards = 524288, 128, 3, 8
    mlp = make_mlp(6, a, seed=0)
    sigma = 0.25 * np.eye(a)
    cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, seed=1, **kw)
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    U_in = (0.05 * np.random.default_rng(1).standard_normal((H, a))).astype(F32)
    xd = torch.tensor(x, device="cuda")
    hs = [m.Handle(shard_rank=g, shard_count=shards, **cfg) for g in range(shards)]
    n = hs[0].record_size
    assert n == 2 + H * a == 386
    recs = torch.zeros(shards * n, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(shards)]
    for g, hg in enumerate(hs):
        assert hg.k_local == 65536 and hg.k_offset == g * 65536
        hg.set_action_sequence(U_in)
        hg.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
        hg.synchronize()
    for g, hg in enumerate(hs):
        hg.shard_finish(recs.data_ptr(), shards, us[g].data_ptr())
        hg.synchronize()
    Uupd = [hg.debug_get(m.DBG_U_UPDATED) for hg in hs]
    for g in range(1, shards):  # replicated combine: identical bits everywhere
        np.testing.assert_array_equal(us[g].cpu().numpy(), us[0].cpu().numpy())
        np.testing.assert_array_equal(Uupd[g], Uupd[0])
        np.testing.assert_array_equal(hs[g].get_action_sequence(), hs[0].get_action_sequence())
    rec = recs.cpu().numpy().reshape(shards, n).astype(np.float64)
    costs = [hg.debug_get(m.DBG_COSTS).astype(np.float64) for hg in hs]
    beta = min(c.min() for c in costs)
    eta, V = 0.0, np.zeros(H * a)
    for g, hg in enumerate(hs):  # one shard's noise at a time: 100 MB each
        eps = hg.debug_get(m.DBG_NOISE)
        np.testing.assert_allclose(eps[:16], orc.noise(1, 0, hg.k_offset, 16, H, a, sigma), rtol=0, atol=5e-6)
        assert rec[g, 0] == costs[g].min()
        e_loc = np.exp(-(costs[g] - costs[g].min()))
        np.testing.assert_allclose(rec[g, 1], e_loc.sum(), rtol=1e-6)
        e = np.exp(-(costs[g] - beta))
        eta += e.sum()
        V += e @ eps.reshape(hg.k_local, H * a).astype(np.float64)
        if g == shards - 1:  # sampled rollouts of the last shard against the fp64 oracle on the noise it really used
            p64 = orc.Problem(tau=H, s=6, a=a, lam=1.0, sigma=sigma, goal=GOAL3, mlp=mlp, threads=0, dtype=np.float64)
            idx = np.sort(np.random.default_rng(4).choice(hg.k_local, 256, replace=False))
            truth = p64.rollout_cost(x, U_in, eps[idx])
            assert (np.abs(costs[g][idx] - truth) / np.abs(truth)).max() < 2e-5
        del eps
    want = U_in.astype(np.float64) + (V / eta).reshape(H, a)
    err = float(np.abs(Uupd[0].astype(np.float64) - want).max())
    print("configs[4], 8 records of 65536 rollouts: max|U' - fp64 recombination| = %.3g (eta = %.6g)" % (err, eta))
    assert err <= 2e-6
    np.testing.assert_array_equal(us[0].cpu().numpy(), Uupd[0][0])
    assert all(hg.get_step_counter() == 1 for hg in hs)


def test_mlp_unsupported_shapes_fail_loudly(m):
    bad = make_mlp(6, 3, hid=128)
    with pytest.raises(m.MppiError) as e:
        m.Handle(k=64, tau=4, s_dim=6, a_dim=3, mlp=bad)
    assert e.value.status == 4 and "256" in str(e.value)


def test_python_controller_with_upsilon_and_gamma(m):
    """The Python reference's ControllerBase(model, cost, ...) with γ, υ != 1: the sampler draws (υΣ)·z while
    the cost keeps Σ⁻¹ of Σ and the ½[γ(uᵀΣ⁻¹u+2uᵀΣ⁻¹ε)+λ(1-1/υ)εᵀΣ⁻¹ε] action cost
    (controller_base.py:362-368, cost_base.py:114-170)."""
    K, H, a, s = 1024, 12, 2, 4
    lam, gamma, ups = 0.8, 1.5, 2.0
    sigma = np.array([[0.3, 0.0], [0.0, 0.2]], F32)
    goal = np.array([[1.0], [0.0], [0.5], [0.0]], F32)
    Q = np.diag([1.0, 0.5, 2.0, 0.5]).astype(F32)
    model = m.PointMassModel(1.5, 0.1, s, a)
    cost = m.StaticCost(lam, gamma, ups, sigma, goal, Q)
    ctl = m.ControllerBase(model, cost, k=K, tau=H, sDim=s, aDim=a, lam=lam, upsilon=ups, sigma=sigma, seed=3)
    p = orc.Problem(tau=H, s=s, a=a, dt=0.1, mass=1.5, lam=lam, gamma=gamma, upsilon=ups, sigma=sigma, goal=goal.ravel(),
                    Q=Q, action_cost=orc.ACTION_COST_PY)
    x = np.array([[0.1], [0.0], [-0.2], [0.1]], F32)
    u = ctl.next(x)
    eps = ctl._h.debug_get(m.DBG_NOISE)
    np.testing.assert_allclose(eps, orc.noise(3, 0, 0, K, H, a, ups * sigma), rtol=0, atol=1e-5)  # sampler uses υΣ
    u_ref, U_ref, c_ref = p.next_with_noise(x.ravel(), np.zeros((H, a), F32), eps)
    np.testing.assert_array_equal(ctl._h.debug_get(m.DBG_COSTS), c_ref)
    np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)
    np.testing.assert_allclose(ctl._actionSeq[..., 0], U_ref, rtol=0, atol=U_TOL)


def test_reference_shaped_python_entry_point(m, tmp_path):
    """examples/main.py = scripts/main.py's loop (YAML config + task, Simulation, PointMassModel, StaticCost,
    ControllerBase, save) with the MuJoCo-free plant; must reach the goal and write the transition CSV in
    DataBase::toCSV's bytes. `--new -l` leaves config.yaml / task.yaml in <log_dir>/controller (observer_base.py:39-54)
    and `--replay --log_dir` (main.py:19-27,68-69; utile.py:53-59) repeats the run from them: identical CSV bytes."""
    import os
    import re
    import subprocess
    import sys
    from conftest import ROOT
    main_py = os.path.join(ROOT, "examples", "main.py")
    log = tmp_path / "run"
    r = subprocess.run([sys.executable, main_py, "--new",
                        "--config", os.path.join(ROOT, "examples", "config", "point_mass3d.yaml"),
                        "--task", os.path.join(ROOT, "examples", "config", "static_task3d.yaml"),
                        "-s", "80", "-l", "--log_dir", str(log)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    dist = float(r.stdout.split("|x - goal| =")[1].split()[0])
    assert dist < 0.1, r.stdout
    logdir = log / "controller"
    assert (logdir / "config.yaml").exists() and (logdir / "task.yaml").exists()
    first = (logdir / "transitions.csv").read_text()
    rows = first.strip().splitlines()
    assert len(rows) == 81 and rows[0] == "x0,x1,x2,x3,x4,x5,u0,u1,u2,x_next0,x_next1,x_next2,x_next3,x_next4,x_next5,"
    assert all(re.fullmatch(r"(-?\d+\.\d{6},){15}", row) for row in rows[1:])
    csv2 = tmp_path / "replayed.csv"
    r = subprocess.run([sys.executable, main_py, "--replay", "--log_dir", str(logdir), "-s", "80", "--csv", str(csv2)],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert csv2.read_text() == first
    r = subprocess.run([sys.executable, main_py, "--replay", "--log_dir", str(tmp_path)], capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "config.yaml" in r.stderr  # a directory that holds no logged run


def test_transition_log_ring_pairing_and_formats(m, tmp_path):
    """mppi_set_transition_log / mppi_save_next / mppi_to_csv_format: off by default (nothing recorded), a ring of
    max_rows rows, x_next pairs with the LAST (x, u) so a skipped saveNext drops that one row only, and the two formats
    (the reference's "%f," cells, data_base.cpp:36-44, and the round-trip "%.9g")."""
    h = m.Handle(k=256, tau=8, s_dim=2, a_dim=1)
    x = np.array([0.25, -0.5], F32)
    h.next(x)
    with pytest.raises(m.MppiError):
        h.to_csv(str(tmp_path / "off.csv"))          # the log is off until asked for
    h.set_transition_log(3)
    us = []
    for i in range(5):
        us.append(h.next(x + i)[0])
        if i != 3:                                   # skip one saveNext: that row has no successor
            h.save_next(x + i + 0.125)
    f = tmp_path / "ring.csv"
    h.to_csv(str(f), m.CSV_ROUNDTRIP)
    rows = f.read_text().strip().splitlines()
    assert rows[0] == "x0,x1,u0,x_next0,x_next1"
    got = np.array([[float(v) for v in r.split(",")] for r in rows[1:]], F32)
    want = np.array([[0.25 + i, -0.5 + i, us[i], 0.375 + i, -0.375 + i] for i in (2, 4)], F32)  # ring of 3 = steps 2,3,4; 3 unpaired
    np.testing.assert_array_equal(got, want)
    # what the bounded log left out is reported, never silent (the reference's m_db is unbounded): 5 steps into 3 rows = 2 overwritten
    assert h.transition_log_stats() == dict(held=3, overwritten=2, without_successor=1)
    h.to_csv(str(f))                                 # reference bytes
    rows = f.read_text().splitlines()
    assert rows[0] == "x0,x1,u0,x_next0,x_next1," and rows[1] == "2.250000,1.500000,%f,2.375000,1.625000," % us[2]


def test_written_out_philox_equals_rocrand_engine(m):
    """The default build evaluates Philox4x32-10 with its own round function and hardware-rate Box-Muller; the
    -DMPPI_ROCRAND_NORMALS build (build/variants/, made by tools/ablate.py or build.build_variant) calls rocRAND's
    engine and normal_distribution4 verbatim. Same counters -> the same noise up to the log/sqrt flavour."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    so = os.path.join(ROOT, "build", "variants", "libmppi_hip_rocrand_normals.so")
    if not os.path.exists(so):
        pytest.skip("variant library not built")
    code = ("import sys; sys.path.insert(0, %r); import numpy as np, mppi_tf_amd as m; "
            "h = m.Handle(k=512, tau=20, s_dim=6, a_dim=3, seed=99, sigma=0.5*np.eye(3)); h.next(np.zeros(6, np.float32)); "
            "np.save(sys.argv[1], h.debug_get(m.DBG_NOISE))" % ROOT)
    outs = []
    for tag, env in (("default", {}), ("rocrand", {"MPPI_SO_PATH": so})):
        f = os.path.join(os.environ.get("TMPDIR", "/tmp"), "mppi_noise_%s_%d.npy" % (tag, os.getpid()))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append(np.load(f))
        os.remove(f)
    np.testing.assert_allclose(outs[0], outs[1], rtol=0, atol=5e-6)
    assert np.abs(outs[0]).max() > 1.0  # not degenerate


def test_mlp_kernels_in_the_rocrand_variant_weight_the_noise_they_roll_out():
    """in the -DMPPI_ROCRAND_NORMALS build the MLP kernels drew their rollout noise with the hardware Box-Muller while the
    tile record regenerated it with rocRAND's normal_distribution4. Both now come from normals_of_block: on the variant library a fused
    learned-model step must equal the injected-noise step fed with the noise it exported (costs bitwise, U' to rounding)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    var = os.path.join(ROOT, "build", "variants", "libmppi_hip_rocrand_normals.so")
    code = """
import sys, numpy as np
sys.path.insert(0, %r)
import mppi_tf_amd as m
rng = np.random.default_rng(0)
def net(dims):
    return dict(W=[(rng.uniform(-1, 1, (dims[i], dims[i + 1])) / np.sqrt(dims[i])).astype(np.float32) * (0.1 if i == len(dims) - 2 else 1) for i in range(len(dims) - 1)],
                b=[np.zeros(dims[i + 1], np.float32) for i in range(len(dims) - 1)])
for dims in ([9, 256, 256, 6], [9, 32, 32, 32, 6]):
    mlp = net(dims)
    cfg = dict(k=2048, tau=12, s_dim=6, a_dim=3, sigma=0.25 * np.eye(3), goal=[1, 0, .5, 0, .75, 0], mlp=mlp, seed=3)
    a, b = m.Handle(**cfg), m.Handle(**cfg)
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], np.float32)
    ua = a.next(x)
    eps = a.debug_get(m.DBG_NOISE)
    ub = b.next_with_noise(x, eps)
    assert np.array_equal(a.debug_get(m.DBG_COSTS), b.debug_get(m.DBG_COSTS)), a.rollout_kernel_name()
    assert np.abs(ua - ub).max() <= 2e-6, (a.rollout_kernel_name(), ua, ub)
print("VARIANT_MLP_OK")
""" % ROOT
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, env=dict(os.environ, MPPI_SO_PATH=var), cwd=ROOT)
    assert r.returncode == 0 and "VARIANT_MLP_OK" in r.stdout, r.stdout[-2000:] + r.stderr[-3000:]


def test_native_cpp_sharded_host_with_rccl(m):
    """examples/host_loop_sharded.cpp: mppi_shard_partial -> ncclAllGather -> mppi_shard_finish from C++ over every
    visible GPU (one here), closed loop on the host plant; replicated controls must agree bit for bit."""
    import os
    import subprocess
    from conftest import ROOT
    exe = os.path.join(ROOT, "examples", "host_loop_sharded")
    assert os.path.exists(exe), "run __graft_entry__.build() first"
    for mode in ("rccl", "p2p"):
        r = subprocess.run([exe, "8192", "32", "3", "80", mode], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout + r.stderr
        assert "rollouts/s" in r.stdout


# =============================================================== direct record exchange (mppi_shard_p2p_*)
@pytest.mark.parametrize("shards", [1, 2, 3])
def test_direct_exchange_equals_allgather_path_bit_for_bit(m, shards):
    """`shards` handles on this one GPU, each on its own stream, exchange their records as packets stored into each
    other's inboxes from inside the finish kernel; U and u must equal the partial -> gather -> finish path bitwise."""
    import torch
    from concurrent.futures import ThreadPoolExecutor
    K, H, a = 4096, 32, 3
    x = np.array([0.2, 0.1, -0.3, 0, 0.5, -0.1], F32)
    xd = torch.tensor(x, device="cuda")
    ref = [make_pair(m, K, H, a, seed=33, shard_rank=g, shard_count=shards)[0] for g in range(shards)]
    hs = [make_pair(m, K, H, a, seed=33, shard_rank=g, shard_count=shards)[0] for g in range(shards)]
    ptrs = [h.p2p_export(want_ipc=False)[0] for h in hs]
    for h in hs:
        h.p2p_attach(ptrs, timeout_ms=500)
    with ThreadPoolExecutor(shards) as ex:  # a probe synchronises its stream: all ranks must be in flight together
        for _ in range(3):
            assert all(ex.map(lambda h: h.p2p_probe(), hs))
    n = ref[0].record_size
    recs = torch.zeros(shards * n, device="cuda")
    u_ref = torch.zeros(a, device="cuda")
    us = [torch.zeros(a, device="cuda") for _ in range(shards)]
    for step in range(4):
        for g, h in enumerate(ref):
            h.shard_partial(xd.data_ptr(), recs[g * n:(g + 1) * n].data_ptr())
            h.synchronize()
        for h in ref:
            h.shard_finish(recs.data_ptr(), shards, u_ref.data_ptr())
            h.synchronize()
        for g, h in enumerate(hs):          # enqueue only: the kernels of all shards meet on the GPU
            h.p2p_step(xd.data_ptr(), us[g].data_ptr())
        for h in hs:
            h.synchronize()
            assert not h.p2p_timed_out()
        for g, h in enumerate(hs):
            np.testing.assert_array_equal(us[g].cpu().numpy(), u_ref.cpu().numpy())
            np.testing.assert_array_equal(h.get_action_sequence(), ref[0].get_action_sequence())
            assert h.get_step_counter() == step + 1


def test_direct_exchange_with_eight_shards_in_one_process():
    """The direct exchange at configs[4]'s shard count: 8 handles in ONE process on this GPU, each on its own stream, meet inside
    their finish kernels (every workgroup stores to 8 inboxes and spins for 8 packets). The kernels of all 8 shards must be in
    flight together, so the worker runs with 16 hardware queues (GPU_MAX_HW_QUEUES; the default 4 would queue shard 5's
    kernels behind shard 1's spinning ones) — on a real node every rank has its own GPU. Bitwise equal to partial -> gather -> finish."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, GPU_MAX_HW_QUEUES="16")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "p2p8_worker.py")], capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0 and "P2P8_WORKER_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


def test_direct_exchange_deadline_instead_of_hang(m):
    """A rank whose peer never sends must come back with the timed-out flag, not hang — and without garbage: the step
    applies a ZERO update (U' = U, then the shift), the Philox step counter still advances, a step already queued behind
    it skips the exchange (no second deadline), and the next mppi_shard_p2p_step is refused with MPPI_ERR_EXCHANGE."""
    import time
    import torch
    K, H, a = 1024, 8, 3
    hs = [make_pair(m, K, H, a, seed=3, shard_rank=g, shard_count=2)[0] for g in range(2)]
    ptrs = [h.p2p_export(want_ipc=False)[0] for h in hs]
    hs[0].p2p_attach(ptrs, timeout_ms=300)
    U0 = (0.1 * np.random.default_rng(0).standard_normal((H, a))).astype(F32)
    hs[0].set_action_sequence(U0)
    xd, u = torch.zeros(6, device="cuda"), torch.zeros(a, device="cuda")
    t0 = time.perf_counter()
    hs[0].p2p_step(xd.data_ptr(), u.data_ptr())   # shard 1 never steps: this one runs into the deadline
    hs[0].p2p_step(xd.data_ptr(), u.data_ptr())   # enqueued before the host can know: must not wait a second time
    hs[0].synchronize()
    el = time.perf_counter() - t0
    assert hs[0].p2p_timed_out() and 0.25 < el < 0.55, el
    np.testing.assert_array_equal(u.cpu().numpy(), U0[1])                       # u = U'[0] of the second zero-update step
    np.testing.assert_array_equal(hs[0].get_action_sequence(), np.vstack([U0[2:], np.zeros((2, a), F32)]))
    assert hs[0].get_step_counter() == 2
    with pytest.raises(m.MppiError) as e:
        hs[0].p2p_step(xd.data_ptr(), u.data_ptr())
    assert e.value.status == 8
    # the all-gather path stays open on the same handle (what ShardedController.resync continues with)
    rec = torch.zeros(hs[0].record_size, device="cuda")
    hs[0].shard_partial(xd.data_ptr(), rec.data_ptr())
    hs[0].shard_finish(rec.data_ptr(), 1, u.data_ptr())
    hs[0].synchronize()
    assert np.isfinite(u.cpu().numpy()).all() and hs[0].get_step_counter() == 3
    with pytest.raises(m.MppiError):
        hs[1].p2p_step(xd.data_ptr(), u.data_ptr())  # not attached
    with pytest.raises(m.MppiError):
        hs[1].p2p_attach(ptrs[::-1], timeout_ms=50)  # own entry must be the own inbox


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def test_direct_exchange_across_processes_over_hipipc():
    """Two processes on this GPU (gloo rendezvous): inboxes exported with hipIpcGetMemHandle, mapped by the peer,
    self-test + vote, then ShardedController steps; both ranks must produce the single-process result."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MPPI_EXCHANGE="p2p")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "p2p_worker.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("P2P_WORKER_OK") == 2, r.stdout[-3000:] + r.stderr[-3000:]


def test_allgather_path_across_processes_is_stream_ordered():
    """Two processes on this GPU (gloo): ShardedController's torch path on torch's default stream gives, step for step and bit for bit,
    the single-process result — it used to gather records one step stale (tests/gather_worker.py says why)."""
    import os
    import subprocess
    import sys
    from conftest import ROOT
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    env.pop("MPPI_EXCHANGE", None)
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                        "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "gather_worker.py")], capture_output=True, text=True,
                       timeout=600, env=env, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.count("GATHER_WORKER_OK") == 2, r.stdout[-3000:] + r.stderr[-3000:]


# =============================================================== options of the Python update
def test_action_limits_clip_the_updated_sequence(m):
    """clip_act (controller_base.py:500-504): U' = clip(U + Σ w eps, a_min, a_max) row-wise, u = U'[0]."""
    K, H, a = 512, 10, 2
    h, p = make_pair(m, K, H, a, seed=4)
    lo, hi = np.array([-0.05, -0.3], F32), np.array([0.1, 0.02], F32)
    h.set_action_limits(lo, hi)
    x = np.array([0.3, 0.0, -0.2, 0.1], F32)
    U = np.zeros((H, a), F32)
    for step in range(3):
        u = h.next(x)
        eps = h.debug_get(m.DBG_NOISE)
        u_ref, U_sh, _ = p.next_with_noise(x, U, eps)  # unclipped update of the same U and noise
        ref = {"U_new": np.concatenate([u_ref.reshape(1, a), U_sh[:-1]])}
        Uc = np.clip(ref["U_new"].astype(F32), lo, hi)
        np.testing.assert_allclose(h.debug_get(m.DBG_U_UPDATED), Uc, rtol=0, atol=2e-6)
        np.testing.assert_allclose(u, Uc[0], rtol=0, atol=2e-6)
        U = h.get_action_sequence()
        np.testing.assert_allclose(U[:-1], h.debug_get(m.DBG_U_UPDATED)[1:], rtol=0, atol=0)
        assert (U >= lo - 1e-7).all() and (U <= hi + 1e-7).all()
    assert (np.abs(ref["U_new"]) > np.maximum(np.abs(lo), np.abs(hi))).any(), "the limits never bound: weak test"
    h.set_action_limits(None, None)
    h.next(x)
    with pytest.raises(m.MppiError):
        h.set_action_limits(hi, lo)


@pytest.mark.parametrize("window,order", [(5, 3), (11, 9), (7, 2), (1, 0)])
def test_sequence_filter_matches_scipy_savgol(m, window, order):
    """filterSeq (controller_base.py:277-291): the stored sequence after a step = savgol_filter(shifted U', window,
    order, axis=0) (mode 'interp'); u and U' themselves are not filtered."""
    from scipy.signal import savgol_filter
    K, H, a = 256, 24, 3
    h, p = make_pair(m, K, H, a, seed=6)
    h.set_sequence_filter(window, order)
    x = np.array([0.3, 0.0, -0.2, 0.1, 0.5, 0.0], F32)
    U = np.zeros((H, a), F32)
    for step in range(3):
        u = h.next(x)
        eps = h.debug_get(m.DBG_NOISE)
        u_ref, U_sh, _ = p.next_with_noise(x, U, eps)
        ref = {"U_new": np.concatenate([u_ref.reshape(1, a), U_sh[:-1]])}
        np.testing.assert_allclose(h.debug_get(m.DBG_U_UPDATED), ref["U_new"], rtol=0, atol=2e-6)
        np.testing.assert_allclose(u, ref["U_new"][0], rtol=0, atol=2e-6)
        Uupd = h.debug_get(m.DBG_U_UPDATED).astype(np.float64)
        shifted = np.concatenate([Uupd[1:], np.zeros((1, a))])
        want = savgol_filter(shifted, window, order, deriv=0, delta=1.0, axis=0)
        U = h.get_action_sequence()
        np.testing.assert_allclose(U, want, rtol=0, atol=2e-6)
    h.set_sequence_filter(0)
    h.next(x)
    np.testing.assert_array_equal(h.get_action_sequence()[:-1], h.debug_get(m.DBG_U_UPDATED)[1:])
    for bad in [(10, 9), (H + 1, 2), (5, 5)]:
        with pytest.raises(m.MppiError):
            h.set_sequence_filter(*bad)


@pytest.mark.parametrize("fault", ["", "export", "probe"])
def test_sharded_controller_falls_back_when_the_direct_exchange_cannot_come_up(m, monkeypatch, fault):
    """ShardedController on the real backend (one rank, exchange forced): with an injected fault in the inbox export
    or in the probe it must take the all-gather path, and give the same controls either way."""
    import torch
    from mppi_tf_amd.distributed import ShardedController
    monkeypatch.setenv("MPPI_FORCE_EXCHANGE", "1")
    monkeypatch.delenv("MPPI_EXCHANGE", raising=False)
    K, H, a = 4096, 32, 3
    cfg = dict(k=K, tau=H, s_dim=6, a_dim=a, sigma=0.25 * np.eye(a), goal=GOAL3, seed=11)
    ctl = ShardedController(tuning={"p2p_fault": fault}, **cfg)  # mppi_set_tuning(MPPI_TUNE_P2P_FAULT)
    assert ctl.exchange == ("rccl" if fault else "p2p"), ctl.p2p_note
    ref = m.Handle(**cfg)
    x = torch.tensor([0.2, 0.1, -0.3, 0, 0.5, -0.1], device="cuda")
    for _ in range(3):
        u = ctl.next(x)
        torch.cuda.synchronize()
        ctl.check()
        np.testing.assert_allclose(u.cpu().numpy(), ref.next(x.cpu().numpy()), rtol=0, atol=2e-6)
    if fault == "export":
        with pytest.raises(RuntimeError):
            ShardedController(exchange="p2p", tuning={"p2p_fault": fault}, **cfg)


def test_device_resident_steps_replay_from_a_hipgraph(m):
    """mppi_next_device is enqueue-only, and all state that advances (U in its two buffers, the Philox step counter)
    lives on the device: an EVEN number of consecutive steps captured into a hipGraph replays as further control steps
    (include/mppi_c.h). Captured through torch's graph API on its capture stream; compared with plain launches."""
    import torch
    K, H, a = 4096, 32, 3
    hg, _ = make_pair(m, K, H, a, seed=13)
    hd, _ = make_pair(m, K, H, a, seed=13)
    x = torch.tensor([0.2, 0.1, -0.3, 0.0, 0.5, -0.1], device="cuda")
    ug = [torch.zeros(a, device="cuda") for _ in range(2)]
    ud = torch.zeros(a, device="cuda")
    # one plain step first on both: the first step reads U from offset 0, later ones from the shifted offset
    for h, u in ((hg, ug[0]), (hd, ud)):
        h.next_device(x.data_ptr(), u.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        hg.next_device(x.data_ptr(), ug[0].data_ptr(), st)
        hg.next_device(x.data_ptr(), ug[1].data_ptr(), st)
    replays = 3
    for _ in range(replays):
        g.replay()
    torch.cuda.synchronize()
    for i in range(2 * replays):
        hd.next_device(x.data_ptr(), ud.data_ptr(), torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(ug[1].cpu().numpy(), ud.cpu().numpy())
    np.testing.assert_array_equal(hg.get_action_sequence(), hd.get_action_sequence())
    assert hg.get_step_counter() == hd.get_step_counter() == 1 + 2 * replays


def test_roctx_ranges_around_the_step(m):
    """MPPI_TUNE_TRACE (the reference brackets its step with tf.profiler.experimental.start/stop,
    controller_base.py:241-248, 587-595): roctx ranges mppi:step > mppi:rollout / mppi:finish around what a step enqueues, the marker
    library dlopen'ed on first use. Tracing changes nothing about the step; profiles/r04_marker_trace.* holds a rocprofv3 --marker-trace
    timeline of examples/host_loop with it."""
    x = np.array([0.2, 0.1, -0.3, 0, 0.5, -0.1], F32)
    traced, plain = make_pair(m, 2048, 16, 3, seed=5, tuning={"trace": 1})[0], make_pair(m, 2048, 16, 3, seed=5)[0]
    for _ in range(3):
        np.testing.assert_array_equal(traced.next(x), plain.next(x))
    traced.set_tuning("trace", 0)
    np.testing.assert_array_equal(traced.next(x), plain.next(x))
