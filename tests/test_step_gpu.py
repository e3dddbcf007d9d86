"""The whole step in one launch (k_step_pc<.., STEP_FUSE>) and the armed launch (STEP_ARM) against the two-launch step and the
CPU oracle, through the C-ABI. Needs an MI355X: every test is marked `gpu`.

Bars: sample costs BIT-IDENTICAL to the oracle's (same bar as tests/test_parity_gpu.py); the fused / armed step's U', u and
Philox step counter BIT-IDENTICAL to the two-launch step's (same records, same combine order); U' within 1e-5 of the oracle.
"""
import time

import numpy as np
import pytest

from oracle import oracle as orc

pytestmark = pytest.mark.gpu
F32 = np.float32
U_TOL = 1e-5
GOAL3 = [1, 0, .5, 0, .75, 0]


@pytest.fixture(scope="module")
def m():
    import mppi_tf_amd
    assert mppi_tf_amd.load().mppi_device_count() >= 1, "no GPU visible to libmppi_hip.so"
    return mppi_tf_amd


def cfg_of(K, H, a, sigma=None, **kw):
    s = 2 * a
    sigma = np.eye(a) * 0.25 if sigma is None else sigma
    d = dict(k=K, tau=H, s_dim=s, a_dim=a, dt=0.1, mass=1.0, lam=1.0, sigma=sigma, goal=(GOAL3 + [0.25, 0])[:s], Q=np.ones(s), seed=1)
    d.update(kw)
    return d


def plant(x, u, a, dt=0.1):
    x = x.copy()
    for j in range(a):
        x[2 * j] = x[2 * j] + F32(dt) * x[2 * j + 1] + F32(dt * dt / 2) * u[j]
        x[2 * j + 1] = x[2 * j + 1] + F32(dt) * u[j]
    return x


def closed_loop(h, a, steps, x0=None):
    x = np.zeros(2 * a, F32) if x0 is None else np.asarray(x0, F32)
    us = []
    for _ in range(steps):
        u = h.next(x)
        us.append(u.copy())
        x = plant(x, u, a)
    return np.asarray(us), x


SHAPES = [(128, 32, 1), (4096, 64, 2), (3000, 50, 3), (8192, 64, 3), (200, 7, 4), (1000, 100, 2), (64, 4, 2), (65, 64, 3), (5000, 160, 1), (777, 80, 4), (2048, 79, 2), (100, 1, 3)]


@pytest.mark.parametrize("K,H,a", SHAPES)
def test_fused_step_equals_two_launch_step(m, K, H, a):
    """Same x sequence through a fused handle and a two-launch handle: controls, U', costs, step counter — the same bits."""
    c = cfg_of(K, H, a)
    hf, h2 = m.Handle(**c), m.Handle(tuning={"fused_step": 0}, **c)
    assert "k_step_pc" in hf.rollout_kernel_name() and "k_rollout_pc" in h2.rollout_kernel_name()
    x = np.zeros(2 * a, F32)
    for i in range(6):
        uf, u2 = hf.next(x), h2.next(x)
        np.testing.assert_array_equal(uf, u2, err_msg="step %d" % i)
        np.testing.assert_array_equal(hf.debug_get(m.DBG_COSTS), h2.debug_get(m.DBG_COSTS))
        np.testing.assert_array_equal(hf.get_action_sequence(), h2.get_action_sequence())
        assert float(hf.debug_get(m.DBG_BETA)) == float(h2.debug_get(m.DBG_BETA)) and float(hf.debug_get(m.DBG_ETA)) == float(h2.debug_get(m.DBG_ETA))
        x = plant(x, uf, a)
    assert hf.get_step_counter() == h2.get_step_counter() == 6
    hf.close(); h2.close()


@pytest.mark.parametrize("K,H,a", [(4096, 64, 2), (3000, 50, 3), (128, 84, 1), (777, 80, 4), (2048, 33, 2)])
def test_fused_step_seven_and_five_producers_agree(m, K, H, a):
    """H <= 84 runs the one-launch step with seven producer waves per tile (k_step_pc<a, 7, 3, ..>), fused_step = 2 keeps five: the
    same tiles, the same record algebra — the same bits."""
    c = cfg_of(K, H, a)
    h7, h5 = m.Handle(**c), m.Handle(tuning={"fused_step": 2}, **c)
    assert "k_step_pc<%d, 7, 3" % a in h7.rollout_kernel_name() and "k_step_pc<%d, 5," % a in h5.rollout_kernel_name()
    x = np.zeros(2 * a, F32)
    for i in range(5):
        u7, u5 = h7.next(x), h5.next(x)
        np.testing.assert_array_equal(u7, u5, err_msg="step %d" % i)
        np.testing.assert_array_equal(h7.debug_get(m.DBG_COSTS), h5.debug_get(m.DBG_COSTS))
        np.testing.assert_array_equal(h7.get_action_sequence(), h5.get_action_sequence())
        x = plant(x, u7, a)
    h7.close(); h5.close()


@pytest.mark.parametrize("K,H,a", [(4096, 64, 2), (3000, 50, 3), (128, 32, 1)])
def test_fused_step_against_oracle(m, K, H, a):
    """BASELINE configs[1], the reference's default K = 3000 / H = 50, configs[0]: the fused step's costs bit-identical to the oracle's on
    the noise the step drew, U' and u within 1e-5."""
    c = cfg_of(K, H, a)
    h = m.Handle(**c)
    p = orc.Problem(tau=H, s=2 * a, a=a, dt=0.1, mass=1.0, lam=1.0, sigma=c["sigma"], goal=c["goal"], Q=c["Q"], threads=0)
    x, U = np.array([0.1, 0, -0.2, 0, 0.3, 0][:2 * a], F32), np.zeros((H, a), F32)
    for step in range(3):
        u = h.next(x)
        eps = h.debug_get(m.DBG_NOISE)
        np.testing.assert_allclose(eps, orc.noise(1, step, 0, K, H, a, c["sigma"]), rtol=0, atol=5e-6)
        u_ref, U_ref, c_ref = p.next_with_noise(x, U, eps)
        np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_ref)
        np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)
        np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)
        U = h.get_action_sequence()
        x = plant(x, u, a)
    h.close()


def test_fused_step_device_path_and_options(m):
    """mppi_next_device on a fused handle (pipelined, no host in between) and the update's options inside the column waves: action limits,
    a dense sigma."""
    import torch
    sig = np.array([[0.3, 0.05], [0.05, 0.2]], F32)
    c = cfg_of(4096, 64, 2, sigma=sig)
    hf, h2 = m.Handle(**c), m.Handle(tuning={"fused_step": 0}, **c)
    for h in (hf, h2):
        h.set_action_limits([-0.05, -0.02], [0.04, 0.03])
    x = torch.zeros(4, dtype=torch.float32, device="cuda")
    uf, u2 = torch.zeros(2, device="cuda"), torch.zeros(2, device="cuda")
    st = torch.cuda.current_stream().cuda_stream or 1
    for _ in range(25):
        hf.next_device(x.data_ptr(), uf.data_ptr(), st)
        h2.next_device(x.data_ptr(), u2.data_ptr(), st)
    torch.cuda.synchronize()
    np.testing.assert_array_equal(uf.cpu().numpy(), u2.cpu().numpy())
    np.testing.assert_array_equal(hf.get_action_sequence(), h2.get_action_sequence())
    U = hf.get_action_sequence()
    assert U[:, 0].max() <= 0.04 and U[:, 0].min() >= -0.05 and U[:, 1].max() <= 0.03 and U[:, 1].min() >= -0.02
    assert hf.get_step_counter() == 25
    hf.close(); h2.close()


def armable(m):
    h = m.Handle(**cfg_of(128, 8, 1))
    try:
        h.set_tuning("armed_us", 100)
        return True
    except Exception:
        return False
    finally:
        h.close()


@pytest.mark.parametrize("K,H,a", [(4096, 64, 2), (3000, 50, 3), (65536, 64, 3), (16384, 64, 3), (128, 32, 1), (40000, 100, 2)])
def test_armed_step_equals_unarmed_step(m, K, H, a):
    """A closed loop driven as fast as the host goes: with armed launches (fused below 128 tiles, rollout + finish above) and without —
    the same controls bit for bit, the same U, the same step counter."""
    if not armable(m):
        pytest.skip("no large-BAR device: MPPI_TUNE_ARMED_US is unsupported here")
    c = cfg_of(K, H, a)
    ha, hu = m.Handle(tuning={"armed_us": 20000, "armed_always": 1}, **c), m.Handle(**c)
    ua, xa = closed_loop(ha, a, 12)
    uu, xu = closed_loop(hu, a, 12)
    np.testing.assert_array_equal(ua, uu)
    np.testing.assert_array_equal(ha.get_action_sequence(), hu.get_action_sequence())  # (retires the launch armed behind the last call)
    assert ha.get_step_counter() == hu.get_step_counter() == 12
    np.testing.assert_array_equal(ha.debug_get(m.DBG_COSTS), hu.debug_get(m.DBG_COSTS))
    # and on: the handle keeps working after an armed launch was retired by another entry point
    ua2, _ = closed_loop(ha, a, 5, xa)
    uu2, _ = closed_loop(hu, a, 5, xu)
    np.testing.assert_array_equal(ua2, uu2)
    ha.close(); hu.close()


@pytest.mark.parametrize("K,H,a", [(4096, 64, 2), (65536, 64, 3)])
def test_armed_launch_deadline_path(m, K, H, a):
    """The armed launch's x does not come: tile 0 aborts at its soft deadline, nothing is applied, the next call takes the ordinary
    launch and the controls are those of a handle that never armed. Also: x that arrives while the abort is being decided."""
    if not armable(m):
        pytest.skip("no large-BAR device: MPPI_TUNE_ARMED_US is unsupported here")
    c = cfg_of(K, H, a)
    ha, hu = m.Handle(tuning={"armed_us": 300, "armed_always": 1}, **c), m.Handle(**c)
    x = np.zeros(2 * a, F32)
    rng = np.random.default_rng(0)
    for i in range(30):
        ua, uu = ha.next(x), hu.next(x)
        np.testing.assert_array_equal(ua, uu, err_msg="step %d" % i)
        x = plant(x, ua, a)
        # every third call well past the deadline, every third right around it, the rest at once
        if i % 3 == 0:
            time.sleep(0.002)
        elif i % 3 == 1:
            t0 = time.perf_counter()
            while time.perf_counter() - t0 < 300e-6 + rng.uniform(-40e-6, 40e-6):
                pass
    np.testing.assert_array_equal(ha.get_action_sequence(), hu.get_action_sequence())
    assert ha.get_step_counter() == hu.get_step_counter() == 30
    ha.close(); hu.close()


def test_armed_soak_with_a_host_that_dithers_around_the_deadline(m):
    """3000 closed-loop steps with host delays drawn between 0 and 2.5 deadlines (so that every interleaving of "x arrives" and "tile 0
    gives up" occurs many times), the transition log on, an action clip set: the same controls as a handle that never arms, step for step."""
    if not armable(m):
        pytest.skip("no large-BAR device: MPPI_TUNE_ARMED_US is unsupported here")
    c = cfg_of(3000, 50, 3)
    ha, hu = m.Handle(tuning={"armed_us": 60, "armed_always": 1}, log_rows=64, **c), m.Handle(**c)
    for h in (ha, hu):
        h.set_action_limits([-0.3] * 3, [0.3] * 3)
    rng = np.random.default_rng(5)
    x = np.zeros(6, F32)
    for i in range(3000):
        ua, uu = ha.next(x), hu.next(x)
        assert np.array_equal(ua, uu), "step %d: %s vs %s" % (i, ua, uu)
        x = plant(x, ua, 3)
        if i % 500 == 499:
            x = rng.standard_normal(6).astype(F32)  # (keep the loop away from its fixed point)
        d = rng.uniform(0.0, 150e-6)
        t0 = time.perf_counter()
        while time.perf_counter() - t0 < d:
            pass
    np.testing.assert_array_equal(ha.get_action_sequence(), hu.get_action_sequence())
    assert ha.get_step_counter() == hu.get_step_counter() == 3000
    ha.close(); hu.close()


def test_armed_launch_is_retired_by_every_other_entry_point(m):
    """set_goal, debug getters, next_with_noise, next_device, destroy — with a launch armed behind the last mppi_next."""
    if not armable(m):
        pytest.skip("no large-BAR device: MPPI_TUNE_ARMED_US is unsupported here")
    import torch
    c = cfg_of(4096, 64, 2)
    ha, hu = m.Handle(tuning={"armed_us": 50000, "armed_always": 1}, **c), m.Handle(**c)
    x = np.zeros(4, F32)
    for h in (ha, hu):
        h.next(x); h.next(x)
        h.set_goal([0.5, 0, -0.5, 0])
        h.next(x)
        eps = np.random.default_rng(1).standard_normal((4096, 64, 2)).astype(F32) * 0.25
        h.next_with_noise(x, eps)
        h.next(x)
        xd, ud = torch.zeros(4, device="cuda"), torch.zeros(2, device="cuda")
        h.next_device(xd.data_ptr(), ud.data_ptr(), torch.cuda.current_stream().cuda_stream or 1)
        torch.cuda.synchronize()
        h.next(x); h.next(x)
    np.testing.assert_array_equal(ha.get_action_sequence(), hu.get_action_sequence())
    assert ha.get_step_counter() == hu.get_step_counter() == 8
    t0 = time.perf_counter()
    ha.next(x)
    ha.close()  # a launch is armed: destroy cancels it instead of waiting out the 50 ms deadline
    assert time.perf_counter() - t0 < 0.04
    hu.close()


def test_armed_adaptive_rule(m):
    """Without armed_always a slow host loop never arms (nothing spins on the GPU between its calls), a fast one does after two calls."""
    if not armable(m):
        pytest.skip("no large-BAR device: MPPI_TUNE_ARMED_US is unsupported here")
    c = cfg_of(4096, 64, 2)
    ha, hu = m.Handle(tuning={"armed_us": 200}, **c), m.Handle(**c)
    x = np.zeros(4, F32)
    for i in range(20):
        ua, uu = ha.next(x), hu.next(x)
        np.testing.assert_array_equal(ua, uu)
        x = plant(x, ua, 2)
        if i < 5:
            time.sleep(0.001)
    np.testing.assert_array_equal(ha.get_action_sequence(), hu.get_action_sequence())
    ha.close(); hu.close()


@pytest.mark.parametrize("K,H,a", [(65536, 64, 3), (4096, 64, 2), (1000, 100, 1), (20000, 64, 4)])
def test_contracted_instance_against_fp64(m, K, H, a):
    """MPPI_FLAG_FP_CONTRACT (k_rollout_pc<.., PC_COST_DIAG_FMA>): fused multiply-adds in the model step and the costs. Not the reference's
    op-by-op rounding, so not bit-identical to the fp32 oracle — held to the oracle's fp64 evaluation on the noise the step drew instead:
    sample costs within 2e-6 relative, or 1.25x the unfused fp32 evaluation's own distance where that is larger (long horizons), U' and u within 1e-5 (north_star's tolerance)."""
    c = cfg_of(K, H, a)
    h = m.Handle(fp_contract=True, **c)
    assert h.rollout_kernel_name().endswith("3, 0>"), h.rollout_kernel_name()
    kw = dict(tau=H, s=2 * a, a=a, dt=0.1, mass=1.0, lam=1.0, sigma=c["sigma"], goal=c["goal"], Q=c["Q"], threads=0)
    p64, p32 = orc.Problem(dtype=np.float64, **kw), orc.Problem(**kw)
    x, U = np.array([0.1, 0, -0.2, 0, 0.3, 0, 0.05, 0][:2 * a], F32), np.zeros((H, a), F32)
    for step in range(3):
        u = h.next(x)
        eps = h.debug_get(m.DBG_NOISE)
        c_dev = h.debug_get(m.DBG_COSTS).astype(np.float64)
        c64 = np.asarray(p64.rollout_cost(x.astype(np.float64), U.astype(np.float64), eps.astype(np.float64)), np.float64)
        rel = np.abs(c_dev - c64) / np.maximum(np.abs(c64), 1e-30)
        u_ref, U_ref, c32 = p32.next_with_noise(x, U, eps)
        rel32 = np.abs(c32.astype(np.float64) - c64) / np.maximum(np.abs(c64), 1e-30)  # what the reference's own unfused fp32 evaluation is off by
        # 2e-6 relative: a 64..100-step fp32 recurrence and ~200 fp32 additions per sample; the unfused fp32 evaluation itself sits at ~1e-6
        # over 65536 samples (measured 1.2e-6 for the contracted one) — the contracted instance must not be worse than that by more than a rounding or two
        # (H = 100, a = 1: both evaluations sit at 2.5e-6 — the bar that means something is the unfused evaluation's own distance)
        assert rel.max() <= max(2e-6, 1.25 * rel32.max()), "contracted costs off the fp64 evaluation by %.3g relative (unfused fp32: %.3g)" % (rel.max(), rel32.max())
        assert not np.array_equal(c_dev.astype(F32), c32) or K < 100, "the contracted instance should differ from the unfused fp32 costs somewhere"
        np.testing.assert_allclose(u, u_ref, rtol=0, atol=U_TOL)
        np.testing.assert_allclose(h.get_action_sequence(), U_ref, rtol=0, atol=U_TOL)
        U = h.get_action_sequence()
        x = plant(x, u, a)
    h.close()


# ---------------------------------------------------------------------------------------- the pre-launched pipelined step (MPPI_TUNE_PRELAUNCH)
def device_steps(h, x, n, a):
    """n pipelined steps on the handle's OWN stream (stream = None: the only place the pre-launched path serves), then drain"""
    import torch
    xd = torch.tensor(np.asarray(x, F32), device="cuda")
    ud = torch.zeros(a, device="cuda")
    torch.cuda.synchronize()
    for _ in range(n):
        h.next_device(xd.data_ptr(), ud.data_ptr(), None)
    h.synchronize()
    torch.cuda.synchronize()
    return ud.cpu().numpy()


PRE_SHAPES = [(65536, 64, 3), (16384, 64, 3), (8256, 32, 2), (40000, 50, 3), (30000, 100, 1), (65536, 72, 2), (20000, 17, 4), (9000, 64, 2),
              (4096, 64, 2), (3000, 50, 3), (128, 32, 1), (8192, 64, 3), (777, 100, 4), (64, 4, 2)]  # (the last six: <= 128 tiles, the one-launch step pre-launched)


@pytest.mark.parametrize("K,H,a", PRE_SHAPES)
def test_prelaunched_step_equals_plain_step(m, K, H, a):
    """Steps alternating between the handle's two streams, each rollout resident before the previous step's U' exists and fed with it as
    granules: controls, U', costs, beta, eta and the step counter are the plain two-launch step's bits — after 1, 2, 3 steps (the start-up
    order), after many, and again after the pipeline has drained and is entered anew."""
    c = cfg_of(K, H, a)
    hp, h0 = m.Handle(tuning={"prelaunch": 1}, **c), m.Handle(**c)
    x = (np.arange(2 * a) % 3 - 1).astype(F32) * F32(0.1)
    total = 0
    for n in (1, 2, 3, 9, 40):
        up, u0 = device_steps(hp, x, n, a), device_steps(h0, x, n, a)
        total += n
        np.testing.assert_array_equal(up, u0, err_msg="after %d steps" % total)
        np.testing.assert_array_equal(hp.get_action_sequence(), h0.get_action_sequence())
        np.testing.assert_array_equal(hp.debug_get(m.DBG_COSTS), h0.debug_get(m.DBG_COSTS))
        assert float(hp.debug_get(m.DBG_BETA)) == float(h0.debug_get(m.DBG_BETA)) and float(hp.debug_get(m.DBG_ETA)) == float(h0.debug_get(m.DBG_ETA))
        assert hp.get_step_counter() == h0.get_step_counter() == total
    hp.close(); h0.close()


def test_prelaunched_step_against_the_oracle(m):
    """The pre-launched step's sample costs against the oracle's on the noise the step drew: bit for bit; U' within 1e-5."""
    K, H, a = 9000, 20, 3
    c = cfg_of(K, H, a)
    h = m.Handle(tuning={"prelaunch": 1}, **c)
    p = orc.Problem(tau=H, s=2 * a, a=a, dt=0.1, mass=1.0, lam=1.0, sigma=c["sigma"], goal=c["goal"], Q=c["Q"])
    x = np.array([0.1, 0, -0.2, 0, 0.3, 0], F32)
    for step in range(3):
        U = h.get_action_sequence().reshape(H, a).copy()
        device_steps(h, x, 1, a)
        eps = h.debug_get(m.DBG_NOISE)  # the noise the last step used
        np.testing.assert_allclose(eps, orc.noise(1, step, 0, K, H, a, c["sigma"]), rtol=0, atol=5e-6)
        u_ref, U_ref, c_ref = p.next_with_noise(x, U, eps)
        np.testing.assert_array_equal(h.debug_get(m.DBG_COSTS), c_ref)
        np.testing.assert_allclose(h.get_action_sequence().reshape(H, a), U_ref, rtol=0, atol=U_TOL)  # (the shifted sequence, as the oracle returns it)
    h.close()


def test_prelaunched_step_mixes_with_every_other_entry_point(m):
    """Any other entry point drains both streams first: host-synchronous steps, injected noise, a new goal, a new sequence between
    pre-launched steps give what the same calls give on a plain handle."""
    K, H, a = 20000, 32, 3
    c = cfg_of(K, H, a)
    hp, h0 = m.Handle(tuning={"prelaunch": 1}, **c), m.Handle(**c)
    x = np.array([0.2, 0, -0.1, 0, 0.0, 0.1], F32)
    rng = np.random.default_rng(0)
    for h in (hp, h0):
        device_steps(h, x, 3, a)
        h.next(x)
        device_steps(h, x, 2, a)
        h.set_goal(np.array([0.5, 0, 0.5, 0, 0.5, 0], F32))
        device_steps(h, x, 2, a)
        h.set_action_sequence(np.full((H, a), 0.05, F32))
        device_steps(h, x, 4, a)
    np.testing.assert_array_equal(hp.get_action_sequence(), h0.get_action_sequence())
    np.testing.assert_array_equal(hp.debug_get(m.DBG_COSTS), h0.debug_get(m.DBG_COSTS))
    assert hp.get_step_counter() == h0.get_step_counter() == 12
    eps = (0.25 * rng.standard_normal((K, H, a))).astype(F32)
    np.testing.assert_array_equal(hp.next_with_noise(x, eps), h0.next_with_noise(x, eps))
    hp.close(); h0.close()


def test_prelaunch_is_refused_where_it_does_not_apply(m):
    """More than one round of the grid would let a waiting grid starve the running one; normalizeCost is not the step's one pass."""
    for K, H, a in ((300000, 64, 3),):
        h = m.Handle(**cfg_of(K, H, a))
        with pytest.raises(m.MppiError):
            h.set_tuning("prelaunch", 1)
        h.close()
    h = m.Handle(normalize_cost=True, **cfg_of(65536, 64, 3))
    with pytest.raises(m.MppiError):
        h.set_tuning("prelaunch", 1)
    h.close()
