#!/usr/bin/env python3
"""Writes tests/golden/*.json — the reference's own known-answer vectors, as DATA.

Every number below was transcribed by hand from the literals in the reference's unit tests
(read as text; the reference cannot be compiled or imported here — SURVEY.md §8c):
  C++  test/test_controller.cpp, test/test_cost.cpp, test/test_model.cpp, test/test_utile.cpp
  Py   scripts/test.py (TestPointMassModel, TestAUVModel, TestNNAUVModel, TestCost, TestStaticCost, TestElipseCost,
       TestElipse3DCost, TestController)
Where the reference test spells an expectation as an arithmetic expression of literals
(e.g. `(dt*dt)/(2.f*m)`), the same expression is evaluated here in the same precision
(np.float32 for the C++ tests, python float = fp64 for scripts/test.py) and the RESULT is stored.
No reference source text is kept; only inputs and expected outputs.

Run:  python tests/golden/transcribe_reference_vectors.py      (idempotent)
"""
import json
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
f32 = np.float32


def dump(name, obj):
    def conv(o):
        if isinstance(o, np.ndarray):
            return o.tolist()
        if isinstance(o, (np.floating,)):
            return float(o)
        if isinstance(o, (np.integer,)):
            return int(o)
        if isinstance(o, dict):
            return {k: conv(v) for k, v in o.items()}
        if isinstance(o, (list, tuple)):
            return [conv(v) for v in o]
        return o
    with open(os.path.join(HERE, name + ".json"), "w") as fh:
        json.dump(conv(obj), fh, indent=1, sort_keys=True)
        fh.write("\n")


# ------------------------------------------------------------------ controller (A3, A8-A10)
# test_controller.cpp:18-31 fixture (k=5, tau=3, a_dim=2); scripts/test.py:1377-1388 identical.
COST = [3., 10., 0., 1., 5.]
NOISE = [[[1., -0.5], [1., -0.5], [2., 1.]],
         [[0.3, 0.], [2., 0.2], [1.2, 3.]],
         [[0.5, 0.5], [0.5, 0.5], [0.5, 0.5]],
         [[0.6, 0.7], [0.2, -0.3], [0.1, -0.4]],
         [[-2., -3.], [-4., -1.], [0., 0.]]]
ACTION = [[1., 0.5], [2.3, 4.5], [2.1, -0.4]]


def controller():
    # test_controller.cpp:109-167 testUpdate / scripts/test.py:1427-1468
    w = [0.034951787275480706, 3.1871904480408675e-05, 0.7020254138530686,
         0.2582607169364174, 0.004730210030553017]
    n = np.array(NOISE)
    wn = [[sum(w[k] * n[k, t, a] for k in range(5)) for a in range(2)] for t in range(3)]
    dump("controller_update_k5_tau3_a2", dict(
        source="test/test_controller.cpp:24-31,109-167; scripts/test.py:1377-1388,1427-1468",
        tol="EXPECT_FLOAT_EQ (4 ulp fp32) / assertAllClose rtol=atol=1e-6",
        k=5, tau=3, a=2, lam=1.0, cost=COST, noise=NOISE, action=ACTION,
        beta=0.0, exp_arg=[-3., -10., 0., -1., -5.],
        exp=[0.049787068367863944, 4.5399929762484854e-05, 1.0, 0.36787944117144233,
             0.006737946999085467],
        nabla=1.424449856468154, weights=w, weighted_noise=wn, sum_weights=1.0))
    # test_controller.cpp:71-107 testDataPrep / scripts/test.py:1400-1425
    dump("controller_dataprep", dict(
        source="test/test_controller.cpp:71-107; scripts/test.py:1400-1425",
        noise=NOISE, action=ACTION,
        a=[[1., 0.5], [2.3, 4.5], [2.1, -0.4]],
        n=[[[1., -0.5], [0.3, 0.], [0.5, 0.5], [0.6, 0.7], [-2., -3.]],
           [[1., -0.5], [2., 0.2], [0.5, 0.5], [0.2, -0.3], [-4., -1.]],
           [[2., 1.], [1.2, 3.], [0.5, 0.5], [0.1, -0.4], [0., 0.]]]))
    # test_controller.cpp:169-193 testNew, :195-222 testShiftAndInit / test.py:1470-1494
    dump("controller_getnew_shift", dict(
        source="test/test_controller.cpp:169-222; scripts/test.py:1470-1494",
        action=ACTION,
        getnew={"0": [], "1": [[1., 0.5]], "2": [[1., 0.5], [2.3, 4.5]],
                "3": [[1., 0.5], [2.3, 4.5], [2.1, -0.4]]},
        shift=[dict(init=[[1., 0.5]], nb=1, expected=[[2.3, 4.5], [2.1, -0.4], [1., 0.5]]),
               dict(init=[[1., 0.5], [2.3, 4.5]], nb=2,
                    expected=[[2.1, -0.4], [1., 0.5], [2.3, 4.5]])]))


# ------------------------------------------------------------------ cost, C++ form (A5-A7)
def cost_cpp():
    # test_cost.cpp:27-114 fixtures, :169-239 expectations. lambda = 1, Q given as a diagonal.
    sc = [
        dict(name="s2_a2_k1", s=2, a=2, k=1, lam=1.0,
             state=[[0., 1.]], goal=[1., 1.], action=[1., 1.], eps=[[1., 1.]],
             sigma=[[1., 0.], [0., 1.]], q_diag=[1., 1.],
             exp_state=[1.], exp_step=[3.]),
        dict(name="s4_a2_k1", s=4, a=2, k=1, lam=1.0,
             state=[[0., 0.5, 2., 0.]], goal=[1., 1., 1., 2.], action=[0.5, 2.], eps=[[0.5, 1.]],
             sigma=[[1., 0.], [0., 1.]], q_diag=[1., 1., 10., 10.],
             exp_state=[51.25], exp_step=[53.5]),
        dict(name="s4_a3_k5", s=4, a=3, k=5, lam=1.0,
             state=[[0., 0.5, 2., 0.], [0., 2., 0., 0.], [10., 2., 2., 3.], [1., 1., 1., 2.],
                    [3., 4., 5., 6.]],
             goal=[1., 1., 1., 2.], action=[0.5, 2., 0.25],
             eps=[[0.5, 1., 2.], [0.5, 2., 0.25], [-2., -0.2, -1.], [0., 0., 0.], [1., 0.5, 3.]],
             sigma=[[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]], q_diag=[1., 1., 10., 10.],
             exp_state=[51.25, 52., 102., 0., 333.],
             exp_step=[51.25 + 2.75, 52 + 4.3125, 102 - 1.65, 0. + 0, 333 + 2.25]),
    ]
    dump("cost_cpp", dict(source="test/test_cost.cpp:27-114,169-239",
                          tol="EXPECT_FLOAT_EQ (4 ulp fp32)", scenarios=sc))


# ------------------------------------------------------------------ model, C++ (A4) dt=0.01
STATE3 = [[0., 0., 0., 0., 0., 0.], [2., 1., 5., 0., -1., -2.], [0.5] * 6,
          [1., 0., 1., 0., 1., 0.], [-1., 0.5, -3., 2., 0., 0.]]
ACTION3 = [[1., 1., 1.], [2., 0., -1.], [0., 0., 0.], [0.5, -0.5, 0.5], [3., 3., 3.]]
STATE_INIT = [[-1., 0.5, -3., 2., 0., 0.]]
U_COEF = [[1., 1., 1.], [2., 0., -1.], [0., 0., 0.], [0.5, -0.5, 0.5], [3., 3., 3.]]


def model_cpp():
    # test_model.cpp:17-75 fixture; :120-255 expectations, evaluated in float like the test.
    out = []

    def exp_u(m, dt, coef):
        acc = (f32(dt) * f32(dt)) / (f32(2.) * f32(m))
        vel = f32(dt) / f32(m)
        return [[v for c in row for v in (f32(c) * acc, f32(c) * vel)] for row in coef]

    dt = 0.01
    out.append(dict(name="step1_k1_s2_a1", s=2, a=1, mass=1.0, dt=dt, state=[[0., 0.]],
                    action=[[1.]], exp_free=[[0., 0.]], exp_action=exp_u(1., dt, [[1.]])))
    out.append(dict(name="step2_k1_s4_a2", s=4, a=2, mass=2.0, dt=dt, state=[[0.] * 4],
                    action=[[1., 1.]], exp_free=[[0.] * 4], exp_action=exp_u(2., dt, [[1., 1.]])))
    d = f32(dt)
    free3 = [[0.] * 6,
             [f32(2.) + d, 1., 5., 0., f32(-1.) - f32(2.) * d, -2.],
             [f32(.5) + d / f32(2.), .5, f32(.5) + d / f32(2.), .5, f32(.5) + d / f32(2.), .5],
             [1., 0., 1., 0., 1., 0.],
             [f32(-1.) + d / f32(2.), .5, f32(-3.) + f32(2.) * d, 2., 0., 0.]]
    out.append(dict(name="large_k5_s6_a3", s=6, a=3, mass=1.5, dt=dt, state=STATE3,
                    action=ACTION3, exp_free=free3, exp_action=exp_u(1.5, dt, U_COEF)))
    out.append(dict(name="init_k5_s6_a3", s=6, a=3, mass=1.5, dt=dt, state=STATE_INIT,
                    action=ACTION3, exp_free=[free3[4]], exp_action=exp_u(1.5, dt, U_COEF)))
    for sc in out:
        fr, ac = np.array(sc["exp_free"], f32), np.array(sc["exp_action"], f32)
        sc["exp_result"] = (ac + fr).astype(f32)  # broadcast when the state has one row
    dump("model_cpp", dict(source="test/test_model.cpp:17-75,120-255",
                           tol="EXPECT_FLOAT_EQ (4 ulp fp32)", scenarios=out))


# ------------------------------------------------------------------ model, Py (A4) dt=0.1 fp64
def model_py():
    # scripts/test.py:43-218 (TestPointMassModel), fp64, assertAllClose 1e-6.
    dt = 0.1
    out = []

    def exp_u(m, coef):
        acc = dt * dt / (2. * m)
        vel = dt / m
        return [[v for c in row for v in (c * acc, c * vel)] for row in coef]

    out.append(dict(name="step1_k1_s2_a1_m1", s=2, a=1, mass=1.0, dt=dt, state=[[0., 0.]],
                    action=[[1.]], exp_free=[[0., 0.]], exp_action=exp_u(1., [[1.]])))
    out.append(dict(name="step1_k1_s4_a2_m1", s=4, a=2, mass=1.0, dt=dt, state=[[0.] * 4],
                    action=[[1., 1.]], exp_free=[[0.] * 4], exp_action=exp_u(1., [[1., 1.]])))
    free3 = [[0.] * 6,
             [2. + dt, 1., 5., 0., -1. - 2. * dt, -2.],
             [.5 + .5 * dt, .5, .5 + .5 * dt, .5, .5 + .5 * dt, .5],
             [1., 0., 1., 0., 1., 0.],
             [-1. + .5 * dt, .5, -3. + 2. * dt, 2., 0., 0.]]
    out.append(dict(name="step1_k5_s6_a3_m1d5", s=6, a=3, mass=1.5, dt=dt, state=STATE3,
                    action=ACTION3, exp_free=free3, exp_action=exp_u(1.5, U_COEF)))
    out.append(dict(name="init_k5_s6_a3_m1d5", s=6, a=3, mass=1.5, dt=dt, state=STATE_INIT,
                    action=ACTION3, exp_free=[free3[4]], exp_action=exp_u(1.5, U_COEF)))
    for sc in out:
        sc["exp_result"] = (np.array(sc["exp_action"]) + np.array(sc["exp_free"]))
    # scripts/test.py:173-218 test_step3: three consecutive steps with the same action.
    m = 1.5
    acc, vel = dt * dt / (2. * m), dt / m
    Bu = [[v for c in row for v in (c * 3 * (acc + vel * dt), c * vel * 3)] for row in U_COEF]
    ex = [[0.] * 6,
          [2. + dt * 3, 1., 5., 0., -1. - 2. * dt * 3, -2.],
          [.5 + .5 * dt * 3, .5, .5 + .5 * dt * 3, .5, .5 + .5 * dt * 3, .5],
          [1., 0., 1., 0., 1., 0.],
          [-1. + .5 * dt * 3, .5, -3. + 2. * dt * 3, 2., 0., 0.]]
    step3 = dict(name="step3_k5_s6_a3_m1d5", s=6, a=3, mass=m, dt=dt, state=STATE3,
                 action=ACTION3, n_steps=3, exp_result=(np.array(Bu) + np.array(ex)))
    dump("model_py", dict(source="scripts/test.py:43-218", tol="assertAllClose rtol=atol=1e-6",
                          scenarios=out, step3=step3))


# ------------------------------------------------------------------ blockDiag (A4 layout)
def blockdiag():
    # test_utile.cpp:15-27 fixture (m=1.5, dt=0.01), :63-173 expectations.
    m, dt = f32(1.5), f32(0.01)
    A = [[f32(1.), dt], [f32(0.), f32(1.)]]
    B = [[(dt * dt) / (f32(2.) * m)], [dt / m]]
    cases = []
    for n in (1, 2, 3, 4):
        ea = np.zeros((2 * n, 2 * n), f32)
        eb = np.zeros((2 * n, n), f32)
        for b in range(n):
            ea[2 * b:2 * b + 2, 2 * b:2 * b + 2] = A
            eb[2 * b:2 * b + 2, b:b + 1] = B
        cases.append(dict(n=n, exp_a=ea, exp_b=eb))
    dump("blockdiag", dict(source="test/test_utile.cpp:15-27,63-173", A=np.array(A, f32),
                           B=np.array(B, f32), cases=cases))


# ------------------------------------------------------------------ cost, Py γ/υ form (A5-A7)
def cost_elipse():
    # scripts/test.py:1098-1161 TestElipseCost: a = b = 1, centre (0, 0), speed 1, m_state = m_vel = 1;
    # states are [k, 4, 1] = (x, vx, y, vy); expectations are the test's literals / literal sums.
    par = dict(a=1., b=1., cx=0., cy=0., speed=1., m_state=1., m_vel=1.)
    dump("cost_elipse", dict(
        source="scripts/test.py:1098-1161 (TestElipseCost)", tol="assertAllClose rtol=atol=1e-6", ellipse=par,
        scenarios=[
            dict(name="testStepElipseCost_s4_l1_k1", state=[[0., 0.5, 1., 0.]], exp_state_cost=[0.25]),
            dict(name="testStepElipseCost_s4_l1_k5",
                 state=[[0., 0.5, 1., 0.], [0., 2., 0., 0.], [10., 2., 2., 3.], [1., 1., 1., 2.], [3., 4., 5., 6.]],
                 exp_state_cost=[0.25, 2., 103 + 6.788897449072021, 1 + 1.5278640450004208, 33 + 38.57779489814404]),
        ]))


def cost_py():
    S4 = [[0., 0.5, 2., 0.], [0., 2., 0., 0.], [10., 2., 2., 3.], [1., 1., 1., 2.], [3., 4., 5., 6.]]
    E3 = [[0.5, 1., 2.], [0.5, 2., 0.25], [-2., -0.2, -1.], [0., 0., 0.], [1., 0.5, 3.]]
    I2, I3 = np.eye(2).tolist(), np.eye(3).tolist()
    Q4 = np.diag([1., 1., 10., 10.]).tolist()

    def ac(gamma, lam, ups, aa, mix, nn, ups_form="inv"):
        # TestCost spells λ(1-1/υ); TestStaticCost's first three spell λ(1-υ) (same at υ=1).
        f = (1 - 1. / ups) if ups_form == "inv" else (1 - ups)
        return [0.5 * (gamma * (aa + 2. * m) + lam * f * n) for m, n in zip(mix, nn)]

    mix3, nn3 = [2.75, 4.3125, -1.65, 0., 2.25], [5.25, 4.3125, 5.04, 0., 10.25]
    base = []
    # scripts/test.py:689-838 TestCost (CostBase.action_cost only)
    base.append(dict(name="s2_a2_l1", a=2, lam=1., gamma=1., upsilon=1., sigma=I2,
                     action=[1., 1.], noise=[[1., 1.]], exp_action=ac(1., 1., 1., 2., [2.], [0.])))
    base.append(dict(name="s4_a2_l1", a=2, lam=1., gamma=1., upsilon=1., sigma=I2,
                     action=[0.5, 2.], noise=[[0.5, 1.]], exp_action=ac(1., 1., 1., 4.25, [2.25], [1.25])))
    for lam, g, u in ((1., 1., 1.), (10., 2., 3.), (15., 20., 30.)):
        base.append(dict(name="s4_a3_l%g_g%g_u%g" % (lam, g, u), a=3, lam=lam, gamma=g, upsilon=u,
                         sigma=I3, action=[0.5, 2., 0.25], noise=E3,
                         exp_action=ac(g, lam, u, 4.3125, mix3, nn3)))
    # scripts/test.py:841-1096 TestStaticCost (state + action, Q full matrix)
    static = []
    static.append(dict(name="s2_a2_l1", s=2, a=2, lam=1., gamma=1., upsilon=1., sigma=I2,
                       Q=np.eye(2).tolist(), goal=[1., 1.], state=[[0., 1.]], action=[1., 1.],
                       noise=[[1., 1.]], exp_action=ac(1., 1., 1., 2., [2.], [0.], "lin"),
                       exp_state=[1.]))
    static.append(dict(name="s4_a2_l1", s=4, a=2, lam=1., gamma=1., upsilon=1., sigma=I2, Q=Q4,
                       goal=[1., 1., 1., 2.], state=[[0., 0.5, 2., 0.]], action=[0.5, 2.],
                       noise=[[0.5, 1.]], exp_action=ac(1., 1., 1., 4.25, [2.25], [1.25], "lin"),
                       exp_state=[51.25]))
    static.append(dict(name="s4_a3_l1", s=4, a=3, lam=1., gamma=1., upsilon=1., sigma=I3, Q=Q4,
                       goal=[1., 1., 1., 2.], state=S4, action=[0.5, 2., 0.25], noise=E3,
                       exp_action=ac(1., 1., 1., 4.3125, mix3, nn3, "lin"),
                       exp_state=[51.25, 52., 102., 0., 333.]))
    static.append(dict(name="s13_a6_l1", s=13, a=6, lam=1., gamma=1., upsilon=1.,
                       sigma=np.eye(6).tolist(),
                       Q=np.diag([1.] * 7 + [10.] * 6).tolist(),
                       goal=[1., 1., 2., 0., 0., 0., 1., 0., 0., 0., 0., 0., 0.],
                       state=[[0., 0.5, 2., 0., 0., 0., 1., 1., 2., 3., 4., 5., 6.],
                              [0., 2., 0., 0., 0.5, 0.5, 0., 4., 5., 6., 1., 2., 3.]],
                       action=[0.5, 2., 0.25, 4., 1., 1.5],
                       noise=[[0.5, 1., 2., 3., 4., 5.], [0.5, 2., 0.25, 1.25, 2.5, 0.75]],
                       exp_action=ac(1., 1., 1., 23.5625, [26.25, 12.9375], [55.25, 12.6875]),
                       exp_state=[911.25, 917.5]))
    for sc in static:
        sc["exp_step"] = (np.array(sc["exp_action"]) + np.array(sc["exp_state"]))
    dump("cost_py", dict(source="scripts/test.py:685-1096", tol="assertAllClose rtol=atol=1e-6",
                         action_cost=base, static_cost=static))


# ------------------------------------------------------------------ AUVModel, NNAUVModel, ElipseCost3D
# scripts/test.py:237-262 TestAUVModel.setUp: the model parameters every test below uses (dt = 0.1, rk = 2)
AUV_PARAMS = dict(mass=1000., volume=1.5, density=1000., height=1.6, length=2.5, width=1.5, cog=[0., 0., 0.], cob=[0., 0., 0.5],
                  Ma=(500. * np.eye(6)).tolist(), linear_damping=[-70., -70., -700., -300., -300., -100.],
                  quad_damping=[-740., -990., -1800., -670., -770., -520.], linear_damping_forward_speed=[1., 2., 3., 4., 5., 6.],
                  inertial=dict(ixx=650., iyy=750., izz=550., ixy=1., ixz=2., iyz=3.), rk=2, dt=0.1)


def model_auv():
    # test_B2I_transform_and_jacobian (scripts/test.py:264-402): three pose quaternions (x, y, z, w), the rotation matrices
    # "from lib" (literals, 7 digits), and T_q = 0.5 [[w,-z,y],[z,w,-x],[-y,x,w],[-x,-y,-z]] (the test's exp_TB2Iquat rows
    # written in the model's row order rxt, ryt, rzt, rwt — auv_model.py:388-396; the test fills row 0 with (-x,-y,-z), which
    # is the model's LAST row: the test as written does not hold against the model it tests; the model is the specification)
    quats = [[0., 0., 0., 1.],
             [0.0438308910967523, 0.25508068761447, 0.171880267220619, 0.950510320581509],
             [-0.111618880991033, 0.633022223770408, 0.492403876367579, 0.586824089619078]]
    rot_from_lib = [[[1., 0., 0.], [0., 1., 0.], [0., 0., 1.]],
                    [[0.8107820, -0.3043871, 0.4999810], [0.3491088, 0.9370720, 0.0043632], [-0.4698463, 0.1710101, 0.8660254]],
                    [[-0.2863574, -0.7192234, 0.6330222], [0.4365945, 0.4901593, 0.7544065], [-0.8528685, 0.4924039, 0.1736482]]]
    # test_restoring (:404-487): two poses given as quaternions AND as roll/pitch/yaw; the expectation is built from the Euler
    # rotation matrix: W = m g, B = V rho g, f_g = R^T (0,0,-W), f_b = R^T (0,0,B), g = -(f_g + f_b, r_g x f_g + r_b x f_b)
    rq = [[-0.1127657, 0.8086476, 0.0328141, 0.5764513], [-0.4582488, 0.4839407, 0.0503092, 0.7438269]]
    rpy = np.array([[13., 110., 25.], [280., 50., 325.]]) * (np.pi / 180.)
    W, B = 1000. * 9.81, 1.5 * 1000. * 9.81
    rest, rots = [], []
    for roll, pitch, yaw in rpy:
        cr, sr, cp, sp, cy, sy = np.cos(roll), np.sin(roll), np.cos(pitch), np.sin(pitch), np.cos(yaw), np.sin(yaw)
        R = np.array([[cy * cp, -sy * cr + cy * sp * sr, sy * sr + cy * cr * sp],
                      [sy * cp, cy * cr + sr * sp * sy, -cy * sr + sp * sy * cr],
                      [-sp, cp * sr, cp * cr]])
        fbg, fbb = R.T @ np.array([0., 0., -W]), R.T @ np.array([0., 0., B])
        mbg, mbb = np.cross(AUV_PARAMS["cog"], fbg), np.cross(AUV_PARAMS["cob"], fbb)
        rest.append(np.concatenate([-(fbb + fbg), -(mbb + mbg)]))
        rots.append(R)
    # test_damping (:489-503): D = -diag(lin) - v0 diag(fwd) + (-diag(quad)) |v| (the last product broadcast over the rows)
    dvel = [[1., 1., 1., 1., 1., 1.], [2., 1.5, 1., 3., 3.5, 2.5], [-2., -1.5, -1., -3., -3.5, -2.5]]
    damp = []
    for v in dvel:
        D = -np.diag(AUV_PARAMS["linear_damping"]) - v[0] * np.diag(AUV_PARAMS["linear_damping_forward_speed"])
        damp.append(D + (-np.diag(AUV_PARAMS["quad_damping"])) * np.abs(np.array(v))[:, None])
    # test_corrolis (:505-539): vel = (1,1,1,0,0,0); C = C_rb + C_a with m = 1000, Ma = 500 I
    cv = [1., 1., 1., 0., 0., 0.]
    m, Mav = 1000., -(500. * np.array(cv))
    crb = np.array([[0., 0., 0., 0., m * cv[2], -m * cv[1]], [0., 0., 0., -m * cv[2], 0., m * cv[0]], [0., 0., 0., m * cv[1], -m * cv[0], 0.],
                    [0., m * cv[2], -m * cv[1], 0., 0., 0.], [-m * cv[2], 0., m * cv[0], 0., 0., 0.], [m * cv[1], -m * cv[0], 0., 0., 0., 0.]])
    ca = np.array([[0., 0., 0., 0., -Mav[2], Mav[1]], [0., 0., 0., Mav[2], 0., -Mav[0]], [0., 0., 0., -Mav[1], Mav[0], 0.],
                   [0., -Mav[2], Mav[1], 0., -Mav[5], Mav[4]], [Mav[2], 0., -Mav[0], Mav[5], 0., -Mav[3]], [-Mav[1], Mav[0], 0., -Mav[4], Mav[3], 0.]])
    # test_step1_k1 / test_step1_k5 (:541-586) only print: inputs kept (no expectation in the reference)
    step_states = [[0., 0., 0., 0., 0., 0., 1., 0., 0., 0., 0., 0., 0.], [1., 1., 1., 0., 0., 0., 1., 0.1, 2., 2., 1., 2., 3.],
                   [0., 2., 1., 0.2, 0.3, 0., 1., -1., -1., -1., -1., -1., -1.], [5., 0.2, 0., 1.2, 0., 3.1, 1., 0., 0., 0., 0., 0., 0.],
                   [0., 0., 0., 0., 0., 0., 1., 1., 1., 1., 1., 1., 1.]]
    step_actions = [[1.] * 6, [1.] * 6, [-1.] * 6, [2.] * 6, [-1.] * 6]
    dump("model_auv", dict(source="scripts/test.py:237-586 TestAUVModel", tol="assertAllClose rtol=atol=1e-6", params=AUV_PARAMS,
                           b2i=dict(quat=quats, rot_from_lib=rot_from_lib),
                           restoring=dict(quat=rq, exp_rot=rots, exp_restoring=rest),
                           damping=dict(vel=dvel, exp=damp), coriolis=dict(vel=cv, exp=crb + ca),
                           step_inputs=dict(state=step_states, action=step_actions)))


def model_nnauv():
    # scripts/test.py:587-684 TestNNAUVModel (identity normalisation: Xmean = Ymean = 0, Xstd = Ystd = 1)
    t1 = dict(state_t=[[1., 1., .5, 0., 0., 0., 1., 1., 0., .25, 0., 0., 0.]], state_t1=[[2., 1., .75, 0., 0., 0., 1., 3., 3.5, 4.5, 5.5, 6.5, 7.5]],
              action=[[1., 2., 3., 4., 5., 6.]],
              exp_x=[[0., 0., 0., 1., 1., 0., .25, 0., 0., 0., 1., 2., 3., 4., 5., 6.]],
              exp_y=[[1., 0., .25, 0., 0., 0., 0., 2., 3.5, 4.25, 5.5, 6.5, 7.5]])
    n1 = dict(state=[[float(i) for i in range(13)]], action=[[13., 14., 15., 16., 17., 18.]], exp=[[float(i) for i in range(3, 19)]])
    st = [[float(i) for i in range(13)], [18. - i for i in range(13)], [-float(i) for i in range(13)], [-(18. - i) for i in range(13)],
          [-(i + .5) for i in range(13)], [i + .5 for i in range(13)]]
    ac = [[13., 14., 15., 16., 17., 18.], [5., 4., 3., 2., 1., 0.], [-13., -14., -15., -16., -17., -18.], [-5., -4., -3., -2., -1., -0.],
          [-13.5, -14.5, -15.5, -16.5, -17.5, -18.5], [13.5, 14.5, 15.5, 16.5, 17.5, 18.5]]
    exp = [[float(i) for i in range(3, 19)], [15. - i for i in range(16)], [-float(i) for i in range(3, 19)], [-(15. - i) for i in range(16)],
           [-(i + .5) for i in range(3, 19)], [i + .5 for i in range(3, 19)]]
    dump("model_nnauv", dict(source="scripts/test.py:587-684 TestNNAUVModel", training_n1=t1, prepare_n1=n1,
                             prepare_n6=dict(state=st, action=ac, exp=exp)))


def cost_elipse3d():
    # scripts/test.py:1164-1360 TestElipse3DCost: axis = (2, 1.5), speed = 1, m_state = m_vel = 1, lambda = gamma = upsilon = 1
    base = dict(axis=[2., 1.5], speed=1., m_state=1., m_vel=1.)
    prep = [dict(normal=[0., 0., 1.], aVec=[1., 0., 0.], center=[0., 0., 0.], exp_R=np.eye(3)),
            dict(normal=[0., 1., 1.], aVec=[1., 0., 0.], center=[0., 1., -2.], exp_R=np.array([[1., 0., 0.], [0., .5, -.5], [0., .5, .5]]).T)]
    plane = dict(normal=[0., 1., 1.], aVec=[1., 0., 0.], center=[0., 1., -2.])
    pos = dict(position=[[.1, .4, .2], [1., 1., -2.], [2., 1., 0.]], exp=[0.8863888888888889, 3.6944444444444446, 0.4444444444444444])
    ori = dict(pose=[[.1, .4, .2, 0., 0., 0., 1.], [1., 1., -2., 0.48038446, 0.32025631, 0.16012815, 0.80064077],
                     [2., 1., -2., 0.20628425, -0.30942637, -0.92827912, 0.]],
               exp=[3.0018837793006306, 2.4098026419889416, 1.1216620246733544])
    vel = dict(velocity=[[.1, .4, .2, 0., 0., 0.], [1., 1., -2., .3, .2, .1], [2., 1., -2., .2, -.3, -.9]],
               exp=[abs(0.21 - 1), abs(6 - 1), abs(9 - 1)])
    # test_state_cost (:1317-1338) has no expectation; inputs kept
    states = [[.1, .4, .2, 0., 0., 0., 1., .3, .7, 2., 1., 2.4, 5.], [1., 1., -2., .3, .2, .1, .5, .4, 2.7, 2., 0., 0., 0.],
              [2., 1., -2., .2, -.3, -.9, 0., 2.3, 1.7, 0., .1, .4, .01]]
    # test_tf_rot (:1340-1359): the two tensorflow_graphics quaternion calls the cost makes
    tf_rot = dict(q=[0., 0.7071068, 0., 0.7071068], position=[1., 2., 3.], quat=[0.7071068, 0., 0., 0.7071068],
                  exp_pos=[3., 2., -1.], exp_quat=[.5, .5, -.5, .5])
    dump("cost_elipse3d", dict(source="scripts/test.py:1164-1360 TestElipse3DCost", tol="assertAllClose rtol=atol=1e-6", base=base,
                               prep_const=prep, plane=plane, position_error=pos, orientation_error=ori, velocity_error=vel,
                               state_cost_inputs=states, tf_rot=tf_rot))


if __name__ == "__main__":
    controller()
    cost_cpp()
    model_cpp()
    model_py()
    blockdiag()
    cost_py()
    cost_elipse()
    model_auv()
    model_nnauv()
    cost_elipse3d()
    print("wrote", sorted(f for f in os.listdir(HERE) if f.endswith(".json")))
