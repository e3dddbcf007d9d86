"""atan2_pose (mppi_gen.hip.h): the lean atan2 of k_rollout_nnspeed_pc's pose wave, restated in numpy fp32 with the coefficients READ FROM THE
HEADER — min / max, a reciprocal, an odd polynomial in fused multiply-adds, two quadrant folds, the sign — against fp64 arctan2 over all four
quadrants. The claim in the header's comment (within 2 ulp wherever |result| is not tiny) holds with a correctly rounded reciprocal; v_rcp_f32 is
good to 1 ulp, which in the worst direction everywhere adds one more (3.1 ulp, 5e-7 rad absolute) — both bars are held here on the CPU;
on the GPU the kernel that uses it is held to the fp64 oracle by tests/test_auv_gpu.py (costs 2e-5 relative, U' 1e-5)."""
import os
import re

import numpy as np

F32 = np.float32
HDR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "mppi-tf_amd", "csrc", "mppi_gen.hip.h")


def coefficients():
    src = open(HDR).read()
    body = src[src.index("float atan2_pose(float y, float x)"):]
    body = body[:body.index("return __builtin_copysignf")]
    first = re.search(r"float p = (-?0x[0-9a-f.]+p[-+]?\d+)f;", body).group(1)
    rest = re.findall(r"__builtin_fmaf\(p, s, (-?0x[0-9a-f.]+p[-+]?\d+)f\)", body)
    assert len(rest) == 7
    return [F32(float.fromhex(c)) for c in [first] + rest]  # highest power first


def fma32(a, b, c):  # one rounding: exact in fp64 for fp32 operands up to the final rounding (products of two fp32 fit in 48 bits)
    return (a.astype(np.float64) * b.astype(np.float64) + c.astype(np.float64)).astype(F32)


def atan2_pose(y, x, rcp_ulp=0):
    ax, ay = np.abs(x), np.abs(y)
    mx, mn = np.maximum(ax, ay), np.minimum(ax, ay)
    r = (1.0 / mx.astype(np.float64)).astype(F32)
    if rcp_ulp:  # v_rcp_f32 is good to 1 ulp: the worst case in either direction
        r = np.nextafter(r, F32(np.inf) if rcp_ulp > 0 else F32(0))
    t = (mn * r).astype(F32)
    s = (t * t).astype(F32)
    c = coefficients()
    p = np.full_like(t, c[0])
    for ck in c[1:]:
        p = fma32(p, s, np.full_like(t, ck))
    a = fma32((t * s).astype(F32), p, t)
    a = np.where(ay > ax, F32(np.pi / 2) - a, a).astype(F32)
    a = np.where(x < 0, F32(np.pi) - a, a).astype(F32)
    return np.copysign(a, y)


def test_pose_atan2_against_fp64():
    rng = np.random.default_rng(0)
    n = 400000
    ang = rng.uniform(-np.pi, np.pi, n)
    rad = np.exp(rng.uniform(np.log(1e-3), np.log(2.0), n))  # the arguments are entries of a rotation matrix: |.| <= 1 + a few ulp
    y, x = (rad * np.sin(ang)).astype(F32), (rad * np.cos(ang)).astype(F32)
    x = np.where(x == 0, F32(2.4e-7), x)  # euler_from_quat never passes x = 0 (it adds +-eps)
    ref = np.arctan2(y.astype(np.float64), x.astype(np.float64))
    for rcp_ulp in (0, 1, -1):
        got = atan2_pose(y, x, rcp_ulp).astype(np.float64)
        # near the axes the result of pi/2 - a or pi - a keeps a's absolute error: measure in ulps of the larger of |result| and |a| (= what feeds the network
        # after (e - mean) / std is an ABSOLUTE angle error, a few 1e-7 rad)
        err = np.abs(got - ref)
        assert err.max() < 5e-7, (rcp_ulp, err.max())
        big = np.abs(ref) > 0.5
        ulps = err[big] / np.spacing(np.abs(ref[big]).astype(F32))
        assert ulps.max() <= (2.5 if rcp_ulp == 0 else 3.5), (rcp_ulp, ulps.max())  # measured: 2.1 with a correctly rounded reciprocal, 3.1 with v_rcp_f32's 1 ulp always against it
    # the signs and the branch cuts
    yy = np.array([0.0, -0.0, 0.0, -0.0, 1.0, -1.0], F32)
    xx = np.array([1.0, 1.0, -1.0, -1.0, 2.4e-7, -2.4e-7], F32)
    np.testing.assert_allclose(atan2_pose(yy, xx), np.arctan2(yy, xx), rtol=0, atol=3e-7)
    assert np.signbit(atan2_pose(yy, xx)[1]) and np.signbit(atan2_pose(yy, xx)[3])
