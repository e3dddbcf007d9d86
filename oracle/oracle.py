"""ctypes/numpy front-end of the CPU oracle (oracle/mppi_oracle.c).

TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
`cpu_baseline` leg may import this module. Nothing under mppi-tf_amd/ does.

The arithmetic lives in C (each C function cites the reference file:line it restates); this
file only marshals numpy arrays.  `dtype` selects the fp32 (parity target, C++ reference is
DT_FLOAT) or fp64 (Python reference / truth bound) instantiation.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libmppi_oracle.so")

MAX_LAYERS = 8
ACTION_COST_CPP, ACTION_COST_PY = 0, 1
MODEL_POINT_MASS, MODEL_MLP, MODEL_AUV, MODEL_NNAUV, MODEL_NNAUV_SPEED = 0, 1, 2, 3, 4
STATE_COST_QUADRATIC, STATE_COST_ELLIPSE, STATE_COST_QUAT, STATE_COST_ELLIPSE3D = 0, 1, 2, 3


def build(force=False):
    """Compile the oracle with gcc (recipe: oracle/Makefile)."""
    srcs = [os.path.join(_HERE, f) for f in
            ("mppi_oracle.c", "mppi_oracle_impl.inc", "mppi_oracle_auv.inc", "mppi_oracle_decl.inc", "mppi_oracle.h", "Makefile")]
    if (not force and os.path.exists(_SO)
            and all(os.path.getmtime(_SO) >= os.path.getmtime(s) for s in srcs)):
        return _SO
    subprocess.check_call(["make", "-C", _HERE, "-B", "libmppi_oracle.so"],
                          stdout=subprocess.DEVNULL)
    return _SO


def _structs(real):
    P = C.POINTER(real)

    class Ellipse3d(C.Structure):
        _fields_ = [("q", real * 4), ("axis", real * 3), ("mapping", real * 3), ("gv", real), ("mS", real), ("mV", real)]

    class Auv(C.Structure):
        _fields_ = [("mass", real), ("volume", real), ("density", real), ("gravity", real), ("dt", real), ("rk", C.c_int),
                    ("cog", real * 3), ("cob", real * 3), ("mtot", real * 36), ("inv_mtot", real * 36),
                    ("lin_damp", real * 36), ("lin_damp_fwd", real * 36), ("quad_damp", real * 6)]

    class Cost(C.Structure):
        _fields_ = [("s", C.c_int), ("a", C.c_int), ("action_cost_kind", C.c_int),
                    ("lam", real), ("gamma", real), ("upsilon", real),
                    ("goal", P), ("Q", P), ("sigma_inv", P),
                    ("state_cost_kind", C.c_int), ("ellipse", real * 7), ("e3", Ellipse3d)]

    class Mlp(C.Structure):
        _fields_ = [("s", C.c_int), ("a", C.c_int), ("n_layers", C.c_int),
                    ("widths", C.c_int * MAX_LAYERS),
                    ("W", P * MAX_LAYERS), ("b", P * MAX_LAYERS),
                    ("xmean", P), ("xstd", P), ("ymean", P), ("ystd", P)]

    class Problem(C.Structure):
        _fields_ = [("cost", Cost), ("tau", C.c_int), ("model_kind", C.c_int),
                    ("threads", C.c_int), ("A", P), ("B", P), ("mlp", Mlp), ("auv", Auv)]

    return Cost, Mlp, Problem, Ellipse3d, Auv


class _Inst:
    """One precision instantiation of the C oracle."""

    def __init__(self, lib, suffix, real, npdt):
        self.lib, self.suffix, self.real, self.npdt = lib, suffix, real, np.dtype(npdt)
        self.P = C.POINTER(real)
        self.Cost, self.Mlp, self.Problem, self.Ellipse3d, self.Auv = _structs(real)
        self.real = real

    def fn(self, name, restype=None):
        f = getattr(self.lib, name + self.suffix)
        f.restype = restype
        return f

    def arr(self, x, shape=None):  # noqa: E301
        a = np.ascontiguousarray(np.asarray(x, dtype=self.npdt))
        if shape is not None:
            a = a.reshape(shape)
        return a

    def ptr(self, a):
        return a.ctypes.data_as(self.P) if a is not None else None


_lib = None
_insts = {}


def _get(dtype):
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        # size the OpenMP team to the CPUs this job may really use (see usable_cpus)
        _lib.orc_num_threads.restype = C.c_int
        _lib.orc_set_num_threads.argtypes = [C.c_int]
        _lib.orc_set_num_threads(min(int(_lib.orc_num_threads()), usable_cpus()))
    key = np.dtype(dtype).name
    if key not in _insts:
        if key == "float32":
            _insts[key] = _Inst(_lib, "_f32", C.c_float, np.float32)
        elif key == "float64":
            _insts[key] = _Inst(_lib, "_f64", C.c_double, np.float64)
        else:
            raise ValueError(dtype)
    return _insts[key]


def usable_cpus():
    """CPUs this process may actually use: the smaller of the affinity mask and the cgroup CPU quota (a GPU box gives one
    job a share of a many-core host; OpenMP's default of one thread per visible core then oversubscribes it badly)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = open(path).read().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return max(1, n)


def num_threads():
    _get(np.float32)
    _lib.orc_num_threads.restype = C.c_int
    return int(_lib.orc_num_threads())


def set_num_threads(n):
    """OpenMP thread count of the following calls (noise generation included)"""
    _get(np.float32)
    _lib.orc_set_num_threads.argtypes = [C.c_int]
    _lib.orc_set_num_threads(int(n))


# --------------------------------------------------------------------------- model (A4)
def block_diag(mat, nb, dtype=np.float32):
    I = _get(dtype)
    m = I.arr(mat)
    r, c = m.shape
    out = np.zeros((nb * r, nb * c), I.npdt)
    I.fn("orc_block_diag")(I.ptr(m), r, c, nb, I.ptr(out))
    return out


def pm_matrices(dt, mass, s, a, dtype=np.float32):
    I = _get(dtype)
    A = np.zeros((s, s), I.npdt)
    B = np.zeros((s, a), I.npdt)
    I.fn("orc_pm_matrices")(I.real(dt), I.real(mass), s, a, I.ptr(A), I.ptr(B))
    return A, B


def model_free_step(A, x, dtype=np.float32):
    I = _get(dtype)
    A = I.arr(A)
    s = A.shape[0]
    x = I.arr(x, (-1, s))
    out = np.zeros_like(x)
    I.fn("orc_model_free_step")(I.ptr(A), s, I.ptr(x), x.shape[0], I.ptr(out))
    return out


def model_action_step(B, v, dtype=np.float32):
    I = _get(dtype)
    B = I.arr(B)
    s, a = B.shape
    v = I.arr(v, (-1, a))
    out = np.zeros((v.shape[0], s), I.npdt)
    I.fn("orc_model_action_step")(I.ptr(B), s, a, I.ptr(v), v.shape[0], I.ptr(out))
    return out


def model_step(A, B, x, v, dtype=np.float32):
    I = _get(dtype)
    A, B = I.arr(A), I.arr(B)
    s, a = B.shape
    x, v = I.arr(x, (-1, s)), I.arr(v, (-1, a))
    out = np.zeros((v.shape[0], s), I.npdt)
    I.fn("orc_model_step")(I.ptr(A), I.ptr(B), s, a, I.ptr(x), x.shape[0], I.ptr(v), v.shape[0], I.ptr(out))
    return out


# --------------------------------------------------------------------------- cost (A5-A7)
def mat_inverse(S, dtype=np.float32):
    I = _get(dtype)
    S = I.arr(S)
    n = S.shape[0]
    out = np.zeros_like(S)
    rc = I.fn("orc_mat_inverse", C.c_int)(I.ptr(S), n, I.ptr(out))
    if rc:
        raise np.linalg.LinAlgError("singular sigma")
    return out


def _q_full(I, Q, s):
    Q = I.arr(Q)
    if Q.ndim == 1:  # Diag(in_Q), cost_base.cpp:40
        Q = I.arr(np.diag(Q))
    assert Q.shape == (s, s)
    return Q


class Problem:
    """Holds the numpy buffers + the C struct of one oracle problem instance."""

    def __init__(self, tau, s, a, dt=0.1, mass=1.0, lam=1.0, sigma=None, goal=None, Q=None,
                 gamma=1.0, upsilon=1.0, action_cost=ACTION_COST_CPP, mlp=None, threads=1,
                 dtype=np.float32, ellipse=None, auv=None, nnauv=None, quat_cost=False, ellipse3d=None, nnauv_speed=None):
        """ellipse: dict(a, b, cx, cy, speed, m_state, m_vel) selects ElipseCost's state cost (elipse_cost.py:9-85)
        instead of the quadratic one.
        auv: the reference's AUVModel `parameters` dict (auv_model.py:85-245; + "dt") selects the Fossen model (s=13, a=6);
        nnauv: an mlp dict with input width s+a-3 selects NNAUVModel (nn_model.py:215-304);
        nnauv_speed: an mlp dict with 15 inputs (3 Euler angles, 6 velocities, 6 forces) and 6 outputs selects NNAUVModelSpeed
        (nn_model.py:307-588), stepped with `dt`;
        quat_cost: StaticQuatCost (static_cost.py:73-159): goal [13], Q [10,10] (or its diagonal);
        ellipse3d: dict(normal, aVec, axis, speed, m_state, m_vel) selects ElipseCost3D (elipse_cost.py:101-246)."""
        I = self.I = _get(dtype)
        self.tau, self.s, self.a = tau, s, a
        self.sigma = I.arr(np.eye(a) if sigma is None else sigma, (a, a))
        self.sigma_inv = mat_inverse(self.sigma, dtype)
        if goal is None:  # controller_base.cpp:43-46: (1,0) per axis
            goal = np.tile([1.0, 0.0], s // 2)
        self.goal = I.arr(goal, (s,))
        self.Q = _q_full(I, np.ones(10 if quat_cost else s) if Q is None else Q, 10 if quat_cost else s)
        self.lam = float(lam)
        p = self.c = I.Problem()
        p.cost.s, p.cost.a, p.cost.action_cost_kind = s, a, action_cost
        p.cost.lam, p.cost.gamma, p.cost.upsilon = lam, gamma, upsilon
        p.cost.goal, p.cost.Q, p.cost.sigma_inv = I.ptr(self.goal), I.ptr(self.Q), I.ptr(self.sigma_inv)
        if ellipse is not None:
            p.cost.state_cost_kind = 1  # ORC_STATE_COST_ELLIPSE
            for i, key in enumerate(("a", "b", "cx", "cy", "speed", "m_state", "m_vel")):
                p.cost.ellipse[i] = ellipse[key]
        if quat_cost:
            p.cost.state_cost_kind = STATE_COST_QUAT
        if ellipse3d is not None:
            p.cost.state_cost_kind = STATE_COST_ELLIPSE3D
            self.R3 = ellipse3d_prepare(p.cost.e3, dtype=dtype, **ellipse3d)
        p.tau, p.threads = tau, threads
        self._keep = []
        if auv is not None:
            p.model_kind = MODEL_AUV
            fill_auv(I, p.auv, auv, dt)
        elif nnauv is not None:
            p.model_kind = MODEL_NNAUV
            self._fill_mlp(p.mlp, nnauv, n_in=s + a - 3)
        elif nnauv_speed is not None:
            p.model_kind = MODEL_NNAUV_SPEED
            self._fill_mlp(p.mlp, nnauv_speed, n_in=15, n_out=6)
            p.auv.dt = dt
        elif mlp is None:
            p.model_kind = MODEL_POINT_MASS
            self.A, self.B = pm_matrices(dt, mass, s, a, dtype)
            p.A, p.B = I.ptr(self.A), I.ptr(self.B)
        else:
            p.model_kind = MODEL_MLP
            self._fill_mlp(p.mlp, mlp)

    def _fill_mlp(self, m, mlp, n_in=None, n_out=None):
        I = self.I
        Ws, bs = mlp["W"], mlp["b"]
        m.s, m.a, m.n_layers = self.s, self.a, len(Ws)
        for l, (W, b) in enumerate(zip(Ws, bs)):
            W, b = I.arr(W), I.arr(b)
            self._keep += [W, b]
            m.widths[l] = W.shape[1]
            m.W[l], m.b[l] = I.ptr(W), I.ptr(b)
        n_in = self.s + self.a if n_in is None else n_in
        n_out = self.s if n_out is None else n_out
        for name, default, n in (("xmean", 0.0, n_in), ("xstd", 1.0, n_in),
                                 ("ymean", 0.0, n_out), ("ystd", 1.0, n_out)):
            v = I.arr(mlp.get(name, np.full(n, default)), (n,))
            self._keep.append(v)
            setattr(m, name, I.ptr(v))

    # A5
    def state_cost(self, x):
        I = self.I
        x = I.arr(x, (-1, self.s))
        out = np.zeros(x.shape[0], I.npdt)
        I.fn("orc_final_cost")(C.byref(self.c.cost), I.ptr(x), x.shape[0], I.ptr(out))
        return out

    # A6
    def action_cost(self, u, eps):
        I = self.I
        u, eps = I.arr(u, (self.a,)), I.arr(eps, (-1, self.a))
        if self.c.cost.action_cost_kind == ACTION_COST_PY:
            f = I.fn("orc_action_cost_py", I.real)
            return np.array([f(I.ptr(u), I.ptr(e), I.ptr(self.sigma_inv), self.a, I.real(self.c.cost.lam),
                               I.real(self.c.cost.gamma), I.real(self.c.cost.upsilon)) for e in eps], I.npdt)
        f = I.fn("orc_action_cost_cpp", I.real)
        return np.array([f(I.ptr(u), I.ptr(e), I.ptr(self.sigma_inv), self.a, I.real(self.c.cost.lam))
                         for e in eps], I.npdt)

    # A7 (one step)
    def step_cost(self, x, u, eps):
        I = self.I
        x, u, eps = I.arr(x, (-1, self.s)), I.arr(u, (self.a,)), I.arr(eps, (-1, self.a))
        out = np.zeros(x.shape[0], I.npdt)
        I.fn("orc_step_cost")(C.byref(self.c.cost), I.ptr(x), I.ptr(u), I.ptr(eps), x.shape[0], I.ptr(out))
        return out

    def model_next(self, x, v):
        """one step x [k,s], v [k,a] -> x' [k,s] of whatever model the problem holds"""
        I = self.I
        x, v = I.arr(x, (-1, self.s)), I.arr(v, (-1, self.a))
        out = np.zeros_like(x)
        f = I.fn("orc_model_next")
        for i in range(x.shape[0]):
            xi, vi, oi = np.ascontiguousarray(x[i]), np.ascontiguousarray(v[i]), np.zeros(self.s, I.npdt)
            f(C.byref(self.c), I.ptr(xi), I.ptr(vi), I.ptr(oi))
            out[i] = oi
        return out

    def mlp_step(self, x, v):
        I = self.I
        x, v = I.arr(x, (self.s,)), I.arr(v, (self.a,))
        out = np.zeros(self.s, I.npdt)
        I.fn("orc_mlp_step")(C.byref(self.c.mlp), I.ptr(x), I.ptr(v), I.ptr(out))
        return out

    # A7 (full recurrence)
    def rollout_cost(self, x0, U, eps, traj=False):
        I = self.I
        x0, U = I.arr(x0, (self.s,)), I.arr(U, (self.tau, self.a))
        eps = I.arr(eps, (-1, self.tau, self.a))
        k = eps.shape[0]
        out = np.zeros(k, I.npdt)
        tr = np.zeros((k, self.tau, self.s), I.npdt) if traj else None
        I.fn("orc_rollout_cost")(C.byref(self.c), I.ptr(x0), I.ptr(U), I.ptr(eps), k, I.ptr(out), I.ptr(tr))
        return (out, tr) if traj else out

    # A1 with injected noise; returns (u, U_next, cost)
    def next_with_noise(self, x, U, eps, normalize=False):
        I = self.I
        x, U = I.arr(x, (self.s,)), I.arr(U, (self.tau, self.a)).copy()
        eps = I.arr(eps, (-1, self.tau, self.a))
        k = eps.shape[0]
        u = np.zeros(self.a, I.npdt)
        c = np.zeros(k, I.npdt)
        I.fn("orc_next_with_noise")(C.byref(self.c), I.ptr(x), I.ptr(U), I.ptr(eps), k, int(normalize),
                                    I.ptr(u), I.ptr(c))
        return u, U, c


# --------------------------------------------------------------------------- update (A8-A10)
def update(cost, eps, U, lam, normalize=False, acc_double=True, dtype=np.float32):
    """Returns dict(beta, arg, exp, nabla, w, wn, Unew)."""
    I = _get(dtype)
    cost = I.arr(cost, (-1,))
    k = cost.shape[0]
    U = I.arr(U)
    tau, a = U.shape
    eps = I.arr(eps, (k, tau, a))
    beta, nabla = np.zeros(1, I.npdt), np.zeros(1, I.npdt)
    arg, e, w = (np.zeros(k, I.npdt) for _ in range(3))
    wn, Un = np.zeros((tau, a), I.npdt), np.zeros((tau, a), I.npdt)
    I.fn("orc_update")(I.ptr(cost), I.ptr(eps), I.ptr(U), I.real(lam), k, tau, a, int(normalize),
                       int(acc_double), I.ptr(beta), I.ptr(arg), I.ptr(e), I.ptr(nabla), I.ptr(w),
                       I.ptr(wn), I.ptr(Un))
    return dict(beta=beta[0], arg=arg, exp=e, nabla=nabla[0], w=w, wn=wn, Unew=Un)


def get_new(U, nb, dtype=np.float32):
    I = _get(dtype)
    U = I.arr(U)
    tau, a = U.shape
    out = np.zeros((nb, a), I.npdt)
    I.fn("orc_get_new")(I.ptr(U), tau, a, nb, I.ptr(out))
    return out


def shift(U, init, nb, dtype=np.float32):
    I = _get(dtype)
    U = I.arr(U)
    tau, a = U.shape
    init = I.arr(init, (-1, a))
    out = np.zeros((tau - nb + init.shape[0], a), I.npdt)
    I.fn("orc_shift")(I.ptr(U), tau, a, I.ptr(init), init.shape[0], nb, I.ptr(out))
    return out


def combine_records(rec, U, lam, dtype=np.float32):
    I = _get(dtype)
    U = I.arr(U)
    tau, a = U.shape
    rec = I.arr(rec, (-1, 2 + tau * a))
    out = np.zeros((tau, a), I.npdt)
    I.fn("orc_combine_records")(I.ptr(rec), rec.shape[0], I.ptr(U), I.real(lam), tau, a, I.ptr(out))
    return out


# --------------------------------------------------------------------------- noise (A2)
def philox4x32_10(ctr, key):
    _get(np.float32)
    c = (C.c_uint32 * 4)(*[int(v) for v in ctr])
    k = (C.c_uint32 * 2)(*[int(v) for v in key])
    o = (C.c_uint32 * 4)()
    _lib.orc_philox4x32_10(c, k, o)
    return [int(v) for v in o]


def normals(seed, step, k_offset, k, tau, a):
    _get(np.float32)
    out = np.zeros((k, tau, a), np.float32)
    _lib.orc_normals(C.c_uint64(seed), C.c_uint64(step), C.c_uint64(k_offset), k, tau, a,
                     out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


def noise(seed, step, k_offset, k, tau, a, sigma):
    _get(np.float32)
    sigma = np.ascontiguousarray(np.asarray(sigma, np.float32).reshape(a, a))
    out = np.zeros((k, tau, a), np.float32)
    _lib.orc_noise(C.c_uint64(seed), C.c_uint64(step), C.c_uint64(k_offset), k, tau, a,
                   sigma.ctypes.data_as(C.POINTER(C.c_float)), out.ctypes.data_as(C.POINTER(C.c_float)))
    return out


# --------------------------------------------------------------------------- AUV family, quaternion / 3D-ellipse costs
def skew(v):
    """auv_model.py:7-36 skew_op"""
    return np.array([[0.0, -v[2], v[1]], [v[2], 0.0, -v[0]], [-v[1], v[0], 0.0]])


def auv_matrices(params):
    """auv_model.py:186-262 in fp64 (the reference's dtype): -> dict(mtot, inv_mtot, lin_damp, lin_damp_fwd, quad_damp), matrices [6,6].
    rigid-body mass = [[m I, -m S(cog)], [m S(cog), inertial]] (:247-251), total = rigid body + added mass (:253-254)."""
    r32 = lambda v: np.asarray(v, np.float32).astype(np.float64)  # the parameters ARE fp32 numbers (the C-ABI's mppi_auv_desc carries floats)
    m = float(r32(params["mass"]))
    cog = r32(params["cog"])
    i = params["inertial"]
    inertial = r32([[i["ixx"], i["ixy"], i["ixz"]], [i["ixy"], i["iyy"], i["iyz"]], [i["ixz"], i["iyz"], i["izz"]]])
    lower = m * skew(cog)
    rb = np.block([[m * np.eye(3), -lower], [lower, inertial]])
    mtot = rb + r32(params.get("Ma", np.zeros((6, 6))))

    def mat(key):
        d = r32(params.get(key, np.zeros(6)))
        return np.diag(d) if d.shape == (6,) else d
    # the inverse through orc_mat_inverse (Gauss-Jordan, partial pivoting, double): deterministic, no LAPACK in the loop
    return dict(mtot=mtot, inv_mtot=mat_inverse(mtot, np.float64), lin_damp=mat("linear_damping"), lin_damp_fwd=mat("linear_damping_forward_speed"),
                quad_damp=r32(params.get("quad_damping", np.zeros(6))))


def fill_auv(I, c, params, dt):
    mats = auv_matrices(params)
    r32 = lambda v: float(np.float32(v))
    c.mass, c.volume, c.density = r32(params["mass"]), r32(params["volume"]), r32(params["density"])
    c.gravity, c.dt, c.rk = r32(9.81), r32(params.get("dt", dt)), int(params.get("rk", 1))  # auv_model.py:236, :111-114
    for i in range(3):
        c.cog[i], c.cob[i] = r32(params["cog"][i]), r32(params["cob"][i])
    for key in ("mtot", "inv_mtot", "lin_damp", "lin_damp_fwd"):
        flat = mats[key].ravel()
        arr = getattr(c, key)
        for i in range(36):
            arr[i] = flat[i]
    for i in range(6):
        c.quad_damp[i] = mats["quad_damp"][i]


class AuvModel:
    """AUVModel's pieces (the reference's tests call them one by one, scripts/test.py:237-586)."""

    def __init__(self, params, dt=0.1, dtype=np.float64):
        I = self.I = _get(dtype)
        self.c = I.Auv()
        fill_auv(I, self.c, params, dt)

    def b2i(self, quat):
        I = self.I
        q = I.arr(quat, (4,))
        rot, T = np.zeros((3, 3), I.npdt), np.zeros((4, 3), I.npdt)
        I.fn("orc_auv_b2i")(I.ptr(q), I.ptr(rot), I.ptr(T))
        return rot, T

    def restoring(self, quat):
        I = self.I
        rot, _ = self.b2i(quat)
        g = np.zeros(6, I.npdt)
        I.fn("orc_auv_restoring")(C.byref(self.c), I.ptr(rot), I.ptr(g))
        return g

    def damping(self, vel):
        I = self.I
        v = I.arr(vel, (6,))
        D = np.zeros((6, 6), I.npdt)
        I.fn("orc_auv_damping")(C.byref(self.c), I.ptr(v), I.ptr(D))
        return D

    def coriolis(self, vel):
        I = self.I
        v = I.arr(vel, (6,))
        Cm = np.zeros((6, 6), I.npdt)
        I.fn("orc_auv_coriolis")(C.byref(self.c), I.ptr(v), I.ptr(Cm))
        return Cm

    def state_dot(self, x, u):
        I = self.I
        x, u = I.arr(x, (13,)), I.arr(u, (6,))
        out = np.zeros(13, I.npdt)
        I.fn("orc_auv_state_dot")(C.byref(self.c), I.ptr(x), I.ptr(u), I.ptr(out))
        return out

    def step(self, x, u):
        I = self.I
        x, u = I.arr(x, (13,)), I.arr(u, (6,))
        out = np.zeros(13, I.npdt)
        I.fn("orc_auv_step")(C.byref(self.c), I.ptr(x), I.ptr(u), I.ptr(out))
        return out


def ellipse3d_prepare(e3, normal, aVec, axis, speed, m_state, m_vel, dtype=np.float64, center=None):
    """fills the C struct e3; returns R [3,3] (elipse_cost.py:163-167 self.R). `center` is accepted and unused, as in the reference."""
    I = _get(dtype)
    n, a, ax = I.arr(normal, (3,)), I.arr(aVec, (3,)), I.arr(axis, (2,))
    R = np.zeros((3, 3), I.npdt)
    I.fn("orc_ellipse3d_prepare")(I.ptr(n), I.ptr(a), I.ptr(ax), I.real(speed), I.real(m_state), I.real(m_vel), C.byref(e3), I.ptr(R))
    return R


class Ellipse3D:
    """ElipseCost3D's pieces (scripts/test.py:1164-1360)."""

    def __init__(self, normal, aVec, axis, speed, m_state, m_vel, dtype=np.float64, center=None):
        I = self.I = _get(dtype)
        self.c = I.Ellipse3d()
        self.R = ellipse3d_prepare(self.c, normal, aVec, axis, speed, m_state, m_vel, dtype)
        self.q = np.array(list(self.c.q), I.npdt)

    def position_error(self, p):
        I = self.I
        p = I.arr(p, (3,))
        return I.fn("orc_ellipse3d_position_error", I.real)(C.byref(self.c), I.ptr(p))

    def orientation_error(self, pose):
        I = self.I
        p = I.arr(pose, (7,))
        return I.fn("orc_ellipse3d_orientation_error", I.real)(C.byref(self.c), I.ptr(p))

    def velocity_error(self, vel):
        I = self.I
        v = I.arr(vel, (6,))
        return I.fn("orc_ellipse3d_velocity_error", I.real)(C.byref(self.c), I.ptr(v))

    def state_cost(self, x):
        I = self.I
        x = I.arr(x, (13,))
        return I.fn("orc_state_cost_ellipse3d", I.real)(C.byref(self.c), I.ptr(x))


def quat_rotate(p, q, dtype=np.float64):
    I = _get(dtype)
    p, q = I.arr(p, (3,)), I.arr(q, (4,))
    o = np.zeros(3, I.npdt)
    I.fn("orc_quat_rotate")(I.ptr(p), I.ptr(q), I.ptr(o))
    return o


def quat_multiply(q1, q2, dtype=np.float64):
    I = _get(dtype)
    q1, q2 = I.arr(q1, (4,)), I.arr(q2, (4,))
    o = np.zeros(4, I.npdt)
    I.fn("orc_quat_multiply")(I.ptr(q1), I.ptr(q2), I.ptr(o))
    return o


def quat_dist(x, goal, dtype=np.float64):
    """StaticQuatCost.dist (static_cost.py:141-159): [10]"""
    I = _get(dtype)
    x, g = I.arr(x, (13,)), I.arr(goal, (13,))
    d = np.zeros(10, I.npdt)
    I.fn("orc_quat_dist")(I.ptr(x), I.ptr(g), I.ptr(d))
    return d


def euler_from_quaternion(q, dtype=np.float64):
    """tensorflow_graphics euler.from_quaternion as NNAUVModelSpeed.to_euler calls it (nn_model.py:566-588): [k,4] (x,y,z,w) -> [k,3]"""
    I = _get(dtype)
    q = I.arr(q, (-1, 4))
    out = np.zeros((q.shape[0], 3), I.npdt)
    f = I.fn("orc_euler_from_quat")
    for i in range(q.shape[0]):
        qi, ei = np.ascontiguousarray(q[i]), np.zeros(3, I.npdt)
        f(I.ptr(qi), I.ptr(ei))
        out[i] = ei
    return out


def nnauv_speed_prepare_data(state, action, xmean=None, xstd=None, dtype=np.float64):
    """NNAUVModelSpeed.prepare_data (nn_model.py:438-461): concat(euler(state)[3:], action) normalised. [k,13],[k,6] -> [k,15]"""
    st, ac = np.asarray(state, np.float64).reshape(-1, 13), np.asarray(action, np.float64).reshape(-1, 6)
    X = np.concatenate([euler_from_quaternion(st[:, 3:7], dtype).astype(np.float64), st[:, 7:], ac], axis=1)
    return (X - (0.0 if xmean is None else xmean)) / (1.0 if xstd is None else xstd)


def nnauv_speed_prepare_training_data(state_t, state_t1, action, dtype=np.float64):
    """NNAUVModelSpeed.prepare_training_data (nn_model.py:382-436), norm=False: X as prepare_data, Y = velocity delta [k,6]"""
    st, st1 = np.asarray(state_t, np.float64).reshape(-1, 13), np.asarray(state_t1, np.float64).reshape(-1, 13)
    return nnauv_speed_prepare_data(st, action, dtype=dtype), st1[:, 7:] - st[:, 7:]


def nnauv_prepare_data(state, action, xmean=None, xstd=None):
    """NNAUVModel.prepare_data (nn_model.py:289-293): concat(state[3:], action) normalised. [k,13],[k,6] -> [k,16]"""
    st, ac = np.asarray(state, np.float64).reshape(-1, 13), np.asarray(action, np.float64).reshape(-1, 6)
    X = np.concatenate([st[:, 3:], ac], axis=1)
    return (X - (0.0 if xmean is None else xmean)) / (1.0 if xstd is None else xstd)


def nnauv_prepare_training_data(state_t, state_t1, action, mask=None):
    """NNAUVModel.prepare_training_data (nn_model.py:241-287), norm=False form: X = concat(stateT[3:], action), Y = stateT1 - stateT
    in the frame whose origin is stateT's position (the masked components cancel in the difference)."""
    st, st1 = np.asarray(state_t, np.float64).reshape(-1, 13), np.asarray(state_t1, np.float64).reshape(-1, 13)
    ac = np.asarray(action, np.float64).reshape(-1, 6)
    mask = np.array([1, 1, 1] + [0] * 10, np.float64) if mask is None else np.asarray(mask, np.float64).reshape(13)
    t_from = mask * st
    return np.concatenate([st[:, 3:], ac], axis=1), (st1 - t_from) - (st - t_from)
