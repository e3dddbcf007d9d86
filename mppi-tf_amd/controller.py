"""Host-side mirror of the reference's plugin surface for the control-step path.

Same class and method names, argument meaning and shapes (trailing singleton kept:
state [k,sDim,1], action [aDim,1], noise [k,tau,aDim,1]) as the reference's Python package —
  scripts/src/models/point_mass_model.py   PointMassModel
  scripts/src/costs/cost_base.py           CostBase
  scripts/src/costs/static_cost.py         StaticCost
  scripts/src/controllers/controller_base.py ControllerBase
— and as the C++ ControllerBase(k, tau, dt, mass, s_dim, a_dim) (include/controller_base.hpp:60-109),
so parity tests read like the reference's own tests.  Every numeric method is a call into
libmppi_hip.so (HIP kernels); nothing is computed in Python/numpy and there is no fallback.
Arithmetic is fp32 (the C++ reference's DT_FLOAT); the Python reference is fp64.

`scope` arguments are accepted and ignored (TensorFlow name scopes).
"""
import numpy as np

from . import _lib
from ._lib import ACTION_COST_CPP, ACTION_COST_PY, Handle


def _col(x):
    """[..., n] -> [..., n, 1] (the reference's trailing singleton)."""
    return np.asarray(x)[..., None]


def _flat(x, n):
    return np.asarray(x, np.float32).reshape(-1, n)


# capacity of the transition log the controller mirrors switch on (the reference's m_db grows without bound)
LOG_ROWS = 1 << 16

class PointMassModel:
    """x' = A x + (B/m) u ; A = I⊗[[1,dt],[0,1]], B = I⊗[[dt²/2],[dt]]  (point_mass_model.py:28-151,
    src/model_base.cpp:53-82). Positional signature is the one the reference's tests use
    (scripts/test.py:52: PointMassModel(m, dt, s, a))."""

    def __init__(self, mass=1.0, dt=0.1, stateDim=2, actionDim=1, name="point_mass", device=0):
        self._mass, self._dt, self._stateDim, self._actionDim = float(mass), float(dt), int(stateDim), int(actionDim)
        self._name = name
        self._h = Handle(k=1, tau=1, s_dim=stateDim, a_dim=actionDim, dt=dt, mass=mass, device=device)

    def get_name(self):
        return self._name

    def get_state_dim(self):
        return self._stateDim

    def get_action_dim(self):
        return self._actionDim

    def get_mass(self):
        return self._mass

    def build_free_step_graph(self, scope, state):
        fr, _, _ = self._h.model_step(_flat(state, self._stateDim), np.zeros((np.asarray(state).shape[0], self._actionDim)))
        return _col(fr)

    def build_action_step_graph(self, scope, action):
        a = _flat(action, self._actionDim)
        _, ac, _ = self._h.model_step(np.zeros((a.shape[0], self._stateDim)), a)
        return _col(ac)

    def build_step_graph(self, scope, state, action):
        _, _, nx = self._h.model_step(_flat(state, self._stateDim), _flat(action, self._actionDim))
        return _col(nx)

    def predict(self, state, action):
        return self.build_step_graph("predict", state, action)


class CostBase:
    """Action-cost part shared by every cost (cost_base.py:6-209). `state_cost` is abstract."""

    _action_cost_kind = ACTION_COST_PY

    def __init__(self, lam, gamma, upsilon, sigma, device=0):
        self.lam = float(np.asarray(lam).ravel()[0])
        self.gamma, self.upsilon = float(gamma), float(upsilon)
        self.sigma = np.asarray(sigma, np.float32)
        if self.sigma.ndim != 2:
            raise AssertionError("The noise covariance matrix needs to be a semi definit positive.")
        self.sig_shape = self.sigma.shape
        self._device = device
        self._aDim = self.sig_shape[0]
        self._ha = None

    def _action_handle(self):
        if self._ha is None:
            self._ha = Handle(k=1, tau=1, s_dim=2 * self._aDim, a_dim=self._aDim, lam=self.lam, gamma=self.gamma,
                              upsilon=self.upsilon, sigma=self.sigma, action_cost=self._action_cost_kind,
                              device=self._device)
        return self._ha

    def action_cost(self, scope, action, noise):
        out = self._action_handle().action_cost(np.asarray(action).reshape(-1), _flat(noise, self._aDim))
        return out.reshape(-1, 1, 1)

    def state_cost(self, scope, state):
        raise NotImplementedError

    def build_step_cost_graph(self, scope, state, action, noise):
        a = np.asarray(action)
        if a.shape != (self._aDim, 1):
            raise AssertionError("Bad shape for the action tensor, should be [%d, 1], got shape %s" % (self._aDim, a.shape))
        n = np.asarray(noise)
        if n.ndim != 3 or n.shape[1:] != (self._aDim, 1):
            raise AssertionError("Bad shape for the noise tensor, should be [k/1, %d, 1], got shape %s" % (self._aDim, n.shape))
        return self.add_cost(scope, self.state_cost(scope, state), self.action_cost(scope, action, noise))

    def build_final_step_cost_graph(self, scope, state):
        return self.state_cost(scope, state)

    def add_cost(self, scope, currentCost, newCost):
        return (np.asarray(currentCost, np.float32) + np.asarray(newCost, np.float32)).astype(np.float32)


class StaticCost(CostBase):
    """(x-goal)ᵀ Q (x-goal) + action cost (static_cost.py:6-71; C++ src/cost_base.cpp:37-68)."""

    def __init__(self, lam, gamma, upsilon, sigma, goal, Q, diag=False, device=0, cpp_action_cost=False):
        CostBase.__init__(self, lam, gamma, upsilon, sigma, device)
        if cpp_action_cost:
            self._action_cost_kind = ACTION_COST_CPP
        self.Q = np.asarray(Q, np.float32)
        if diag:
            self.Q = np.diag(self.Q.ravel())
        self.q_shape = self.Q.shape
        self._sDim = self.q_shape[0]
        self._h = None
        self.setGoal(goal)

    def setGoal(self, goal):
        g = np.asarray(goal, np.float32)
        if g.shape != (self.q_shape[0], 1):
            raise AssertionError("Goal tensor shape error, expected: [%d, 1], got %s" % (self.q_shape[0], g.shape))
        self.goal = g
        if self._h is not None:
            self._h.set_goal(g.ravel())

    set_goal = setGoal

    def getGoal(self):
        return self.goal

    def _handle(self):
        if self._h is None:
            self._h = Handle(k=1, tau=1, s_dim=self._sDim, a_dim=self._aDim, lam=self.lam, gamma=self.gamma,
                             upsilon=self.upsilon, sigma=self.sigma, goal=self.goal.ravel(), Q=self.Q, q_is_full=True,
                             action_cost=self._action_cost_kind, device=self._device)
        return self._h

    def action_cost(self, scope, action, noise):
        out = self._handle().action_cost(np.asarray(action).reshape(-1), _flat(noise, self._aDim))
        return out.reshape(-1, 1, 1)

    def state_cost(self, scope, state):
        return self._handle().state_cost(_flat(state, self._sDim)).reshape(-1, 1, 1)

    def build_step_cost_graph(self, scope, state, action, noise):
        a, n = np.asarray(action), np.asarray(noise)
        if a.shape != (self._aDim, 1) or n.ndim != 3 or n.shape[1:] != (self._aDim, 1):
            raise AssertionError("Bad shape for the action/noise tensor")
        out = self._handle().step_cost(_flat(state, self._sDim), a.reshape(-1), _flat(noise, self._aDim))
        return out.reshape(-1, 1, 1)

    def dist(self, state):
        return np.asarray(state, np.float32) - self.goal


class ElipseCost(CostBase):
    """2D elliptic track cost (costs/elipse_cost.py:9-104), state (x, vx, y, vy):
    m_state·|((x-cx)/a)² + ((y-cy)/b)² - 1| + m_vel·(sqrt(vx²+vy²) - speed)² + action cost. The reference's spelling
    and constructor argument order are kept."""

    def __init__(self, lam, gamma, upsilon, sigma, a, b, center_x, center_y, speed, m_state, m_vel, device=0):
        CostBase.__init__(self, lam, gamma, upsilon, sigma, device)
        self.a, self.b, self.cx, self.cy, self.gv = float(a), float(b), float(center_x), float(center_y), float(speed)
        self.mx, self.mv = float(m_state), float(m_vel)
        self._sDim = 4
        self._h = None

    @property
    def ellipse(self):
        return dict(a=self.a, b=self.b, cx=self.cx, cy=self.cy, speed=self.gv, m_state=self.mx, m_vel=self.mv)

    def _handle(self):
        if self._h is None:
            self._h = Handle(k=1, tau=1, s_dim=self._sDim, a_dim=self._aDim, lam=self.lam, gamma=self.gamma,
                             upsilon=self.upsilon, sigma=self.sigma, action_cost=self._action_cost_kind,
                             ellipse=self.ellipse, device=self._device)
        return self._h

    def state_cost(self, scope, state):
        st = np.asarray(state)
        if st.ndim != 3 or st.shape[1:] != (4, 1):  # elipse_cost.py:66-67
            raise AssertionError("State tensor doesn't have the expected shape.\n Expected [k/1, 4, 1], got {}".format(st.shape))
        return self._handle().state_cost(_flat(state, 4)).reshape(-1, 1, 1)

    def draw_goal(self):  # elipse_cost.py:87-91
        alpha = np.linspace(0, 2 * np.pi, 1000)
        return self.a * np.cos(alpha), self.b * np.sin(alpha)

    def dist(self, state):  # elipse_cost.py:93-104
        x, vx, y, vy = (np.asarray(state).reshape(-1)[i] for i in range(4))
        v = np.sqrt(vx ** 2 + vy ** 2)
        return {"x_dist": (((x - self.cx) / self.a) ** 2 + ((y - self.cy) / self.b) ** 2) - 1, "v_dist": np.abs(v - self.gv)}


class ControllerBase:
    """MPPI controller (controller_base.py:17-597). `next(state)` is one control step on the GPU.

    Follows the C++ reference where the two differ: the shifted sequence is kept as
    the warm start (controller_base.cpp:144), and the cost object decides the action-cost form.
    """

    def __init__(self, model, cost, k=1, tau=1, sDim=1, aDim=1, lam=1., upsilon=1., sigma=np.array([]),
                 initSeq=np.array([]), normalizeCost=False, filterSeq=False, log=False, logPath=None,
                 graphMode=False, configDict=None, taskDict=None, modelDict=None, debug=False,
                 seed=1, device=0):
        self._model, self._cost = model, cost
        self._k, self._tau, self._sDim, self._aDim = int(k), int(tau), int(sDim), int(aDim)
        self._lam, self._upsilon = float(lam), float(upsilon)
        self._normalizeCost = bool(normalizeCost)
        sigma = np.asarray(sigma, np.float32)
        if sigma.size == 0:
            sigma = np.eye(aDim, dtype=np.float32)
        # build_noise: noises = (upsilon * sigma) @ rng   (controller_base.py:362-368)
        self._sigma = sigma
        # the model_base slot: the point-mass model (dt, mass), or a 13-state model that brings its own description
        model_kw = {}
        if hasattr(model, "handle_kw"):           # NNAUVModelSpeed (auv.py): brings its own Handle keyword
            model_kw = model.handle_kw()
        elif hasattr(model, "handle_parameters"):   # AUVModel (auv.py)
            model_kw = dict(auv=model.handle_parameters())
        elif hasattr(model, "mlp"):               # NNAUVModel (auv.py)
            model_kw = dict(nnauv=model.mlp())
        self._h = Handle(k=self._k, tau=self._tau, s_dim=self._sDim, a_dim=self._aDim,
                         dt=model._dt, mass=(1.0 if model_kw else model._mass), lam=self._lam, **model_kw,
                         gamma=getattr(cost, "gamma", 1.0), upsilon=getattr(cost, "upsilon", 1.0),
                         sigma=sigma, **self._state_cost_args(cost),
                         action_cost=cost._action_cost_kind, normalize_cost=self._normalizeCost,
                         seed=seed, device=device,
                         # build_noise: noises = (υΣ)·z (controller_base.py:362-368); the cost keeps Σ⁻¹ of Σ
                         upsilon_scales_noise=True, log_rows=LOG_ROWS)
        if abs(getattr(cost, "upsilon", self._upsilon) - self._upsilon) > 0:
            raise AssertionError("controller and cost must be built with the same upsilon")
        initSeq = np.asarray(initSeq, np.float32)
        if initSeq.size:
            if initSeq.shape != (tau, aDim, 1):
                raise AssertionError
            self._h.set_action_sequence(initSeq[..., 0])
        # filterSeq (controller_base.py:277-291). The reference's literals are savgol_filter(seq, 10, 9): an even
        # window ('interp' mode does not define it uniquely) whose result it never reads; here filterSeq=True means
        # the neighbouring odd window (11, 9) when the horizon allows it, or filterSeq=(window, polyorder).
        if filterSeq:
            w, p = filterSeq if isinstance(filterSeq, (tuple, list)) else (11, 9)
            self._h.set_sequence_filter(w, p)
        # clip_act (controller_base.py:500-504): the model's action limits, if it has any
        lo = getattr(model, "_actMin", None)
        hi = getattr(model, "_actMax", None)
        if lo is not None and hi is not None:
            self._h.set_action_limits(np.asarray(lo, np.float32).reshape(-1), np.asarray(hi, np.float32).reshape(-1))
        self._timingDict = {"total": 0., "calls": 0}
        self._steps = 0
        # log=True: save() runs predict() on every transition (controller_base.py:158-160). The reference hands the numbers to its
        # TensorBoard observer (out of scope here); they are kept in `predictions` instead, one dict per transition.
        self._log = bool(log)
        self.predictions = []

    @staticmethod
    def _state_cost_args(cost):
        """what selects the device state cost: (goal, Q) of a StaticCost or the ellipse of an ElipseCost"""
        if isinstance(cost, ElipseCost):
            return dict(ellipse=cost.ellipse)
        if hasattr(cost, "handle_args"):  # StaticQuatCost / ElipseCost3D (auv.py)
            return cost.handle_args()
        return dict(goal=cost.goal.ravel(), Q=cost.Q, q_is_full=True)

    # ---- the step ---------------------------------------------------------------------
    def next(self, state):
        import time
        t0 = time.perf_counter()
        u = self._h.next(np.asarray(state, np.float32).reshape(-1))
        self._timingDict["total"] += time.perf_counter() - t0
        self._timingDict["calls"] += 1
        self._steps += 1
        return u

    def next_with_noise(self, state, noises):
        return self._h.next_with_noise(np.asarray(state, np.float32).reshape(-1), np.asarray(noises, np.float32))

    def save(self, x, u, xNext):
        self._h.save_next(np.asarray(xNext, np.float32).reshape(-1))
        if self._log:  # controller_base.py:158-160
            self.predict(x, u, None, xNext)

    def state_error(self, stateGt, statePred):
        """controller_base.py:162-170, for the 13-state layout it assumes (position [:3], quaternion [3:7], velocities [-6:]):
        -> (|position error|, 1 - <q_gt, q_pred>, |velocity error|, velocity error [6])"""
        gt, pr = np.asarray(stateGt, np.float64).reshape(-1), np.asarray(statePred, np.float64).reshape(-1)
        return (float(np.linalg.norm(gt[:3] - pr[:3])), float(1.0 - np.dot(gt[3:7], pr[3:7])), float(np.linalg.norm(gt[-6:] - pr[-6:])),
                gt[-6:] - pr[-6:])

    def predict(self, x, u, actionSeq, xNext):
        """controller_base.py:172-210: the model's one-step prediction from (x, u) against the observed next state, the cost's distance
        metric and the step's state cost (the reference's rollout over actionSeq is commented out there; so it is here). The prediction
        and the state cost run on the device through the C-ABI helpers (mppi_model_step / mppi_state_cost). -> the predicted next state
        [1, sDim, 1]; the numbers go to `self.predictions`."""
        x, u = np.asarray(x, np.float32).reshape(1, self._sDim, 1), np.asarray(u, np.float32).reshape(1, self._aDim, 1)
        nextState = self._model.predict(x, u)
        rec = {"error": self.state_error(xNext, nextState)}
        try:
            d = self._cost.dist(x)  # (the reference: tf.squeeze(self._cost.dist(tf.expand_dims(x, axis=0)), axis=0))
            rec["dist"] = d[0] if isinstance(d, np.ndarray) and d.ndim == 3 else d
        except NotImplementedError:  # CostBase.dist is abstract in the reference too (cost_base.py:193-205)
            rec["dist"] = None
        rec["step_cost"] = float(np.asarray(self._cost.state_cost("step_cost", x)).reshape(-1)[0])
        self.predictions.append(rec)
        return nextState

    def predict_trajectory(self, x, actionSeq=None):
        """The part of predict() the reference left commented out (controller_base.py:186-201): the model rolled along the nominal action
        sequence from x — no noise — with the state cost of every state it passes and the terminal cost added once more. One device call
        per step through the model / cost helpers (a diagnostic, not the hot path). -> (trajectory [tau, sDim], cost)"""
        U = self._h.get_action_sequence() if actionSeq is None else np.asarray(actionSeq, np.float32).reshape(self._tau, self._aDim)
        state = np.asarray(x, np.float32).reshape(1, self._sDim, 1)
        traj, cost = [], 0.0
        for t in range(self._tau):
            state = self._model.build_step_graph("predict", state, U[t].reshape(1, self._aDim, 1))
            traj.append(np.asarray(state, np.float32).reshape(-1))
            cost += float(np.asarray(self._cost.state_cost("predict", state)).reshape(-1)[0])
        cost += float(np.asarray(self._cost.state_cost("predict", state)).reshape(-1)[0])  # build_final_step_cost_graph
        return np.asarray(traj), cost

    def update_model(self):
        """Push the model object's current weights and normalisation into the controller (mppi_set_mlp): what happens implicitly in
        the reference, where learner and controller share the model's tf.Variables (learner_base.py:469-496). Learned models only."""
        if not hasattr(self._model, "mlp"):
            raise AssertionError("the controller's model has no learned weights")
        self._h.set_mlp(self._model.mlp())

    def set_goal(self, goal):
        self._cost.set_goal(goal)
        self._h.set_goal(np.asarray(goal, np.float32).reshape(-1))

    @property
    def _actionSeq(self):
        return _col(self._h.get_action_sequence())

    # ---- graph helpers (same names as controller_base.py:436-560) -----------------------
    def build_model(self, scope, k, state, noises, actionSeq):
        n = np.asarray(noises, np.float32).reshape(self._k, self._tau, self._aDim)
        c = self._h.rollout_cost(np.asarray(state, np.float32).reshape(-1), np.asarray(actionSeq, np.float32).reshape(self._tau, self._aDim), n)
        return c.reshape(-1, 1, 1)

    def _upd(self, cost, noises=None):
        c = np.asarray(cost, np.float32).reshape(-1)
        n = np.zeros((self._k, self._tau, self._aDim), np.float32) if noises is None else \
            np.asarray(noises, np.float32).reshape(self._k, self._tau, self._aDim)
        return self._h.update(c, n, self._h.get_action_sequence())

    def update_terms(self, cost, noises):
        """Every intermediate of `update` (controller_base.py:436-498: beta, exp_arg, exp, nabla, weights,
        weighted_noise, update) for given costs [k,1,1] and noises [k,tau,aDim,1], in one device call
        (mppi_update). The reference exposes them as seven chained elementwise ops; here the chain is
        fused, so it is inspected as a whole."""
        r = self._upd(cost, noises)
        return dict(beta=np.array([[r["beta"]]]), exp_arg=r["arg"].reshape(-1, 1, 1), exp=r["exp"].reshape(-1, 1, 1),
                    nabla=np.array([[r["nabla"]]]), weights=r["w"].reshape(-1, 1, 1), weighted_noise=_col(r["wn"]),
                    update=_col(r["Unew"]))

    def beta(self, scope, cost):
        return np.array([[self._upd(cost)["beta"]]], np.float32)

    def update(self, scope, cost, noises, normalize=False):
        if normalize:
            raise NotImplementedError("construct ControllerBase(normalizeCost=True) and call next()")
        return self.update_terms(cost, noises)["update"]

    def prepare_action(self, scope, actions, timestep):
        return np.asarray(actions)[timestep]

    def prepare_noise(self, scope, noises, timestep):
        return np.asarray(noises)[:, timestep]

    def shift(self, scope, actionSeq, init, length):
        a = np.asarray(actionSeq, np.float32)
        return _col(_lib.shift(a[..., 0], np.asarray(init, np.float32)[..., 0], length))

    def get_next(self, scope, current, length):
        return _col(_lib.get_new(np.asarray(current, np.float32)[..., 0], length))

    def init_zeros(self, scope, size):
        return np.zeros((size, self._aDim, 1), np.float32)


class ControllerBaseCpp:
    """The C++ reference's constructor and host-loop calls (include/controller_base.hpp:60-109):
    ControllerBase(k, tau, dt, mass, s_dim, a_dim); next(x) -> u; setGoal; saveNext; toCSV.
    Defaults as controller_base.cpp:37-69: λ=1, Σ=I, Q=1, goal=(1,0)…; the model mass is the
    literal 1. the reference passes (controller_base.cpp:68), NOT the `mass` argument."""

    def __init__(self, k, tau, dt, mass, s_dim, a_dim, device=0, seed=1, **overrides):
        self.m_k, self.m_tau, self.m_s_dim, self.m_a_dim, self.m_dt, self.m_mass = k, tau, s_dim, a_dim, dt, mass
        kw = dict(k=k, tau=tau, s_dim=s_dim, a_dim=a_dim, dt=dt, mass=1.0, device=device, seed=seed)
        kw.update(overrides)
        kw.setdefault("log_rows", LOG_ROWS)  # m_db.addX / addU on every next(), controller_base.cpp:146-147
        self._h = Handle(**kw)

    def next(self, x):
        return self._h.next(x).tolist()

    def setGoal(self, goal):
        try:
            self._h.set_goal(goal)
        except _lib.MppiError as e:
            if e.status == _lib.ERR_INVALID_ARG:
                return False  # "Wrong goal size" -> false, controller_base.cpp:126-130
            raise
        return True

    def saveNext(self, x_next):
        self._h.save_next(x_next)

    def toCSV(self, filename):
        self._h.to_csv(filename)  # DataBase::toCSV's bytes (data_base.cpp:36-71)
        st = self._h.transition_log_stats()
        if st["overwritten"] or st["without_successor"]:  # the reference's m_db is unbounded: say what the bounded log left out
            import warnings
            warnings.warn("toCSV: %d transitions were overwritten (log capacity %d rows) and %d rows had no saveNext; they are not in %s"
                          % (st["overwritten"], LOG_ROWS, st["without_successor"], filename))
