"""Host-side mirror of the reference's 13-state AUV family for the control-step path:

  scripts/src/models/auv_model.py   AUVModel         Fossen 6-DOF rigid body, quaternion attitude, rk1 / rk2 / "rk4"
  scripts/src/models/nn_model.py    NNAUVModel       x' = x + denorm(nn(norm(concat(x[3:], u))))
  scripts/src/costs/static_cost.py  StaticQuatCost   (pos, 2 acos<q, g_q>, vel)-distance, Q [10,10]
  scripts/src/costs/elipse_cost.py  ElipseCost3D     elliptic track in a tilted plane

Same class and method names, argument meaning and shapes (trailing singleton kept: state [k,13,1], action [k,6,1]) as the
reference, so the parity tests read like scripts/test.py. Every numeric method is a call into libmppi_hip.so (HIP kernels);
nothing is computed in numpy and there is no fallback. fp32 on the device (the Python reference is fp64).
"""
import os

import numpy as np

from ._lib import ACTION_COST_PY, Handle
from .controller import CostBase, _col, _flat

S_DIM, A_DIM = 13, 6


class AUVModel:
    """auv_model.py:80-562. `parameters` is the reference's dict (mass, volume, density, cog, cob, Ma, linear_damping,
    quad_damping, linear_damping_forward_speed, inertial, rk); modelDict is accepted for signature compatibility."""

    def __init__(self, modelDict=None, inertialFrameId="world", actionDim=6, limMax=None, limMin=None, name="AUV", k=1, dt=0.1,
                 rk=2, parameters=None, device=0):
        if parameters is None:
            parameters = modelDict or {}
        assert inertialFrameId in ("world", "world_ned")
        for key in ("mass", "volume", "density"):  # auv_model.py:122-139
            assert parameters.get(key, 0) > 0, "%s has to be positive." % key.capitalize()
        for key in ("cog", "cob"):
            if key not in parameters:
                raise AssertionError("need to define the center of %s in the body frame" % ("gravity" if key == "cog" else "buoyancy"))
        self._parameters = dict(parameters)
        self._rk = int(parameters.get("rk", 1))  # auv_model.py:111-114 (the constructor's rk argument is "Deprecated")
        self._dt, self._k, self._name = float(dt), int(k), name
        self._stateDim, self._actionDim = S_DIM, int(actionDim)
        self._actMax, self._actMin = limMax, limMin
        self._mass, self._volume, self._density, self._gravity = parameters["mass"], parameters["volume"], parameters["density"], 9.81
        self._cog, self._cob = np.asarray(parameters["cog"], np.float64), np.asarray(parameters["cob"], np.float64)
        self._device = device
        self._h = Handle(k=1, tau=1, s_dim=S_DIM, a_dim=A_DIM, dt=dt, sigma=np.eye(A_DIM), goal=np.zeros(S_DIM), auv=self.handle_parameters(),
                         device=device)
        self._pose = None

    def handle_parameters(self, rk=None):
        p = dict(self._parameters)
        p["rk"] = self._rk if rk is None else rk
        return p

    # ---- model_base.py surface
    def get_name(self):
        return self._name

    def get_state_dim(self):
        return self._stateDim

    def get_action_dim(self):
        return self._actionDim

    def set_k(self, k):
        self._k = int(k)

    def build_step_graph(self, scope, state, action):
        return self.step(scope, state, action, rk=self._rk)

    def predict(self, state, action):
        return self.build_step_graph("predict", state, action)

    def step(self, scope, state, action, rk=1):
        h = self._h if rk == self._rk else Handle(k=1, tau=1, s_dim=S_DIM, a_dim=A_DIM, dt=self._dt, sigma=np.eye(A_DIM), goal=np.zeros(S_DIM),
                                                  auv=self.handle_parameters(rk), device=self._device)
        return _col(h.model_next(_flat(state, S_DIM), _flat(action, A_DIM)))

    def _pieces(self, state, action=None):
        st = _flat(state, S_DIM)
        ac = np.zeros((st.shape[0], A_DIM), np.float32) if action is None else _flat(action, A_DIM)
        return self._h.auv_pieces(st, ac)

    def state_dot(self, state, action):
        return _col(self._pieces(state, action)["xdot"])

    def prepare_data(self, state):
        s = np.asarray(state)
        return s[:, 0:7], s[:, 7:13]

    def body2inertial_transform(self, pose):
        """pose [k,7,1] (or a full state): sets _rotBtoI [k,3,3] and _TBtoIquat [k,4,3] (auv_model.py:353-398)"""
        p = np.asarray(pose, np.float32)[..., 0]
        st = np.zeros((p.shape[0], S_DIM), np.float32)
        st[:, :p.shape[1]] = p
        pc = self._pieces(st)
        self._pose = st
        self._rotBtoI, self._TBtoIquat = pc["rot"], pc["T"]

    def get_jacobian(self):
        k = self._rotBtoI.shape[0]
        jac = np.zeros((k, 7, 6), np.float32)
        jac[:, 0:3, 0:3], jac[:, 3:7, 3:6] = self._rotBtoI, self._TBtoIquat
        return jac

    def restoring_forces(self, scope):
        """g(eta) [k,6,1] of the pose last given to body2inertial_transform (auv_model.py:450-480)"""
        return _col(self._pieces(self._pose)["g"])

    def _vel_state(self, vel):
        v = _flat(vel, 6)
        st = np.zeros((v.shape[0], S_DIM), np.float32)
        st[:, 6], st[:, 7:] = 1.0, v
        return st

    def damping_matrix(self, scope, vel=None):
        return self._pieces(self._vel_state(vel))["D"]

    def coriolis_matrix(self, scope, vel=None):
        return self._pieces(self._vel_state(vel))["C"]

    def normalize_quat(self, pose):
        p = np.array(pose, np.float64)
        p[:, 3:7] /= np.linalg.norm(p[:, 3:7], axis=1, keepdims=True)
        return p


class NNAUVModel:
    """nn_model.py:179-304. weights: dict(W=[...], b=[...]) of Dense(16|32, relu) x 1..3 + Dense(13) with input 16 =
    concat(state[3:], action); set_Xmean_Xstd / set_Ymean_Ystd as in the reference (identity by default)."""

    def __init__(self, modelDict=None, inertialFrameId="world", k=1, stateDim=13, actionDim=6, mask=None, name="auv_nn_model",
                 weightFile=None, weights=None, dt=0.1, device=0):
        self._stateDim, self._actionDim, self._k, self._name, self._dt = int(stateDim), int(actionDim), int(k), name, float(dt)
        self.mask = np.array([[[1]] * 3 + [[0]] * 10], np.float64) if mask is None else np.asarray(mask, np.float64).reshape(1, 13, 1)
        n_in = stateDim + actionDim - 3
        self.Xmean, self.Xstd = np.zeros(n_in), np.ones(n_in)
        self.Ymean, self.Ystd = np.zeros(stateDim), np.ones(stateDim)
        if weights is None:  # the reference's architecture (nn_model.py:54-60) with Keras' default glorot-uniform kernels, zero biases
            rng = np.random.default_rng(0)
            dims = [n_in, 32, 32, 32, stateDim]
            lim = [np.sqrt(6.0 / (dims[i] + dims[i + 1])) for i in range(4)]
            weights = dict(W=[rng.uniform(-lim[i], lim[i], (dims[i], dims[i + 1])).astype(np.float32) for i in range(4)],
                           b=[np.zeros(dims[i + 1], np.float32) for i in range(4)])
        self._weights = dict(W=[np.asarray(w, np.float32) for w in weights["W"]], b=[np.asarray(b, np.float32) for b in weights["b"]])
        self._device, self._h = device, None

    def get_name(self):
        return self._name

    def get_state_dim(self):
        return self._stateDim

    def get_action_dim(self):
        return self._actionDim

    def set_k(self, k):
        self._k = int(k)

    def set_Xmean_Xstd(self, mean, std):
        self.Xmean, self.Xstd, self._h = np.asarray(mean, np.float64).reshape(-1), np.asarray(std, np.float64).reshape(-1), None

    def set_Ymean_Ystd(self, mean, std):
        self.Ymean, self.Ystd, self._h = np.asarray(mean, np.float64).reshape(-1), np.asarray(std, np.float64).reshape(-1), None

    def get_weights(self):
        out = []
        for w, b in zip(self._weights["W"], self._weights["b"]):
            out += [w.copy(), b.copy()]
        return out

    weights = get_weights

    def update_weights(self, var, msg=False):
        self._weights = dict(W=[np.asarray(v, np.float32) for v in var[0::2]], b=[np.asarray(v, np.float32) for v in var[1::2]])
        self._h = None

    def normalisation(self):
        return dict(xmean=self.Xmean, xstd=self.Xstd, ymean=self.Ymean, ystd=self.Ystd)

    # nn_model.py:137-142 (a Keras SavedModel under path/weights_step<N> there): the flat file of include/mppi_c.h's mppi_learner_save —
    # weights + this model's normalisation; the Adam slots are zero and the step count 0 when the MODEL writes it (LearnerBase.save_params
    # writes the same format with the optimizer's state). Written through the device library, like everything numeric here.
    def save_params(self, path, step):
        from ._lib import Learner
        os.makedirs(path, exist_ok=True)
        f = os.path.join(path, "weights_step{}".format(step))
        lrn = Learner(dict(W=self._weights["W"], b=self._weights["b"]), device=self._device)
        lrn.save(f, self.normalisation())
        lrn.close()
        return f

    def load_params(self, path):
        from ._lib import Learner
        lrn, norm = Learner.from_file(path, device=self._device)
        w = lrn.get_weights()
        lrn.close()
        self._weights, self._h = dict(W=w["W"], b=w["b"]), None
        if norm is not None:
            self.set_Xmean_Xstd(norm["xmean"], norm["xstd"])
            self.set_Ymean_Ystd(norm["ymean"], norm["ystd"])

    def mlp(self):
        """the dict Handle(nnauv=...) takes"""
        return dict(W=self._weights["W"], b=self._weights["b"], xmean=self.Xmean, xstd=self.Xstd, ymean=self.Ymean, ystd=self.Ystd)

    def _handle(self):
        if self._h is None:
            self._h = Handle(k=1, tau=1, s_dim=S_DIM, a_dim=A_DIM, dt=self._dt, sigma=np.eye(A_DIM), goal=np.zeros(S_DIM), nnauv=self.mlp(),
                             device=self._device)
        return self._h

    def build_step_graph(self, scope, state, action):
        return _col(self._handle().model_next(_flat(state, S_DIM), _flat(action, A_DIM)))

    def predict(self, state, action):
        return self.build_step_graph("predict", state, action)

    # data preparation is host bookkeeping around the path (numpy), nn_model.py:241-304
    def prepare_training_data(self, stateT, stateT1, action, norm=True):
        stateT, stateT1, action = (np.asarray(v, np.float64) for v in (stateT, stateT1, action))
        tFrom = self.mask * stateT
        poseBIt, poseBIt1 = stateT - tFrom, stateT1 - tFrom
        X = np.concatenate([stateT[:, 3:], action], axis=1)[..., 0]
        Y = (poseBIt1 - poseBIt)[..., 0]
        if norm:
            X, Y = (X - self.Xmean) / self.Xstd, (Y - self.Ymean) / self.Ystd
        return X, Y

    def prepare_data(self, state, action):
        data = np.concatenate([np.asarray(state, np.float64)[:, 3:], np.asarray(action, np.float64)], axis=1)[..., 0]
        return (data - self.Xmean) / self.Xstd

    def denormalizeY(self, normY):
        return normY * self.Ystd + self.Ymean

    def denormalizeX(self, normX):
        return normX * self.Xstd + self.Xmean

    def next_state(self, state, delta):
        return np.asarray(state) + np.asarray(delta)


def euler_from_quaternion(q):
    """tensorflow_graphics.geometry.transformation.euler.from_quaternion (what NNAUVModelSpeed.to_euler calls, nn_model.py:566-588), numpy,
    float64: [k,4] (x, y, z, w) -> [k,3] (theta_x, theta_y, theta_z), R = Rz Ry Rx. Host bookkeeping (training data); the device
    kernels carry their own fp32 form."""
    q = np.asarray(q, np.float64).reshape(-1, 4)
    eps = 2.0 * np.finfo(np.float64).eps
    shr = 1.0 - 4.0 * eps
    x, y, z, w = q.T
    tx, ty, tz = 2 * x * shr, 2 * y * shr, 2 * z * shr
    twx, twy, twz, txx, txy, txz, tyy, tyz, tzz = tx * w, ty * w, tz * w, tx * x, ty * x, tz * x, ty * y, tz * y, tz * z
    r00, r10, r21, r22 = (1 - (tyy + tzz)) * shr, (txy + twz) * shr, (tyz + twx) * shr, (1 - (txx + tyy)) * shr
    r20, r01, r02 = (txz - twy) * shr, (txy - twz) * shr, (txz + twy) * shr
    sgn = lambda v: np.where(v >= 0, 1.0, -1.0)
    th_y = -np.arcsin(r20)
    sc = sgn(np.cos(th_y))
    r00g, r22g = sgn(r00) * eps + r00, sgn(r22) * eps + r22
    general = np.stack([np.arctan2(r21 * sc, r22g * sc), th_y, np.arctan2(r10 * sc, r00g * sc)], axis=-1)
    s20 = sgn(r20)
    lock = np.stack([np.arctan2(-s20 * r01, -s20 * (sgn(r02) * eps + r02)), -s20 * (np.pi / 2), np.zeros_like(r20)], axis=-1)
    return np.where((np.abs(np.abs(r20) - 1.0) < 1e-6)[:, None], lock, general)


class NNAUVModelSpeed(NNAUVModel):
    """nn_model.py:307-588: the network — Dense(16, relu) x 3 + Dense(6) on 15 inputs (3 Euler angles, 6 velocities, 6 forces) — predicts
    the velocity delta; the pose is integrated with the quaternion kinematics over dt and renormalised."""

    def __init__(self, modelDict=None, inertialFrameId="world", k=1, stateDim=13, actionDim=6, mask=None, name="auv_nn_model",
                 weightFile=None, weights=None, dt=0.1, device=0):
        if weights is None:  # the reference's architecture (nn_model.py:340-346) with Keras' default glorot-uniform kernels, zero biases
            rng = np.random.default_rng(0)
            dims = [stateDim + actionDim - 3 - 1, 16, 16, 16, 6]
            lim = [np.sqrt(6.0 / (dims[i] + dims[i + 1])) for i in range(4)]
            weights = dict(W=[rng.uniform(-lim[i], lim[i], (dims[i], dims[i + 1])).astype(np.float32) for i in range(4)],
                           b=[np.zeros(dims[i + 1], np.float32) for i in range(4)])
        NNAUVModel.__init__(self, modelDict, inertialFrameId, k, stateDim, actionDim, mask, name, weightFile, weights, dt, device)
        n_in = stateDim + actionDim - 3 - 1  # position dropped, quaternion -> Euler angles (nn_model.py:348-352)
        self.Xmean, self.Xstd = np.zeros(n_in), np.ones(n_in)
        self.Ymean, self.Ystd = np.zeros(6), np.ones(6)

    def handle_kw(self):
        """the Handle keyword ControllerBase passes on"""
        return dict(nnauv_speed=self.mlp())

    def _handle(self):
        if self._h is None:
            self._h = Handle(k=1, tau=1, s_dim=S_DIM, a_dim=A_DIM, dt=self._dt, sigma=np.eye(A_DIM), goal=np.zeros(S_DIM), device=self._device,
                             **self.handle_kw())
        return self._h

    def to_euler(self, stateQ):  # :566-588 [k,13,1] -> [k,12,1]
        st = np.asarray(stateQ, np.float64)
        return np.concatenate([st[:, 0:3], euler_from_quaternion(st[:, 3:7, 0])[..., None], st[:, 7:]], axis=1)

    def prepare_data(self, state, action):  # :438-461 -> [k,15]
        X = np.concatenate([self.to_euler(state)[:, 3:], np.asarray(action, np.float64)], axis=1)[..., 0]
        return (X - self.Xmean) / self.Xstd

    def prepare_training_data(self, stateT, stateT1, action, norm=True):  # :382-436 -> X [k,15], Y [k,6] (velocity delta)
        stateT, stateT1, action = (np.asarray(v, np.float64) for v in (stateT, stateT1, action))
        X = np.concatenate([self.to_euler(stateT)[:, 3:], action], axis=1)[..., 0]
        Y = (stateT1 - stateT)[:, 7:, 0]
        if norm:
            X, Y = (X - self.Xmean) / self.Xstd, (Y - self.Ymean) / self.Ystd
        return X, Y

    def next_state(self, state, delta):  # :463-472, on the device: the step with a zero network output would need the weights; host form
        st = np.asarray(state, np.float64)
        out = st.copy()
        for i in range(st.shape[0]):
            x, y, z, w = st[i, 3:7, 0]
            rot = np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                            [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                            [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])
            T = 0.5 * np.array([[-x, -y, -z], [w, -z, y], [z, w, -x], [-y, x, w]])  # this class's row order (:545-555)
            pose = st[i, 0:7, 0] + np.concatenate([rot @ st[i, 7:10, 0], T @ st[i, 10:13, 0]]) * self._dt
            pose[3:7] /= np.sqrt(max(float(pose[3:7] @ pose[3:7]), 1e-12))
            out[i, 0:7, 0] = pose
            out[i, 7:13, 0] = st[i, 7:13, 0] + np.asarray(delta, np.float64)[i].reshape(6)
        return out


class StaticQuatCost(CostBase):
    """static_cost.py:73-159: goal [13,1], Q [10,10] (or its diagonal with diag=True)."""

    def __init__(self, lam, gamma, upsilon, sigma, goal, Q, diag=False, device=0):
        CostBase.__init__(self, lam, gamma, upsilon, sigma, device)
        self.Q = np.asarray(Q, np.float32)
        if diag:
            self.Q = np.diag(self.Q.ravel())
        self.q_shape = self.Q.shape
        if self.q_shape != (10, 10):
            raise AssertionError("Goal tensor shape error, expected: [10, 10], got {}".format(self.q_shape))
        self._sDim, self._h = S_DIM, None
        self.set_goal(goal)

    def set_goal(self, goal):
        g = np.asarray(goal, np.float32)
        if g.shape != (13, 1):
            raise AssertionError("Goal tensor shape error, expected: [{}, 1], got {}".format(self.q_shape[0], g.shape))
        self.goal = g
        if self._h is not None:
            self._h.set_goal(g.ravel())

    setGoal = set_goal

    def get_goal(self):
        return self.goal

    def handle_args(self):
        return dict(goal=self.goal.ravel(), Q=self.Q, quat_cost=True)

    def _handle(self):
        if self._h is None:
            self._h = Handle(k=1, tau=1, s_dim=S_DIM, a_dim=self._aDim, lam=self.lam, gamma=self.gamma, upsilon=self.upsilon, sigma=self.sigma,
                             action_cost=self._action_cost_kind, device=self._device, **self.handle_args())
        return self._h

    def state_cost(self, scope, state):
        return self._handle().state_cost(_flat(state, S_DIM)).reshape(-1, 1, 1)

    def action_cost(self, scope, action, noise):
        return self._handle().action_cost(np.asarray(action).reshape(-1), _flat(noise, self._aDim)).reshape(-1, 1, 1)

    def dist(self, state):
        """[k,10,1]; host bookkeeping (plots / logs in the reference), static_cost.py:141-159"""
        st, g = np.asarray(state, np.float64)[..., 0], self.goal.astype(np.float64)[:, 0]
        theta = 2 * np.arccos(st[:, 3:7] @ g[3:7])
        return np.concatenate([st[:, :3] - g[:3], theta[:, None], st[:, -6:] - g[-6:]], axis=1)[..., None]


class ElipseCost3D(CostBase):
    """elipse_cost.py:101-246. Constructor arguments in the reference's order; `center` and `v_speed` are stored by the
    reference and never enter the cost — kept for signature compatibility."""

    def __init__(self, lam, gamma, upsilon, sigma, normal, aVec, axis, center, speed, v_speed, mState, mVel, device=0):
        CostBase.__init__(self, lam, gamma, upsilon, sigma, device)
        self.normal, self.aVec = np.asarray(normal, np.float64).reshape(3), np.asarray(aVec, np.float64).reshape(3)
        self.axis2, self.center = np.asarray(axis, np.float64).reshape(2), np.asarray(center, np.float64).reshape(3, 1)
        self.gv, self.mS, self.mV = float(speed), float(mState), float(mVel)
        self.t = self.center
        self._sDim, self._h = S_DIM, None
        # R = inv([aVec bVec normal])^T (prepare_consts, :163-167): host bookkeeping, the device builds its own copy from the same inputs
        N = np.stack([self.aVec, np.cross(self.normal, self.aVec), self.normal], axis=-1)
        self.R = np.linalg.inv(N).T

    def handle_args(self):
        return dict(ellipse3d=dict(normal=self.normal, aVec=self.aVec, axis=self.axis2, speed=self.gv, m_state=self.mS, m_vel=self.mV))

    def _handle(self):
        if self._h is None:
            self._h = Handle(k=1, tau=1, s_dim=S_DIM, a_dim=self._aDim, lam=self.lam, gamma=self.gamma, upsilon=self.upsilon, sigma=self.sigma,
                             action_cost=self._action_cost_kind, device=self._device, **self.handle_args())
        return self._h

    def state_cost(self, scope, state):
        return self._handle().state_cost(_flat(state, S_DIM)).reshape(-1, 1, 1)

    def _terms(self, pos=None, quat=None, vel=None):
        """the three error terms of plane-frame quantities (mppi_ellipse3d_terms, in_plane_frame = 1) -> [k,3]"""
        k = next(v for v in (pos, quat, vel) if v is not None).shape[0]
        st = np.zeros((k, S_DIM), np.float32)
        st[:, 6] = 1.0
        if pos is not None:
            st[:, :3] = pos
        if quat is not None:
            st[:, 3:7] = quat
        if vel is not None:
            st[:, 7:13] = vel
        return self._handle().ellipse3d_terms(st, in_plane_frame=True)

    def position_error(self, position):
        """points in the plane frame [k,3,1] -> [k,1,1] (:169-190)"""
        return self._terms(pos=np.asarray(position, np.float32)[..., 0])[:, 0].reshape(-1, 1, 1)

    def orientation_error(self, pose):
        """poses in the plane frame [k,7,1] -> [k] (:192-222)"""
        ps = np.asarray(pose, np.float32)[..., 0]
        return self._terms(pos=ps[:, :3], quat=ps[:, 3:7])[:, 1]

    def velocity_error(self, velocity):
        """[k,6,1] -> [k,1,1] (:224-246)"""
        return self._terms(vel=np.asarray(velocity, np.float32)[..., 0])[:, 2].reshape(-1, 1, 1)


def auv_task(H, learned=False):
    """The reference's AUV task as one Handle configuration: rexrov2 parameters (config/models/rexrov2.default.yaml), the static 13-state
    goal with its diagonal Q (config/tasks/static_cost_auv.yaml), noise 1500 N on every axis, dt 0.1 (config/envs/uuv_sim.default.yaml)."""
    params = dict(mass=1862.87, volume=1.8121303501945525, density=1028.0, cog=[0, 0, 0], cob=[0, 0, 0.3],
                  Ma=[[779.79, -6.8773, -103.32, 8.5426, -165.54, -7.8033], [-6.8773, 1222, 51.29, 409.44, -5.8488, 62.726],
                      [-103.32, 51.29, 3659.9, 6.1112, -386.42, 10.774], [8.5426, 409.44, 6.1112, 534.9, -10.027, 21.019],
                      [-165.54, -5.8488, -386.42, -10.027, 842.69, -1.1162], [-7.8033, 62.726, 10.775, 21.019, -1.1162, 224.32]],
                  linear_damping=[-74.82, -69.48, -728.4, -268.8, -309.77, -105], quad_damping=[-748.22, -992.53, -1821.01, -672, -774.44, -523.27],
                  inertial=dict(ixx=525.39, iyy=794.2, izz=691.23, ixy=1.44, ixz=33.41, iyz=2.6), rk=2)
    cfg = dict(tau=H, s_dim=S_DIM, a_dim=A_DIM, dt=0.1, lam=1.0, sigma=(1500.0 * np.eye(A_DIM)).astype(np.float32),
               goal=[1.0, 2.0, -10.0, 0.0, 0.0, 0.0, 1.0] + [0.0] * 6, Q=np.array([1e4] * 3 + [100.0] * 4 + [1.0] * 6, np.float32), seed=1,
               x0=[0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 1.0] + [0.0] * 6)
    if not learned:
        cfg["auv"] = params
    return cfg
