"""ctypes binding of libmppi_hip.so (include/mppi_c.h) — the only way Python reaches the kernels.

There is no CPU fallback: a missing library or a missing GPU raises (MppiError / OSError).
torch is imported before the library is loaded on purpose: torch's wheel bundles its own
libamdhip64.so.7; loading it first makes our library bind to that same HIP runtime instance
(same soname) instead of bringing a second runtime into the process.
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
# MPPI_SO_PATH selects a variant build of the SAME library (tools/ablate.py); never a different backend.
SO_PATH = os.environ.get("MPPI_SO_PATH") or os.path.join(HERE, "libmppi_hip.so")

OK, ERR_INVALID_ARG, ERR_NO_DEVICE, ERR_HIP, ERR_UNSUPPORTED, ERR_SINGULAR_SIGMA, ERR_ALLOC, ERR_IO, ERR_EXCHANGE = range(9)
MODEL_POINT_MASS, MODEL_MLP, MODEL_AUV, MODEL_NN_AUV, MODEL_NN_AUV_SPEED = 0, 1, 2, 3, 4
STATE_COST_QUADRATIC, STATE_COST_ELLIPSE, STATE_COST_QUAT, STATE_COST_ELLIPSE3D = 0, 1, 2, 3
ACTION_COST_CPP, ACTION_COST_PY = 0, 1
DBG_COSTS, DBG_BETA, DBG_ETA, DBG_WEIGHTS, DBG_NOISE, DBG_U_UPDATED, DBG_AUX = range(7)
CSV_REFERENCE, CSV_ROUNDTRIP = 0, 1
# mppi_set_tuning items (diagnostics; the library reads no environment variable)
TUNING = {"force_tile_kernel": 0, "pc_producers": 1, "pc_balance": 2, "pc_lds_min": 3, "sync_spin": 4, "p2p_fault": 5, "mlp_v1": 6, "mlp32_valu": 7,
          "trace": 8, "gen_one_wave": 9, "fused_step": 10, "armed_us": 11, "armed_always": 12, "prelaunch": 13}
P2P_FAULTS = {"": 0, "export": 1, "probe": 2}

FP = C.POINTER(C.c_float)
DP = C.POINTER(C.c_double)


class MppiError(RuntimeError):
    def __init__(self, status, text):
        super().__init__("mppi status %d: %s" % (status, text))
        self.status = status


class MlpDesc(C.Structure):
    _fields_ = [("n_layers", C.c_int32), ("widths", C.POINTER(C.c_int32)),
                ("W", C.POINTER(FP)), ("b", C.POINTER(FP)),
                ("xmean", FP), ("xstd", FP), ("ymean", FP), ("ystd", FP)]


class AuvDesc(C.Structure):
    _fields_ = [("mass", C.c_float), ("volume", C.c_float), ("density", C.c_float), ("gravity", C.c_float),
                ("cog", C.c_float * 3), ("cob", C.c_float * 3), ("inertial", C.c_float * 6),
                ("added_mass", FP), ("linear_damping", FP), ("linear_damping_forward_speed", FP), ("quad_damping", FP),
                ("rk", C.c_int32)]


class Config(C.Structure):
    _fields_ = [("struct_size", C.c_uint32), ("k", C.c_int32), ("tau", C.c_int32),
                ("s_dim", C.c_int32), ("a_dim", C.c_int32), ("dt", C.c_float), ("mass", C.c_float),
                ("lam", C.c_float), ("gamma", C.c_float), ("upsilon", C.c_float),
                ("action_cost_kind", C.c_int32), ("normalize_cost", C.c_int32),
                ("sigma", FP), ("goal", FP), ("Q", FP), ("q_is_full", C.c_int32),
                ("seed", C.c_uint64), ("model_kind", C.c_int32), ("mlp", C.POINTER(MlpDesc)),
                ("device", C.c_int32), ("shard_rank", C.c_int32), ("shard_count", C.c_int32),
                ("flags", C.c_int32), ("state_cost_kind", C.c_int32), ("ellipse", FP),
                ("auv", C.POINTER(AuvDesc)), ("quat_Q", FP), ("ellipse3d", FP)]


class Collectives(C.Structure):
    """mppi_collectives: the caller's all-gather / all-reduce with ncclAllGather's / ncclAllReduce's signatures + the communicator"""
    _fields_ = [("all_gather", C.c_void_p), ("all_reduce", C.c_void_p), ("comm", C.c_void_p)]


# name -> (restype, argtypes); must list EVERY symbol include/mppi_c.h declares
# (tests/test_capi_symbols.py parses the header and compares).
_H = C.c_void_p
SIGNATURES = {
    "mppi_abi_version": (C.c_int, []),
    "mppi_version": (C.c_char_p, []),
    "mppi_status_string": (C.c_char_p, [C.c_int]),
    "mppi_device_count": (C.c_int, []),
    "mppi_config_init": (C.c_int, [C.POINTER(Config), C.c_int, C.c_int, C.c_float, C.c_float, C.c_int, C.c_int]),
    "mppi_create": (C.c_int, [C.POINTER(Config), C.POINTER(_H)]),
    "mppi_destroy": (None, [_H]),
    "mppi_last_error": (C.c_char_p, [_H]),
    "mppi_set_goal": (C.c_int, [_H, FP, C.c_int]),
    "mppi_set_mlp": (C.c_int, [_H, C.POINTER(MlpDesc)]),
    "mppi_next": (C.c_int, [_H, FP, C.c_int, FP, C.c_int]),
    "mppi_next_with_noise": (C.c_int, [_H, FP, C.c_int, FP, C.c_size_t, FP, C.c_int]),
    "mppi_save_next": (C.c_int, [_H, FP, C.c_int]),
    "mppi_to_csv": (C.c_int, [_H, C.c_char_p]),
    "mppi_to_csv_format": (C.c_int, [_H, C.c_char_p, C.c_int]),
    "mppi_set_transition_log": (C.c_int, [_H, C.c_int]),
    "mppi_transition_log_stats": (C.c_int, [_H, C.POINTER(C.c_uint64), C.POINTER(C.c_uint64), C.POINTER(C.c_uint64)]),
    "mppi_set_tuning": (C.c_int, [_H, C.c_int, C.c_int]),
    "mppi_get_action_sequence": (C.c_int, [_H, FP, C.c_int]),
    "mppi_set_action_sequence": (C.c_int, [_H, FP, C.c_int]),
    "mppi_get_step_counter": (C.c_int, [_H, C.POINTER(C.c_uint64)]),
    "mppi_set_step_counter": (C.c_int, [_H, C.c_uint64]),
    "mppi_debug_get": (C.c_int, [_H, C.c_int, FP, C.c_size_t]),
    "mppi_model_step": (C.c_int, [_H, FP, C.c_int, FP, C.c_int, FP, FP, FP]),
    "mppi_auv_pieces": (C.c_int, [_H, FP, FP, C.c_int, FP]),
    "mppi_ellipse3d_terms": (C.c_int, [_H, FP, C.c_int, C.c_int, FP]),
    "mppi_state_cost": (C.c_int, [_H, FP, C.c_int, FP]),
    "mppi_action_cost": (C.c_int, [_H, FP, FP, C.c_int, FP]),
    "mppi_step_cost": (C.c_int, [_H, FP, FP, FP, C.c_int, FP]),
    "mppi_rollout_cost": (C.c_int, [_H, FP, FP, FP, FP]),
    "mppi_update": (C.c_int, [_H, FP, FP, FP, FP, FP, FP, FP, FP, FP, FP]),
    "mppi_get_new": (C.c_int, [FP, C.c_int, C.c_int, C.c_int, FP]),
    "mppi_shift": (C.c_int, [FP, C.c_int, C.c_int, FP, C.c_int, C.c_int, FP]),
    "mppi_record_size": (C.c_int, [_H]),
    "mppi_local_samples": (C.c_int, [_H]),
    "mppi_sample_offset": (C.c_int, [_H]),
    "mppi_next_device": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_shard_partial": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_shard_finish": (C.c_int, [_H, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p]),
    "mppi_shard_cost_range": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_shard_partial_normalized": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_shard_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.POINTER(Collectives), C.c_void_p]),
    "mppi_synchronize": (C.c_int, [_H]),
    "mppi_set_action_limits": (C.c_int, [_H, FP, FP, C.c_int]),
    "mppi_set_sequence_filter": (C.c_int, [_H, C.c_int, C.c_int]),
    "mppi_shard_p2p_export": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mppi_shard_p2p_open": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_void_p)]),
    "mppi_shard_p2p_attach": (C.c_int, [_H, C.POINTER(C.c_void_p), C.c_int, C.c_int]),
    "mppi_shard_p2p_probe": (C.c_int, [_H, C.c_void_p, C.POINTER(C.c_int)]),
    "mppi_shard_p2p_step": (C.c_int, [_H, C.c_void_p, C.c_void_p, C.c_void_p]),
    "mppi_shard_p2p_status": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "mppi_profile_begin": (C.c_int, [_H, C.c_int]),
    "mppi_profile_end": (C.c_int, [_H, FP, FP, C.POINTER(C.c_int)]),
    "mppi_rollout_kernel_name": (C.c_int, [_H, C.c_char_p, C.c_size_t]),
    # the learner (LearnerBase.train / _train_step)
    "mppi_learner_create": (C.c_int, [C.c_int, C.POINTER(C.c_int32), C.POINTER(FP), C.POINTER(FP), C.c_int, C.POINTER(_H)]),
    "mppi_learner_destroy": (None, [_H]),
    "mppi_learner_last_error": (C.c_char_p, [_H]),
    "mppi_learner_set_data": (C.c_int, [_H, FP, FP, C.c_int]),
    "mppi_learner_train": (C.c_int, [_H, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, FP, FP]),
    "mppi_learner_evaluate": (C.c_int, [_H, FP, FP, FP]),
    "mppi_learner_get_weights": (C.c_int, [_H, C.POINTER(FP), C.POINTER(FP)]),
    "mppi_learner_set_weights": (C.c_int, [_H, C.POINTER(FP), C.POINTER(FP)]),
    "mppi_learner_reset_optimizer": (C.c_int, [_H]),
    "mppi_learner_get_step": (C.c_int, [_H, C.POINTER(C.c_int)]),
    "mppi_learner_save": (C.c_int, [_H, C.c_char_p, DP, DP, DP, DP]),
    "mppi_learner_load": (C.c_int, [_H, C.c_char_p, DP, DP, DP, DP, C.POINTER(C.c_int)]),
    "mppi_learner_peek": (C.c_int, [C.c_char_p, C.POINTER(C.c_int), C.POINTER(C.c_int32)]),
}

_lib = None


def load():
    """Load libmppi_hip.so (built in-tree by build.py). Raises OSError when it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    import torch  # noqa: F401  (one HIP runtime per process — see module docstring)
    if not os.path.exists(SO_PATH):
        raise OSError("%s is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    lib = C.CDLL(SO_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.mppi_abi_version() != 5:
        raise OSError("libmppi_hip.so ABI version mismatch")
    _lib = lib
    return lib


def f32(x, shape=None):
    a = np.ascontiguousarray(np.asarray(x, dtype=np.float32))
    return a.reshape(shape) if shape is not None else a


def fp(a):
    return a.ctypes.data_as(FP) if a is not None else None


def _mlp_desc(mlp, n_in, n_out):
    """mppi_mlp_desc of an mlp dict(W, b[, xmean, xstd, ymean, ystd]) -> (desc, objects that must outlive its use)"""
    Ws = [f32(w) for w in mlp["W"]]
    bs = [f32(b).ravel() for b in mlp["b"]]
    held = Ws + bs
    desc = MlpDesc()
    desc.n_layers = len(Ws)
    widths = (C.c_int32 * len(Ws))(*[w.shape[1] for w in Ws])
    Wp = (FP * len(Ws))(*[fp(w) for w in Ws])
    bp = (FP * len(bs))(*[fp(b) for b in bs])
    desc.widths, desc.W, desc.b = widths, Wp, bp
    for name, n in (("xmean", n_in), ("xstd", n_in), ("ymean", n_out), ("ystd", n_out)):
        if mlp.get(name) is not None:
            held.append(f32(mlp[name], (n,)))
            setattr(desc, name, fp(held[-1]))
    return desc, held + [desc, widths, Wp, bp]


class Handle:
    """RAII wrapper of one mppi_handle (one controller on one GPU)."""

    def __init__(self, k, tau, s_dim, a_dim, dt=0.1, mass=1.0, lam=1.0, gamma=1.0, upsilon=1.0,
                 sigma=None, goal=None, Q=None, q_is_full=None, action_cost=ACTION_COST_CPP,
                 normalize_cost=False, seed=1, device=0, shard_rank=0, shard_count=1, mlp=None,
                 upsilon_scales_noise=False, mlp_bf16x3=False, fp_contract=False, tuning=None, log_rows=0, ellipse=None,
                 auv=None, nnauv=None, quat_cost=False, ellipse3d=None, nnauv_speed=None):
        """mlp: dict(W=[W1,W2,W3], b=[b1,b2,b3], xmean=, xstd=, ymean=, ystd=) selects the learned
        model_base (Dense(256,relu) x2 + Dense(s_dim); Keras [in x out] kernels).
        tuning: dict of diagnostic switches (keys of TUNING) applied with mppi_set_tuning right after creation.
        log_rows: capacity of the transition log (mppi_set_transition_log); 0 = off.
        auv: the reference's AUVModel `parameters` dict (auv_model.py:85-245: mass, volume, density, cog, cob, Ma, linear_damping,
        quad_damping, linear_damping_forward_speed, inertial{ixx..iyz}, rk) selects the Fossen model (s_dim 13, a_dim 6);
        nnauv: an mlp dict whose first kernel has s+a-3 = 16 rows selects NNAUVModel (nn_model.py:179-304);
        nnauv_speed: an mlp dict with 15 input rows (Euler angles, velocities, forces) and 6 outputs selects NNAUVModelSpeed (nn_model.py:307-588);
        quat_cost: StaticQuatCost (static_cost.py:73-159) with goal [13] and Q [10,10] (or its 10 diagonal entries);
        ellipse3d: dict(normal, aVec, axis, speed, m_state, m_vel) selects ElipseCost3D (elipse_cost.py:101-246)."""
        lib = self.lib = load()
        cfg = Config()
        self._check(lib.mppi_config_init(C.byref(cfg), k, tau, dt, mass, s_dim, a_dim), None)
        cfg.lam, cfg.gamma, cfg.upsilon = lam, gamma, upsilon
        cfg.action_cost_kind, cfg.normalize_cost = action_cost, int(bool(normalize_cost))
        cfg.seed, cfg.device, cfg.shard_rank, cfg.shard_count = seed, device, shard_rank, shard_count
        cfg.flags = (1 if upsilon_scales_noise else 0) | (2 if mlp_bf16x3 else 0) | (4 if fp_contract else 0)  # MPPI_FLAG_UPSILON_SCALES_NOISE | _MLP_BF16X3 | _FP_CONTRACT
        keep = []
        if ellipse is not None:  # ElipseCost (elipse_cost.py:9-85): dict or the 7 numbers a, b, cx, cy, speed, m_state, m_vel
            e = [ellipse[k] for k in ("a", "b", "cx", "cy", "speed", "m_state", "m_vel")] if isinstance(ellipse, dict) else ellipse
            keep.append(f32(e, (7,)))
            cfg.state_cost_kind, cfg.ellipse = 1, fp(keep[-1])
        if auv is not None:
            d = AuvDesc()
            d.mass, d.volume, d.density, d.gravity = auv["mass"], auv["volume"], auv["density"], auv.get("gravity", 0.0)
            d.rk = int(auv.get("rk", 1))  # auv_model.py:111-114: rk defaults to 1 when the parameters do not carry it
            for i in range(3):
                d.cog[i], d.cob[i] = auv["cog"][i], auv["cob"][i]
            for i, key in enumerate(("ixx", "iyy", "izz", "ixy", "ixz", "iyz")):
                d.inertial[i] = auv["inertial"][key]

            def mat6(key):  # a 6-vector is the diagonal (auv_model.py:186-195)
                if auv.get(key) is None:
                    return None
                m6 = np.asarray(auv[key], np.float32)
                keep.append(f32(np.diag(m6) if m6.shape == (6,) else m6, (36,)))
                return fp(keep[-1])
            d.added_mass, d.linear_damping = mat6("Ma"), mat6("linear_damping")
            d.linear_damping_forward_speed = mat6("linear_damping_forward_speed")
            if auv.get("quad_damping") is not None:
                keep.append(f32(auv["quad_damping"], (6,)))
                d.quad_damping = fp(keep[-1])
            keep.append(d)
            cfg.model_kind, cfg.auv = MODEL_AUV, C.pointer(d)
        if quat_cost:
            q = f32(Q if Q is not None else np.ones(10))
            keep.append(f32(np.diag(q) if q.ndim == 1 else q, (100,)))
            cfg.state_cost_kind, cfg.quat_Q = STATE_COST_QUAT, fp(keep[-1])
            Q = None
        if ellipse3d is not None:
            e = ellipse3d
            keep.append(f32(list(np.ravel(e["normal"])) + list(np.ravel(e["aVec"])) + list(np.ravel(e["axis"])) + [e["speed"], e["m_state"], e["m_vel"]], (11,)))
            cfg.state_cost_kind, cfg.ellipse3d = STATE_COST_ELLIPSE3D, fp(keep[-1])
        if nnauv is not None:
            mlp = nnauv
        if nnauv_speed is not None:
            mlp = nnauv_speed
        if sigma is not None:
            keep.append(f32(sigma, (a_dim, a_dim)))
            cfg.sigma = fp(keep[-1])
        if goal is not None:
            keep.append(f32(goal, (s_dim,)))
            cfg.goal = fp(keep[-1])
        if Q is not None:
            q = f32(Q)
            if q_is_full is None:
                q_is_full = q.ndim == 2
            keep.append(f32(q, (s_dim, s_dim) if q_is_full else (s_dim,)))
            cfg.Q, cfg.q_is_full = fp(keep[-1]), int(q_is_full)
        self._mlp_io = None
        if mlp is not None:
            n_in = 15 if nnauv_speed is not None else s_dim + a_dim - (3 if nnauv is not None else 0)
            self._mlp_io = (n_in, 6 if nnauv_speed is not None else s_dim)
            desc, held = _mlp_desc(mlp, *self._mlp_io)
            keep += held
            cfg.model_kind = MODEL_NN_AUV_SPEED if nnauv_speed is not None else (MODEL_NN_AUV if nnauv is not None else MODEL_MLP)
            cfg.mlp = C.pointer(desc)
        self.h = _H()
        self.k, self.tau, self.s, self.a = k, tau, s_dim, a_dim
        st = lib.mppi_create(C.byref(cfg), C.byref(self.h))
        if st != OK:
            self.h = _H()
            self._check(st, None)
        self.k_local = lib.mppi_local_samples(self.h)
        self._stage = None
        self.k_offset = lib.mppi_sample_offset(self.h)
        self.record_size = lib.mppi_record_size(self.h)
        for key, val in (tuning or {}).items():
            self.set_tuning(key, val)
        if log_rows:
            self.set_transition_log(log_rows)

    def set_tuning(self, key, value):
        if key == "p2p_fault" and isinstance(value, str):
            value = P2P_FAULTS[value]
        self._check(self.lib.mppi_set_tuning(self.h, TUNING[key], int(value)))

    def set_transition_log(self, max_rows):
        self._check(self.lib.mppi_set_transition_log(self.h, int(max_rows)))

    def _check(self, st, h=True):
        if st != OK:
            txt = self.lib.mppi_last_error(self.h if (h and self.h) else None) or b""
            raise MppiError(st, "%s (%s)" % (txt.decode(), self.lib.mppi_status_string(st).decode()))

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.mppi_destroy(h)

    def __del__(self):
        # At interpreter shutdown the module globals and the ctypes function objects may already be torn down (VERDICT r02:
        # "TypeError: 'NoneType' object is not callable" from __del__ = close): a finaliser must never raise.
        try:
            self.close()
        except Exception:
            pass

    # ---- host loop -------------------------------------------------------------------
    def set_goal(self, goal):
        g = f32(goal).ravel()
        self._check(self.lib.mppi_set_goal(self.h, fp(g), g.size))

    def set_mlp(self, mlp):
        """mppi_set_mlp: new weights / normalisation for a learned-model handle (same layer widths), from the next step on"""
        if self._mlp_io is None:
            raise MppiError(1, "not a learned-model handle")
        desc, held = _mlp_desc(mlp, *self._mlp_io)
        self._check(self.lib.mppi_set_mlp(self.h, C.byref(desc)))
        del held

    def next(self, x):
        # persistent staging arrays with ready-made ctypes pointers: building them per call costs more host time
        # (~5 us) than the launch of the step itself
        if self._stage is None:
            xs, us = np.zeros(self.s, np.float32), np.zeros(self.a, np.float32)
            self._stage = (xs, us, fp(xs), fp(us))
        xs, us, xp, up = self._stage
        xin = np.asarray(x).reshape(-1)
        if xin.size != self.s:  # let the C-ABI report it (MPPI_ERR_INVALID_ARG, as the reference's size checks)
            xin = f32(xin)
            self._check(self.lib.mppi_next(self.h, fp(xin), xin.size, up, self.a))
        xs[:] = xin
        st = self.lib.mppi_next(self.h, xp, self.s, up, self.a)
        if st != OK:
            self._check(st)
        return us.copy()

    def next_with_noise(self, x, eps):
        x, eps = f32(x).ravel(), f32(eps).ravel()
        u = np.zeros(self.a, np.float32)
        self._check(self.lib.mppi_next_with_noise(self.h, fp(x), x.size, fp(eps), eps.size, fp(u), u.size))
        return u

    def save_next(self, x_next):
        x = f32(x_next).ravel()
        self._check(self.lib.mppi_save_next(self.h, fp(x), x.size))

    def transition_log_stats(self):
        """-> dict(held, overwritten, without_successor): what to_csv will and will not write"""
        a, b, c = C.c_uint64(0), C.c_uint64(0), C.c_uint64(0)
        self._check(self.lib.mppi_transition_log_stats(self.h, C.byref(a), C.byref(b), C.byref(c)))
        return dict(held=int(a.value), overwritten=int(b.value), without_successor=int(c.value))

    def to_csv(self, filename, fmt=CSV_REFERENCE):
        """DataBase::toCSV; fmt = CSV_REFERENCE (the reference's bytes) or CSV_ROUNDTRIP (%.9g, no trailing commas)"""
        self._check(self.lib.mppi_to_csv_format(self.h, os.fsencode(filename), fmt))

    # ---- state -----------------------------------------------------------------------
    def get_action_sequence(self):
        U = np.zeros((self.tau, self.a), np.float32)
        self._check(self.lib.mppi_get_action_sequence(self.h, fp(U), U.size))
        return U

    def set_action_sequence(self, U):
        U = f32(U).ravel()
        self._check(self.lib.mppi_set_action_sequence(self.h, fp(U), U.size))

    def get_step_counter(self):
        v = C.c_uint64(0)
        self._check(self.lib.mppi_get_step_counter(self.h, C.byref(v)))
        return int(v.value)

    def set_step_counter(self, step):
        self._check(self.lib.mppi_set_step_counter(self.h, step))

    def debug_get(self, what):
        n = {DBG_COSTS: self.k_local, DBG_BETA: 1, DBG_ETA: 1, DBG_WEIGHTS: self.k_local,
             DBG_NOISE: self.k_local * self.tau * self.a, DBG_U_UPDATED: self.tau * self.a, DBG_AUX: 8}[what]
        out = np.zeros(n, np.float32)
        self._check(self.lib.mppi_debug_get(self.h, what, fp(out), n))
        if what == DBG_NOISE:
            return out.reshape(self.k_local, self.tau, self.a)
        if what == DBG_U_UPDATED:
            return out.reshape(self.tau, self.a)
        return out if n > 1 else out[0]

    # ---- graph helpers ---------------------------------------------------------------
    def model_next(self, x, v):
        """next state only (works for the learned model too) -> [k,s]"""
        x, v = f32(x, (-1, self.s)), f32(v, (-1, self.a))
        nx = np.zeros((v.shape[0], self.s), np.float32)
        self._check(self.lib.mppi_model_step(self.h, fp(x), x.shape[0], fp(v), v.shape[0], None, None, fp(nx)))
        return nx

    def model_step(self, x, v):
        """-> (free [kx,s], action [k,s], next [k,s])"""
        x, v = f32(x, (-1, self.s)), f32(v, (-1, self.a))
        kx, k = x.shape[0], v.shape[0]
        fr, ac, nx = np.zeros((kx, self.s), np.float32), np.zeros((k, self.s), np.float32), np.zeros((k, self.s), np.float32)
        self._check(self.lib.mppi_model_step(self.h, fp(x), kx, fp(v), k, fp(fr), fp(ac), fp(nx)))
        return fr, ac, nx

    def auv_pieces(self, x, u):
        """AUVModel's intermediates for k (state, action) pairs -> dict(rot [k,3,3], T [k,4,3], Cv, Dv, g [k,6], xdot [k,13])"""
        x, u = f32(x, (-1, 13)), f32(u, (-1, 6))
        k = x.shape[0]
        out = np.zeros((k, 124), np.float32)
        self._check(self.lib.mppi_auv_pieces(self.h, fp(x), fp(u), k, fp(out)))
        return dict(rot=out[:, :9].reshape(k, 3, 3), T=out[:, 9:21].reshape(k, 4, 3), Cv=out[:, 21:27], Dv=out[:, 27:33], g=out[:, 33:39],
                    xdot=out[:, 39:52], D=out[:, 52:88].reshape(k, 6, 6), C=out[:, 88:124].reshape(k, 6, 6))

    def ellipse3d_terms(self, x, in_plane_frame=False):
        """ElipseCost3D's (position, orientation, velocity) errors of k states -> [k,3]"""
        x = f32(x, (-1, 13))
        out = np.zeros((x.shape[0], 3), np.float32)
        self._check(self.lib.mppi_ellipse3d_terms(self.h, fp(x), x.shape[0], int(bool(in_plane_frame)), fp(out)))
        return out

    def state_cost(self, x):
        x = f32(x, (-1, self.s))
        out = np.zeros(x.shape[0], np.float32)
        self._check(self.lib.mppi_state_cost(self.h, fp(x), x.shape[0], fp(out)))
        return out

    def action_cost(self, u, eps):
        u, eps = f32(u, (self.a,)), f32(eps, (-1, self.a))
        out = np.zeros(eps.shape[0], np.float32)
        self._check(self.lib.mppi_action_cost(self.h, fp(u), fp(eps), eps.shape[0], fp(out)))
        return out

    def step_cost(self, x, u, eps):
        x, u, eps = f32(x, (-1, self.s)), f32(u, (self.a,)), f32(eps, (-1, self.a))
        out = np.zeros(x.shape[0], np.float32)
        self._check(self.lib.mppi_step_cost(self.h, fp(x), fp(u), fp(eps), x.shape[0], fp(out)))
        return out

    def rollout_cost(self, x, U, eps):
        x, U = f32(x, (self.s,)), f32(U, (self.tau, self.a))
        eps = f32(eps, (self.k_local, self.tau, self.a))
        out = np.zeros(self.k_local, np.float32)
        self._check(self.lib.mppi_rollout_cost(self.h, fp(x), fp(U), fp(eps), fp(out)))
        return out

    def update(self, cost, eps, U):
        """-> dict(beta, arg, exp, nabla, w, wn, Unew) (exp_arg is 'arg', weights 'w', weighted noise 'wn')"""
        K = self.k_local
        cost, U = f32(cost, (K,)), f32(U, (self.tau, self.a))
        eps = f32(eps, (K, self.tau, self.a))
        beta, nabla = np.zeros(1, np.float32), np.zeros(1, np.float32)
        arg, e, w = (np.zeros(K, np.float32) for _ in range(3))
        wn, Un = np.zeros((self.tau, self.a), np.float32), np.zeros((self.tau, self.a), np.float32)
        self._check(self.lib.mppi_update(self.h, fp(cost), fp(eps), fp(U), fp(beta), fp(arg), fp(e), fp(nabla),
                                         fp(w), fp(wn), fp(Un)))
        return dict(beta=beta[0], arg=arg, exp=e, nabla=nabla[0], w=w, wn=wn, Unew=Un)

    # ---- device-resident step (pointers are ints: tensor.data_ptr()) -----
    @staticmethod
    def _stream(stream):
        """The C-ABI's `void *stream` from what a Python caller holds. None = the handle's own stream (NULL in C). A torch stream object or its
        `.cuda_stream` integer names that stream — and torch's DEFAULT stream, whose handle is 0, is passed as hipStreamLegacy (1): a raw 0 that
        came out of torch must never select the handle's own non-blocking stream, which nothing of torch's is ordered against (the stale-record
        race of r04; ADVICE r04: mapped once, here, for every caller)."""
        if stream is None:
            return 0
        v = getattr(stream, "cuda_stream", stream)
        return int(v) or 1

    def next_device(self, x_ptr, u_ptr, stream=None):
        self._check(self.lib.mppi_next_device(self.h, x_ptr, u_ptr, self._stream(stream)))

    def shard_partial(self, x_ptr, record_ptr, stream=None):
        self._check(self.lib.mppi_shard_partial(self.h, x_ptr, record_ptr, self._stream(stream)))

    def shard_finish(self, records_ptr, n_records, u_ptr, stream=None):
        self._check(self.lib.mppi_shard_finish(self.h, records_ptr, n_records, u_ptr, self._stream(stream)))

    def shard_cost_range(self, x_ptr, range_ptr, stream=None):
        """normalize_cost on a sharded handle, first half: {-min, max} of this shard's sample costs -> range_ptr[2]"""
        self._check(self.lib.mppi_shard_cost_range(self.h, x_ptr, range_ptr, self._stream(stream)))

    def shard_partial_normalized(self, x_ptr, range_ptr, record_ptr, stream=None):
        """... second half: range_ptr[2] = the {-min, max} all ranks agreed on (all-reduce MAX) -> this shard's record"""
        self._check(self.lib.mppi_shard_partial_normalized(self.h, x_ptr, range_ptr, record_ptr, self._stream(stream)))

    # ---- options of the Python reference's update ---------------------------------------------
    def shard_step(self, x_ptr, u_ptr, coll=None, stream=None):
        """mppi_shard_step: the whole sharded step in ONE call; coll = a Collectives (or None on an unsharded handle: record -> finish)"""
        self._check(self.lib.mppi_shard_step(self.h, x_ptr, u_ptr, C.byref(coll) if coll is not None else None, self._stream(stream)))

    def set_action_limits(self, a_min=None, a_max=None):
        """clip_act (controller_base.py:500-504): clamp U' rows to [a_min, a_max]; None, None = off"""
        if a_min is None and a_max is None:
            self._check(self.lib.mppi_set_action_limits(self.h, None, None, 0))
            return
        lo, hi = f32(a_min, (self.a,)), f32(a_max, (self.a,))
        self._check(self.lib.mppi_set_action_limits(self.h, fp(lo), fp(hi), self.a))

    def set_sequence_filter(self, window, polyorder=3):
        """filterSeq (controller_base.py:277-291): Savitzky-Golay smoothing of the stored sequence; window 0 = off"""
        self._check(self.lib.mppi_set_sequence_filter(self.h, int(window), int(polyorder)))

    # ---- direct record exchange (mppi_shard_p2p_*, include/mppi_c.h) -------------------------
    def p2p_export(self, want_ipc=True):
        """-> (inbox device pointer, 64-byte hipIpcMemHandle_t as bytes or None)"""
        buf = C.create_string_buffer(64) if want_ipc else None
        ptr = C.c_void_p()
        self._check(self.lib.mppi_shard_p2p_export(self.h, buf, C.byref(ptr)))
        return ptr.value, (buf.raw if want_ipc else None)

    def p2p_open(self, ipc_handle):
        """map another process's inbox -> device pointer valid here"""
        ptr = C.c_void_p()
        self._check(self.lib.mppi_shard_p2p_open(self.h, C.create_string_buffer(ipc_handle, 64), C.byref(ptr)))
        return ptr.value

    def p2p_attach(self, inbox_ptrs, timeout_ms=2000):
        arr = (C.c_void_p * len(inbox_ptrs))(*inbox_ptrs)
        self._check(self.lib.mppi_shard_p2p_attach(self.h, arr, len(inbox_ptrs), timeout_ms))

    def p2p_probe(self, stream=None):
        ok = C.c_int(0)
        self._check(self.lib.mppi_shard_p2p_probe(self.h, self._stream(stream), C.byref(ok)))
        return bool(ok.value)

    def p2p_step(self, x_ptr, u_ptr, stream=None):
        self._check(self.lib.mppi_shard_p2p_step(self.h, x_ptr, u_ptr, self._stream(stream)))

    def p2p_timed_out(self):
        t = C.c_int(0)
        self._check(self.lib.mppi_shard_p2p_status(self.h, C.byref(t)))
        return bool(t.value)

    def synchronize(self):
        self._check(self.lib.mppi_synchronize(self.h))

    def rollout_kernel_name(self):
        buf = C.create_string_buffer(128)
        self._check(self.lib.mppi_rollout_kernel_name(self.h, buf, 128))
        return buf.value.decode()

    def profile_begin(self, max_steps):
        self._check(self.lib.mppi_profile_begin(self.h, max_steps))

    def profile_end(self):
        """-> (rollout kernel ms avg, finish kernel ms avg, steps recorded); HIP events on the launch stream."""
        r, f, n = C.c_float(0), C.c_float(0), C.c_int(0)
        self._check(self.lib.mppi_profile_end(self.h, C.byref(r), C.byref(f), C.byref(n)))
        return r.value, f.value, n.value


def get_new(U, nb):
    lib = load()
    U = f32(U)
    tau, a = U.shape
    out = np.zeros((nb, a), np.float32)
    st = lib.mppi_get_new(fp(U), tau, a, nb, fp(out))
    if st != OK:
        raise MppiError(st, "mppi_get_new")
    return out


def shift(U, init, nb):
    lib = load()
    U = f32(U)
    tau, a = U.shape
    init = f32(init, (-1, a))
    out = np.zeros((tau - nb + init.shape[0], a), np.float32)
    st = lib.mppi_shift(fp(U), tau, a, fp(init), init.shape[0], nb, fp(out))
    if st != OK:
        raise MppiError(st, "mppi_shift")
    return out


class Learner:
    """RAII wrapper of one mppi_learner: full-batch Adam on the MSE of a small Dense network, on the GPU (mppi_learner.hip).
    weights: dict(W=[...], b=[...]) (Keras [in x out] kernels); widths <= 32, 1-4 layers."""

    def __init__(self, weights, device=0):
        lib = self.lib = load()
        self.Ws = [f32(w) for w in weights["W"]]
        self.bs = [f32(b).ravel() for b in weights["b"]]
        self.widths = [self.Ws[0].shape[0]] + [w.shape[1] for w in self.Ws]
        wd = (C.c_int32 * len(self.widths))(*self.widths)
        self.h = _H()
        st = lib.mppi_learner_create(len(self.Ws), wd, self._ptrs(self.Ws), self._ptrs(self.bs), device, C.byref(self.h))
        if st != OK:
            self.h = _H()
            raise MppiError(st, (lib.mppi_learner_last_error(None) or b"").decode())
        self.n = 0

    @staticmethod
    def _ptrs(arrs):
        return (FP * len(arrs))(*[fp(a) for a in arrs])

    def _check(self, st):
        if st != OK:
            raise MppiError(st, (self.lib.mppi_learner_last_error(self.h) or b"").decode())

    def close(self):
        h, self.h = getattr(self, "h", None), None
        if h:
            self.lib.mppi_learner_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_data(self, X, Y):
        X, Y = f32(X, (-1, self.widths[0])), f32(Y, (-1, self.widths[-1]))
        assert X.shape[0] == Y.shape[0]
        self.n = X.shape[0]
        self._check(self.lib.mppi_learner_set_data(self.h, fp(X), fp(Y), self.n))

    def train(self, steps, lr, beta1=0.9, beta2=0.999, eps=1e-7):
        """`steps` Adam steps (Keras defaults) -> (loss of the first step's forward pass, loss of the last step's)"""
        first, last = np.zeros(1, np.float32), np.zeros(1, np.float32)
        self._check(self.lib.mppi_learner_train(self.h, int(steps), lr, beta1, beta2, eps, fp(first), fp(last)))
        return float(first[0]), float(last[0])

    def evaluate(self, grads=False, pred=False):
        """loss of the current weights on the training set [, gradients as dict(W=[...], b=[...])] [, predictions]"""
        loss = np.zeros(1, np.float32)
        g = np.zeros((len(self.Ws), 33, 32), np.float32) if grads else None
        p = np.zeros((self.n, self.widths[-1]), np.float32) if pred else None
        self._check(self.lib.mppi_learner_evaluate(self.h, fp(loss), fp(g), fp(p)))
        out = [float(loss[0])]
        if grads:
            out.append(dict(W=[g[l, :self.widths[l], :self.widths[l + 1]].copy() for l in range(len(self.Ws))],
                            b=[g[l, 32, :self.widths[l + 1]].copy() for l in range(len(self.Ws))]))
        if pred:
            out.append(p)
        return out[0] if len(out) == 1 else tuple(out)

    def get_weights(self):
        Ws = [np.zeros_like(w) for w in self.Ws]
        bs = [np.zeros_like(b) for b in self.bs]
        self._check(self.lib.mppi_learner_get_weights(self.h, self._ptrs(Ws), self._ptrs(bs)))
        return dict(W=Ws, b=bs)

    def set_weights(self, weights):
        Ws, bs = [f32(w) for w in weights["W"]], [f32(b).ravel() for b in weights["b"]]
        self._check(self.lib.mppi_learner_set_weights(self.h, self._ptrs(Ws), self._ptrs(bs)))

    def reset_optimizer(self):
        self._check(self.lib.mppi_learner_reset_optimizer(self.h))

    # persistence (mppi_learner_save / _load / _peek): weights, both Adam moments, the step count and — the model's, handed through —
    # the normalisation, in the flat format include/mppi_c.h documents
    def save(self, filename, norm=None):
        """norm: dict(xmean, xstd, ymean, ystd) or None"""
        held = [None] * 4
        if norm is not None:
            n_in, n_out = self.widths[0], self.widths[-1]
            held = [np.ascontiguousarray(np.asarray(norm[k], np.float64).reshape(n)) for k, n in
                    (("xmean", n_in), ("xstd", n_in), ("ymean", n_out), ("ystd", n_out))]
        self._check(self.lib.mppi_learner_save(self.h, os.fsencode(filename), *[a.ctypes.data_as(DP) if a is not None else None for a in held]))

    def load(self, filename):
        """-> the file's normalisation dict(xmean, xstd, ymean, ystd), or None when it holds none"""
        n_in, n_out = self.widths[0], self.widths[-1]
        out = [np.zeros(n_in), np.zeros(n_in), np.zeros(n_out), np.zeros(n_out)]
        has = C.c_int(0)
        self._check(self.lib.mppi_learner_load(self.h, os.fsencode(filename), *[a.ctypes.data_as(DP) for a in out], C.byref(has)))
        return dict(zip(("xmean", "xstd", "ymean", "ystd"), out)) if has.value else None

    @classmethod
    def from_file(cls, filename, device=0):
        """a learner of the file's network with the file's weights, Adam moments and step count -> (learner, normalisation or None)"""
        lib = load()
        n, w = C.c_int(0), (C.c_int32 * 5)()
        st = lib.mppi_learner_peek(os.fsencode(filename), C.byref(n), w)
        if st != OK:
            raise MppiError(st, (lib.mppi_learner_last_error(None) or b"").decode())
        widths = [int(w[i]) for i in range(n.value + 1)]
        zero = dict(W=[np.zeros((widths[i], widths[i + 1]), np.float32) for i in range(n.value)],
                    b=[np.zeros(widths[i + 1], np.float32) for i in range(n.value)])
        lrn = cls(zero, device=device)
        return lrn, lrn.load(filename)

    def step_count(self):
        v = C.c_int(0)
        self._check(self.lib.mppi_learner_get_step(self.h, C.byref(v)))
        return int(v.value)
