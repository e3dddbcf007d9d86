"""mppi-tf_amd — the MI355X-native MPPI control step behind the reference's plugin surface.

Scope: ONE hot path of NicolayP/mppi-tf: perturb K control sequences, roll the
model H steps, cost every step, soft-min weight, reduce to the new nominal sequence — as
hand-written HIP for gfx950 in csrc/, reached through the C-ABI of include/mppi_c.h.

  csrc/            HIP kernels + the C-ABI implementation  -> libmppi_hip.so (build.py)
  _lib.py          ctypes binding (Handle)
  controller.py    the reference's ControllerBase / StaticCost / PointMassModel interface
  auv.py           the 13-state family: AUVModel, NNAUVModel, NNAUVModelSpeed, StaticQuatCost, ElipseCost3D
  learner.py       LearnerBase: replay buffer + full-batch Adam on the GPU (csrc/mppi_learner.hip) -> the learned model's weights
  distributed.py   K-sharding across GPUs (one process per GPU, one all-gather per step)
"""
from . import build as _build  # noqa: F401
from ._lib import (ACTION_COST_CPP, ACTION_COST_PY, CSV_REFERENCE, CSV_ROUNDTRIP, DBG_AUX, DBG_BETA, DBG_COSTS, DBG_ETA, DBG_NOISE,
                   DBG_U_UPDATED, DBG_WEIGHTS, Handle, MppiError, load)
from .controller import ControllerBase, ControllerBaseCpp, CostBase, ElipseCost, PointMassModel, StaticCost
from .auv import AUVModel, ElipseCost3D, NNAUVModel, NNAUVModelSpeed, StaticQuatCost
from .learner import LearnerBase
from ._lib import Learner

__all__ = ["Handle", "MppiError", "load", "ControllerBase", "ControllerBaseCpp", "CostBase", "PointMassModel",
           "StaticCost", "ElipseCost", "AUVModel", "NNAUVModel", "NNAUVModelSpeed", "StaticQuatCost", "ElipseCost3D", "LearnerBase", "Learner", "ACTION_COST_CPP", "ACTION_COST_PY", "DBG_COSTS", "DBG_BETA", "DBG_ETA", "DBG_WEIGHTS",
           "DBG_NOISE", "DBG_U_UPDATED", "DBG_AUX", "CSV_REFERENCE", "CSV_ROUNDTRIP"]
