"""rustsolver_amd -- MI355X-native CFR regret / strategy-update engine behind RustSolver's solver API.

The product is the HIP shared library (csrc/, built by `python -m rustsolver_amd.build`) and its
C ABI (include/rustsolver_amd.h).  This package is the ctypes host mirror used by tests and bench.py.
Importing it fails loudly if the library has not been built; there is no CPU fallback.
"""
import sys as _sys

# `python -m rustsolver_amd.build` must be able to run while the .so is missing or stale
_BUILDING = "rustsolver_amd.build" in getattr(_sys, "orig_argv", [])

from . import _lib  # noqa: E402
from ._lib import (ACT_BET, ACT_CALL, ACT_CHECK, ACT_FOLD, ACT_RAISE, CHANCE_ENUM, CHANCE_PASS, F16, F32, I32,
                   LEAF_SIGN, LEAF_UNCONTESTED, LEAF_UTIL, OPP_FULL, OPP_SAMPLE, NODE_ACTION, NODE_PRIVATE_CHANCE, NODE_PUBLIC_CHANCE,
                   NODE_TERMINAL, TERM_ALLIN, TERM_SHOWDOWN, TERM_UNCONTESTED, UPD_CLAMP_I64, UPD_PRUNE, UPD_RMPLUS,
                   UPD_WRAP_I32, RsError, FORM_DEFAULT, FORM_ON, FORM_OFF, FAN_DEFAULT, FAN_NONE, FAN_EXPAND, SHADOW_RULE, SHADOW_ALL)

if not _BUILDING:
    _lib.load()  # ImportError if librustsolver_amd.so is missing

from .solver import (DealTrainer, DeviceBuffer, GameTree, Infoset, InfosetTable, MCCFRTrainer, Options,  # noqa: E402
                     build_game_tree, create_infosets, deal_buffer, deal_pitch, default_flop, device_count, discount_factor,
                     jit_check_tree, jit_check_tree_deals, showdown_sign,
                     three_street_options, tree_from_nodes)
from . import synth  # noqa: E402
from . import abstraction  # noqa: E402

__all__ = ["Options", "default_flop", "three_street_options", "build_game_tree", "tree_from_nodes", "create_infosets",
           "InfosetTable", "Infoset", "GameTree", "MCCFRTrainer", "DealTrainer", "DeviceBuffer", "device_count", "discount_factor",
           "synth", "RsError"]
