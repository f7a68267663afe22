"""cart-pole-mpc_amd: MI355X-native batched cart-pole MPC hot path.

The directory name is not a Python identifier; import it with
    import importlib; cpmpc = importlib.import_module("cart-pole-mpc_amd")
from the repository root.

Contents:
    csrc/      HIP kernels (gfx950) + the C-ABI implementation (include/cpmpc.h)
    lib/       libcpmpc.so (built by build.py / __graft_entry__.build(); git-ignored)
    capi.py    ctypes binding of the C-ABI
    batch.py   batched host API on torch tensors (BatchOptimization, BatchSimulator, ClosedLoop)
    host/      C++ facade with the reference's class API + the pypendulum binding
"""
from . import capi  # noqa: F401
from .capi import CpmpcError, Params, SolverOpts, default_params, default_solver_opts  # noqa: F401


def __getattr__(name):
    # torch-dependent pieces are imported lazily so the C-ABI can be inspected without torch
    if name in ("BatchOptimization", "BatchSimulator", "BatchOutputs", "ClosedLoop", "dynamics_batch", "rk4_batch"):
        from . import batch
        return getattr(batch, name)
    raise AttributeError(name)


def pypendulum():
    """The `pypendulum` extension module (names of wrapper/wrapper.cc:40-98) built in lib/ by build.build_host()."""
    import importlib
    import os
    import sys
    lib_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "lib")
    if lib_dir not in sys.path:
        sys.path.insert(0, lib_dir)
    return importlib.import_module("pypendulum")
