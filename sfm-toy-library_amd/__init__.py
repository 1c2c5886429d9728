"""MI355X-native bundle-adjustment back end behind SfMBundleAdjustmentUtils::adjustBundle().

Layout
  csrc/   hand-written HIP kernels (gfx950) + the C ABI of include/sfmba.h  -> libsfmba_hip.so
  host/   C++ mirror of the reference interface (sfmtoylib::SfMBundleAdjustmentUtils::adjustBundle)
  capi.py ctypes binding of the C ABI (tests, bench.py and the multi-GPU driver go through it)
  synthetic.py / problem_io.py   seeded synthetic BA problems and the problem dump format
  sharded.py   one-process-per-GPU LM driver (torch.distributed / RCCL) for point-sharded and row-sharded problems

Nothing in this package imports oracle/: the product path fails loudly when the HIP library
or a GPU is missing, it never falls back to a CPU implementation.
"""
from .structs import (SfmbaOptions, SfmbaSummary, SfmbaIteration, TERMINATION_NAMES,
                      CONVERGENCE, NO_CONVERGENCE, FAILURE, LINEAR_CHOLESKY, LINEAR_PCG, LINEAR_AUTO,
                      PRECISION_F64, PRECISION_F32J, CREATE_DETERMINISTIC, CREATE_ROW_SHARDED, CREATE_NO_PAIR_LIST)
from .synthetic import make_problem, BAProblem, CONFIGS
from .problem_io import save_problem, load_problem, save_bal, load_bal

__all__ = [
    "SfmbaOptions", "SfmbaSummary", "SfmbaIteration", "TERMINATION_NAMES",
    "CONVERGENCE", "NO_CONVERGENCE", "FAILURE", "LINEAR_CHOLESKY", "LINEAR_PCG", "LINEAR_AUTO",
    "PRECISION_F64", "PRECISION_F32J", "CREATE_DETERMINISTIC", "CREATE_ROW_SHARDED", "CREATE_NO_PAIR_LIST",
    "make_problem", "BAProblem", "CONFIGS", "save_problem", "load_problem", "save_bal", "load_bal",
]
