"""ctypes mirrors of the structs in include/sfmba.h (field order must match the header)."""
import ctypes as C

CONVERGENCE, NO_CONVERGENCE, FAILURE = 0, 1, 2
TERMINATION_NAMES = {0: "CONVERGENCE", 1: "NO_CONVERGENCE", 2: "FAILURE"}
LINEAR_CHOLESKY, LINEAR_PCG, LINEAR_AUTO = 0, 1, 2
PRECISION_F64, PRECISION_F32J = 0, 1
CREATE_DETERMINISTIC = 1
CREATE_ROW_SHARDED = 2      # every rank holds the whole problem (sharded.HipRowShardBackend)
CREATE_NO_PAIR_LIST = 4     # matrix-free solve: no list of observation pairs, the reduced matrix is never formed


class SfmbaOptions(C.Structure):
    _fields_ = [
        ("max_iters", C.c_int),
        ("max_seconds", C.c_double),
        ("function_tolerance", C.c_double),
        ("gradient_tolerance", C.c_double),
        ("parameter_tolerance", C.c_double),
        ("initial_radius", C.c_double),
        ("max_radius", C.c_double),
        ("min_radius", C.c_double),
        ("min_relative_decrease", C.c_double),
        ("min_lm_diagonal", C.c_double),
        ("max_lm_diagonal", C.c_double),
        ("jacobi_scaling", C.c_int),
        ("max_consecutive_invalid_steps", C.c_int),
        ("linear_solver", C.c_int),
        ("precision", C.c_int),
        ("pcg_tolerance", C.c_double),
        ("pcg_max_iters", C.c_int),
        ("verbose", C.c_int),
        ("pcg_anchored", C.c_int),
        # ABI v4: behaviour switches (0 = library default, 1 = on, -1 = off; the SFMBA_* environment variable overrides)
        ("pcg_coarse_space", C.c_int),
        ("pcg_symmetric", C.c_int),
        ("pcg_f32_matrix", C.c_int),
        ("early_linearise", C.c_int),
        ("shard_two_phase", C.c_int),
        ("shard_f32_exchange", C.c_int),
        ("shard_distributed_cg", C.c_int),
    ]

    @classmethod
    def defaults(cls, **overrides):
        """Reference options: BA.cpp:171-177 + Ceres defaults (same values as sfmba_options_default)."""
        o = cls(max_iters=500, max_seconds=10.0, function_tolerance=1e-6, gradient_tolerance=1e-10,
                parameter_tolerance=1e-8, initial_radius=1e4, max_radius=1e16, min_radius=1e-32,
                min_relative_decrease=1e-3, min_lm_diagonal=1e-6, max_lm_diagonal=1e32,
                jacobi_scaling=1, max_consecutive_invalid_steps=5, linear_solver=LINEAR_AUTO,
                precision=PRECISION_F64, pcg_tolerance=1e-8, pcg_max_iters=0, verbose=0, pcg_anchored=1)
        for k, v in overrides.items():
            if not hasattr(o, k):
                raise AttributeError(k)
            setattr(o, k, v)
        return o


class SfmbaSummary(C.Structure):
    _fields_ = [
        ("termination", C.c_int),
        ("iterations", C.c_int),
        ("successful_steps", C.c_int),
        ("unsuccessful_steps", C.c_int),
        ("residual_evals", C.c_int),
        ("jacobian_evals", C.c_int),
        ("linear_iters", C.c_int),
        ("initial_cost", C.c_double),
        ("final_cost", C.c_double),
        ("seconds", C.c_double),
        ("setup_seconds", C.c_double),
        ("message", C.c_char * 128),
        ("cholesky_fallbacks", C.c_int),
    ]

    def as_dict(self):
        d = {k: getattr(self, k) for k, _ in self._fields_}
        d["message"] = self.message.decode(errors="replace")
        d["termination_name"] = TERMINATION_NAMES.get(self.termination, "?")
        return d


class SfmbaIteration(C.Structure):
    _fields_ = [
        ("iteration", C.c_int),
        ("step_is_valid", C.c_int),
        ("step_is_successful", C.c_int),
        ("linear_iters", C.c_int),
        ("cost", C.c_double),
        ("cost_change", C.c_double),
        ("gradient_max_norm", C.c_double),
        ("step_norm", C.c_double),
        ("relative_decrease", C.c_double),
        ("trust_region_radius", C.c_double),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}
