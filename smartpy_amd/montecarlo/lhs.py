"""Latin hypercube sampling run (counterpart of smartpy/montecarlo/lhs.py)."""
from .montecarlo import MonteCarlo
from ..sampling import latin_hypercube


class LHS(MonteCarlo):
    """Sample the parameter space with a Latin hypercube (McKay et al.) and simulate every set.

    Same constructor as the reference (lhs.py:36-38); `run()` evaluates the whole sample on the GPU(s)."""

    def __init__(self, catchment, root_f, in_format, out_format,
                 sample_size,
                 parallel='seq', save_sim=False, settings_filename=None):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='lhs', settings_filename=settings_filename)
        self.lhs_params = self._get_params_from_lh(sample_size)
        self._set_sample(self.lhs_params)

    def _get_params_from_lh(self, sample_size):
        """lhs.py:133-167: bounds from model.parameters.ranges, NumPy's global legacy random stream."""
        return latin_hypercube(sample_size, self.model.parameters.ranges, self.param_names)
