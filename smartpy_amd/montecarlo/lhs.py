"""Latin hypercube sampling run (counterpart of smartpy/montecarlo/lhs.py)."""
from .montecarlo import MonteCarlo
from ..sampling import latin_hypercube, latin_hypercube_device


class LHS(MonteCarlo):
    """Sample the parameter space with a Latin hypercube (McKay et al.) and simulate every set.

    Same constructor as the reference (lhs.py:36-38); `run()` evaluates the whole sample on the GPU(s).

    device_sampling=True (not in the reference) draws the sample on the GPU instead (sampling.latin_hypercube_device:
    same plan, torch's generator, `seed` for reproducibility) and leaves it there for the launch -- for N >= 1e6 the
    host sampler's ~1.5 s is comparable to the whole ensemble run.  The default is the host sampler, which
    reproduces the reference's legacy-NumPy random stream bit for bit."""

    def __init__(self, catchment, root_f, in_format, out_format,
                 sample_size,
                 parallel='seq', save_sim=False, settings_filename=None,
                 device_sampling=False, seed=None):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='lhs', settings_filename=settings_filename)
        self.device_sampling, self.seed = device_sampling, seed
        drawn = self._get_params_from_lh(sample_size)
        self._set_sample(drawn)
        self.lhs_params = self._sample          # the [N, 10] float64 host matrix, as in the reference (lhs.py:114)

    def _get_params_from_lh(self, sample_size):
        """lhs.py:133-167: bounds from model.parameters.ranges; NumPy's global legacy random stream on the host, or
        torch's generator on the device."""
        ranges = self.model.parameters.ranges
        if self.device_sampling:
            return latin_hypercube_device(sample_size, ranges, self.param_names, seed=self.seed)
        return latin_hypercube(sample_size, ranges, self.param_names, seed=self.seed)
