"""Re-simulate a whole previous sampling on another period (counterpart of smartpy/montecarlo/total.py)."""
from .montecarlo import MonteCarlo
from .selection import SecondStage


class Total(SecondStage, MonteCarlo):
    def __init__(self, catchment, root_f, in_format, out_format,
                 parallel='seq', save_sim=False, settings_filename=None, decompression_csv=False, sampling=None):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='total', settings_filename=settings_filename)
        self._load_sampling(catchment, decompression_csv, sampling)
        self._set_sample(self.sampled_params)
