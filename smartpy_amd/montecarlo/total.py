"""Re-simulate a whole previous sampling on another period (counterpart of smartpy/montecarlo/total.py)."""
from .montecarlo import MonteCarlo


class Total(MonteCarlo):
    def __init__(self, catchment, root_f, in_format, out_format,
                 parallel='seq', save_sim=False, settings_filename=None, decompression_csv=False):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='total', settings_filename=settings_filename)
        self.sampling_run_file = \
            ''.join([self.model.out_f, catchment, '.SMART.lhs.nc']) if self.out_format == 'netcdf' else \
            ''.join([self.model.out_f, catchment, '.SMART.lhs'])
        self.sampled_params, self.sampled_obj_fns = self._get_sampled_sets_from_file(
            self.sampling_run_file, self.param_names, self.obj_fn_names, decompression_csv)
        self._set_sample(self.sampled_params)
