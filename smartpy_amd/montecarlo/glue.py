"""GLUE conditioning run (counterpart of smartpy/montecarlo/glue.py): keep the behavioural sets of a previous
LHS sampling, re-simulate them (typically on another period)."""
from .montecarlo import MonteCarlo
from .selection import condition_mask, check_shapes


class GLUE(MonteCarlo):
    def __init__(self, catchment, root_f, in_format, out_format,
                 conditioning,
                 parallel='seq', save_sim=False, settings_filename=None,
                 decompression_csv=False):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='glue', settings_filename=settings_filename)
        self.sampling_run_file = \
            ''.join([self.model.out_f, catchment, '.SMART.lhs.nc']) if self.out_format == 'netcdf' else \
            ''.join([self.model.out_f, catchment, '.SMART.lhs'])
        self.sampled_params, self.sampled_obj_fns = self._get_sampled_sets_from_file(
            self.sampling_run_file, self.param_names, self.obj_fn_names, decompression_csv)
        try:
            self.objective_fn_indices = [self.obj_fn_names.index(fn) for fn in conditioning]
        except ValueError:
            raise Exception("One of the names of objective functions for conditioning in GLUE is not recognised."
                            "Please check for typos and case sensitive issues.")
        self.conditions_types = [conditioning[fn][0] for fn in conditioning]
        self.conditions_values = [conditioning[fn][1] for fn in conditioning]
        self.behavioural_params = self._get_behavioural_sets(self.sampled_params,
                                                             self.sampled_obj_fns[:, self.objective_fn_indices],
                                                             self.conditions_values, self.conditions_types)
        self._set_sample(self.behavioural_params)

    @staticmethod
    def _get_behavioural_sets(params, obj_fns, conditions_val, conditions_typ):
        """glue.py:222-289 -> the rows of params (float32, possibly none) that meet every condition."""
        check_shapes(params, obj_fns, conditions_val, conditions_typ, 'objective')
        return params[condition_mask(obj_fns, conditions_val, conditions_typ), :]
