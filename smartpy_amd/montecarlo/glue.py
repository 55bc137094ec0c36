"""GLUE conditioning run (counterpart of smartpy/montecarlo/glue.py): keep the behavioural sets of a previous
LHS sampling, re-simulate them (typically on another period)."""
from .montecarlo import MonteCarlo
from .selection import condition_mask, check_shapes, SecondStage


class GLUE(SecondStage, MonteCarlo):
    """Constructor of the reference (glue.py:34-37) plus `sampling=`: a finished sampling run of this process (an LHS
    object after run()) whose objective functions are still on the GPU -- the behavioural mask is then evaluated
    there and only the selected parameter rows travel, instead of re-reading and parsing the database file."""

    def __init__(self, catchment, root_f, in_format, out_format,
                 conditioning,
                 parallel='seq', save_sim=False, settings_filename=None,
                 decompression_csv=False, sampling=None):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='glue', settings_filename=settings_filename)
        self.objective_fn_indices = self._columns_of(
            conditioning, "One of the names of objective functions for conditioning in GLUE is not recognised."
                          "Please check for typos and case sensitive issues.")
        self.conditions_types = [conditioning[fn][0] for fn in conditioning]
        self.conditions_values = [conditioning[fn][1] for fn in conditioning]
        self._load_sampling(catchment, decompression_csv, sampling)
        if sampling is not None:
            keep = condition_mask(self._device_obj_fns[:, self.objective_fn_indices], self.conditions_values,
                                  self.conditions_types)
            self.behavioural_params = self._rows_as_stored(keep)
        else:
            self.behavioural_params = self._get_behavioural_sets(
                self.sampled_params, self.sampled_obj_fns[:, self.objective_fn_indices],
                self.conditions_values, self.conditions_types)
        self._set_sample(self.behavioural_params)

    @staticmethod
    def _get_behavioural_sets(params, obj_fns, conditions_val, conditions_typ):
        """glue.py:222-289 -> the rows of params (float32, possibly none) that meet every condition."""
        check_shapes(params, obj_fns, conditions_val, conditions_typ, 'objective')
        return params[condition_mask(obj_fns, conditions_val, conditions_typ), :]
