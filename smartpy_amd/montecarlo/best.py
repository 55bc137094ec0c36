"""Best-n conditioning run (counterpart of smartpy/montecarlo/best.py)."""
from .montecarlo import MonteCarlo
from .selection import condition_mask, check_shapes, best_rows, SecondStage


class Best(SecondStage, MonteCarlo):
    """Constructor of the reference (best.py:34-38) plus `sampling=` (see GLUE): constraints and the top-n selection
    run on the GPU over the objective functions of a finished sampling run of this process."""

    def __init__(self, catchment, root_f, in_format, out_format,
                 target, nb_best, constraining=None,
                 parallel='seq', save_sim=False, settings_filename=None,
                 decompression_csv=False, sampling=None):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='{}best'.format(nb_best),
                            settings_filename=settings_filename)
        self._load_sampling(catchment, decompression_csv, sampling)
        self.target_fn_index = self._columns_of(
            [target], "The objective function {} for conditioning in Best is not recognised."
                      "Please check for typos and case sensitive issues.".format(target))
        constraining = constraining or {}
        self.constraints_indices = self._columns_of(
            constraining, "One of the names of constraints in Best is not recognised."
                          "Please check for typos and case sensitive issues.")
        self.constraints_types = [constraining[fn][0] for fn in constraining]
        self.constraints_values = [constraining[fn][1] for fn in constraining]
        if sampling is not None:
            fns = self._device_obj_fns
            if nb_best > fns.shape[0]:
                raise Exception('The number of best models requested is higher than the sample size.')
            allowed = condition_mask(fns[:, self.constraints_indices], self.constraints_values, self.constraints_types)
            if nb_best > int(allowed.sum()):
                raise Exception('The number of best models requested is higher than the restrained sample size.')
            self.best_params = self._rows_as_stored(best_rows(fns[:, self.target_fn_index[0]], allowed, nb_best))
        else:
            self.best_params = self._get_best_sets(self.sampled_params,
                                                   self.sampled_obj_fns[:, self.constraints_indices],
                                                   self.constraints_values, self.constraints_types,
                                                   self.sampled_obj_fns[:, self.target_fn_index], nb_best)
        self._set_sample(self.best_params)

    @staticmethod
    def _get_best_sets(params, constraints_fns, constraints_val, constraints_typ, sort_fn, nb_best):
        """best.py:221-287: apply the constraints, then keep the nb_best rows with the LARGEST target value, in
        ascending order of it -- whatever the target (RMSE and PBias included), as the reference does."""
        check_shapes(params, constraints_fns, constraints_val, constraints_typ, 'constraint')
        n = params.shape[0]
        if sort_fn.shape[0] != n:
            raise Exception('The matrices containing objective functions and parameters have different sample sizes.')
        if nb_best > n:
            raise Exception('The number of best models requested is higher than the sample size.')
        allowed = condition_mask(constraints_fns, constraints_val, constraints_typ)
        if nb_best > int(allowed.sum()):
            raise Exception('The number of best models requested is higher than the restrained sample size.')
        return params[best_rows(sort_fn[:, 0], allowed, nb_best)]
