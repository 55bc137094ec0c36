"""Best-n conditioning run (counterpart of smartpy/montecarlo/best.py)."""
from .montecarlo import MonteCarlo
from .selection import condition_mask, check_shapes


class Best(MonteCarlo):
    def __init__(self, catchment, root_f, in_format, out_format,
                 target, nb_best, constraining=None,
                 parallel='seq', save_sim=False, settings_filename=None,
                 decompression_csv=False):
        MonteCarlo.__init__(self, catchment, root_f, in_format, out_format,
                            parallel=parallel, save_sim=save_sim, func='{}best'.format(nb_best),
                            settings_filename=settings_filename)
        self.sampling_run_file = \
            ''.join([self.model.out_f, catchment, '.SMART.lhs.nc']) if self.out_format == 'netcdf' else \
            ''.join([self.model.out_f, catchment, '.SMART.lhs'])
        self.sampled_params, self.sampled_obj_fns = self._get_sampled_sets_from_file(
            self.sampling_run_file, self.param_names, self.obj_fn_names, decompression_csv)
        try:
            self.target_fn_index = [self.obj_fn_names.index(target)]
        except ValueError:
            raise Exception("The objective function {} for conditioning in Best is not recognised."
                            "Please check for typos and case sensitive issues.".format(target))
        if constraining:
            try:
                self.constraints_indices = [self.obj_fn_names.index(fn) for fn in constraining]
            except ValueError:
                raise Exception("One of the names of constraints in Best is not recognised."
                                "Please check for typos and case sensitive issues.")
            self.constraints_types = [constraining[fn][0] for fn in constraining]
            self.constraints_values = [constraining[fn][1] for fn in constraining]
        else:
            self.constraints_indices, self.constraints_types, self.constraints_values = [], [], []
        self.best_params = self._get_best_sets(self.sampled_params,
                                               self.sampled_obj_fns[:, self.constraints_indices],
                                               self.constraints_values, self.constraints_types,
                                               self.sampled_obj_fns[:, self.target_fn_index], nb_best)
        self._set_sample(self.best_params)

    @staticmethod
    def _get_best_sets(params, constraints_fns, constraints_val, constraints_typ, sort_fn, nb_best):
        """best.py:221-287: apply the constraints, sort ascending on the target and keep the LAST nb_best rows --
        i.e. the largest values whatever the target, as the reference does."""
        check_shapes(params, constraints_fns, constraints_val, constraints_typ, 'constraint')
        if sort_fn.shape[0] != params.shape[0]:
            raise Exception('The matrices containing objective functions and parameters have different sample sizes.')
        if nb_best > params.shape[0]:
            raise Exception('The number of best models requested is higher than the sample size.')
        constrained = condition_mask(constraints_fns, constraints_val, constraints_typ)
        sort_fn_constrained = sort_fn[constrained, :]
        param_constrained = params[constrained, :]
        if nb_best > param_constrained.shape[0]:
            raise Exception('The number of best models requested is higher than the restrained sample size.')
        return param_constrained[sort_fn_constrained[:, 0].argsort()][-nb_best:]
