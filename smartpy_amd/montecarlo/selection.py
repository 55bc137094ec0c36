"""Conditions on objective functions shared by GLUE and Best (glue.py:222-289, best.py:221-287).

`condition_mask` and `best_rows` accept numpy arrays (what the file-based second stages of the reference hand them)
as well as torch tensors on any device: on an `[N, 8]` objective matrix that is still on the GPU the behavioural
mask, its count and the top-k rows are computed there and only the selected rows travel.
"""
import ctypes
import os
import tempfile

import numpy as np


def _is_torch(x):
    return type(x).__module__.startswith('torch')


def condition_mask(obj_fns, conditions_val, conditions_typ):
    """Boolean mask of the rows meeting every condition.  obj_fns [N, k]; kinds 'equal', 'min', 'max', 'inside',
    'outside' with the reference's semantics -- including 'outside', written there as
    (value <= lower) & (value >= upper), which no value satisfies."""
    if _is_torch(obj_fns):
        import torch
        mask = torch.ones((obj_fns.shape[0],), dtype=torch.bool, device=obj_fns.device)
        columns = obj_fns.t()
    else:
        mask = np.ones((obj_fns.shape[0],), dtype=bool)
        columns = obj_fns.T
    for obj_fn, values, kind in zip(columns, conditions_val, conditions_typ):
        if kind in ('equal', 'min', 'max'):
            if len(values) != 1:
                raise Exception("The tuple for \"{}\" condition does not contain one and only one element.".format(kind))
            if kind == 'equal':
                selection = obj_fn == values[0]
            elif kind == 'min':
                selection = obj_fn >= values[0]
            else:
                selection = obj_fn <= values[0]
        elif kind in ('inside', 'outside'):
            if len(values) != 2:
                raise Exception("The tuple for \"{}\" condition does not contain two and only two "
                                "elements.".format(kind))
            if not values[1] > values[0]:
                raise Exception("The two elements of the tuple for \"{}\" are inconsistent.".format(kind))
            if kind == 'inside':
                selection = (obj_fn >= values[0]) & (obj_fn <= values[1])
            else:
                selection = (obj_fn <= values[0]) & (obj_fn >= values[1])
        else:
            raise Exception("The type of threshold \"{}\" is not in the database.".format(kind))
        mask &= selection
    return mask


def check_shapes(params, fns, values, kinds, what):
    if fns.ndim != 2:
        raise Exception('The matrix containing the {} functions is not 2D.'.format(what))
    if params.ndim != 2:
        raise Exception('The matrix containing the parameters is not 2D.')
    if fns.shape[0] != params.shape[0]:
        raise Exception('The matrices containing {} functions and parameters have different sample '
                        'sizes.'.format(what))
    if not ((fns.shape[1] == len(values)) and (fns.shape[1] == len(kinds))):
        raise Exception('The {} function matrix and the conditions matrices '
                        'do not have compatible dimensions.'.format(what))


def best_rows(sort_fn, constrained, nb_best):
    """Indices (into the unconstrained sample) of the nb_best LARGEST values of sort_fn among the rows where
    `constrained` is true, in ascending order of the value -- best.py:287 `argsort()[-nb_best:]`.

    Equal keys: the reference calls numpy's default argsort, an unstable sort whose order among equal keys is a detail
    of the numpy build (KAT-12: the ten best by a GW flag of zeros and ones come back as rows 16, 15, 14, 12, ... --
    neither ascending nor descending row order).  On the host this function makes the very same call.  On a device
    tensor the top-k is a stable sort there -- unless keys tie inside the selection or across its edge, the one case
    where the sort's inner order shows: then the constrained keys (4 bytes per row) are sorted on the host by the same
    numpy call, so that a second stage built from a finished run picks the rows the one built from its database
    file picks."""
    if _is_torch(sort_fn):
        import torch
        idx = torch.nonzero(constrained, as_tuple=False)[:, 0]
        keys = sort_fn[idx]
        order = torch.argsort(keys, stable=True)
        picked = order[-nb_best:]       # (nb_best = 0: every constrained row, like best.py:287's `[-0:]` on the host)
        edge = keys[order[-nb_best - 1:]] if 0 < nb_best < order.numel() else keys[picked]
        tied = ((edge[1:] == edge[:-1]) | (torch.isnan(edge[1:]) & torch.isnan(edge[:-1]))).any() \
            if edge.numel() > 1 else False
        if bool(tied):
            host = np.argsort(keys.cpu().numpy())[-nb_best:]
            picked = torch.from_numpy(host).to(idx.device)
        return idx[picked]
    idx = np.nonzero(constrained)[0]
    return idx[np.argsort(sort_fn[idx])][-nb_best:]


def as_stored(matrix):
    """What a float64 matrix becomes on its way through a sampling database: cast to float32, printed '%.6e', parsed
    back to float32 (montecarlo.py:225-231, :262) -- 7 significant digits, not a float32 round trip.  Done with the
    library's own writer and reader on a scratch file, so a selection made on the device hands the second stage
    exactly the rows the file-based path would have read."""
    from .. import _lib
    table = np.ascontiguousarray(matrix, dtype=np.float32)
    if table.size == 0:
        return table
    fd, path = tempfile.mkstemp(prefix='smart_rows_')
    os.close(fd)
    try:
        L = _lib.lib()
        _lib.check(L.smart_db_append_rows(path.encode('utf8'), table.ctypes.data_as(ctypes.POINTER(ctypes.c_float)),
                                          table.shape[0], table.shape[1], 0))
        with open(path, 'rb') as f:
            text = f.read()
    finally:
        os.remove(path)
    out = np.empty_like(table)
    cols = np.arange(table.shape[1], dtype=np.int32)
    n = L.smart_db_parse_rows(text, len(text), table.shape[1], cols.ctypes.data_as(ctypes.POINTER(ctypes.c_int32)),
                              table.shape[1], out.ctypes.data_as(ctypes.POINTER(ctypes.c_float)), table.shape[0], 0)
    if n != table.shape[0]:
        raise Exception("as_stored: %d of %d rows came back" % (n, table.shape[0]))
    return out


class SecondStage(object):
    """What GLUE, Best and Total share: where the sample of a previous run comes from.  Either its database file,
    `<out>/<catchment>.SMART.lhs[.nc]` (glue.py:182-188, best.py:172-178, total.py:122-128), or -- `sampling=` -- the
    finished run itself, whose parameter rows and objective functions are still on the GPU."""

    def _load_sampling(self, catchment, decompression_csv, sampling):
        self.sampling_run_file = '{}{}.SMART.lhs{}'.format(self.model.out_f, catchment,
                                                           '.nc' if self.out_format == 'netcdf' else '')
        self._device_obj_fns = self._device_rows = None
        self._stored_params = self._stored_obj_fns = self._host_sample = None
        if sampling is None:
            self.sampled_params, self.sampled_obj_fns = self._get_sampled_sets_from_file(
                self.sampling_run_file, self.param_names, self.obj_fn_names, decompression_csv)
            return
        if sampling.device_obj_fns is None:
            raise Exception("The sampling run handed to {} has not been run yet.".format(type(self).__name__))
        if list(sampling.obj_fn_names) != list(self.obj_fn_names):
            raise Exception("The sampling run handed to {} does not hold the same objective "
                            "functions.".format(type(self).__name__))
        import torch
        # The objective functions as the database hands them to the file-based constructors: float32, printed
        # '%.6e', parsed back (7 significant digits -- not a float32 round trip).  The conditions are evaluated on
        # exactly these numbers, uploaded again (4 bytes x 8 per sample), so that a threshold that falls between a
        # value and its printed form, or two values that only tie in print, select the same rows either way.
        device = sampling.device_obj_fns.device
        self._stored_obj_fns = as_stored(sampling.obj_fns)
        self._device_obj_fns = torch.from_numpy(self._stored_obj_fns).to(device)
        self._device_rows = sampling.device_sample if sampling.device_sample is not None \
            else torch.from_numpy(sampling._sample).to(device)
        self._host_sample = sampling._sample
        self._stored_params = None       # the [N, 10] host view is made when somebody looks at it

    @property
    def sampled_obj_fns(self):
        return self._stored_obj_fns

    @sampled_obj_fns.setter
    def sampled_obj_fns(self, value):
        self._stored_obj_fns = value

    @property
    def sampled_params(self):
        """float32 [N, 10]: what _get_sampled_sets_from_file returns for the sampling run's database."""
        if self._stored_params is None and getattr(self, '_host_sample', None) is not None:
            self._stored_params = as_stored(self._host_sample)
        return self._stored_params

    @sampled_params.setter
    def sampled_params(self, value):
        self._stored_params = value

    def _rows_as_stored(self, which):
        """Parameter rows picked on the device (boolean mask or index tensor) -> float32 host matrix with the
        rounding of the database."""
        return as_stored(self._device_rows[which].cpu().numpy()).reshape(-1, len(self.param_names))

    def _columns_of(self, names, problem):
        try:
            return [self.obj_fn_names.index(name) for name in names]
        except ValueError:
            raise Exception(problem)
