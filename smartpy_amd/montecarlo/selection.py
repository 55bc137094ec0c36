"""Conditions on objective functions shared by GLUE and Best (glue.py:222-289, best.py:221-287).

`condition_mask` and `best_rows` accept numpy arrays (what the file-based second stages of the reference hand them)
as well as torch tensors on any device: on an `[N, 8]` objective matrix that is still on the GPU the behavioural
mask, its count and the top-k rows are computed there and only the selected rows travel.
"""
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
    `constrained` is true, in ascending order of the value -- best.py:287 `argsort()[-nb_best:]`."""
    if _is_torch(sort_fn):
        import torch
        idx = torch.nonzero(constrained, as_tuple=False)[:, 0]
        order = torch.argsort(sort_fn[idx], stable=True)
        return idx[order][-nb_best:]
    idx = np.nonzero(constrained)[0]
    return idx[np.argsort(sort_fn[idx])][-nb_best:]
