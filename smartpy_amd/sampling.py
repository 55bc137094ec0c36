"""Latin hypercube sampling of the SMART parameter space -- host side (montecarlo/lhs.py:133-167).

Same algorithm and the same random stream as the reference: one rand(n, 10) draw and ten permutation(n)
draws from NumPy's legacy global RNG, (permutation + rand) / n, then the inverse CDF of the uniform
distribution on [lo, hi], lo + u * (hi - lo) (what scipy.stats.uniform.ppf evaluates).  With the same
np.random.seed the matrix is bit-identical to the reference's (tests/test_host_logic.py, KAT-7).
"""
import numpy as np

PARAMETER_NAMES = ['T', 'C', 'H', 'D', 'S', 'Z', 'SK', 'FK', 'GK', 'RK']     # parameters.py:25


def latin_hypercube(sample_size, ranges, names=None, seed=None):
    """[sample_size, len(names)] float64.  seed=None draws from the current global stream like the reference;
    an integer seeds NumPy's legacy generator first (the reference leaves seeding to the caller)."""
    names = names or PARAMETER_NAMES
    if seed is not None:
        np.random.seed(seed)
    bounds = np.asarray([[ranges[p][0], ranges[p][1]] for p in names], dtype=np.float64)
    nb = len(names)
    random_matrix = np.random.rand(sample_size, nb)
    plan = np.empty((sample_size, nb), dtype=np.float64)
    for p in range(nb):
        plan[:, p] = np.random.permutation(sample_size)
    plan += random_matrix
    plan /= sample_size
    return bounds[:, 0] + plan * (bounds[:, 1] - bounds[:, 0])


def latin_hypercube_device(sample_size, ranges, names=None, seed=None, device=None):
    """The same sampling plan built on the GPU with torch (for N >= 1e6, where the host sampler's 1.5 s would be
    comparable to the whole ensemble launch): one rand(n, k) draw, one random permutation of the n strata per
    parameter, (stratum + rand) / n, inverse CDF of the uniform distribution.  Same algorithm as
    montecarlo/lhs.py:133-167 but NOT the same random stream as NumPy's legacy generator: use
    `latin_hypercube` when the sample has to be reproduced bit for bit."""
    import torch
    names = names or PARAMETER_NAMES
    device = torch.device(device) if device is not None else torch.device('cuda' if torch.cuda.is_available() else 'cpu')
    gen = torch.Generator(device=device)
    if seed is not None:
        gen.manual_seed(int(seed))
    else:
        gen.seed()
    nb = len(names)
    lo = torch.tensor([ranges[p][0] for p in names], dtype=torch.float64, device=device)
    hi = torch.tensor([ranges[p][1] for p in names], dtype=torch.float64, device=device)
    rnd = torch.rand((sample_size, nb), dtype=torch.float64, device=device, generator=gen)
    # a random permutation per column: argsort of iid uniforms
    strata = torch.argsort(torch.rand((sample_size, nb), dtype=torch.float64, device=device, generator=gen), dim=0)
    plan = (strata.to(torch.float64) + rnd) / sample_size
    return lo + plan * (hi - lo)
