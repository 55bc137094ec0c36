"""Sharding of the sample axis over the GPUs of a node, one process per GPU.

The reference parallelises the same axis through spotpy's MPI job farm (lhs.py:75-89, montecarlo.py:153):
rank 0 hands one parameter vector to a worker and gets (discharge, [gw]) back, pickled, one sample at a
time.  Here every rank runs a contiguous block of ceil(N / world) rows of the sample matrix in one launch
and the only exchange is a single all-gather of the per-sample results ([N_local, 8] objective functions and
[N_local] groundwater ratios: 72 bytes per sample) over RCCL (torch.distributed backend "nccl" on ROCm).
Discharge series are never gathered (29 GB at N = 1e6): each rank keeps / writes its own shard.

The helpers work on CPU tensors with the gloo backend too, which is how tests/test_dist_gloo.py covers them.
"""
import os

import torch
import torch.distributed as dist


def shard_bounds(n_rows, world_size, rank):
    """Contiguous block [lo, hi) of rank `rank`: ceil(n / world) rows each, the last blocks possibly shorter/empty."""
    per = -(-n_rows // world_size)
    lo = min(rank * per, n_rows)
    return lo, min(lo + per, n_rows)


def shard_counts(n_rows, world_size):
    return [shard_bounds(n_rows, world_size, r)[1] - shard_bounds(n_rows, world_size, r)[0]
            for r in range(world_size)]


def env_world():
    """(rank, world_size, local_rank) from the torchrun environment; (0, 1, 0) when not launched distributed."""
    return (int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')),
            int(os.environ.get('LOCAL_RANK', '0')))


def init(backend=None):
    """Initialise torch.distributed from the environment (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE).
    Returns (rank, world_size, device).  backend defaults to nccl (= RCCL) when a GPU is visible, else gloo."""
    rank, world, local = env_world()
    backend = backend or os.environ.get('SMART_DIST_BACKEND') or None      # e.g. gloo: two ranks sharing one GPU
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    device = torch.device('cuda', torch.cuda.current_device()) if use_gpu else torch.device('cpu')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        # lazy communicator creation (no device_id): the first collective binds RCCL to the current device, set above
        dist.init_process_group(backend=backend or ('nccl' if use_gpu else 'gloo'), rank=rank, world_size=world)
    return rank, world, device


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_staged():
    """gloo moves host memory: device tensors are staged through the CPU (test set-ups only; RCCL takes them as is)."""
    return dist.get_backend() == 'gloo'


def rank_world():
    """(rank, world_size) of the initialised process group, (0, 1) otherwise."""
    if is_distributed():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def barrier():
    if is_distributed():
        if dist.get_backend() == 'nccl':
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def gather_rows(local, n_rows_total):
    """All-gather row blocks produced under shard_bounds(): local [n_local, ...] -> [n_rows_total, ...] on every
    rank.  One collective: blocks are padded to ceil(N / world) rows so that all_gather_into_tensor applies."""
    if not is_distributed():
        return local
    world = dist.get_world_size()
    per = -(-n_rows_total // world)
    tail = tuple(local.shape[1:])
    if local.shape[0] != per:
        pad = torch.zeros((per,) + tail, dtype=local.dtype, device=local.device)
        pad[:local.shape[0]] = local
        local = pad
    device = local.device
    if device.type != 'cpu' and _host_staged():
        local = local.cpu()
    out = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out[:n_rows_total].to(device)


def max_over_ranks(value, device):
    """Scalar max-reduce (timings)."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device='cpu' if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device):
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device='cpu' if _host_staged() else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
