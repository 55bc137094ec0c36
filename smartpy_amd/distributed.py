"""Sharding of an ensemble over the GPUs of a node, one process per GPU: along the sample axis, or -- for a
catchment x sample batch -- along the catchment axis.

The reference parallelises the same axis through spotpy's MPI job farm (lhs.py:75-89, montecarlo.py:153):
rank 0 hands one parameter vector to a worker and gets (discharge, [gw]) back, pickled, one sample at a
time.  Here every rank runs a contiguous block of ceil(N / world) rows of the sample matrix in one launch
and the only exchange is a single all-gather of the per-sample results ([N_local, 8] objective functions and
[N_local] groundwater ratios: 72 bytes per sample) over RCCL (torch.distributed backend "nccl" on ROCm).
Discharge series are never gathered (29 GB at N = 1e6): each rank keeps / writes its own shard.

A catchment x sample batch (BASELINE config 5: 64 catchments x 1e4 samples) is cut along the CATCHMENT axis instead:
rank r holds the forcing, areas and observations of its own block of catchments only (the forcing is the one large
per-catchment input: 1.4 MB each), runs them as one launch with the catchment index on grid.y, and the same single
all-gather returns [C, N, 9].  ShardedEnsemble below does either.

The helpers work on CPU tensors with the gloo backend too, which is how tests/test_dist_gloo.py covers them.
"""
import os

# RCCL between the processes of a node hands device memory from rank to rank through IPC handles, and the host driver
# of the MI355X pool this was built on supports dmabuf IPC only: with the legacy mode left on, hipIpcGetMemHandle
# fails ("invalid argument") and with it the first collective of any N > 1 run.  The pool's own environment exports
# HSA_ENABLE_IPC_MODE_LEGACY=0 for that reason (the build notes say so; that, not a measurement of ours, decides it:
# no run with more than one GPU was available to the builder).  The HSA runtime reads the variable when it starts,
# i.e. at this process's first GPU call -- so it is set here, at import, ahead of anything that could make one, for
# every way the ranks may have been started (torchrun, mpirun, srun, bench.py's own launcher).  A value the user set
# stands.
os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')

import torch                            # noqa: E402
import torch.distributed as dist        # noqa: E402


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


#: how device tensors travel in this process group: None = as the group's backend takes them (RCCL for device tensors);
#: True = staged through the host over gloo, because two ranks of the group sit on ONE device, which RCCL refuses.
#: Decided by init(); a group initialised elsewhere (the tests' own gloo groups) is read off its backend.
_STAGED = None


def device_identity(device):
    """What tells one physical GPU from another across the processes of a job: host name + the device's UUID (the PCI
    bus id where the runtime has no UUID to give).  Independent of how the launcher numbered or masked the devices."""
    props = torch.cuda.get_device_properties(device)
    uuid = str(getattr(props, 'uuid', '') or '')
    return '%s/%s' % (os.uname().nodename, uuid or 'pci-%s' % getattr(props, 'pci_bus_id', device.index))


def shares_a_device(identities):
    """Do two of the ranks (one device_identity() each) sit on the same physical GPU?  RCCL refuses that; what the answer
    does NOT depend on is how many devices any one rank can see."""
    return len(set(identities)) < len(identities)


def init(backend=None):
    """Initialise torch.distributed from the environment (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE).
    Returns (rank, world_size, device).

    With GPUs the group is created with BOTH backends (gloo for host tensors, RCCL for device tensors; RCCL's
    communicator is only built by the first device collective), the ranks then tell each other which physical device
    each of them sits on (host name + UUID, over gloo), and only if two of them share one -- a one-GPU box running the
    N-rank code path -- are device tensors staged through the host instead of handed to RCCL, which refuses two ranks
    on a device.  The decision does not look at WORLD_SIZE against device_count(): a launch with one visible device
    per rank (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES per task, `srun --gpus-per-task=1`) or over several nodes has
    more ranks than visible devices with every rank on a GPU of its own, and keeps RCCL.  `backend` /
    SMART_DIST_BACKEND ('nccl' | 'gloo') overrides."""
    global _STAGED
    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # (see the top of the module; before the first GPU call)
    rank, world, local = env_world()
    backend = backend or os.environ.get('SMART_DIST_BACKEND') or None      # e.g. gloo: two ranks sharing one GPU
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    device = torch.device('cuda', torch.cuda.current_device()) if use_gpu else torch.device('cpu')
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        if backend is not None or not use_gpu:
            backend = backend or 'gloo'
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
            _STAGED = backend == 'gloo'
        else:
            # lazy communicator creation (no device_id): the first device collective binds RCCL to the current device
            dist.init_process_group(backend='cpu:gloo,cuda:nccl', rank=rank, world_size=world)
            where = [None] * world
            dist.all_gather_object(where, device_identity(device))         # host objects: gloo
            _STAGED = shares_a_device(where)
            if _STAGED and rank == 0:
                import warnings
                warnings.warn("smartpy_amd.distributed: %d ranks on %d physical device(s): device tensors are staged "
                              "through the host over gloo, not handed to RCCL" % (world, len(set(where))))
            if not _STAGED:
                _STAGED = not rccl_answers(device)
    return rank, world, device


#: why rccl_answers() gave up on RCCL in this process (the text of the exception, or what came back wrong); None otherwise
rccl_failure = None


def _first_device_collective(device):
    """One all-reduce of a double on the device: what makes RCCL build its communicator (and open its IPC handles)."""
    t = torch.ones(1, dtype=torch.float64, device=device)
    dist.all_reduce(t)
    torch.cuda.current_stream(device).synchronize()
    return float(t.item())


def rccl_answers(device, probe=None):
    """Run the group's first device collective NOW, and let the ranks agree (over gloo) on whether it worked.

    What travels between the ranks of this package is small (72 bytes per sample, once per ensemble; the compute never
    waits for a peer), so a node whose RCCL cannot start -- a driver without the IPC mode it wants, a masked xGMI link
    -- does not have to lose the run: if the collective RAISES on any rank, every rank stages its result blocks through
    the host over gloo instead, rank 0 says so in a warning, `rccl_failure` keeps the reason and bench.py's line reports
    backend 'gloo' with it.  A collective that hangs is not caught here (RCCL's own watchdog ends the job);
    SMART_DIST_BACKEND=nccl asks for RCCL alone and never comes this way."""
    global rccl_failure
    world = dist.get_world_size()
    try:
        got = (probe or _first_device_collective)(device)
        if got != float(world):
            rccl_failure = 'the first all-reduce over %d ranks returned %r' % (world, got)
    except Exception as e:      # noqa: BLE001 -- whatever the backend raises: the decision below is the handling
        rccl_failure = '%s: %s' % (type(e).__name__, str(e).strip().splitlines()[0] if str(e).strip() else '')
    bad = torch.tensor([0.0 if rccl_failure is None else 1.0])
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)                    # a host tensor: gloo
    if bad.item() > 0.0:
        if rccl_failure is None:
            rccl_failure = 'RCCL failed on another rank'
        if dist.get_rank() == 0:
            import warnings
            warnings.warn("smartpy_amd.distributed: RCCL did not start (%s): result blocks are staged through the host "
                          "over gloo" % rccl_failure)
        return False
    return True


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_staged():
    """Are device tensors staged through the CPU for the collectives (gloo moves host memory; test set-ups and ranks that
    share a device only: RCCL takes device tensors as they are)?"""
    if _STAGED is not None:
        return _STAGED
    return dist.get_backend() == 'gloo'        # a group somebody else initialised


def data_backend():
    """What carries the result blocks between the ranks: 'nccl' (= RCCL) or 'gloo'; None without a group."""
    if not is_distributed():
        return None
    return 'gloo' if _host_staged() else 'nccl'


def rank_world():
    """(rank, world_size) of the initialised process group, (0, 1) otherwise."""
    if is_distributed():
        return dist.get_rank(), dist.get_world_size()
    return 0, 1


def barrier():
    """Every rank has got here.  One small all-reduce in the memory the group's data path uses (a device tensor over
    RCCL: the barrier then also orders behind the collectives queued on the device; a host tensor over gloo)."""
    if is_distributed():
        t = torch.zeros(1, dtype=torch.float64,
                        device='cpu' if _host_staged() else torch.device('cuda', torch.cuda.current_device()))
        dist.all_reduce(t)
        if t.is_cuda:
            torch.cuda.current_stream(t.device).synchronize()


def gather_rows(local, n_rows_total, out=None):
    """All-gather row blocks produced under shard_bounds(): local [n_local, ...] -> [n_rows_total, ...] on every
    rank.  One collective: blocks are padded to ceil(N / world) rows so that all_gather_into_tensor applies.
    `out`: a [world * ceil(N / world), ...] buffer to gather into (repeated calls then allocate nothing when the
    local block already has ceil(N / world) rows)."""
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
        out = None
    if out is None:
        out = torch.empty((world * per,) + tail, dtype=local.dtype, device=local.device)
    dist.all_gather_into_tensor(out, local.contiguous())
    return out[:n_rows_total].to(device)


#: bytes the last collect_rows() moved through THIS rank's buffers (sent or received): what the tests hold the
#: "discharge travels to rank 0 only" promise against
last_collect_bytes = 0


def collect_rows(local, n_rows_total, dst=0):
    """Row blocks produced under shard_bounds() -> ONE host matrix [n_rows_total, ...] on rank `dst` (a numpy array of
    local's dtype), None on the others.  Point to point, one block at a time into one staging buffer: rank r sends its
    block and is done; rank dst never holds more than its own block plus one peer's on the device.

    This is how the simulated series of MonteCarlo.run(save_sim=True) reach the rank that writes the database -- the
    reference's MPI farm sends each series to the master only, too (montecarlo.py:211-231) -- where an all-gather
    would put the whole [N, R] matrix (29 GB of fp64 at N = 1e6) on every GPU and every host."""
    global last_collect_bytes
    import numpy as np
    last_collect_bytes = 0
    if not is_distributed():
        return local.detach().cpu().numpy()
    rank, world = dist.get_rank(), dist.get_world_size()
    staged = _host_staged() or not local.is_cuda
    mine = local.contiguous()
    if staged:
        mine = mine.cpu()
    tail = tuple(mine.shape[1:])
    if rank != dst:
        lo, hi = shard_bounds(n_rows_total, world, rank)
        if hi > lo:
            dist.send(mine[:hi - lo].contiguous(), dst=dst)
            last_collect_bytes = (hi - lo) * int(np.prod(tail, dtype=np.int64)) * mine.element_size()
        return None
    out = np.empty((n_rows_total,) + tail, dtype=torch.empty(0, dtype=mine.dtype).numpy().dtype)
    per = -(-n_rows_total // world)
    buf = torch.empty((per,) + tail, dtype=mine.dtype, device=mine.device)
    for r in range(world):
        lo, hi = shard_bounds(n_rows_total, world, r)
        if hi <= lo:
            continue
        if r == rank:
            out[lo:hi] = mine[:hi - lo].cpu().numpy()
        else:
            dist.recv(buf[:hi - lo], src=r)
            out[lo:hi] = buf[:hi - lo].cpu().numpy()
            last_collect_bytes += (hi - lo) * int(np.prod(tail, dtype=np.int64)) * mine.element_size()
    return out


def agree_or_raise(failure, device=None):
    """All ranks get here; `failure` is this rank's exception (or None).  If ANY rank failed, EVERY rank raises, in this
    call: the failed one its own exception, the others a SmartEngineError that says so -- no rank is left waiting in a
    collective its failed peer never joins."""
    from .engine import SmartEngineError
    worst = max_over_ranks(1.0 if failure is not None else 0.0, device)
    if failure is not None:
        raise failure
    if worst > 0.0:
        raise SmartEngineError(-6, "smartpy_amd: another rank's launch failed; the gathered results would not be complete")


def broadcast_matrix(matrix, src=0):
    """Every rank gets rank `src`'s copy of a numpy matrix (shape and dtype included).  The Monte-Carlo classes call
    this on their sample before sharding it: the reference's sampler draws from NumPy's unseeded global stream
    (lhs.py:149,154), so processes started side by side would otherwise each hold a different matrix -- in the
    reference's MPI mode only the master draws (spotpy's mpi sampler), which is what this restores."""
    import numpy as np
    if not is_distributed():
        return matrix
    device = torch.device('cpu') if _host_staged() else torch.device('cuda', torch.cuda.current_device())
    head = [matrix.shape, str(matrix.dtype)] if dist.get_rank() == src else [None, None]
    dist.broadcast_object_list(head, src=src)
    if dist.get_rank() == src:
        t = torch.from_numpy(np.ascontiguousarray(matrix)).to(device)
    else:
        t = torch.empty(tuple(head[0]), dtype=getattr(torch, head[1]), device=device)
    dist.broadcast(t, src=src)
    return t.cpu().numpy()


class ShardedEnsemble(object):
    """This rank's part of an ensemble cut over the process group, prepared once (engine.prepare_ensemble) so that
    step() = one launch + the one all-gather, nothing else.

    axis='samples'     params [N, 10] is cut into contiguous row blocks; forcing / area / obs are shared.
                       local_block=True: params already IS this rank's block (every rank drew its own rows, all
                       blocks equally long): nothing is cut, the gathered matrix is the blocks in rank order.
    axis='catchments'  the catchments are cut into contiguous blocks; `forcing` is either the full [C, T, 2] array or
                       a callable c -> [T, 2] that is only asked for this rank's catchments; area_m2, obs, gw_obs,
                       extra may be scalars / per-catchment arrays of the FULL batch (they are sliced here).
    step() returns [N, 9] (samples) or [C, N, 9] (catchments) on every rank: the 8 objective functions and the
    groundwater ratio of every run.  Without observations the objective columns are NaN."""

    def __init__(self, params, forcing, area_m2, delta_sec, n_warm, report_gap, axis='samples', n_catchments=None,
                 obs=None, gw_obs=None, extra=None, device=None, local_block=False, **kw):
        import numpy as np
        from . import engine
        rank, world = rank_world()
        self.axis = axis
        self.device = torch.device(device) if device is not None else engine.default_device()
        if axis == 'samples':
            self.n_total = params.shape[0] * (world if local_block else 1)
            lo, hi = (0, params.shape[0]) if local_block else shard_bounds(self.n_total, world, rank)
            self.n_local = hi - lo
            local_params = params[lo:hi] if hi > lo else params[:1]       # an empty shard still joins the collective
            self._prep = engine.prepare_ensemble(local_params, forcing, area_m2, delta_sec, n_warm, report_gap,
                                                 obs=obs, gw_obs=gw_obs, extra=extra, device=self.device, **kw)
            n_run = local_params.shape[0]
            tail = (9,)
        elif axis == 'catchments':
            C = int(n_catchments if n_catchments is not None else forcing.shape[0])
            self.n_total = C
            lo, hi = shard_bounds(C, world, rank)
            self.n_local = hi - lo
            cs = list(range(lo, hi)) if hi > lo else [0]

            def per_catchment(x, shared_ndim):
                """a scalar / shared array stays; an array with a leading catchment axis is cut to this rank's block"""
                if x is None or isinstance(x, dict) or np.ndim(x) <= shared_ndim:
                    return x
                return x[cs]
            f_local = np.stack([np.asarray(forcing(c), dtype=np.float64) for c in cs]) if callable(forcing) \
                else forcing[cs]
            self._prep = engine.prepare_ensemble(
                per_catchment(params, 2), f_local, per_catchment(area_m2, 0), delta_sec, n_warm, report_gap,
                obs=per_catchment(obs, 1), gw_obs=per_catchment(gw_obs, 0), extra=per_catchment(extra, 1),
                device=self.device, **kw)
            n_run = len(cs)
            tail = (params.shape[-2], 9)
        else:
            raise Exception("axis must be 'samples' or 'catchments'")
        per = -(-self.n_total // world)
        self._packed = torch.full((max(per, n_run),) + tail, float('nan'), dtype=torch.float64, device=self.device)
        self._n_run = n_run
        self._gathered = torch.empty((world * per,) + tail, dtype=torch.float64, device=self.device) \
            if is_distributed() and not _host_staged() else None

    def step(self):
        self._prep.enqueue()
        return self._pack_and_gather(self._prep.result())

    def _pack_and_gather(self, out):
        k = self.n_local
        if k:
            if self.axis == 'samples':
                if out.objfn is not None:
                    self._packed[:k, :8].copy_(out.objfn[:k])
                self._packed[:k, 8].copy_(out.gw[:k])
            else:
                objfn = out.objfn if out.objfn is None or out.objfn.dim() == 3 else out.objfn.unsqueeze(0)
                gw = out.gw if out.gw.dim() == 2 else out.gw.unsqueeze(0)
                if objfn is not None:
                    self._packed[:k, :, :8].copy_(objfn[:k])
                self._packed[:k, :, 8].copy_(gw[:k])
        per = -(-self.n_total // rank_world()[1])
        return gather_rows(self._packed[:per], self.n_total, out=self._gathered)

    def verify(self):
        """Status word of this rank's last launch (engine.PreparedEnsemble.verify), agreed over the ranks.

        Returns None when the matrix step() returned stands on every rank; the freshly gathered [N, 9] / [C, N, 9]
        matrix when ANY rank had to repeat its launch (a time slice that timed out, a stale plan: the matrix step()
        gathered holds that rank's NaN rows, so every rank packs and gathers again).  (Not the EnsembleResult that
        PreparedEnsemble.verify() returns: this rank's own outputs are `prepared.result()`.)

        A rank whose repeated launch is not clean either raises SmartEngineError -- and so does every other rank, in
        the same call: the outcome travels in the one reduction this method makes, so that no rank is left waiting in
        a collective its failed peer never joins."""
        from .engine import SmartEngineError
        failure, out = None, None
        try:
            out = self._prep.verify()
        except SmartEngineError as e:
            failure = e
        code = 2.0 if failure is not None else (1.0 if self._prep.repeated else 0.0)
        worst = max_over_ranks(code, self.device)
        if failure is not None:
            raise failure
        if worst >= 2.0:
            raise SmartEngineError(-6, "smartpy_amd: another rank's repeated launch still reports a status; the "
                                       "gathered results are not complete")
        if worst >= 1.0:
            return self._pack_and_gather(out)
        return None

    @property
    def prepared(self):
        return self._prep


def _scalar_device(device=None):
    """Where a one-number reduction lives: the host when the group stages through it, else the given (or current) GPU."""
    if _host_staged() or not torch.cuda.is_available():
        return torch.device('cpu')
    return device if device is not None else torch.device('cuda', torch.cuda.current_device())


def max_over_ranks(value, device=None):
    """Scalar max-reduce (timings, outcomes)."""
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value, device=None):
    if not is_distributed():
        return float(value)
    t = torch.tensor([float(value)], dtype=torch.float64, device=_scalar_device(device))
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())
