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
import threading
import time
from datetime import timedelta

import torch
import torch.distributed as dist

# HSA_ENABLE_IPC_MODE_LEGACY.  RCCL between the processes of a node hands device memory from rank to rank through IPC
# handles; on a host driver that supports dmabuf IPC only, the legacy mode makes hipIpcGetMemHandle fail ("invalid
# argument") and with it the first collective of any N > 1 run.  The MI355X pool this was built on exports
# HSA_ENABLE_IPC_MODE_LEGACY=0 for that reason -- and a driver that knows the legacy mode only needs the opposite.  So
# this module does NOT touch the variable on import (round 5 did; advisor: it changed every user process that imported
# smartpy_amd.montecarlo): what the launcher exported stands; SMART_DIST_IPC_DMABUF=1 asks init() to set it to 0 (before
# this process's first GPU call, when the HSA runtime reads it); bench.py's own launcher sets it for the ranks it starts.
# INTEGRATION.md section 5 has the note.  RCCL with more than one rank has not run on hardware available to the builder:
# the N > 1 RCCL path below is UNVERIFIED until a multi-GPU run exists; everything around it (sharding, agreement,
# host staging) runs in tests/test_dist_gloo.py.


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


#: how device tensors travel in this process group: None = as the group's backend takes them (a group somebody else
#: initialised: read off its backend); False = through the RCCL subgroup init() made (_DATA_GROUP); True = staged
#: through the host over gloo -- two ranks of the group sit on ONE device (which RCCL refuses), or RCCL did not answer.
_STAGED = None
#: the RCCL subgroup device tensors travel through (init() makes it beside the gloo default group); None: the default group
_DATA_GROUP = None
#: a communicator that raised or never answered was left behind in this process: finish() must not wait for it
_ABANDONED = False
#: why init() gave up on RCCL in this process (the text of the exception, or what came back wrong); None otherwise
rccl_failure = None


def timeouts():
    """(seconds a host-side collective may wait for a peer, seconds the first device collective may take, seconds RCCL's
    own watchdog allows a device collective) -- SMART_DIST_TIMEOUT (120), SMART_DIST_PROBE_TIMEOUT (three quarters of it),
    SMART_DIST_RCCL_TIMEOUT (1800).  A rank that never arrives makes its peers RAISE after the first; a first device
    collective that hangs makes every rank fall back to host staging after the second; the third is what torch's watchdog
    ends the process with when a LATER device collective hangs (long on purpose: a communicator that was given up on must
    not take the process down while it works through the host)."""
    host = float(os.environ.get('SMART_DIST_TIMEOUT', '120'))
    probe = float(os.environ.get('SMART_DIST_PROBE_TIMEOUT', str(0.75 * host)))
    return host, probe, float(os.environ.get('SMART_DIST_RCCL_TIMEOUT', '1800'))


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


def init(backend=None, probe=None):
    """Initialise torch.distributed from the environment (MASTER_ADDR / MASTER_PORT / RANK / WORLD_SIZE).
    Returns (rank, world_size, device).

    The DEFAULT group is gloo, with a timeout (timeouts()): host objects, agreements on outcomes, and the data path of
    last resort travel through it, and a rank that never arrives makes its peers raise instead of wait.  With GPUs the
    ranks then tell each other which physical device each of them sits on (host name + UUID); if no two share one, an
    RCCL SUBGROUP over all ranks is made for the device tensors (communicator built by its first collective), and that
    first collective is run here, at once, with a bound (rccl_answers): if it raises or does not answer on ANY rank,
    every rank drops the subgroup and stages its result blocks through the host over gloo -- 72 bytes per sample, once
    per ensemble -- `rccl_failure` keeps the reason and bench.py's line reports it.  Two ranks on one device (a one-GPU
    box running the N-rank code path) stage through the host from the start.  The decision does not look at WORLD_SIZE
    against device_count(): a launch with one visible device per rank (ROCR_VISIBLE_DEVICES / HIP_VISIBLE_DEVICES per
    task, `srun --gpus-per-task=1`) has more ranks than visible devices with every rank on a GPU of its own.
    `backend` / SMART_DIST_BACKEND: 'gloo' = host staging, no RCCL; 'nccl' = RCCL for everything or nothing (one group,
    no fallback).  `probe`: tests put a stand-in for the first device collective here."""
    global _STAGED, _DATA_GROUP, _ABANDONED, rccl_failure
    # (a group initialised earlier in this process, by this function or by somebody else, leaves nothing behind here)
    _STAGED, _DATA_GROUP, _ABANDONED = None, None, False
    rccl_failure = os.environ.get('SMART_DIST_RCCL_FAILURE') or None   # (bench.py's launcher: why it restarted on gloo)
    if os.environ.get('SMART_DIST_IPC_DMABUF') == '1':
        os.environ['HSA_ENABLE_IPC_MODE_LEGACY'] = '0'         # (see the top of the module; before the first GPU call)
    rank, world, local = env_world()
    backend = backend or os.environ.get('SMART_DIST_BACKEND') or None      # e.g. gloo: two ranks sharing one GPU
    use_gpu = torch.cuda.is_available()
    if use_gpu:
        torch.cuda.set_device(local % max(torch.cuda.device_count(), 1))
    device = torch.device('cuda', torch.cuda.current_device()) if use_gpu else torch.device('cpu')
    # (SMART_DIST_SINGLE=1: a group of ONE rank -- what a one-GPU box can run of the RCCL subgroup's code, tests only)
    if (world > 1 or os.environ.get('SMART_DIST_SINGLE') == '1') and not dist.is_initialized():
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29500')
        t_host, _, t_rccl = timeouts()
        if backend == 'nccl':
            dist.init_process_group(backend='nccl', rank=rank, world_size=world, timeout=timedelta(seconds=t_rccl))
            _STAGED = False
            return rank, world, device
        dist.init_process_group(backend='gloo', rank=rank, world_size=world, timeout=timedelta(seconds=t_host))
        _STAGED = True
        if backend is None and use_gpu:
            where = [None] * world
            dist.all_gather_object(where, device_identity(device))
            if shares_a_device(where):
                if rank == 0:
                    import warnings
                    warnings.warn("smartpy_amd.distributed: %d ranks on %d physical device(s): device tensors are staged "
                                  "through the host over gloo, not handed to RCCL" % (world, len(set(where))))
            else:
                group = dist.new_group(backend='nccl', timeout=timedelta(seconds=t_rccl))
                if rccl_answers(device, group=group, probe=probe):
                    _DATA_GROUP, _STAGED = group, False
    return rank, world, device


def _first_device_collective(device, group=None):
    """One all-reduce of a double on the device: what makes RCCL build its communicator (and open its IPC handles).
    Asynchronous, and polled: nothing here makes the device's default stream wait for a collective that may never end."""
    torch.cuda.set_device(device)           # (the current device is a per-thread setting)
    t = torch.ones(1, dtype=torch.float64, device=device)
    work = dist.all_reduce(t, group=group, async_op=True)
    while not work.is_completed():
        time.sleep(0.002)
    work.wait()
    return float(t.item())


def rccl_answers(device, group=None, probe=None):
    """Run the group's first device collective NOW, in a thread of its own and with a bound (timeouts()[1]), and let the
    ranks agree (over the gloo default group) on whether it worked.

    What travels between the ranks of this package is small (72 bytes per sample, once per ensemble; the compute never
    waits for a peer), so a node whose RCCL cannot start -- a driver without the IPC mode it wants, a masked xGMI link
    -- does not have to lose the run: if the collective RAISES, returns something wrong or DOES NOT ANSWER within the
    bound on any rank, every rank stages its result blocks through the host instead, rank 0 says so in a warning and
    `rccl_failure` keeps the reason.  The communicator that failed is left alone (its peers may sit inside its set-up
    for good; destroying it could wait for them): finish() ends such a process without waiting for it."""
    global rccl_failure, _ABANDONED
    world = dist.get_world_size()
    bound = timeouts()[1]
    box = {}

    def run():
        try:
            box['got'] = (probe or _first_device_collective)(device, group)
        except Exception as e:      # noqa: BLE001 -- whatever the backend raises: the decision below is the handling
            box['error'] = '%s: %s' % (type(e).__name__, str(e).strip().splitlines()[0] if str(e).strip() else '')

    th = threading.Thread(target=run, name='smartpy_amd-first-collective', daemon=True)
    t0 = time.monotonic()
    th.start()
    th.join(bound)
    mine = None
    if th.is_alive():
        mine = 'the first device collective over %d ranks did not answer within %.0f s' % (world, bound)
    elif 'error' in box:
        mine = box['error']
    elif box.get('got') != float(world):
        mine = 'the first all-reduce over %d ranks returned %r' % (world, box.get('got'))
    bad = torch.tensor([0.0 if mine is None else 1.0])
    dist.all_reduce(bad, op=dist.ReduceOp.MAX)                    # a host tensor over the gloo default group
    if bad.item() > 0.0:
        rccl_failure = mine if mine is not None else 'RCCL failed on another rank'
        _ABANDONED = True
        if dist.get_rank() == 0:
            import warnings
            warnings.warn("smartpy_amd.distributed: RCCL did not start (%s; %.1f s): result blocks are staged through the "
                          "host over gloo" % (rccl_failure, time.monotonic() - t0))
        return False
    return True


def finish(code=0):
    """The end of a program that called init(): destroy the process group -- unless a communicator that failed or hung
    was left behind (rccl_answers), in which case tearing it down could wait for peers that never come: then the
    standard streams are flushed and the process ends at once with `code`."""
    import sys
    if dist.is_available() and dist.is_initialized():
        if _ABANDONED:
            sys.stdout.flush()
            sys.stderr.flush()
            os._exit(code)
        dist.destroy_process_group()


def is_distributed():
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def _host_staged():
    """Are device tensors staged through the CPU for the collectives (gloo moves host memory: test set-ups, ranks that
    share a device, an RCCL that did not answer; RCCL takes device tensors as they are)?"""
    if _STAGED is not None:
        return _STAGED
    return dist.get_backend() == 'gloo'        # a group somebody else initialised


def _host_scalars():
    """Do one-number agreements (outcomes, timings) travel as host tensors?  Whenever the default group can carry them
    (init()'s gloo default group; any group with a gloo part): an agreement must not depend on the device data path."""
    return _host_staged() or 'gloo' in str(dist.get_backend()) or not torch.cuda.is_available()


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
        dist.all_reduce(t, group=_DATA_GROUP if t.is_cuda else None)
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
    dist.all_gather_into_tensor(out, local.contiguous(), group=_DATA_GROUP if local.is_cuda else None)
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
    group = None if staged else _DATA_GROUP
    tail = tuple(mine.shape[1:])
    # the receiving rank makes room FIRST (the whole [N, R] matrix on its host: 15 GB of float32 at N = 1e6), and the
    # ranks agree that it could before anybody sends: a sender must not sit in dist.send() towards a rank that has
    # raised MemoryError (round 5's advisor)
    failure, out, buf = None, None, None
    if rank == dst:
        try:
            out = np.empty((n_rows_total,) + tail, dtype=torch.empty(0, dtype=mine.dtype).numpy().dtype)
            buf = torch.empty((-(-n_rows_total // world),) + tail, dtype=mine.dtype, device=mine.device)
        except Exception as e:      # noqa: BLE001 -- MemoryError, a torch allocation error: the peers hear of either
            failure = e
    agree_or_raise(failure)
    if rank != dst:
        lo, hi = shard_bounds(n_rows_total, world, rank)
        if hi > lo:
            dist.send(mine[:hi - lo].contiguous(), dst=dst, group=group)
            last_collect_bytes = (hi - lo) * int(np.prod(tail, dtype=np.int64)) * mine.element_size()
        return None
    for r in range(world):
        lo, hi = shard_bounds(n_rows_total, world, r)
        if hi <= lo:
            continue
        if r == rank:
            out[lo:hi] = mine[:hi - lo].cpu().numpy()
        else:
            dist.recv(buf[:hi - lo], src=r, group=group)
            out[lo:hi] = buf[:hi - lo].cpu().numpy()
            last_collect_bytes += (hi - lo) * int(np.prod(tail, dtype=np.int64)) * mine.element_size()
    return out


def agree_or_raise(failure, device=None):
    """All ranks get here; `failure` is this rank's exception (or None; ANY exception: a status word that did not clear,
    a HIP out-of-memory, a ValueError).  If ANY rank failed, EVERY rank raises, in this call: the failed one its own
    exception, the others a SmartEngineError that says so -- no rank is left waiting in a collective its failed peer
    never joins."""
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
    dist.broadcast(t, src=src, group=_DATA_GROUP if t.is_cuda else None)
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
    """Where a one-number reduction lives: the host whenever the default group carries host tensors (init()'s does), else
    the given (or current) GPU."""
    if _host_scalars():
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
