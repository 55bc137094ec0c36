"""Host side of the ensemble engine: torch tensors in, one HIP launch through the C ABI, torch tensors out.

PyTorch is plumbing here (device memory, streams, torch.distributed); all arithmetic of the hot path is in
smartpy_amd/csrc/*.hip behind include/smart_amd.h.  Nothing in this module computes model steps on the CPU
and nothing imports the test oracle.
"""
import ctypes
import math
import threading
import warnings
import weakref

import numpy as np
import torch

from . import _lib
from ._lib import SmartEngineError, REPORT_SUMMARY, REPORT_RAW, MATH_LITERAL, MATH_FAST  # noqa: F401

OBJ_FN_NAMES = ['NSE', 'KGE', 'KGEc', 'KGEa', 'KGEb', 'PBias', 'RMSE', 'GW']   # montecarlo.py:71-74
VARIABLES = ['Q_aeva', 'Q_ove', 'Q_dra', 'Q_int', 'Q_sgw', 'Q_dgw', 'Q_out',
             'V_ove', 'V_dra', 'V_int', 'V_sgw', 'V_dgw',
             'V_ly1', 'V_ly2', 'V_ly3', 'V_ly4', 'V_ly5', 'V_ly6', 'V_river']  # structure.py:78-82

_REPORT = {'summary': REPORT_SUMMARY, 'raw': REPORT_RAW}
_MATH = {'literal': MATH_LITERAL, 'fast': MATH_FAST}
_LITERAL_FORMS = {'auto': _lib.LITERAL_FORM_AUTO, 'rows': _lib.LITERAL_FORM_ROWS, 'lanes': _lib.LITERAL_FORM_LANES}


def report_code(report):
    """structure.py:65-70."""
    try:
        return _REPORT[report]
    except KeyError:
        raise Exception("Reporting type '{}' unknown.".format(report))


def default_device():
    if not torch.cuda.is_available():
        raise SmartEngineError(-6, "smartpy_amd needs a HIP device (torch.cuda.is_available() is False); "
                                   "there is no CPU fallback")
    return torch.device('cuda', torch.cuda.current_device())


#: bytes this module has copied from the host to a device since it was imported (as_device and SingleRun): what the
#: tests hold "a repeated simulate() uploads its ten parameters and nothing else" against
h2d_bytes = 0


def as_device(x, device, shape=None):
    """numpy / list / tensor -> contiguous fp64 tensor on device."""
    global h2d_bytes
    if x is None:
        return None
    if not isinstance(x, torch.Tensor):
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(x, dtype=np.float64)))
    if not x.is_cuda and torch.device(device).type == 'cuda':
        h2d_bytes += x.numel() * 8
    x = x.to(device=device, dtype=torch.float64).contiguous()
    if shape is not None:
        x = x.reshape(shape)
    return x


def extra_vector(extra):
    """{'aar', 'r-o_ratio', 'r-o_split'} -> the 7 doubles of include/smart_amd.h (structure.py:100-112)."""
    if not extra:
        return None
    return [float(extra['aar']), float(extra['r-o_ratio'])] + [float(v) for v in extra['r-o_split']]


class EnsembleResult(object):
    """Outputs of one launch.  discharge is a [C, N, R] *view* of the sample-minor buffer the kernel writes."""

    def __init__(self, discharge_rn, gw, objfn, final_vars, n_samples, squeeze):
        self._dis = discharge_rn        # [C, R, ld] or None
        self._n = n_samples
        self._squeeze = squeeze
        self.gw = gw[0] if squeeze else gw
        self.objfn = None if objfn is None else (objfn[0] if squeeze else objfn)
        self.final_vars = None if final_vars is None else (final_vars[0] if squeeze else final_vars)

    @property
    def discharge_report_major(self):
        """[C, R, N] (or [R, N]) exactly as stored: row r holds every sample's discharge of report step r."""
        if self._dis is None:
            return None
        d = self._dis[:, :, :self._n]
        return d[0] if self._squeeze else d

    @property
    def discharge(self):
        """[C, N, R] (or [N, R]): discharge[n] is what SMART.simulate(row n)[0] returns (smart.py:208)."""
        d = self.discharge_report_major
        return None if d is None else d.transpose(-1, -2)


def n_reports(n_steps, gap, report_type):
    return int(_lib.lib().smart_n_reports(n_steps, gap, report_type))


def variant_classes(params, delta_sec, initial=None, area=None):
    """Which arithmetic variant of the fast kernels a parameter row needs -- the rules of wave_class() in
    csrc/smart_fast_model.h: 0 regular, 1 stiff (some k*3600 < dt: clamps / river rule reachable), 2 guarded
    (S outside [0, 0.5], C < 0 or Z <= 0), 3 ill-conditioned (dt / (RK*3600) > 2, the river: literal arithmetic) --
    and any row with a NaN or an infinite parameter, or a share or a residence time that is none (D or H outside [0, 1], T < 0, a k <= 0), for the
    literal arithmetic to decide what comes of it; likewise a
    row whose INITIAL states (initial [C, N, 12] or [N, 12], with the catchments' areas) hold a NaN, an infinity, a
    negative volume, or soil so far above its capacity that S * sum(levels) / Z starts beyond 0.5 or H * sum(levels) / Z
    beyond 1 (in any catchment)."""
    k = params[:, 6:10] * 3600.0
    cls = torch.zeros(params.shape[0], dtype=torch.int64, device=params.device)
    cls[~(k >= delta_sec).all(dim=1)] = 1
    cls[~((params[:, 4] >= 0.0) & (params[:, 4] <= 0.5) & (params[:, 1] >= 0.0) & (params[:, 5] > 0.0))] = 2
    cls[~(k[:, 3] >= 0.5 * delta_sec)] = 3
    cls[~torch.isfinite(params).all(dim=1)] = 3
    # shares that are none (D or H outside [0, 1], a negative T): negative inflows, the reference's clamps -- literal too
    cls[~((params[:, 3] >= 0.0) & (params[:, 3] <= 1.0) & (params[:, 2] >= 0.0) & (params[:, 2] <= 1.0) &
          (params[:, 0] >= 0.0))] = 3
    cls[~(params[:, 6:10] > 0.0).all(dim=1)] = 3      # ... and residence times that are none (k <= 0)
    cls[~(params[:, 5] > 0.0)] = 3                    # ... and a soil without capacity (Z <= 0: quotients by it)
    cls[~(params[:, 0] >= 0.2) | ~(params[:, 5] <= 1.0e3)] = 3   # ... and a discharge orders below the rain's
    cls[~(params[:, 5] >= 1.0)] = 3                   # ... and a soil of less than a millimetre
    if initial is not None:
        st = initial.reshape(-1, params.shape[0], 12)
        ar = torch.as_tensor(area, dtype=torch.float64, device=params.device).reshape(-1, 1)
        bad = (~torch.isfinite(st) | (st < 0.0)).any(dim=2)                  # (-0.0 is a zero like any other)
        lay = torch.zeros_like(st[:, :, 5])         # summed in wave_class()'s order: 0.0 + ly1 + ... + ly6, left to right
        for i in range(5, 11):
            lay = lay + st[:, :, i]
        fill = (lay / ar * 1e3) / params[:, 5].unsqueeze(0)                    # tot / Z of the first step
        s_init = params[:, 4].unsqueeze(0) * fill
        h_init = params[:, 2].unsqueeze(0) * fill     # the overland share of the first rainy step's excess: beyond one it
        bad = bad | ~(s_init <= 0.5) | ~(h_init <= 1.0)     # leaves the filling a negative excess (wave_class has the story)
        cls[bad.any(dim=0)] = 3
    return cls


def _variant_grouping(params, delta_sec, sort_rows=False, initial=None, area=None):
    """A wavefront runs ONE variant for its 64 lanes, the most general one any of its rows needs.  To keep a row's
    arithmetic (and cost) independent of its neighbours, rows are grouped by variant before the launch, each group
    padded to whole wavefronts with copies of its last row.  Returns (gather [N_run], inverse [N]) or None when the
    matrix needs no reordering (one variant only -- always the case for hourly steps with the default ranges).

    sort_rows: within a variant, order the rows so that the 64 samples of a wavefront behave alike -- by T (the
    rainfall correction factor) in 64 bins, then by S * Z.  Wet or dry is decided by the sign of rain * T - peva, so a
    wavefront whose rows have nearly the same T takes one side of that branch together on (almost) every step
    instead of executing both; and how far the rain excess of a step gets down the soil column depends on the deficit
    the leaks have left in the top layers, which goes with S * Z -- rows alike in it let the wave-uniform early exits
    of the filling cascade fire.  -2.5 % launch time at 1e5 samples, -5 % at 125,000, -8 % at 1e6
    (tools/debug/sort_rows.py).  Only asked for when no discharge matrix is stored -- its columns would have to be
    permuted back, which costs more than the launch gains; the per-sample results are permuted back on the way out."""
    cls = variant_classes(params, delta_sec, initial, area)
    mixed = int(cls.min()) != int(cls.max())
    if not mixed and not sort_rows:
        return None
    pieces = []
    for c in range(4):
        idx = torch.nonzero(cls == c)[:, 0] if mixed else (torch.arange(params.shape[0], device=params.device)
                                                            if c == int(cls[0]) else cls[:0])
        if idx.numel():
            if sort_rows:
                t, sz = params[idx, 0], params[idx, 4] * params[idx, 5]
                span = lambda v: (v - v.min()) / torch.clamp(v.max() - v.min(), min=1e-300)     # noqa: E731
                idx = idx[torch.argsort(torch.clamp((span(t) * 64).floor(), max=63) * 2.0 + span(sz), stable=True)]
            pad = (-idx.numel()) % 64 if mixed else 0
            pieces.append(torch.cat([idx, idx[-1:].expand(pad)]) if pad else idx)
    gather = torch.cat(pieces)
    inverse = torch.empty(params.shape[0], dtype=torch.int64, device=params.device)
    inverse[gather] = torch.arange(gather.numel(), device=params.device)    # duplicates hold identical results
    return gather, inverse


class _Memo(object):
    """What has been worked out about a (params, forcing) pair: the variant grouping of the rows and the launch plan
    (which kernels the rows and the forcing need, smart_plan_ensemble).  Entries are tied to the tensor OBJECTS the
    caller passed (weak references) and their in-place version counters, never to addresses: a fresh tensor that
    happens to reuse a freed address starts from nothing.  One list for the process, guarded by a lock (the C side
    is thread-aware as well: DeviceCtx::mu)."""
    _entries = []
    _lock = threading.Lock()
    SIZE = 8

    @classmethod
    def _matches(cls, ent, kind, tensors, key, refs):
        return (ent[0] == kind and ent[3] == key and len(refs) == len(tensors)
                and all(r is t for r, t in zip(refs, tensors)))

    @classmethod
    def lookup(cls, kind, tensors, key):
        with cls._lock:
            alive = []
            hit = None
            for ent in cls._entries:
                refs = [r() for r in ent[1]]
                if any(t is None for t in refs):
                    continue
                alive.append(ent)
                if cls._matches(ent, kind, tensors, key, refs) and ent[2] == tuple(t._version for t in tensors):
                    hit = ent
            cls._entries[:] = alive
            return None if hit is None else hit[4]

    @classmethod
    def store(cls, kind, tensors, key, value):
        with cls._lock:
            cls._entries.append((kind, [weakref.ref(t) for t in tensors], tuple(t._version for t in tensors), key,
                                 (value,)))
            del cls._entries[:-cls.SIZE]

    @classmethod
    def forget(cls, kind, tensors, key):
        """Drop what is remembered about these tensors: it turned out stale without a version bump (a write through
        ctypes, a foreign kernel, `.data`), and would be handed out again to the next prepare_ensemble()."""
        with cls._lock:
            cls._entries[:] = [ent for ent in cls._entries
                               if not cls._matches(ent, kind, tensors, key, [r() for r in ent[1]])]


class PreparedEnsemble(object):
    """One ensemble call, made ready: inputs on the device, output buffers, the workspace and the launch plan, all
    owned here -- launch() only enqueues the kernels (no allocation, no synchronisation), so a caller that repeats the
    call (a benchmark, a HIP-graph capture, a calibration loop over observation sets) pays for the set-up once.
    Build one with prepare_ensemble()."""

    repeated = False    # did the last verify() have to repeat the launch?

    def enqueue(self):
        """The kernels onto torch's current stream of the device, nothing else: no result object is built, nothing is
        permuted or copied (launch() = enqueue() + result())."""
        with torch.cuda.device(self.device):
            self._e.stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_lib.lib().smart_run_ensemble_hip(ctypes.byref(self._e)))

    def launch(self):
        """Enqueue on torch's current stream of the device.  Returns the EnsembleResult (views of this object's
        buffers: the next launch() overwrites them)."""
        self.enqueue()
        return self._result()

    def result(self):
        """The EnsembleResult of the last enqueue(): rows back in the caller's order (one permutation of the outputs
        when the rows were grouped; the stored discharge matrix included)."""
        return self._result()

    def status(self):
        """SMART_STATUS_* bits of the last launch (synchronises with the stream)."""
        word = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            self._e.stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_lib.lib().smart_launch_status(ctypes.byref(self._e), ctypes.byref(word)))
        return int(word.value)

    def describe(self):
        """The kernels launch() enqueues on this device, as text (smart_describe_launch)."""
        buf = ctypes.create_string_buffer(512)
        with torch.cuda.device(self.device):
            _lib.check(_lib.lib().smart_describe_launch(ctypes.byref(self._e), buf, len(buf)))
        return buf.value.decode()

    def verify(self):
        """Read the status word of the last launch and repair what it reports: a time slice that gave up waiting for
        its predecessor (never seen on an idle GPU; possible when the queue is preempted) -> the launch is repeated
        unsliced; a plan that no longer matches the inputs -> re-planned and repeated.  Raises if the second launch
        is not clean either.  Returns the result to use."""
        word = self.status()
        self.repeated = word != 0
        if word == 0:
            return self._result()
        warnings.warn("smartpy_amd: launch status %#x (%s); repeating the launch %s" % (
            word, ' + '.join(n for b, n in ((_lib.STATUS_SLICE_TIMEOUT, 'a time slice timed out'),
                                            (_lib.STATUS_STALE_PLAN, 'stale plan'),
                                            (_lib.STATUS_NONFINITE_FORCING, 'a NaN or an infinity in the forcing'))
                             if word & b),
            'in literal arithmetic' if word & _lib.STATUS_NONFINITE_FORCING else (
                'without time slices' if word & _lib.STATUS_SLICE_TIMEOUT else 'with a fresh plan')))
        if word & _lib.STATUS_NONFINITE_FORCING:
            # what the reference's branches make of a NaN only the literal kernel reproduces (forcing that came from the
            # host never gets here: prepare_ensemble looked at it)
            self._e.math_mode = MATH_LITERAL
        if word & _lib.STATUS_SLICE_TIMEOUT:
            self._e.time_slices = 1
        if word & _lib.STATUS_STALE_PLAN:
            ordered = self._e.plan & _lib.PLAN_ROWS_ORDERED
            self._e.plan = 0
            self._e.plan = self._make_plan() | ordered
            if self._memo_of is not None:      # what was remembered about these tensors is what went stale
                tensors, key = self._memo_of
                _Memo.forget('fast', tensors, key)
                _Memo.store('fast', tensors, key, (self._grouping, int(self._e.plan)))
        self.enqueue()
        word = self.status()
        if word != 0:
            raise SmartEngineError(-6, "smartpy_amd: the repeated launch reports status %#x as well" % word)
        return self._result()

    def _make_plan(self):
        plan = ctypes.c_int32(0)
        with torch.cuda.device(self.device):
            self._e.stream = torch.cuda.current_stream(self.device).cuda_stream
            _lib.check(_lib.lib().smart_plan_ensemble(ctypes.byref(self._e), ctypes.byref(plan)))
        return int(plan.value)

    def _result(self):
        dis, gw, objfn, fin = self._dis, self._gw, self._objfn, self._fin
        if self._grouping is not None:        # back to the caller's row order
            inverse = self._grouping[1]
            gw = gw[:, inverse]
            objfn = None if objfn is None else objfn[:, inverse]
            fin = None if fin is None else fin[:, inverse]
            if dis is not None:
                if self._caller_out is not None:
                    self._caller_out[:, :, :self.n_samples].copy_(torch.index_select(dis, 2, inverse))
                    dis = self._caller_out
                else:
                    dis = torch.index_select(dis, 2, inverse)
        return EnsembleResult(dis, gw, objfn, fin, self.n_samples, self._squeeze)


def prepare_ensemble(params, forcing, area_m2, delta_sec, n_warm, report_gap, report='summary', extra=None,
                     initial=None, obs=None, gw_obs=None, math_mode='fast', want_discharge=True, want_objfn=None,
                     want_final=False, device=None, discharge_out=None, group_variants=True, time_slices=0,
                     literal_form='auto'):
    """Everything of run_ensemble() short of the launch: see PreparedEnsemble.  Arguments as run_ensemble()."""
    L = _lib.lib()
    device = torch.device(device) if device is not None else (
        params.device if isinstance(params, torch.Tensor) and params.is_cuda else default_device())
    params_in, forcing_in = params, forcing
    params = as_device(params, device)
    forcing = as_device(forcing, device)
    squeeze = forcing.dim() == 2
    if squeeze:
        forcing = forcing.unsqueeze(0)
    C, T = forcing.shape[0], forcing.shape[1]
    if forcing.shape[2] != 2:
        raise Exception("forcing must be [T, 2] or [C, T, 2] (rain, peva)")
    if params.dim() == 2:
        pstride = 0
        N = params.shape[0]
    else:
        if params.shape[0] != C:
            raise Exception("params [C, N, 10] must have one block per catchment")
        N = params.shape[1]
        pstride = N * 10
    if params.shape[-1] != 10:
        raise Exception("params must have 10 columns (T, C, H, D, S, Z, SK, FK, GK, RK)")
    rtype = report_code(report)
    try:
        mmode = _MATH[math_mode]
    except KeyError:
        raise Exception("math mode '{}' unknown.".format(math_mode))
    # A NaN or an infinity in the forcing (a gap someone filled with 'nan'): the reference's branches see it -- a NaN
    # excess is "not wet", the evaporation cascade then empties all six layers (structure.py:359, :409-419) -- and only
    # the literal kernel takes those decisions; the fast kernels are compiled for numbers.  Forcing that arrives from
    # the host is looked at here; a caller who keeps it on the device says math_mode='literal' for such data.
    if mmode == MATH_FAST and not (isinstance(forcing_in, torch.Tensor) and forcing_in.is_cuda):
        if not bool(np.isfinite(np.asarray(forcing_in, dtype=np.float64)).all()):
            mmode = MATH_LITERAL
    R = n_reports(T, report_gap, rtype)

    area = as_device(np.full(C, area_m2, dtype=np.float64) if np.ndim(area_m2) == 0 else area_m2, device, (C,))
    if isinstance(extra, dict):
        extra = extra_vector(extra)
    if extra is not None:
        extra = as_device(extra, device)
        extra = extra.reshape(1, 7).expand(C, 7).contiguous() if extra.numel() == 7 else extra.reshape(C, 7)
    if initial is not None:
        initial = as_device(initial, device)
        initial = initial.reshape(1, N, 12).expand(C, N, 12).contiguous() if initial.numel() == N * 12 \
            else initial.reshape(C, N, 12)
    if want_objfn is None:
        want_objfn = obs is not None
    if obs is not None:
        obs = as_device(obs, device)
        obs = obs.reshape(1, R).expand(C, R).contiguous() if obs.numel() == R else obs.reshape(C, R)
    elif want_objfn:
        raise Exception("objective functions need observations")
    if gw_obs is not None:
        gw_obs = as_device(np.full(C, gw_obs, dtype=np.float64) if np.ndim(gw_obs) == 0 else gw_obs, device, (C,))

    # what is already known about these very tensors (only when the caller handed over device tensors: anything
    # converted above is a fresh object, and a fresh object is classified afresh)
    memo_on = [t for t in (params_in, forcing_in) if isinstance(t, torch.Tensor) and t.is_cuda]
    sort_rows = not want_discharge and discharge_out is None
    memo_key = (float(delta_sec), int(report_gap), rtype, C, N, T, bool(group_variants), sort_rows)
    if initial is not None:
        memo_on = []          # the rows' classes depend on the initial states as well: nothing is remembered across calls
    memo = _Memo.lookup('fast', memo_on, memo_key) if len(memo_on) == 2 and mmode == MATH_FAST else None

    # rows grouped by arithmetic variant (fast mode, one shared [N, 10] matrix spanning more than one wavefront)
    p = PreparedEnsemble()
    p.device, p.n_samples, p._squeeze = device, N, squeeze
    p._grouping, p._caller_out = None, None
    p._memo_of = (memo_on, memo_key) if len(memo_on) == 2 and mmode == MATH_FAST else None
    if group_variants and mmode == MATH_FAST and pstride == 0 and N > 64:
        p._grouping = memo[0][0] if memo else _variant_grouping(params, float(delta_sec), sort_rows, initial, area)
        if p._grouping is not None:
            gather = p._grouping[0]
            params = params[gather].contiguous()
            if initial is not None:
                initial = initial[:, gather].contiguous()
            N = gather.numel()
    if p._grouping is not None and discharge_out is not None:
        p._caller_out, discharge_out, want_discharge = discharge_out, None, True

    ld = N
    dis = None
    if discharge_out is not None:
        dis = discharge_out
        assert dis.is_contiguous() and dis.dtype == torch.float64 and dis.shape[:2] == (C, R) and dis.shape[2] >= N
        ld = dis.shape[2]
    elif want_discharge:
        dis = torch.empty((C, R, ld), dtype=torch.float64, device=device)
    p._dis = dis
    p._gw = torch.empty((C, N), dtype=torch.float64, device=device)
    p._objfn = torch.empty((C, N, 8), dtype=torch.float64, device=device) if want_objfn else None
    p._fin = torch.empty((C, N, 19), dtype=torch.float64, device=device) if want_final else None

    def ptr(t):
        return None if t is None else t.data_ptr()

    e = p._e = _lib.SmartEnsemble()
    e.n_catchments, e.n_samples, e.n_steps, e.n_warm, e.report_gap = C, N, T, int(n_warm), int(report_gap)
    e.report_type, e.math_mode, e.delta_sec = rtype, mmode, float(delta_sec)
    e.area_m2, e.forcing, e.params, e.params_catchment_stride = ptr(area), ptr(forcing), ptr(params), pstride
    e.extra, e.initial, e.obs, e.gw_obs = ptr(extra), ptr(initial), ptr(obs), ptr(gw_obs)
    e.discharge, e.discharge_ld, e.gw, e.objfn = ptr(dis), ld, ptr(p._gw), ptr(p._objfn)
    e.final_vars = ptr(p._fin)
    e.time_slices = int(time_slices)
    try:
        e.literal_form = _LITERAL_FORMS[literal_form]
    except KeyError:
        raise Exception("literal_form '{}' unknown ('auto', 'rows' or 'lanes').".format(literal_form))
    # the caller of the C ABI owns every buffer, the library's scratch included: header, observation statistics,
    # slice hand-over
    with torch.cuda.device(device):
        n_ws = int(L.smart_workspace_bytes(ctypes.byref(e)))
    p._ws = torch.empty((n_ws + 7) // 8, dtype=torch.float64, device=device) if n_ws > 0 else None
    e.workspace, e.workspace_bytes = ptr(p._ws), n_ws
    _lib.check(L.smart_check_ensemble(ctypes.byref(e)))
    if mmode == MATH_FAST and p._ws is not None:
        ordered = _lib.PLAN_ROWS_ORDERED if (sort_rows and p._grouping is not None) else 0
        if memo:
            e.plan = memo[0][1]
        elif torch.cuda.is_current_stream_capturing():
            e.plan = 0      # planning synchronises: inside a graph capture every kernel the call could need is launched
        else:
            e.plan = p._make_plan() | ordered
            if len(memo_on) == 2:
                _Memo.store('fast', memo_on, memo_key, (p._grouping, int(e.plan)))
    p._keep = (area, forcing, params, extra, initial, obs, gw_obs)     # alive for as long as the struct points at them
    return p


def run_ensemble(params, forcing, area_m2, delta_sec, n_warm, report_gap, report='summary', extra=None,
                 initial=None, obs=None, gw_obs=None, math_mode='fast', want_discharge=True, want_objfn=None,
                 want_final=False, device=None, discharge_out=None, group_variants=True, time_slices=0, verify=True,
                 literal_form='auto'):
    """One launch of the whole ensemble: the batched form of the spotpy loop over MonteCarlo.simulation /
    objectivefunction (montecarlo.py:153-154,179-209).

    params   [N, 10] or [C, N, 10]    forcing [T, 2] or [C, T, 2] (rain, peva per step)
    area_m2  scalar or [C]            extra   dict / 7-vector / [C, 7] / None
    initial  [N, 12] / [C, N, 12]     obs     [R] or [C, R], NaN = missing    gw_obs scalar / [C] / None

    verify: read the launch's status word afterwards (synchronises with the stream) and repeat the launch if a time
    slice timed out; skipped while the stream is being captured into a HIP graph.  Callers that pipeline launches
    use prepare_ensemble() / launch() and call verify() when they synchronise anyway.

    literal_form: how the rows that need the reference's own operation order (class 3: dt / RK > 2 -- a tenth of a daily
    ensemble -- or a parameter that is none) are laid over the wavefronts: 'rows' (one sample per DPP row, the latency
    form), 'lanes' (one per lane, the throughput form) or 'auto' (from the number of such blocks the plan counted).
    The same bits either way.
    """
    p = prepare_ensemble(params, forcing, area_m2, delta_sec, n_warm, report_gap, report=report, extra=extra,
                         initial=initial, obs=obs, gw_obs=gw_obs, math_mode=math_mode,
                         want_discharge=want_discharge, want_objfn=want_objfn, want_final=want_final, device=device,
                         discharge_out=discharge_out, group_variants=group_variants, time_slices=time_slices,
                         literal_form=literal_form)
    p.enqueue()
    if verify and p._ws is not None and p._e.math_mode == MATH_FAST and not torch.cuda.is_current_stream_capturing():
        out = p.verify()        # (builds the result once: after the status word has been read)
    else:
        out = p.result()
    out._prepared = p       # the result's tensors live in the prepared call's buffers
    return out


class SingleRun(object):
    """ONE parameter set at a time over a fixed forcing series -- SMART.simulate() (smart.py:154-210), the call a
    calibration loop written against the reference's per-sample protocol makes thousands of times
    (montecarlo.py:179-186) -- made ready once: the forcing on the device, a [1, 10] parameter buffer, the outputs, the
    workspace and what smart_plan_ensemble found out about the forcing all live here.  run(params) then copies 80
    bytes to the device, enqueues the one kernel the row's class needs, and brings [R] + 1 doubles back; nothing is
    allocated, the forcing is not touched, no planning kernel runs.  (Round 4 gave the smartcpp hook this treatment and
    left the package's own single-run entry stacking, uploading and planning on every call.)"""

    def __init__(self, forcing, area_m2, delta_sec, n_warm, report_gap, report='summary', extra=None, device=None,
                 math_mode='fast'):
        self.device = torch.device(device) if device is not None else default_device()
        self.delta_sec = float(delta_sec)
        self.params = torch.zeros((1, 10), dtype=torch.float64, device=self.device)
        self._host = torch.zeros((1, 10), dtype=torch.float64).pin_memory() if self.device.type == 'cuda' \
            else torch.zeros((1, 10), dtype=torch.float64)
        self.forcing = forcing if isinstance(forcing, torch.Tensor) and forcing.is_cuda else as_device(forcing, self.device)
        self._args = (area_m2, delta_sec, n_warm, report_gap)
        self._kw = dict(report=report, extra=extra, math_mode=math_mode, device=self.device)
        self._prep = None
        self._forcing_bits = 0

    def run(self, params):
        """params: the ten values (T, C, H, D, S, Z, SK, FK, GK, RK).  -> (discharge ndarray [R], gw float)."""
        global h2d_bytes
        row = np.ascontiguousarray(np.asarray(params, dtype=np.float64).reshape(1, 10))
        self._host.copy_(torch.from_numpy(row))
        self.params.copy_(self._host, non_blocking=True)        # the 80 bytes of this call
        h2d_bytes += 80
        if self._prep is None:
            # the first call plans: which kinds of forcing the series holds comes back once and is kept
            self._prep = prepare_ensemble(self.params, self.forcing, *self._args, **self._kw)
            self._forcing_bits = self._prep._e.plan & (_lib.PLAN_FORCING_PIECEWISE | _lib.PLAN_FORCING_VARYING |
                                                       _lib.PLAN_FORCING_RUNS)
        p = self._prep
        if p._e.math_mode == MATH_FAST and p._ws is not None:
            # the row's class on the host (ten numbers, the rules of wave_class): the plan names its kernel and no other
            cls = int(_lib.lib().smart_row_class(row.ctypes.data, self.delta_sec, None, 0.0))
            p._e.plan = _lib.PLAN_VALID | self._forcing_bits | _lib.PLAN_CLASS_BITS[cls]
            p.enqueue()
            out = p.verify()
        else:
            p.enqueue()
            out = p.result()
        return np.ascontiguousarray(out.discharge.cpu().numpy()[0]), float(out.gw.cpu().numpy()[0])


def objective_functions(discharge_report_major, obs, gw_sim=None, gw_obs=None):
    """montecarlo.py:193-209 for every column of a stored [R, N] discharge matrix (one pass over it, HBM-bound)."""
    L = _lib.lib()
    sim = discharge_report_major
    if not (isinstance(sim, torch.Tensor) and sim.is_cuda):
        sim = as_device(sim, default_device())
    if sim.stride(-1) != 1:
        sim = sim.contiguous()
    R, N = sim.shape
    ld = sim.stride(0)
    obs = as_device(obs, sim.device, (R,))
    gw_sim = as_device(gw_sim, sim.device, (N,)) if gw_sim is not None else None
    out = torch.empty((N, 8), dtype=torch.float64, device=sim.device)
    with torch.cuda.device(sim.device):
        _lib.check(L.smart_objfn_hip(N, R, sim.data_ptr(), ld, obs.data_ptr(),
                                     None if gw_sim is None else gw_sim.data_ptr(),
                                     float('nan') if gw_obs is None else float(gw_obs), out.data_ptr(),
                                     torch.cuda.current_stream(sim.device).cuda_stream))
    return out


def allsteps(area_m2, delta_sec, length_simu, nd_rain, nd_peva, nd_parameters, nd_initial, report_type, report_gap):
    """smartcpp.allsteps: same arguments and results as run_all_steps (structure.py:149-152,197); host arrays."""
    L = _lib.lib()
    rain = np.ascontiguousarray(nd_rain, dtype=np.float64)
    peva = np.ascontiguousarray(nd_peva, dtype=np.float64)
    par = np.ascontiguousarray(nd_parameters, dtype=np.float64)
    ini = np.ascontiguousarray(nd_initial, dtype=np.float64)
    length_simu, report_gap = int(length_simu), int(report_gap)
    if len(rain) < length_simu or len(peva) < length_simu or len(par) != 10 or len(ini) != 19:
        raise Exception("allsteps: inconsistent argument sizes")
    if report_type == REPORT_SUMMARY and report_gap > 0 and length_simu % report_gap:
        raise ValueError("cannot reshape array of size {} into shape ({})".format(length_simu, report_gap))
    R = n_reports(length_simu, report_gap, report_type)
    dis = np.empty(max(R, 0), dtype=np.float64)
    gw = ctypes.c_double(math.nan)
    fin = np.empty(19, dtype=np.float64)
    _lib.check(L.smart_allsteps_hip(float(area_m2), float(delta_sec), length_simu, rain.ctypes.data, peva.ctypes.data,
                                    par.ctypes.data, ini.ctypes.data, int(report_type), report_gap, dis.ctypes.data,
                                    ctypes.addressof(gw), fin.ctypes.data))
    return dis, gw.value, fin


def hook_counters():
    """{calls, allocations, forcing_bytes_uploaded, fast_calls, plans} of the smartcpp.allsteps stand-in since the library
    was loaded (smart_hook_counters)."""
    c = (ctypes.c_int64 * 5)()
    _lib.check(_lib.lib().smart_hook_counters(c, 5))
    return dict(zip(('calls', 'allocations', 'forcing_bytes_uploaded', 'fast_calls', 'plans'), (int(v) for v in c)))


def onestep(*args):
    """smartcpp.onestep: the 26 positional floats of run_one_step (structure.py:200-206) -> 19 floats."""
    if len(args) != 26:
        raise TypeError("onestep() takes exactly 26 positional arguments ({} given)".format(len(args)))
    return onestep_batch(np.asarray(args, dtype=np.float64).reshape(1, 26))[0]


def onestep_batch(inputs):
    """[n, 26] -> [n, 19]: n independent single steps in one launch."""
    L = _lib.lib()
    x = np.ascontiguousarray(inputs, dtype=np.float64)
    out = np.empty((x.shape[0], 19), dtype=np.float64)
    _lib.check(L.smart_onestep_hip(x.shape[0], x.ctypes.data, out.ctypes.data))
    return out


def river_step_batch(inputs):
    """[n, 4] (time_gap_sec, r_in_q_riv, r_p_rk [h], r_s_v_riv) -> [n, 2] (r_out_q_riv, r_s_v_riv): n independent
    calls of run_one_step_river (structure.py:461-503) in one launch."""
    L = _lib.lib()
    x = np.ascontiguousarray(inputs, dtype=np.float64)
    out = np.empty((x.shape[0], 2), dtype=np.float64)
    _lib.check(L.smart_river_step_hip(x.shape[0], x.ctypes.data, out.ctypes.data))
    return out
