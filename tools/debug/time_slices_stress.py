"""Repeat the time-sliced launch and compare every result with the unsliced one bit for bit.  The output buffers are
poisoned before every launch (a row that was not written cannot pass for the previous launch's) and the status word
must read 0 every time.  With `busy` as third argument a second stream keeps the GPU busy with unrelated kernels
(matrix products of varying size) while the sliced launches run: workgroups of the ensemble kernel then start in
whatever order and on whatever CUs the competing work leaves free -- the situation the ticket scheme is for.

With `flat` the forcing varies inside the day: the step loop with deferred evaporation, whose pending demand travels
in the hand-over.

With `runs` the forcing is constant over runs of six steps: the run engine.

Round 4: `raw` -- report='raw' (smart_fast_intervals_raw, or with `flat` smart_fast_steps_raw); `every` -- a report
every step (smart_fast_steps_every); `half` -- instead of bursts of matrix products, ONE long-running kernel of 128
workgroups beside every launch (a row-wise cumulative sum: each row one workgroup for milliseconds), which keeps about
half the CUs away from the ensemble for the whole launch: uneven load, slices of a block start on whatever CU comes
free.  The hand-over buffers are the prepared call's own, launch after launch at the same addresses: a consumer slice
that lands on a CU which read that block's hand-over in an earlier launch finds its L1 WARM with that launch's lines --
the case MI355X_MICROARCH.md says hides every missing acquire when left out of a test.

    python tools/debug/time_slices_stress.py <n_samples> <launches> [busy | half] [flat | runs] [raw | every]
"""
import sys, time
sys.path.insert(0, '.')
import torch
import bench
from smartpy_amd import engine, sampling
from smartpy_amd.parameters import Parameters
dev = torch.device('cuda:0')
forcing = bench.synthetic_forcing(0, True)[0]
flat = 'flat' in sys.argv[3:]
runs = 'runs' in sys.argv[3:]
if flat:
    forcing = bench.hourly_varying_forcing(forcing)
if runs:
    forcing = bench.six_hourly_forcing(forcing)
T, W = forcing.shape[0], 8760
ft = torch.as_tensor(forcing, device=dev)
obs = torch.rand(T // 24, dtype=torch.float64, device=dev) + 0.5
n, reps = int(sys.argv[1]), int(sys.argv[2])
busy = 'busy' in sys.argv[3:]
half = 'half' in sys.argv[3:]
raw = 'raw' in sys.argv[3:]
every = 'every' in sys.argv[3:]
gap = 1 if every else 24
if every:
    obs = torch.rand(T, dtype=torch.float64, device=dev) + 0.5
params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=11), device=dev)
kw = dict(extra=bench.EXTRA, obs=obs, gw_obs=0.12667, want_discharge=False, report='raw' if raw else 'summary')
plain = engine.prepare_ensemble(params, ft, bench.AREA, 3600.0, W, gap, time_slices=1, **kw)
ref = plain.launch()
ref = (ref.gw.clone(), ref.objfn.clone())
assert plain.status() == 0
p = engine.prepare_ensemble(params, ft, bench.AREA, 3600.0, W, gap, **kw)
side = torch.cuda.Stream()
mats = [torch.randn(m, m, device=dev) for m in (512, 1024, 2048, 3072)]
rows = torch.rand(128, 3000000, dtype=torch.float64, device=dev) if half else None
bad = timeouts = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(reps):
    p._gw.fill_(float('nan'))
    p._objfn.fill_(float('nan'))
    if half:
        with torch.cuda.stream(side):
            torch.cumsum(rows, dim=1)
    if busy:
        with torch.cuda.stream(side):
            for j in range(4):
                m = mats[(i + j) % 4]
                torch.mm(m, m)
    out = p.launch()
    word = p.status()
    timeouts += word != 0
    if not (torch.equal(out.gw.view(torch.int64), ref[0].view(torch.int64))
            and torch.equal(out.objfn.view(torch.int64), ref[1].view(torch.int64))):
        bad += 1
torch.cuda.synchronize()
print('N=%d%s: %s; %d launches, %.2f ms each incl. poison + compare, %d differ, %d with a non-zero status word'
      % (n, (' sub-daily forcing' if flat else '') + (' 6-hourly forcing' if runs else '') + (' raw reports' if raw else '') + (' a report every step' if every else '') + (' + competing stream' if busy else '') + (' + half the CUs taken' if half else ''), p.describe(), reps, (time.perf_counter() - t0) / reps * 1e3, bad,
         timeouts))
