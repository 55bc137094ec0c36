"""Repeat the time-sliced launch and compare every result with the unsliced one bit for bit.  The output buffers are
poisoned before every launch (a row that was not written cannot pass for the previous launch's) and the status word
must read 0 every time.  With `busy` as third argument a second stream keeps the GPU busy with unrelated kernels
(matrix products of varying size) while the sliced launches run: workgroups of the ensemble kernel then start in
whatever order and on whatever CUs the competing work leaves free -- the situation the ticket scheme is for.

With `flat` the forcing varies inside the day: the step loop with deferred evaporation, whose pending demand travels
in the hand-over.

With `runs` the forcing is constant over runs of six steps: the run engine.

    python tools/debug/time_slices_stress.py <n_samples> <launches> [busy] [flat | runs]
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
params = torch.as_tensor(sampling.latin_hypercube(n, Parameters().ranges, seed=11), device=dev)
kw = dict(extra=bench.EXTRA, obs=obs, gw_obs=0.12667, want_discharge=False)
plain = engine.prepare_ensemble(params, ft, bench.AREA, 3600.0, W, 24, time_slices=1, **kw)
ref = plain.launch()
ref = (ref.gw.clone(), ref.objfn.clone())
assert plain.status() == 0
p = engine.prepare_ensemble(params, ft, bench.AREA, 3600.0, W, 24, **kw)
side = torch.cuda.Stream()
mats = [torch.randn(m, m, device=dev) for m in (512, 1024, 2048, 3072)]
bad = timeouts = 0
torch.cuda.synchronize()
t0 = time.perf_counter()
for i in range(reps):
    p._gw.fill_(float('nan'))
    p._objfn.fill_(float('nan'))
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
      % (n, (' sub-daily forcing' if flat else '') + (' 6-hourly forcing' if runs else '') + (' + competing stream' if busy else ''), p.describe(), reps, (time.perf_counter() - t0) / reps * 1e3, bad,
         timeouts))
