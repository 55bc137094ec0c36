import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
import bench
from smartpy_amd import engine
from smartpy_amd.sampling import latin_hypercube_device
from smartpy_amd.parameters import Parameters
from oracle import smart_oracle as so, objfn_oracle
f = bench.synthetic_forcing(0, True)[0][:24 * 400]
T, W = f.shape[0], 24 * 40
obs = np.abs(np.random.default_rng(0).normal(2, 1, T // 24))
for n in (3000000, 4000000, 10000000):
    p = latin_hypercube_device(n, Parameters().ranges, seed=1)
    prep = engine.prepare_ensemble(p, f, bench.AREA, 3600.0, W, 24, extra=bench.EXTRA, obs=obs, gw_obs=0.12667, want_discharge=False)
    torch.cuda.synchronize(); t0 = time.perf_counter(); out = prep.launch(); torch.cuda.synchronize(); dt = time.perf_counter() - t0
    assert prep.status() == 0
    rows = np.array([0, n // 3, n - 1])
    pr = p[torch.from_numpy(rows).cuda()].cpu().numpy()
    dis, gw, _ = so.run_batch(bench.AREA, 3600.0, T, W, f[:, 0].copy(), f[:, 1].copy(), pr, bench.EXTRA, so.REPORT_SUMMARY, 24)
    want = objfn_oracle.objective_matrix(dis, obs, gw, 0.12667)
    got = out.objfn[torch.from_numpy(rows).cuda()].cpu().numpy()
    err = np.max(np.abs(got[:, :7] - want[:, :7]) / np.maximum(np.abs(want[:, :7]), 1e-12))
    print('N=%d: %s  %.1f ms  %.3g steps/s  workspace %.1f MB  max rel err of 3 rows %.2e' % (n, prep.describe(), dt * 1e3, n * (T + W) / dt, prep._e.workspace_bytes / 1e6, err), flush=True)
    del prep, out, p
    torch.cuda.empty_cache()
