"""profiles/<name>.json (written by tools/summarize_profile.py) -> profiles/traffic_latest.json, the PMC figures
bench.py quotes per launch: HBM bytes (FETCH_SIZE corrected as MI355X_MICROARCH.md prescribes + WRITE_SIZE) and the
executed vector-ALU instruction count of the dominant kernel."""
import json, sys
src = sys.argv[1] if len(sys.argv) > 1 else 'profiles/r01_final.json'
d = json.load(open(src))
k = d['pmc'].get('smart::smart_ensemble_fast', {})
out = {
    'hbm_bytes_per_launch': d['hbm_bytes_per_launch'],
    'hbm_bytes_per_launch_uncorrected': d['hbm_bytes_per_launch_uncorrected'],
    'fetch_bytes_raw': d['hbm_read_bytes_raw'], 'write_bytes': d['hbm_write_bytes'],
    'valu_insts_per_launch': k.get('SQ_INSTS_VALU'),
    'salu_insts_per_launch': k.get('SQ_INSTS_SALU'),
    'source': src.replace('.json', '.md') + ': rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes) on '
              '`python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline`; FETCH_SIZE doubled per MI355X_MICROARCH.md '
              '(gfx950 counts 128-B requests as 64 B), an upper bound here since the reads are scalar loads and '
              '8-B/lane rows; SQ_INSTS_VALU from its own pass',
    'kernel': 'smart_ensemble_fast', 'avg_ms_kernel_trace': d['full_size_dispatch_ms']['avg'],
}
json.dump(out, open('profiles/traffic_latest.json', 'w'), indent=1)
print(out)
