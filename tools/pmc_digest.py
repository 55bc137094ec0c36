"""Counters of a rocprofv3 --pmc run, per time-loop kernel: mean per dispatch of every counter in <dir>/**/*counter_collection.csv
(dispatches of a few wavefronts -- warm-ups on small inputs -- left out).  usage: pmc_digest.py <dir>"""
import csv, glob, sys, collections
rows = collections.defaultdict(lambda: collections.defaultdict(dict))
for f in glob.glob(sys.argv[1] + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0]
        if not k.startswith('smart::smart_fast'):
            continue
        d = rows[k][r['Dispatch_Id']]
        d[r['Counter_Name']] = d.get(r['Counter_Name'], 0.0) + float(r['Counter_Value'])
for k, disp in rows.items():
    big = max(d.get('SQ_WAVES', 0.0) for d in disp.values())
    keep = [d for d in disp.values() if d.get('SQ_WAVES', 0.0) >= 0.5 * big]
    names = sorted({n for d in keep for n in d})
    mean = {n: sum(d.get(n, 0.0) for d in keep) / len(keep) for n in names}
    print('  %s: %d dispatches; ' % (k, len(keep)) + ', '.join('%s %.4g' % (n, mean[n]) for n in names))
    if mean.get('SQC_ICACHE_REQ'):
        print('    misses / requests = %.4f (duplicates %.4f)' % (mean.get('SQC_ICACHE_MISSES', 0.0) / mean['SQC_ICACHE_REQ'],
                                                                 mean.get('SQC_ICACHE_MISSES_DUPLICATE', 0.0) / mean['SQC_ICACHE_REQ']))
