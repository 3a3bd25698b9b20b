"""Summarise a tools/profile_round.sh directory: per stk:: kernel calls, avg duration (kernel trace), and
FETCH_SIZE / WRITE_SIZE per dispatch (KiB as rocprofv3 reports them -> bytes). FETCH_SIZE on gfx950 counts
128-byte read requests as 64 bytes for wide coalesced streams (MI355X_MICROARCH.md §HBM): both the raw and
the x2-corrected figure are listed; the elementwise scale_kernel (known 4 B read + 4 B write per float) in the
same run serves as the calibration row."""
import csv, glob, json, sys, collections
d = sys.argv[1]
res = collections.defaultdict(dict)
for f in glob.glob(d + '/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'stk::' in r['Name']:
            k = r['Name'].split('(')[0].replace('void ', '')
            res[k].update(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3, total_ms=float(r['TotalDurationNs']) / 1e6)
for name, key in (('pmc_fetch', 'FETCH_SIZE'), ('pmc_write', 'WRITE_SIZE')):
    agg = collections.defaultdict(float); cnt = collections.Counter()
    for f in glob.glob(d + '/' + name + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != key or 'stk::' not in r['Kernel_Name']: continue
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[k] += float(r['Counter_Value']); cnt[k] += 1
    for k in agg:
        res[k][key + '_bytes_per_dispatch'] = agg[k] / cnt[k] * 1024.0
        res[k][key + '_dispatches'] = cnt[k]
for k, v in res.items():
    if 'FETCH_SIZE_bytes_per_dispatch' in v:
        v['hbm_read_bytes_x2_corrected'] = 2 * v['FETCH_SIZE_bytes_per_dispatch']
        v['hbm_bytes_per_dispatch_corrected'] = v['hbm_read_bytes_x2_corrected'] + v.get('WRITE_SIZE_bytes_per_dispatch', 0.0)
print(json.dumps(res, indent=1, sort_keys=True))
