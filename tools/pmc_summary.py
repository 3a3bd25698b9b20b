"""Summarise a tools/profile_round.sh directory: per stk:: kernel calls and average duration (kernel trace),
FETCH_SIZE / WRITE_SIZE per dispatch (KiB as rocprofv3 reports them -> bytes) and the SQ issue / stall counters.

FETCH_SIZE on gfx950 counts the 128-byte read requests of a wide coalesced stream as 64 bytes (MI355X_MICROARCH.md, HBM):
the guide says to double it for 16-B-per-lane streams and to calibrate other access widths on a known byte count. The
calibration rows are in the same run: `scale_kernel` (float4 stream, 4 B read per float), `grey_blur_u8c3_kernel`
(aligned dword loads of a BGR byte stream: 3 B/px read, 4 B/px written) and `warp_accumulate_u8c3_kernel` (unaligned
8-byte gathers: ~3 B/px/frame). `_calibration` lists raw FETCH_SIZE / known bytes for each.
`_kernel_source_sha256` pins the summary to the ECC sources it was measured with (kernels AND the host schedule: slots per launch, workgroups per frame) — kernels_ecc_col.hip, kernels_ecc_solve.hip, kernels_ecc.hip, stacker.cpp,
ecc_solve_body.h, concatenated — (bench.py ignores a stale one)."""
import collections
import csv
import glob
import hashlib
import json
import os
import sys

d = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
res = collections.defaultdict(dict)
for f in glob.glob(d + '/trace/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'stk::' in r['Name']:
            k = r['Name'].split('(')[0].replace('void ', '')
            res[k].update(calls=int(r['Calls']), avg_us=float(r['AverageNs']) / 1e3, total_ms=float(r['TotalDurationNs']) / 1e6)
sq = ('SQ_WAVE_CYCLES', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_VALU', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_INSTS_VALU',
      'SQ_INSTS_VMEM_RD', 'SQ_WAVES')
ta = ('TA_BUSY_avr', 'TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_PENDING_STALL_CYCLES_sum')
for name, keys in (('pmc_fetch', ('FETCH_SIZE',)), ('pmc_write', ('WRITE_SIZE',)), ('pmc_sq', sq), ('pmc_ta', ta)):
    agg = collections.defaultdict(float)
    cnt = collections.Counter()
    for f in glob.glob(d + '/' + name + '/*/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] not in keys or 'stk::' not in r['Kernel_Name']:
                continue
            k = r['Kernel_Name'].split('(')[0].replace('void ', '')
            agg[(k, r['Counter_Name'])] += float(r['Counter_Value'])
            cnt[(k, r['Counter_Name'])] += 1
    for (k, c), v in agg.items():
        if c in ('FETCH_SIZE', 'WRITE_SIZE'):
            res[k][c + '_bytes_per_dispatch'] = v / cnt[(k, c)] * 1024.0
            res[k][c + '_dispatches'] = cnt[(k, c)]
        else:
            res[k][c + '_per_dispatch'] = v / cnt[(k, c)]
for k, v in res.items():
    if 'FETCH_SIZE_bytes_per_dispatch' in v:
        v['hbm_read_bytes_x2_corrected'] = 2 * v['FETCH_SIZE_bytes_per_dispatch']
        v['hbm_bytes_per_dispatch_corrected'] = v['hbm_read_bytes_x2_corrected'] + v.get('WRITE_SIZE_bytes_per_dispatch', 0.0)
    if 'SQ_WAVE_CYCLES_per_dispatch' in v and v['SQ_WAVE_CYCLES_per_dispatch'] > 0:
        wc = v['SQ_WAVE_CYCLES_per_dispatch']
        v['valu_issue_frac_of_wave_cycles'] = v.get('SQ_ACTIVE_INST_VALU_per_dispatch', 0.0) / wc
        v['parked_frac_of_wave_cycles'] = v.get('SQ_WAIT_ANY_per_dispatch', 0.0) / wc
        v['issue_stall_frac_of_wave_cycles'] = v.get('SQ_WAIT_INST_ANY_per_dispatch', 0.0) / wc
out = dict(res)
# pins the summary to the ECC sources it was measured with: iteration pass, solve / init kernels, solve routine
h = hashlib.sha256()
for name in ('kernels_ecc_col.hip', 'kernels_ecc_solve.hip', 'ecc_solve_body.h', 'kernels_ecc.hip', 'stacker.cpp'):
    h.update(open(os.path.join(root, 'libstacker_rs_amd', 'csrc', name), 'rb').read())
out['_kernel_source_sha256'] = h.hexdigest()
out['_kernel_source_files'] = ['kernels_ecc_col.hip', 'kernels_ecc_solve.hip', 'ecc_solve_body.h', 'kernels_ecc.hip', 'stacker.cpp']
print(json.dumps(out, indent=1, sort_keys=True))
