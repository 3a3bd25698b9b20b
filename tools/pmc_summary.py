"""Summarise a tools/profile_round.sh directory: per stk:: kernel calls and average duration (kernel trace),
FETCH_SIZE / WRITE_SIZE per dispatch (KiB as rocprofv3 reports them -> bytes) and the SQ issue / stall counters.

FETCH_SIZE on gfx950 counts the 128-byte read requests of a wide coalesced stream as 64 bytes (MI355X_MICROARCH.md, HBM):
the guide says to double it for 16-B-per-lane streams and to calibrate other access widths on a known byte count. The
calibration rows are in the same run: `scale_kernel` (float4 stream, 4 B read per float), `grey_blur_u8c3_kernel`
(aligned dword loads of a BGR byte stream: 3 B/px read, 4 B/px written) and `warp_accumulate_u8c3_kernel` (unaligned
8-byte gathers: ~3 B/px/frame). `_calibration` lists raw FETCH_SIZE / known bytes for each.
`_kernel_source_sha256` pins the summary to the sources it was measured with — argv[2] = ecc (default: the ECC kernels AND
the host schedule, slots per launch and workgroups per frame: kernels_ecc_col.hip, kernels_ecc_solve.hip, ecc_solve_body.h,
kernels_ecc.hip, stacker.cpp), keypoint (kernels_orb.hip, orb_device.h, kernels_orb_small.hip, keypoint.cpp, kernels_warp.hip, kernels_homography.hip) or hybrid (both
sets) — concatenated (bench.py ignores a stale one).
Round 4 adds, per kernel: SQ_INSTS_SALU / LDS / BRANCH / SMEM, the clock held (GRBM_GUI_ACTIVE / 8 / mean duration) and, for the
ECC pass, `_issue_slots_per_px`: the STATIC cost of the ring loop's row in issue slots (tools/isa_loops.py on the kernel's ISA,
built here with the Makefile's flags; a plain f32 / simple integer instruction = 1 slot = 2.15 SIMD cycles, packed / conversion /
three-operand / DPP ... = 2, v_rcp_f32 = 4: tools/valu_rates.hip) per row = per pixel of a lane, plus the strip's set-up and
fold spread over its 112 rows — what roofline.valu of the bench line is computed from."""
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
sq2 = ('SQ_INSTS_SALU', 'SQ_INSTS_LDS', 'SQ_INSTS_BRANCH', 'SQ_INSTS_SMEM', 'SQ_BUSY_CYCLES', 'GRBM_GUI_ACTIVE')
ta = ('TA_BUSY_avr', 'TCP_TOTAL_CACHE_ACCESSES_sum', 'TCP_TCC_READ_REQ_sum', 'TCP_PENDING_STALL_CYCLES_sum')
for name, keys in (('pmc_fetch', ('FETCH_SIZE',)), ('pmc_write', ('WRITE_SIZE',)), ('pmc_sq', sq), ('pmc_ta', ta), ('pmc_sq2', sq2)):
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
    if 'GRBM_GUI_ACTIVE_per_dispatch' in v and v.get('avg_us'):
        v['_clock_ghz_held'] = v['GRBM_GUI_ACTIVE_per_dispatch'] / 8.0 / (v['avg_us'] * 1e3)      # cycles per XCD / ns
out = dict(res)
# pins the summary to the sources it was measured with
ECC = ['kernels_ecc_col.hip', 'kernels_ecc_solve.hip', 'ecc_solve_body.h', 'kernels_ecc.hip', 'stacker.cpp']
KP = ['kernels_orb.hip', 'orb_device.h', 'kernels_orb_small.hip', 'keypoint.cpp', 'kernels_warp.hip', 'kernels_homography.hip']
pin = sys.argv[2] if len(sys.argv) > 2 else 'ecc'
files = {'ecc': ECC, 'keypoint': KP, 'hybrid': ECC + KP}[pin]
h = hashlib.sha256()
for name in files:
    h.update(open(os.path.join(root, 'libstacker_rs_amd', 'csrc', name), 'rb').read())
out['_kernel_source_sha256'] = h.hexdigest()
out['_kernel_source_files'] = files
# the ECC pass: static issue-slot cost of the ring loop (per pixel) from the ISA of THIS source
ek = next((k for k in out if k.startswith('stk::ecc_iter_col_kernel<3>')), None)
if ek:
    try:
        import re
        import subprocess
        import tempfile
        src = os.path.join(root, 'libstacker_rs_amd', 'csrc', 'kernels_ecc_col.hip')
        with tempfile.TemporaryDirectory() as td:
            asm = os.path.join(td, 'col.s')
            subprocess.run(['/opt/rocm/bin/hipcc', '-O3', '-std=c++17', '-fPIC', '--offload-arch=gfx950', '-ffp-contract=off', '-fno-slp-vectorize',
                            '-S', '--cuda-device-only', '-o', asm, src], check=True, capture_output=True, cwd=os.path.dirname(src))
            rep = subprocess.run([sys.executable, os.path.join(root, 'tools', 'isa_loops.py'), asm, 'ecc_iter_col_kernelILi3E'], capture_output=True, text=True).stdout
        # the ring loop is the inner loop that reads its taps from LDS: four rows x 7 LDS reads
        best = None
        for m in re.finditer(r'VALU (\d+) instructions = (\d+) units.*?SALU (\d+), LDS (\d+), VMEM (\d+)', rep):
            valu, units, salu, lds, vmem = (int(x) for x in m.groups())
            if 20 <= lds <= 40 and (best is None or units > best[1]):      # (the strip loop around it has all the kernel's LDS instructions)
                best = (valu, units, salu, lds, vmem)
        if best:
            rows = 4                                      # the loop body is four rows (template slots fixed at compile time)
            out[ek]['_ring_loop_static'] = {'rows': rows, 'valu_instructions': best[0], 'issue_slots': best[1], 'salu_static': best[2],
                                            'lds': best[3], 'vmem_static': best[4]}
            # + strip set-up (corner tests, ring fill) and the 66-sum lane fold, once per strip of ~112 rows: ~700 slots (isa_loops: the
            # depth-1 loop minus its inner loops), i.e. ~6 per row
            out[ek]['_issue_slots_per_px'] = best[1] / rows + 6.0      # per pixel = per lane and row, like SQ_INSTS_VALU x 64 / pixels
    except Exception as e:                                # no hipcc on this machine: the dynamic counters stand alone
        out[ek]['_issue_slots_note'] = 'static cost not computed: %s' % e
    if 'SQ_INSTS_VALU_per_dispatch' in out[ek] and 'SQ_WAVES_per_dispatch' in out[ek]:
        pass
print(json.dumps(out, indent=1, sort_keys=True))
