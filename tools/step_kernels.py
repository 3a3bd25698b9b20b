"""Every kernel of the LAST step of a rocprofv3 --kernel-trace run, in start order: start offset, duration, gap before it.
usage: python tools/step_kernels.py <dir with *kernel_trace.csv> <first kernel of a step, e.g. grey_blur_u8c3> [min_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'stk::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
skip = int(sys.argv[4]) if len(sys.argv) > 4 else 0          # take the (skip+1)-th last occurrence
seq = rows[idx[-1 - skip]:(idx[-skip] if skip else None)]
min_us = float(sys.argv[3]) if len(sys.argv) > 3 else 0
t0 = int(seq[0]['Start_Timestamp']); prev = t0
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    name = r['Kernel_Name'].split('(')[0].replace('void ', '').replace('stk::', '')[:48]
    if (e - s) / 1e3 >= min_us:
        print(f"{(s - t0) / 1e3:9.1f} us  +{(e - s) / 1e3:8.1f} us  gap {(s - prev) / 1e3:7.1f}  grid {r.get('Grid_Size_X', r.get('Grid_Size', '?')):>8}  {name}")
    prev = max(prev, e)
print(f"step total {(prev - t0) / 1e3:.1f} us")
