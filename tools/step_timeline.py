"""Timeline of the LAST step of a rocprofv3 --kernel-trace run: kernels in start order with the idle gaps between them.
usage: python tools/step_timeline.py <dir with *kernel_trace.csv> <first kernel of a step, e.g. grey_u8x4>"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'stk::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if sys.argv[2] in r['Kernel_Name']]
seq = rows[idx[-1]:]
t0 = int(seq[0]['Start_Timestamp']); prev = t0; busy = 0; gaps = []
for r in seq:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    if s - prev > 15000: gaps.append(((prev - t0) / 1e3, (s - prev) / 1e3, r['Kernel_Name'].split('(')[0][-40:]))
    busy += e - max(s, prev) if e > prev else 0
    prev = max(prev, e)
tot = (prev - t0) / 1e3
print(f"step {tot:.0f} us, kernels busy {busy / 1e3:.0f} us ({100 * busy / 1e3 / tot:.0f} %), idle {tot - busy / 1e3:.0f} us")
for at, g, nxt in gaps: print(f"  gap {g:7.0f} us at {at:7.0f} us before {nxt}")
