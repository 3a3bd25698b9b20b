import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = [r for r in csv.DictReader(open(f)) if 'stk::' in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'grey_u8x4' in r['Kernel_Name']]
# steps begin where a grey kernel follows a scale/warp kernel; take the last full step: find the last group of 3 grey launches
starts=[i for i in idx]
# group greys that are within 300us
groups=[]
for i in starts:
    t=int(rows[i]['Start_Timestamp'])
    if groups and t-groups[-1][-1][1] < 400000: groups[-1].append((i,t))
    else: groups.append([(i,t)])
g=groups[-2] if len(groups)>1 else groups[-1]
i0=g[0][0]; i1=groups[-1][0][0] if len(groups)>1 else len(rows)
t0=int(rows[i0]['Start_Timestamp'])
for r in rows[i0:i1]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    name=r['Kernel_Name'].split('(')[0].replace('stk::','').replace('void ','')[-34:]
    print(f"{(s-t0)/1e3:8.1f} {(e-s)/1e3:7.1f}  q{r.get('Queue_Id','?'):>3} {name}")
