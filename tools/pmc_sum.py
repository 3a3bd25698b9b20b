import csv, sys, glob, collections
d = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for f in glob.glob(d + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].split('(')[0][:60]
        agg[k][r['Counter_Name']] += float(r['Counter_Value']); 
        cnt[(k, r['Counter_Name'])] += 1
for k in agg:
    if 'stk::' not in k: continue
    print(k)
    for c, v in agg[k].items(): print('   %-28s total %.4g  per-dispatch %.4g  (n=%d)' % (c, v, v / cnt[(k, c)], cnt[(k, c)]))
