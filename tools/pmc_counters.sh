#!/bin/bash
# Run on the GPU box: tools/pmc_counters.sh <tag> "<COUNTER ...>" <python script and args> — one rocprofv3 counter pass
# (own run, kernel trace only beside it), per-kernel means printed and kept under gpurun_out/<tag>.txt
tag=$1; ctrs=$2; shift 2
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --pmc $ctrs --kernel-trace --output-format csv -d $out/pmc -- python3 "$@" > $out/run.log 2>&1 || echo "pmc pass failed"
python3 - $out <<'PY' | tee $out.txt
import collections, csv, glob, sys
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/pmc/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if 'stk::' not in r['Kernel_Name']: continue
        k = (r['Kernel_Name'].split('(')[0].replace('void ', '')[:70], r['Counter_Name'])
        agg[k] += float(r['Counter_Value']); cnt[k] += 1
for k in sorted(agg): print(f"{k[0]:70s} {k[1]:28s} {agg[k] / cnt[k]:16.1f} per dispatch ({cnt[k]} dispatches)")
PY
rm -rf $out/pmc
