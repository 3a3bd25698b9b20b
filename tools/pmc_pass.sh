#!/bin/bash
# Run on the GPU box: tools/pmc_pass.sh "<COUNTER ...>" [bench args...] — one rocprofv3 --pmc pass (with a kernel trace,
# nothing else) over a short bench run; prints the per-dispatch mean of every counter for the stk:: kernels named in
# $KERNELS (default: the ECC iteration kernels).
counters=$1; shift
out=/tmp/pmc_$$
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc $counters --kernel-trace --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --no-cpu-baseline --host-fed-steps 0 --profile-launches 0 "$@" > $out.log 2>&1 || { echo "pass failed"; tail -5 $out.log; }
python3 - $out "${KERNELS:-ecc_iter}" <<'PY'
import collections, csv, glob, sys
agg = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(sys.argv[1] + '/*/*counter_collection.csv'):
    for r in csv.DictReader(open(f)):
        if sys.argv[2] in r['Kernel_Name']:
            k = (r['Kernel_Name'].split('(')[0].replace('void ', '')[:40], r['Counter_Name'])
            agg[k] += float(r['Counter_Value']); cnt[k] += 1
for k in sorted(agg):
    print(k[0], k[1], '%.4g' % (agg[k] / cnt[k]), 'n', cnt[k])
PY
rm -rf $out
