#!/bin/bash
# Run on the GPU box: tools/kernel_time.sh <lib.so> [bench args] — mean duration of every stk:: kernel of one bench run
# (rocprofv3 kernel trace), also when the run itself ends in an error (ablation builds that break convergence).
lib=$1; shift
out=/tmp/kt_$$
cd /tmp && export TMPDIR=/tmp
STACKER_AMD_LIB=$GRAFT_REPO_ROOT/$lib timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --host-fed-steps 0 --profile-launches 0 "$@" > $out.log 2>&1
python3 - $out <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + '/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'ecc_iter' in r['Name'] or 'warp_acc' in r['Name']:
            print(r['Name'].split('(')[0][:60], 'calls', r['Calls'], 'avg_us', round(float(r['AverageNs']) / 1e3, 1))
PY
