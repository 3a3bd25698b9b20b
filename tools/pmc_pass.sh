#!/bin/bash
# usage: tools_pmc.sh <outdir> <counters...> -- bench args
set -e
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/$out -- python3 $GRAFT_REPO_ROOT/bench.py "$@" > $GRAFT_REPO_ROOT/gpurun_out/$out.log 2>&1
