#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  tools/profile_round.sh r01 [bench args...]
# Produces under gpurun_out/<tag>/: kernel-trace stats, one PMC pass for FETCH_SIZE, one for WRITE_SIZE
# (separate passes, as MI355X_MICROARCH.md prescribes), and a JSON summary per kernel.
tag=${1:-r01}; shift
args="--steps 2 --warmup 1 --no-cpu-baseline $*"
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/trace.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_fetch.log 2>&1 || echo "fetch pass failed"
timeout -k 10 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_write.log 2>&1 || echo "write pass failed"
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out > $out/summary.json && cat $out/summary.json | head -60
