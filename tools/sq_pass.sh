#!/bin/bash
# Run on the GPU box: tools/sq_pass.sh <tag> [bench args...] — one rocprofv3 pass with the SQ issue / stall counters and a
# kernel trace of the same command, summarised per kernel (tools/pmc_summary.py).
tag=${1:-sq}; shift
args="--steps 2 --warmup 1 --no-cpu-baseline --host-fed-steps 0 --profile-launches 0 $*"
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/trace.log 2>&1 || echo "trace pass failed"
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/pmc_sq.log 2>&1 || echo "sq pass failed"
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out $PIN > $out/summary.json
cp $out/trace/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/trace $out/pmc_sq
