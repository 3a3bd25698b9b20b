#!/bin/bash
# Run on the GPU box (through gpurun) from the repo root:  tools/profile_round.sh <tag> [bench args...]
# Produces under gpurun_out/<tag>/: kernel-trace stats, one PMC pass for FETCH_SIZE, one for WRITE_SIZE (separate
# passes, as MI355X_MICROARCH.md prescribes: the two do not fit the TCC slots together), one for the SQ issue / stall
# counters, one for the vector-memory front end (TA busy, L1 tag accesses, L1->L2 requests), and a JSON summary per kernel. rocprofv3 gets the program itself after `--` (python3 bench.py ...).
tag=${1:-r03}; shift
args="--steps 2 --warmup 1 --no-cpu-baseline --host-fed-steps 0 $*"
out=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $GRAFT_REPO_ROOT/bench.py $args > $out/trace.log 2>&1 || echo "trace pass failed"
echo "trace done"
timeout -k 10 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $out/pmc_fetch -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_fetch.log 2>&1 || echo "fetch pass failed"
echo "fetch done"
timeout -k 10 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $out/pmc_write -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_write.log 2>&1 || echo "write pass failed"
echo "write done"
timeout -k 10 400 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_WAVES --kernel-trace --output-format csv -d $out/pmc_sq -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_sq.log 2>&1 || echo "sq pass failed"
echo "sq done"
timeout -k 10 400 rocprofv3 --pmc TA_BUSY_avr TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum --kernel-trace --output-format csv -d $out/pmc_ta -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_ta.log 2>&1 || echo "ta pass failed"
echo "ta done"
# round 4: the scalar unit, LDS and branch instruction counts (the ECC pass was co-limited by its scalar bookkeeping) and
# the clock the chip holds under the kernel (GRBM_GUI_ACTIVE / 8 XCDs / duration)
timeout -k 10 400 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_BRANCH SQ_INSTS_SMEM SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $out/pmc_sq2 -- python3 $GRAFT_REPO_ROOT/bench.py $args --profile-launches 0 > $out/pmc_sq2.log 2>&1 || echo "sq2 pass failed"
echo "sq2 done"
cd $GRAFT_REPO_ROOT && python3 tools/pmc_summary.py $out $PIN > $out/summary.json && head -c 1500 $out/summary.json
# keep what is judged (per-kernel stats + summary), drop the raw per-dispatch traces (tens of MB; gpurun_out is capped)
cp $out/trace/*/*kernel_stats.csv $out/kernel_stats.csv 2>/dev/null
rm -rf $out/trace $out/pmc_fetch $out/pmc_write $out/pmc_sq $out/pmc_ta $out/pmc_sq2
