#!/bin/bash
# Run on the GPU box: tools/ab_bench.sh "<lib or ''> ..." [bench args] — the same bench command with each library in turn,
# twice round robin, one line per run (ms per step, frames/s, mean ECC launch): boxes differ by a few per cent, so only
# figures from one call compare.
libs=$1; shift
for r in 1 2; do
  for l in $libs; do
    if [ "$l" = "default" ]; then unset STACKER_AMD_LIB; else export STACKER_AMD_LIB=$GRAFT_REPO_ROOT/libstacker_rs_amd/ab/lib$l.so; fi
    python3 bench.py --no-cpu-baseline --host-fed-steps 0 "$@" 2>/dev/null | python3 -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get('roofline', {})
ks = {k['kernel'][:12]: k['ms_per_step'] for k in d.get('kernels', [])}
print('$l', 'ms_per_step', d['ms_per_step'], 'value', d['value'], 'ecc_launch_ms', r.get('avg_launch_ms'), 'align', d['stages'].get('align_ms_per_step'), 'fold', d['stages'].get('warp_ms_per_step'), 'kernels', ks)"
  done
done
