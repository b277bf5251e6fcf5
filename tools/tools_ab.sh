#!/bin/bash
# usage (on the GPU box): tools_ab.sh "<bench args>" name1 name2 ... — dev: the bench's phase-A kernel times with
# each variant library of build/ab/, interleaved twice on the same box
ARGS=$1; shift
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for n in "$@"; do
    PHYLONIUM_AMD_LIB=$ROOT/build/ab/lib$n.so python $ROOT/bench.py --cpu-sample 0 $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$n', d['ms_per_step'], d.get('ms_per_step_noprofile'), {k:v['avg_ms'] for k,v in d['kernels'].items() if k.startswith('anchor')})"
  done
done
