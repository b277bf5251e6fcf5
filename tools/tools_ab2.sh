#!/bin/bash
# usage (GPU box): tools_ab2.sh "<bench args>" "<kernel names, comma separated>" name1 name2 ... — dev: the named kernels' times with
# each variant library of build/ab/, interleaved twice on the same box
ARGS=$1; KS=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
for rep in 1 2; do
  for n in "$@"; do
    PHYLONIUM_AMD_LIB=$ROOT/build/ab/lib$n.so python $ROOT/bench.py --cpu-sample 0 --no-wallclock $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); ks='$KS'.split(','); print('$n', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ks})"
  done
done
