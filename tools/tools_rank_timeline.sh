#!/bin/bash
# usage (GPU box): tools_rank_timeline.sh <workload> <R/N> <out.txt> [bench args] — dev: one step of rank R of N (emulated on
# this GPU, with the exchange) as a timeline of kernels and copies (rocprofv3 kernel + memory-copy trace, no counters)
WL=$1; RN=$2; OUT=$3; shift 3
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
D=$ROOT/gpurun_out/tl_tmp_$$
cd /tmp && export TMPDIR=/tmp
rm -rf $D
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace -d $D --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 4 --warmup 2 \
  --cpu-sample 0 --no-profile --no-wallclock --emulate-rank $RN --emulate-exchange "$@" > /dev/null 2> $D.err
python3 $ROOT/tools/tools_timeline.py $D > $OUT
rm -rf $D
