#!/bin/bash
# usage: tools_pmc.sh <tag> <workload> "<counters pass 1>" "<counters pass 2>" ... — run on the GPU box via gpurun
# Counters are collected only for the library's kernels (--kernel-include-regex).
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/pmc_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 240 rocprofv3 --kernel-include-regex "phy::" --pmc $set -d $OUT/pmc_$i --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile > $OUT/p${i}_bench.json 2> $OUT/p$i.err
done
python3 $ROOT/tools/tools_prof_summary.py $OUT > $OUT/summary.txt 2>&1
