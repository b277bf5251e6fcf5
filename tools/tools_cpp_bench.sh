#!/bin/bash
# The C++ host's step (phylonium-amd --bench-steps, csrc/group.hip: a rank's pass as one queue) on a synthetic workload,
# one context and N ranks sharing this box's GPU(s):  gpurun -- 'bash tools/tools_cpp_bench.sh c4 "2 8" out.txt'
WL=${1:-c4}; RANKS=${2:-"2 8"}; OUT=${3:-gpurun_out/cpp_bench_$WL.txt}; STEPS=${4:-10}
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
N=$(python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.WORKLOADS['$WL'][0])")
D=/tmp/wallclock_${WL}_$N
mkdir -p $D
python3 $ROOT/tools/tools_wallclock.py --workload $WL --prepare $D/meta.json > /dev/null 2>&1
: > $OUT
( cd $D && $ROOT/phylonium_amd/phylonium-amd --bench-steps $STEPS -r g0000.fasta g*.fasta 2>&1 > /dev/null | grep "bench-steps" >> $ROOT/$OUT )
for n in $RANKS; do
  sleep 2
  ( cd $D && $ROOT/phylonium_amd/phylonium-amd --gpus $n --bench-steps $STEPS --timing -r g0000.fasta g*.fasta 2>&1 > /dev/null | grep "bench-steps\|ranks over" >> $ROOT/$OUT )
done
cat $OUT
