#!/bin/bash
# usage: tools_prof.sh <tag> <workload> — run on the GPU box via gpurun.
# rocprofv3 kernel trace (+stats) and PMC passes for the bench command, restricted
# to the library's kernels; writes gpurun_out/prof_<tag>/summary.json (small).  (--no-wallclock: the bench's wall-clock leg
# starts the C++ driver as child processes, which the profiler would trace as well — one cold launch of every kernel each.)
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
rm -rf $OUT; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
F='--kernel-include-regex phy::'
timeout 300 rocprofv3 $F --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 5 --warmup 1 --cpu-sample 0 --no-profile --no-wallclock "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
cp $(find $OUT/trace -name "*kernel_stats.csv" | head -1) $OUT/kernel_stats.csv 2>/dev/null
i=0
for set in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_WAIT_ANY" \
           "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM SQ_INSTS_BRANCH" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum" \
           "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_64B_sum TCC_EA0_RDREQ_128B_sum" \
           "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum" \
           "FETCH_SIZE" "WRITE_SIZE"; do
  i=$((i+1))
  timeout 240 rocprofv3 $F --pmc $set -d $OUT/pmc_$i --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile --no-wallclock "$@" > $OUT/pmc${i}_bench.json 2> $OUT/pmc$i.err
done
python3 $ROOT/tools/tools_prof_summary.py $OUT > $OUT/summary.txt 2>&1; du -sh $OUT
