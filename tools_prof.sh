#!/bin/bash
# usage: tools_prof.sh <tag> <workload> [extra bench args] — run on the GPU box via gpurun
# Collects: kernel-trace stats, then PMC passes (SQ, TCC) for the bench command.
TAG=$1; WL=$2; shift 2
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT/trace --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 3 --warmup 1 --cpu-sample 0 "$@" > $OUT/trace_bench.json 2> $OUT/trace.err
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_ACTIVE_INST_ANY -d $OUT/pmc_sq --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile "$@" > $OUT/pmc_sq_bench.json 2> $OUT/pmc_sq.err
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_BRANCH -d $OUT/pmc_sq2 --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile "$@" > $OUT/pmc_sq2_bench.json 2> $OUT/pmc_sq2.err
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum -d $OUT/pmc_tcc --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile "$@" > $OUT/pmc_tcc_bench.json 2> $OUT/pmc_tcc.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile "$@" > $OUT/pmc_fetch_bench.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write --output-format csv -- python3 $ROOT/bench.py --workload $WL --steps 1 --warmup 0 --cpu-sample 0 --no-profile "$@" > $OUT/pmc_write_bench.json 2> $OUT/pmc_write.err
python3 $ROOT/tools_prof_summary.py $OUT > $OUT/summary.txt 2>&1; du -sh $OUT
