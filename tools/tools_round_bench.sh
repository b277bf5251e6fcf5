#!/bin/bash
# The round's bench lines, wall-clock runs, timelines and the B0 bandwidth numbers, on the GPU box:
#   gpurun -- 'bash tools/tools_round_bench.sh'
# (files land in gpurun_out/final; the ones kept are copied to profiles/ by hand).
set -x
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests -x -q -m gpu > gpurun_out/final_pytest.log 2>&1; tail -3 gpurun_out/final_pytest.log
R=r06
mkdir -p gpurun_out/final
python bench.py > gpurun_out/final/${R}_bench_c3.json 2> gpurun_out/final/c3.err
python bench.py --workload c4 --steps 20 --warmup 3 --check > gpurun_out/final/${R}_bench_c4.json 2> gpurun_out/final/c4.err
python bench.py --workload c5 --steps 5 --warmup 2 --cpu-sample 0 --check > gpurun_out/final/${R}_bench_c5.json 2> gpurun_out/final/c5.err
python bench.py --workload c3tree --steps 50 --warmup 3 --cpu-sample 0 --check --no-wallclock > gpurun_out/final/${R}_bench_c3tree.json 2> gpurun_out/final/c3tree.err
python bench.py --workload c2like --steps 100 --warmup 5 --cpu-sample 0 --check --no-wallclock > gpurun_out/final/${R}_bench_c2like.json 2> gpurun_out/final/c2like.err
python bench.py --workload c3dup --steps 100 --warmup 5 --cpu-sample 0 --check --no-wallclock > gpurun_out/final/${R}_bench_c3dup.json 2> gpurun_out/final/c3dup.err
python bench.py --gpus 2 --workload small --steps 20 --warmup 3 --no-wallclock > gpurun_out/final/${R}_bench_small_2ranks.json 2> gpurun_out/final/small2.err
for r in 0 3 7; do python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --emulate-rank $r/8 --emulate-exchange > gpurun_out/final/${R}_emulated_c4_rank${r}of8.json 2> gpurun_out/final/emu$r.err; done
python tools/tools_wallclock.py --workload c3 --out gpurun_out/final/${R}_wallclock_c3.json > /dev/null 2>&1
python tools/tools_wallclock.py --workload c4 --gpus 2,8 --out gpurun_out/final/${R}_wallclock_c4.json > /dev/null 2>&1
python tools/tools_wallclock.py --workload c5 --out gpurun_out/final/${R}_wallclock_c5.json > /dev/null 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -1
# one process() as a timeline (kernel + copy trace of a short bench run); a rank's step of C4 / 8 the same way
( cd /tmp && export TMPDIR=/tmp && for wl in c3 c5; do
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/final/tl_$wl
  timeout 300 rocprofv3 --kernel-trace --memory-copy-trace -d $GRAFT_REPO_ROOT/gpurun_out/final/tl_$wl --output-format csv -- python3 $GRAFT_REPO_ROOT/bench.py --workload $wl --steps 4 --warmup 1 --cpu-sample 0 --no-profile --no-wallclock > /dev/null 2>&1
  python3 $GRAFT_REPO_ROOT/tools/tools_timeline.py $GRAFT_REPO_ROOT/gpurun_out/final/tl_$wl > $GRAFT_REPO_ROOT/gpurun_out/final/${R}_timeline_$wl.txt
  rm -rf $GRAFT_REPO_ROOT/gpurun_out/final/tl_$wl
done )
for r in 0 3; do bash tools/tools_rank_timeline.sh c4 $r/8 $GRAFT_REPO_ROOT/gpurun_out/final/${R}_timeline_c4_rank${r}of8.txt; done
# seam B0: the byte kernels' bandwidth (HIP-event spans), and rocprofv3's own figure for the same launches
./build/seqcmp_bw > gpurun_out/final/${R}_seqcmp_bw.json 2> gpurun_out/final/seqcmp_bw.err
./build/seqcmp_bw 256 20 > gpurun_out/final/${R}_seqcmp_bw_256MiB.json 2>> gpurun_out/final/seqcmp_bw.err
( cd /tmp && export TMPDIR=/tmp && rm -rf $GRAFT_REPO_ROOT/gpurun_out/final/prof_seqcmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/final/prof_seqcmp --output-format csv -- $GRAFT_REPO_ROOT/build/seqcmp_bw > /dev/null 2>&1 )
head -4 $(find gpurun_out/final/prof_seqcmp -name "*kernel_stats.csv" | head -1) > gpurun_out/final/${R}_seqcmp_bw_rocprof_stats.csv; rm -rf gpurun_out/final/prof_seqcmp
./build/unaligned 1024 2 > gpurun_out/final/${R}_unaligned_loads.json 2>> gpurun_out/final/seqcmp_bw.err
# the C++ host's pass (phylonium-amd --bench-steps: csrc/group.hip) on c4, one context and 2 / 8 ranks sharing this box's GPU
bash tools/tools_cpp_bench.sh c4 "2 8" gpurun_out/final/${R}_cpp_bench_c4.txt 10 > /dev/null 2>&1
# what a rank's step cannot shed: the kernels behind the chains for 128 ... 1 queries
python tools/tools_floor.py c4 3/8 > gpurun_out/final/${R}_floor_c4_rank3of8.json 2> gpurun_out/final/floor.err
python tools/tools_round_summary.py gpurun_out/final
grep -h "emulated\|check" gpurun_out/final/*.err | head
