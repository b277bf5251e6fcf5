set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05a; mkdir -p $O
python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --no-wallclock > $O/base_c4.json 2> $O/base_c4.err
for r in 0 3; do python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --emulate-rank $r/8 --emulate-exchange > $O/base_emu_c4_rank${r}of8.json 2> $O/base_emu$r.err; done
for r in 0 3; do bash tools/tools_rank_timeline.sh c4 $r/8 $GRAFT_REPO_ROOT/$O/base_timeline_c4_rank${r}of8.txt; done
python bench.py --steps 50 --warmup 3 --cpu-sample 0 --no-wallclock > $O/base_c3.json 2> $O/base_c3.err
grep -h "emulated" $O/*.err
