set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=15 > $O/pytest.log 2>&1; tail -25 $O/pytest.log
./build/seqcmp_bw > $O/seqcmp_bw.json 2> $O/seqcmp_bw.err; cat $O/seqcmp_bw.json
for b in 2 4 8 16; do echo "bpc $b"; PHY_SEQCMP_BPC=$b ./build/seqcmp_bw_dev 64 20 2>&1 | grep long | sed 's/"buffers.*//'; done
( cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/$O/prof_seqcmp --output-format csv -- $GRAFT_REPO_ROOT/build/seqcmp_bw > /dev/null 2>&1 )
f=$(find $O/prof_seqcmp -name "*kernel_stats.csv" | head -1); cat $f | head -12 > $O/seqcmp_bw_rocprof_stats.csv; cat $O/seqcmp_bw_rocprof_stats.csv
rm -rf $O/prof_seqcmp
