set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05c; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu --durations=12 > $O/pytest.log 2>&1; tail -22 $O/pytest.log
./build/seqcmp_bw > $O/seqcmp_bw.json 2> $O/seqcmp_bw.err; cat $O/seqcmp_bw.json
for b in 1 2 3 4 6; do echo "bpc $b"; PHY_SEQCMP_BPC=$b ./build/seqcmp_bw_dev 64 20 2>&1 | grep long | sed 's/"sites.*//'; done
echo 256MiB; ./build/seqcmp_bw 256 20 2>&1 | grep long | sed 's/"sites.*//'
