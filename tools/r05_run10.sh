cd $GRAFT_REPO_ROOT
O=gpurun_out/r05i; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for wl in "--workload c3 --steps 50 --warmup 3" "--workload c4 --steps 20 --warmup 3" "--workload c5 --steps 5 --warmup 2" "--workload c4 --steps 20 --warmup 3 --emulate-rank 3/8 --emulate-exchange"; do
  echo "== $wl"; bash tools/tools_ab2.sh "$wl" "pileup_project,pileup_pairs_mfma,anchor_spec" before pipe pipe5
done 2>&1 | tee $O/ab_projection.txt
