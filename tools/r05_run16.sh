cd $GRAFT_REPO_ROOT
O=gpurun_out/r05m; mkdir -p $O
for wl in "--workload c3 --steps 100 --warmup 3" "--workload c4 --steps 20 --warmup 3" "--workload c4 --steps 20 --warmup 3 --emulate-rank 3/8 --emulate-exchange"; do
  echo "== $wl"; bash tools/tools_ab2.sh "$wl" "anchor_fold,pileup_pairs_mfma,pileup_project,anchor_spec" old base behind
done 2>&1 | tee $O/ab_visited.txt
