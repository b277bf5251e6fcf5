cd $GRAFT_REPO_ROOT
O=gpurun_out/r05l; mkdir -p $O
for wl in "--workload c3 --steps 50 --warmup 3" "--workload c4 --steps 20 --warmup 3" "--workload c5 --steps 5 --warmup 2"; do
  echo "== $wl"; bash tools/tools_ab2.sh "$wl" "anchor_fold,anchor_bridge" base apt8 apt2
done 2>&1 | tee $O/ab_fold.txt
