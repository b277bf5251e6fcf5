cd $GRAFT_REPO_ROOT
O=gpurun_out/r05j; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
for wl in "--workload c3 --steps 100 --warmup 3" "--workload c4 --steps 20 --warmup 3"; do
  echo "== $wl"; for rep in 1 2; do python bench.py $wl --cpu-sample 0 --no-wallclock 2>/dev/null | python -c "import json,sys; d=json.load(sys.stdin); print(d['ms_per_step'], d['ms_per_step_noprofile'], d['value'])"; done
done 2>&1 | tee $O/one_gpu.txt
