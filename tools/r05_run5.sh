set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05e; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "queued_rank or block_exchange or rccl or two_process or group_of_ranks or seqcmp or b0 or result_matrices" > $O/pytest_sel.log 2>&1; tail -30 $O/pytest_sel.log
for p in 1 0; do for b in 1 2 4 8; do echo "pipe $p bpc $b"; PHY_SEQCMP_PIPE=$p PHY_SEQCMP_BPC=$b ./build/seqcmp_bw_dev 64 20 2>&1 | grep "long\|batch" | sed 's/"sites.*//'; done; done
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
for r in 0 3 7; do python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --emulate-rank $r/8 --emulate-exchange > $O/emu_c4_rank${r}of8.json 2> $O/emu$r.err; grep emulated $O/emu$r.err; python -c "
import json; d=json.load(open('$O/emu_c4_rank${r}of8.json')); print(d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; done
bash tools/tools_rank_timeline.sh c4 0/8 $GRAFT_REPO_ROOT/$O/timeline_c4_rank0of8.txt; cat $O/timeline_c4_rank0of8.txt
