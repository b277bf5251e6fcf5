set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05d; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "queued_rank or block_exchange or rccl or two_process or group_of_ranks or seqcmp or b0 or result_matrices" > $O/pytest_sel.log 2>&1; tail -30 $O/pytest_sel.log
./build/seqcmp_bw > $O/seqcmp_bw.json 2> $O/seqcmp_bw.err; cat $O/seqcmp_bw.json
for r in 0 3; do python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --emulate-rank $r/8 --emulate-exchange > $O/emu_c4_rank${r}of8.json 2> $O/emu$r.err; grep emulated $O/emu$r.err; done
bash tools/tools_rank_timeline.sh c4 0/8 $GRAFT_REPO_ROOT/$O/timeline_c4_rank0of8.txt; cat $O/timeline_c4_rank0of8.txt
python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --no-wallclock > $O/c4.json 2> $O/c4.err; python -c "
import json; d=json.load(open('$O/c4.json')); print(d['ms_per_step'], d['roofline_mfma'])"
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
