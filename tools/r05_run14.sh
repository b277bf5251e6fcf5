cd $GRAFT_REPO_ROOT
O=gpurun_out/r05k; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -3 $O/pytest.log
python bench.py --cpu-sample 0 --no-wallclock > $O/c3.json 2> $O/c3.err; python -c "
import json; d=json.load(open('$O/c3.json')); print(d['value'], d['ms_per_step'], d['ms_per_step_noprofile'], d['ms_per_step_all_kernels_timed'], d['roofline']['frac'], d['roofline_mfma']['clock_ghz'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --emulate-rank 3/8 --emulate-exchange > $O/emu3.json 2> $O/emu3.err; python -c "
import json; d=json.load(open('$O/emu3.json')); print(d['ms_per_step'], d['ms_per_step_all_kernels_timed'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"; grep emulated $O/emu3.err
python bench.py --gpus 2 --workload small --steps 10 --warmup 2 --cpu-sample 0 --no-wallclock 2> $O/small2.err | python -c "
import json,sys; d=json.load(sys.stdin); print('2 ranks', d['value'], d['ms_per_step'], d['config']['backend'])"
