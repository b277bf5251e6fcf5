cd $GRAFT_REPO_ROOT
O=gpurun_out/r05g; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
./build/seqcmp_bw > $O/seqcmp_bw.json 2> $O/seqcmp_bw.err; cat $O/seqcmp_bw.json
python bench.py --steps 50 --warmup 3 --cpu-sample 0 --no-wallclock > $O/c3.json 2> $O/c3.err
python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --no-wallclock > $O/c4.json 2> $O/c4.err
python bench.py --workload c5 --steps 5 --warmup 2 --cpu-sample 0 --no-wallclock > $O/c5.json 2> $O/c5.err
for r in 0 3; do python bench.py --workload c4 --steps 20 --warmup 3 --cpu-sample 0 --emulate-rank $r/8 --emulate-exchange > $O/emu_c4_rank${r}of8.json 2> $O/emu$r.err; done
python -c "
import json
for w in ('c3','c4','c5','emu_c4_rank0of8','emu_c4_rank3of8'):
    d=json.load(open('$O/%s.json' % w)); print(w, d['ms_per_step'], d.get('ms_per_step_noprofile'), {k:v['avg_ms'] for k,v in d['kernels'].items()})"
python tools/tools_wallclock.py --workload c3 --out $O/wallclock_c3.json > /dev/null 2> $O/wc3.err
python tools/tools_wallclock.py --workload c4 --out $O/wallclock_c4.json > /dev/null 2> $O/wc4.err
python -c "
import json
for w in ('c3','c4'):
    d=json.load(open('$O/wallclock_%s.json' % w)); print(w, d['floor']['orderly']['wall_s'], d['floor']['quick']['wall_s']); [print(r['label'], r['wall_s_including_exec'], r['timing'][60:], r['matrix_identical']) for r in d['runs']]"
