cd $GRAFT_REPO_ROOT
O=gpurun_out/r05h; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -4 $O/pytest.log
python tools/tools_wallclock.py --workload c4 --out $O/wallclock_c4.json > /dev/null 2> $O/wc4.err
python -c "
import json
for w in ('c4',):
    d=json.load(open('$O/wallclock_%s.json' % w)); print(w, d['floor']['orderly']['wall_s'], d['floor']['quick']['wall_s']); [print(r['label'], r['wall_s_including_exec'], r['timing'][60:], r['matrix_identical']) for r in d['runs']]"
PHY_SLOT_ALIGN_GB=1 PHYLONIUM_AMD_LIB=$PWD/phylonium_amd/libphylonium_amd_dev.so python bench.py --workload c5 --steps 5 --warmup 2 --cpu-sample 0 --no-wallclock 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('c5 slot table on a 1 GiB boundary', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
PHYLONIUM_AMD_LIB=$PWD/phylonium_amd/libphylonium_amd_dev.so python bench.py --workload c5 --steps 5 --warmup 2 --cpu-sample 0 --no-wallclock 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('c5 as allocated', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items()})"
