cd $GRAFT_REPO_ROOT
O=gpurun_out/r05f; mkdir -p $O
timeout 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; tail -5 $O/pytest.log
./build/seqcmp_bw > $O/seqcmp_bw.json 2> $O/seqcmp_bw.err; cat $O/seqcmp_bw.json
for p in 1 0; do for b in 1 2 4 8; do echo "pipe $p bpc $b"; PHY_SEQCMP_PIPE=$p PHY_SEQCMP_BPC=$b ./build/seqcmp_bw_dev 64 20 2>&1 | grep "long" | sed 's/"sites.*//'; done; done
python tools/tools_wallclock.py --workload c3 --out $O/wallclock_c3.json > /dev/null 2> $O/wc3.err
python tools/tools_wallclock.py --workload c4 --gpus 1,2,8 --out $O/wallclock_c4.json > /dev/null 2> $O/wc4.err
python -c "
import json
for w in ('c3','c4'):
    d=json.load(open('$O/wallclock_%s.json' % w)); print(w, d['floor']); [print(r['label'], r['wall_s_including_exec'], r['timing'][:330], r['matrix_identical']) for r in d['runs']]"
