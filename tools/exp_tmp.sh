#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-sample 0 --no-wallclock --workload $WL --steps $ST --warmup 2 --check "$@" 2>gpurun_out/e.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$WL cpw=$PHY_PAIRS_CPW $*', d['ms_per_step'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ('pileup_pairs_mfma','pileup_project')})"; grep -c "check vs oracle.*OK" gpurun_out/e.err; }
WL=c4; ST=6
PHY_PAIRS_CPW=1 run
run
PHY_PAIRS_CPW=2 run
PHY_PAIRS_CPW=8 run
PHY_PAIRS_CPW=16 run
PHY_PAIRS_CPW=4 run --pairs-wchunk 132
WL=c3; ST=30
PHY_PAIRS_CPW=1 run
run
PHY_PAIRS_CPW=2 run
WL=c5; ST=4
PHY_PAIRS_CPW=1 run
run
python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2
