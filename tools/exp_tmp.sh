#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-sample 0 --no-wallclock --workload c3 --steps 30 --warmup 3 --check "$@" 2>gpurun_out/e.err | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$PHY_CHUNK_WEIGHTS_FILE $*', d['ms_per_step'], d['phase_a_plan'], {k:v['avg_ms'] for k,v in d['kernels'].items() if k in ('anchor_spec','anchor_bridge','anchor_fold')})"; grep -c "check vs oracle.*OK" gpurun_out/e.err; }
run
PHY_CHUNK_WEIGHTS_FILE=tools/scratch/weights_life.txt run
PHY_CHUNK_WEIGHTS_FILE=tools/scratch/weights_trips.txt run
run
PHY_CHUNK_WEIGHTS_FILE=tools/scratch/weights_life.txt run
