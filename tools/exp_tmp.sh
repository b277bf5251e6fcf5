#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4j
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r4j/gputests.log 2>&1
tail -4 gpurun_out/r4j/gputests.log
python bench.py --steps 100 --warmup 5 --cpu-sample 0 --no-wallclock 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['ms_per_step_noprofile'])"
