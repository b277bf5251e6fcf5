#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4i
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r4i/gputests.log 2>&1
tail -4 gpurun_out/r4i/gputests.log
python bench.py --workload c5 --steps 4 --warmup 2 --cpu-sample 0 --no-wallclock --check 2>&1 | grep -o "check vs oracle[^\"]*\|\"ms_per_step\": [0-9.]*" | head -3
python bench.py --workload c4 --steps 6 --warmup 2 --cpu-sample 0 --no-wallclock --check --emulate-rank 0/8 --emulate-exchange 2>&1 | grep -o "check[^\"]*OK\|\"ms_per_step\": [0-9.]*" | head -3
