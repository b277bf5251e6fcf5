#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4m
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r4m/gputests.log 2>&1
tail -4 gpurun_out/r4m/gputests.log | head -2
bash tools/tools_prof.sh f3 c3 > /dev/null 2>&1
bash tools/tools_prof.sh f4 c4 > /dev/null 2>&1
bash tools/tools_prof.sh f5 c5 > /dev/null 2>&1
bash tools/tools_prof.sh frank c4 --emulate-rank 0/8 --emulate-exchange > /dev/null 2>&1
ls gpurun_out/prof_f3 | head -3
