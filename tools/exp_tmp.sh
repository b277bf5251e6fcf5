#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4k
( time PHY_FUZZ_SEEDS=150 python -m pytest tests -x -q -m gpu ) > gpurun_out/r4k/gputests.log 2>&1
tail -4 gpurun_out/r4k/gputests.log
