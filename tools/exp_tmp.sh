#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4f
( time python -m pytest tests -x -q -m gpu ) > gpurun_out/r4f/gputests.log 2>&1
tail -5 gpurun_out/r4f/gputests.log
