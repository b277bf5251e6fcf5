#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4g
( time PHY_FUZZ_SEEDS=300 python -m pytest tests -x -q -m gpu ) > gpurun_out/r4g/gputests_fuzz300.log 2>&1
tail -5 gpurun_out/r4g/gputests_fuzz300.log
for b in few-chain-blocks default+recheck; do ( PHY_OPTION_BUNDLE=$b python -m pytest tests/test_gpu_parity.py -x -q -m gpu ) 2>&1 | tail -2; done
