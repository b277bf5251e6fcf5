cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4j
(time PHY_FUZZ_SEEDS=150 timeout 2500 python -m pytest tests -m gpu -q -x) > gpurun_out/r4j/gputests.log 2>&1; tail -4 gpurun_out/r4j/gputests.log; grep -a "^E \|^FAILED" gpurun_out/r4j/gputests.log | head -20
