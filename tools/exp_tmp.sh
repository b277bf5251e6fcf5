cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4m
(time PHY_FUZZ_SEEDS=300 timeout 2800 python -m pytest tests -m gpu -q -x) > gpurun_out/r4m/gputests.log 2>&1; tail -5 gpurun_out/r4m/gputests.log | head -3; grep -a "^E \|^FAILED\|Memory access" gpurun_out/r4m/gputests.log | head
(time timeout 2000 python -m pytest tests -m gpu -q -x) > gpurun_out/r4m/gputests2.log 2>&1; tail -5 gpurun_out/r4m/gputests2.log | head -3; grep -a "^E \|^FAILED\|Memory access" gpurun_out/r4m/gputests2.log | head
