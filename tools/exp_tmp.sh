cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r4g
(time timeout 2000 python -m pytest tests -m gpu -q -x) > gpurun_out/r4g/gputests.log 2>&1; tail -4 gpurun_out/r4g/gputests.log; grep -a "^E \|^FAILED" gpurun_out/r4g/gputests.log | head -20
