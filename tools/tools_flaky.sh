#!/bin/bash
# usage (GPU box): tools_flaky.sh <repeats> <pytest -k expression> — dev: repeat a test selection until it fails, show the failure's head
N=$1; K=$2
cd ${GRAFT_REPO_ROOT:-/root/repo}
for i in $(seq 1 $N); do
  python -X faulthandler -m pytest tests -m gpu -x -q -s -k "$K" > gpurun_out/flaky.txt 2>&1
  if ! tail -3 gpurun_out/flaky.txt | grep -q " passed"; then echo "run $i failed"; grep -v "site-packages\|dist-packages/_pytest\|pluggy" gpurun_out/flaky.txt | head -60; exit 1; fi
  echo "run $i ok: $(tail -1 gpurun_out/flaky.txt)"
done
