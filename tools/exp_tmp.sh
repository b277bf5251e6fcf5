#!/bin/bash
cd $GRAFT_REPO_ROOT
run() { python bench.py --cpu-sample 0 --no-wallclock --no-profile "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$PHY_NO_EAGER/$PHY_OVERLAP_RANGES $*', d['ms_per_step'], d.get('check'))"; }
for wl in "c3 --steps 50 --warmup 5 --check" "c4 --steps 10 --warmup 2 --check" "c5 --steps 5 --warmup 2 --check"; do
for rep in 1 2; do
run --workload $wl
PHY_NO_EAGER=1 run --workload $wl
PHY_NO_EAGER=1 PHY_OVERLAP_RANGES=2 run --workload $wl
PHY_NO_EAGER=1 PHY_OVERLAP_RANGES=4 run --workload $wl
PHY_NO_EAGER=1 PHY_OVERLAP_RANGES=8 run --workload $wl
done; done
