#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/tools_ab2.sh "--workload c5 --steps 5 --warmup 2 --check" "anchor_spec,anchor_bridge" base nt1 nt2
bash tools/tools_ab2.sh "--workload c3 --steps 30 --warmup 3 --check" "anchor_spec,anchor_bridge" base nt1 nt2
bash tools/tools_ab2.sh "--workload c4 --steps 10 --warmup 2" "anchor_spec,anchor_bridge" base nt1 nt2
