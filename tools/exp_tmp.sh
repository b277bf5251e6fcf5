#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/tools_ab2.sh "--workload c3 --steps 30 --warmup 3 --check" "anchor_spec,anchor_bridge,anchor_fold" l32 l64
bash tools/tools_ab2.sh "--workload c4 --steps 8 --warmup 2 --check" "anchor_spec,anchor_bridge,anchor_fold" l32 l64
bash tools/tools_ab2.sh "--workload c5 --steps 4 --warmup 2 --check" "anchor_spec,anchor_bridge,anchor_fold" l32 l64
bash tools/tools_ab2.sh "--workload c2like --steps 50 --warmup 3 --check" "anchor_spec,anchor_bridge,anchor_fold" l32 l64
