#!/bin/bash
cd $GRAFT_REPO_ROOT
bash tools/tools_ab2.sh "--workload c3 --steps 30 --warmup 3 --check" "pileup_project,pileup_pairs_mfma" pj0 pj16
bash tools/tools_ab2.sh "--workload c4 --steps 6 --warmup 2 --check" "pileup_project,pileup_pairs_mfma" pj0 pj16
bash tools/tools_ab2.sh "--workload c5 --steps 4 --warmup 2 --check" "pileup_project,pileup_pairs_mfma" pj0 pj16
