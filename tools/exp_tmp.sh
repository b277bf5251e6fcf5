#!/bin/bash
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/mb
./build/mb/mfma_pairs 1024 > gpurun_out/mb/mfma_pairs_n1024.txt 2>&1
grep -A2 "64x128\|v3 64x64, 8 wave\|v3 64x64, 16 wave\|check MFMA v3 64x128" gpurun_out/mb/mfma_pairs_n1024.txt | head -40
