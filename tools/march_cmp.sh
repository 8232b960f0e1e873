#!/bin/bash
# compare the pair kernels inside ONE gpurun call (box-to-box variation is +-3-4 %)
cd ${GRAFT_REPO_ROOT:-.}
export SF_SWEEP_K=${SF_SWEEP_K:-20}
SIZES=${SIZES:-"256 512 320"}
CFGS=${CFGS:-"0,8,0 1,2,0 1,4,0 1,8,0"}
for cfg in $CFGS; do
  IFS=, read m tj ah <<< "$cfg"
  SF_TAG="march=$m tj=$tj ahead=$ah" SF_MARCH=$m SF_MARCH_TJ=$tj SF_MARCH_AHEAD=$ah timeout -k 10 120 python tools/jacobi_sweep.py $SIZES || exit 1
done
