#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export SF_SWEEP_K=20
SF_TAG="old" SF_MARCH=0 timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
export SF_MARCH=1
for tj in ${TJS:-2 4 6}; do for kc in ${KCS:-0 8 16 32}; do
  SF_TAG="tj=$tj kc=$kc" SF_MARCH_TJ=$tj SF_MARCH_KC=$kc timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
done; done
