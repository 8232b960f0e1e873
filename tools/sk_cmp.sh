#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export SF_SWEEP_K=${SF_SWEEP_K:-20}
SF_TAG="old" SF_MARCH=0 timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
export SF_MARCH=1
for s in ${SS:-3}; do for cfg in ${CFGS:--1 0 3}; do for kc in ${KCS:-0}; do
  SF_TAG="sk S=$s cfg=$cfg kc=$kc" SF_SK_S=$s SF_SK_CFG=$cfg SF_SK_KC=$kc timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
done; done; done
