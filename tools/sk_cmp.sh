#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
export SF_SWEEP_K=${SF_SWEEP_K:-20}
SF_TAG="old" SF_MARCH=0 timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
export SF_MARCH=1
for lib in ${LIBS:-libsfgpu.so}; do for s in ${SS:-3 4}; do for kc in ${KCS:-0}; do
  SF_LIB=$lib SF_TAG="$lib S=$s kc=$kc" SF_SK_S=$s SF_SK_KC=$kc timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
done; done; done
