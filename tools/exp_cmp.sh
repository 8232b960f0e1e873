#!/bin/bash
# timing-only experiments with deliberately broken library variants (results are wrong; only the clock is read)
cd ${GRAFT_REPO_ROOT:-.}
export SF_SWEEP_K=20 SF_MARCH=2 SF_SK_S=3
for lib in ${LIBS:-libsfgpu.so libsfgpu_NOBARRIER.so libsfgpu_NOSTORE.so}; do for kc in ${KCS:-0 64}; do
  SF_LIB=$lib SF_TAG="$lib kc=$kc" SF_SK_KC=$kc timeout -k 10 120 python tools/jacobi_sweep.py ${SIZES:-256 512} || exit 1
done; done
