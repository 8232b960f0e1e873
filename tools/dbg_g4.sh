#!/bin/bash
cd ${GRAFT_REPO_ROOT:-.}
T='tests/test_parity_gpu.py::test_slabs_fused_pairs_two_ghost_planes'
for env in "" "SF_SK_FIRST=0" "SF_GHOST=3" "SF_SK_S=3" "SF_TRAP=0" "SF_SK_FIRST=0 SF_TRAP=0" "SF_SPLIT_FIELDS=0"; do
  echo "== env: $env"
  env $env timeout -k 10 200 python -m pytest "$T" -q -m gpu -k "marching and copy and f32" 2>&1 | grep -E "passed|failed|entries differ" | cut -c1-220
done
