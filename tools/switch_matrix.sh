#!/bin/bash
# tools/switch_matrix.sh — run a parity subset of the GPU suite under every documented SF_* switch (on the GPU box).
# Prints one line per setting; exit code 1 if any setting fails.
sel='full_steps or slabs_fused or slabs_full or bound_sources or randomised or each_operator or test_project or trapezoid'
fail=0
for kv in SF_FUSE2=0 SF_FUSE2=2 SF_JACOBI=0 SF_JACOBI=1 SF_NT=0 SF_NT=1 SF_ISHELL=0 SF_OVL=0 SF_OVL=2 SF_TRAP=0 SF_TRAP=2 \
          SF_SPLIT_FIELDS=0 SF_SPLIT_FIELDS=2 SF_FUSE_SRC=0 SF_GHOST=1 SF_SPLIT=0 SF_STRIP=1 SF_STRIP=2 SF_ZERO_SKIP=0 \
          SF_EVENT_FENCE=1 SF_PREFETCH=0 SF_PREFETCH=3 SF_HALO_STREAM=1 SF_HALO_STREAM=2 SF_HALO_PRIO=1 SF_FUSE_MAXVEC=16 SF_RB=11 SF_GRAPH=1; do
  out=$(env $kv python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "$sel" 2>&1 | tail -1)
  echo "$kv: $out"
  case "$out" in *failed*|*error*) fail=1;; esac
done
exit $fail
