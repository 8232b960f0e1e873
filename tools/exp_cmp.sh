#!/bin/bash
# timing experiments with library variants (SF_LIB): kernel-trace averages of the Jacobi launches of one 20-sweep solve
cd ${GRAFT_REPO_ROOT:-.}
export SF_SWEEP_K=20 SF_SWEEP_REPS=1 TMPDIR=/tmp
for lib in ${LIBS:-libsfgpu.so}; do for n in ${SIZES:-256 512}; do
  d=gpurun_out/exp_${lib%.so}_$n; rm -rf $d
  SF_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 tools/jacobi_sweep.py $n > /dev/null 2>&1 || exit 1
  echo "== $lib N=$n"
  python3 - "$d" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "jacobi_sk" in r["Name"]:
            print("  ", r["Name"].split("(")[0].split("jacobi_sk_kernel")[1], r["Calls"], round(float(r["AverageNs"]) / 1e3, 1))
PY
done; done
