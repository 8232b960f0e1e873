#!/bin/bash
# per-kernel averages of one bench.py run per library variant (SF_LIB): which kernel moved between two builds
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
for lib in ${LIBS:-libsfgpu.so}; do
  d=gpurun_out/bk_${lib%.so}; rm -rf $d
  SF_LIB=$lib rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py --steps ${STEPS:-20} --warmup 3 --no-cpu-baseline --roofline-n -1 > $d.json 2>/dev/null || exit 1
  echo "== $lib $(python3 -c "import json,sys; d=json.loads(open('$d.json').readline()); print(round(d['ms_per_step'],3),'ms/step')")"
  python3 - "$d" <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        rows.append((float(r["TotalDurationNs"]), r["Name"].split("(")[0].replace("void sfk::", "")[:70], int(r["Calls"]), float(r["AverageNs"]) / 1e3))
for t, n, c, a in sorted(rows, reverse=True)[:14]:
    print(f"   {t/1e6:9.2f} ms  {c:5d} x {a:8.1f} us  {n}")
PY
done
