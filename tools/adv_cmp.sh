#!/bin/bash
# advect kernel averages of short bench runs under different SF_ADVECT_ROW settings (one line per setting and kernel)
cd ${GRAFT_REPO_ROOT:-.}
export TMPDIR=/tmp
run() {  # $1 = "ENV=.. ENV=.." ; $2 = bench args
  d=gpurun_out/adv_tmp; rm -rf $d
  env $1 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 bench.py $2 --steps 3 --warmup 1 --no-cpu-baseline --roofline-n -1 > /dev/null 2>&1 || return 1
  python3 - "$d" "$1 | $2" <<'PY'
import csv, glob, sys
for f in glob.glob(sys.argv[1] + "/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "advect" in r["Name"]:
            print(f"{sys.argv[2]:55s} {r['Name'].split('(')[0][-40:]:42s} {float(r['AverageNs'])/1e3:9.1f} us")
PY
}
for a in "--grid 256 --iters 4" "--grid 512 --iters 4" "--grid 256 --iters 4 --dtype f64"; do
  for e in ${ROWS:-0 2 3}; do run "SF_ADVECT_ROW=$e" "$a" || exit 1; done
done
