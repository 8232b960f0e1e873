#!/bin/bash
# the larger BASELINE.json configurations for library variants (SF_LIB) inside ONE gpurun call: LIBS="a.so b.so" bash tools/cfg_cmp.sh
cd ${GRAFT_REPO_ROOT:-.}
for lib in ${LIBS:-libsfgpu.so}; do
  export SF_LIB=$lib
  echo "== $lib"
  python bench.py --grid 512 --iters 40 --dtype f64 --steps 3 --warmup 1 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  config5 512^3 f64 K=40:', round(d['ms_per_step'],2),'ms')"
  python bench.py --grid 256 --dtype f64 --steps 10 --warmup 2 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  256^3 f64 K=20:', round(d['ms_per_step'],3),'ms')"
  if [ -z "$F64_ONLY" ]; then
    python bench.py --grid 512 --iters 40 --steps 5 --warmup 1 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  config3 512^3 f32 K=40:', round(d['ms_per_step'],2),'ms')"
    python bench.py --grid 1024 --iters 20 --steps 3 --warmup 1 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('  config4 1024^3 f32 K=20 (1 GPU):', round(d['ms_per_step'],2),'ms')"
  fi
done
