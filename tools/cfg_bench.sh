#!/bin/bash
# the BASELINE.json configurations that fit one GPU, one bench.py line each (no CPU baseline, few steps)
cd ${GRAFT_REPO_ROOT:-.}
python bench.py --grid 512 --iters 40 --steps 5 --warmup 1 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config3 512^3 f32 K=40:', round(d['ms_per_step'],2),'ms', round(d['value']),'Mcells/s')"
python bench.py --grid 1024 --iters 20 --steps 3 --warmup 1 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config4 1024^3 f32 K=20 (1 GPU):', round(d['ms_per_step'],2),'ms', round(d['value']),'Mcells/s')"
python bench.py --grid 512 --iters 40 --dtype f64 --steps 3 --warmup 1 --no-cpu-baseline --roofline-n -1 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('config5 512^3 f64 K=40:', round(d['ms_per_step'],2),'ms', round(d['value']),'Mcells/s')"
SF_SWEEP_K=20 python tools/jacobi_sweep.py 1024 768
SF_SWEEP_K=20 SF_MARCH=0 SF_TAG=old python tools/jacobi_sweep.py 1024
