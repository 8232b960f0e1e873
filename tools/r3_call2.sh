#!/bin/bash
# round-3 GPU call 2: extended VALU probe; A/B of the marching kernel without v_cndmask selects; parity of the new build
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
./tools/valu_probe > gpurun_out/valu_probe2.txt 2>&1 || exit 1
LIBS="libsfgpu_base.so libsfgpu.so" SIZES="256 512" bash tools/exp_cmp.sh > gpurun_out/exp_cmp2.txt 2>&1 || { tail gpurun_out/exp_cmp2.txt; exit 2; }
timeout -k 10 600 python -m pytest tests/test_parity_gpu.py -k "test_lin_solve or randomised" -x -q > gpurun_out/t2a.log 2>&1; echo "rc=$?" >> gpurun_out/t2a.log
timeout -k 10 300 python -m pytest tests/test_full_size_gpu.py -k "config2 or lin_solve_full" -x -q > gpurun_out/t2b.log 2>&1; echo "rc=$?" >> gpurun_out/t2b.log
timeout -k 10 300 python bench.py --no-cpu-baseline > gpurun_out/bench_r3b.json 2> gpurun_out/bench_r3b.err
cat gpurun_out/valu_probe2.txt gpurun_out/exp_cmp2.txt; tail -3 gpurun_out/t2a.log gpurun_out/t2b.log
