#!/bin/bash
# round-3 GPU call 1: VALU probe, schedule golden traces, the new tests, a bench line
cd ${GRAFT_REPO_ROOT:-.}
mkdir -p gpurun_out
./tools/valu_probe > gpurun_out/valu_probe.txt 2>&1 || exit 1
python tests/golden/make_schedule_golden.py > gpurun_out/sched_golden.log 2>&1 || { tail -20 gpurun_out/sched_golden.log; exit 2; }
python tests/schedule_check.py gpurun_out/schedule_golden/*.jsonl > gpurun_out/sched_check.log 2>&1
timeout -k 10 900 python -m pytest tests/test_full_size_gpu.py tests/test_switches_gpu.py tests/test_oracle_analytic.py -m gpu -x -q --durations=15 > gpurun_out/t_new.log 2>&1
echo "pytest rc=$?" >> gpurun_out/t_new.log
timeout -k 10 400 python bench.py > gpurun_out/bench_r3a.json 2> gpurun_out/bench_r3a.err
echo "bench rc=$?"
tail -5 gpurun_out/t_new.log; cat gpurun_out/valu_probe.txt; head -c 1500 gpurun_out/sched_check.log
