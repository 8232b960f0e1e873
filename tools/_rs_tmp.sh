set -e
python -m pytest tests -m gpu -x -q > gpurun_out/wide_tests.log 2>&1 || { tail -30 gpurun_out/wide_tests.log; exit 1; }
tail -2 gpurun_out/wide_tests.log
SF_OVL=0 python -m pytest tests/test_parity_gpu.py -m gpu -x -q -k "lin_solve or full_steps" > gpurun_out/wide0_tests.log 2>&1 || { tail -30 gpurun_out/wide0_tests.log; exit 1; }
tail -1 gpurun_out/wide0_tests.log
SF_SWEEP_REPS=2 SF_TAG=default timeout -k 10 300 python tools/jacobi_sweep.py 768 1024
SF_SWEEP_REPS=2 SF_SWEEP_DTYPE=f64 SF_TAG=default timeout -k 10 300 python tools/jacobi_sweep.py 512 640
