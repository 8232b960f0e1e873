#!/bin/bash
# tools/pmc_jacobi.sh <tag> <N> — rocprofv3 PMC passes (one counter group per run) + a kernel-trace run for
# the lin_solve sweep. Outputs under gpurun_out/pmc_<tag>/. Run on the GPU box via gpurun.
tag=$1; N=${2:-512}
export TMPDIR=/tmp SF_SWEEP_K=${SF_SWEEP_K:-4} SF_SWEEP_REPS=1
out=gpurun_out/pmc_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/jacobi_sweep.py $N > $out/trace.log 2>&1 &&
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/jacobi_sweep.py $N > $out/fetch.log 2>&1 &&
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/jacobi_sweep.py $N > $out/write.log 2>&1 &&
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- python3 tools/jacobi_sweep.py $N > $out/tcc.log 2>&1 &&
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $out/req -- python3 tools/jacobi_sweep.py $N > $out/req.log 2>&1
echo "pmc $tag done rc=$?"
find $out -name "*.csv" | head -20
