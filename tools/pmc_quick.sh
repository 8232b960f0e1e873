#!/bin/bash
# tools/pmc_quick.sh <tag> <N> — a few PMC passes for the current Jacobi configuration (env knobs apply)
tag=$1; N=${2:-512}
export TMPDIR=/tmp SF_SWEEP_K=4 SF_SWEEP_REPS=1
out=gpurun_out/pmcq_$tag; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/jacobi_sweep.py $N > $out/trace.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 tools/jacobi_sweep.py $N > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 tools/jacobi_sweep.py $N > $out/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --output-format csv -d $out/tcc -- python3 tools/jacobi_sweep.py $N > $out/tcc.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum --output-format csv -d $out/tcp -- python3 tools/jacobi_sweep.py $N > $out/tcp.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD --output-format csv -d $out/sq -- python3 tools/jacobi_sweep.py $N > $out/sq.log 2>&1
rocprofv3 --pmc TA_BUSY_avr TA_TA_BUSY_sum TCP_PENDING_STALL_CYCLES_sum GRBM_GUI_ACTIVE --output-format csv -d $out/ta -- python3 tools/jacobi_sweep.py $N > $out/ta.log 2>&1
python3 tools/pmc_summary.py $out | grep -v fillBuffer
