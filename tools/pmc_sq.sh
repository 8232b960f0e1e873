#!/bin/bash
# tools/pmc_sq.sh <tag> <N> — instruction-issue picture of the Jacobi kernel (separate --pmc passes, K=20 sweeps)
tag=$1; N=${2:-256}
export TMPDIR=/tmp SF_SWEEP_K=20 SF_SWEEP_REPS=1
out=gpurun_out/pmcsq_$tag; rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 tools/jacobi_sweep.py $N > $out/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $out/insts -- python3 tools/jacobi_sweep.py $N > $out/insts.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $out/active -- python3 tools/jacobi_sweep.py $N > $out/active.log 2>&1
rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INST_CYCLES_VMEM SQ_THREAD_CYCLES_VALU SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/wait -- python3 tools/jacobi_sweep.py $N > $out/wait.log 2>&1
python3 tools/pmc_summary.py $out | grep -v "fillBuffer\|copyBuffer"
