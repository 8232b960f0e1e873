#!/bin/bash
# tools/pmc_bench_kernels.sh <tag> — PMC passes over a short bench.py run: traffic and L2 behaviour of every kernel of a step
tag=${1:-x}; out=gpurun_out/pmcb_$tag; rm -rf $out; mkdir -p $out; export TMPDIR=/tmp
A="bench.py --steps 3 --warmup 1 --no-cpu-baseline --roofline-n 0"
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- python3 $A > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- python3 $A > $out/write.log 2>&1
rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/tcc -- python3 $A > $out/tcc.log 2>&1
rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum --output-format csv -d $out/tcp -- python3 $A > $out/tcp.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAVES --output-format csv -d $out/sq -- python3 $A > $out/sq.log 2>&1
python3 tools/pmc_summary.py $out | grep -E "==|advect|project|add_source" | cut -c1-150
