#!/bin/bash
# tools/collect_profiles.sh <round-tag> — run ON THE GPU BOX (via gpurun). Produces under gpurun_out/profiles_<tag>/:
#   bench.json                      the default bench.py line
#   bench_stats/                    rocprofv3 --kernel-trace --stats of the same command (fewer steps)
#   pmc_<N>/{trace,fetch,write,tcc,req}   separate --pmc passes for a 20-sweep lin_solve at N = 256 and 512
#   sq_<N>/{insts,active,wait}      instruction-issue counters of the same solve (separate --pmc passes)
#   sq_old_<N>/...                  the same with SF_MARCH=0 (the register-blocked pair kernel of round 1)
#   rank_share.jsonl                one rank's share of the 1024^3 strong-scaling and of the weak-scaling grids
# Counters are collected in their own runs with --pmc only (no trace domains); see MI355X_MICROARCH.md, HBM section.
# Copy the summaries into profiles/ afterwards with tools/summarize_profiles.py.
tag=${1:-r02}
out=gpurun_out/profiles_$tag
rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline --roofline-n -1 > $out/bench_stats.log 2>&1 || exit 1
export SF_SWEEP_K=20 SF_SWEEP_REPS=1
for N in 256 512; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/pmc_$N/trace -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.trace.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_$N/fetch -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$N/write -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.write.log 2>&1 || exit 1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_$N/tcc -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.tcc.log 2>&1 || exit 1
  rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $out/pmc_$N/req -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.req.log 2>&1 || exit 1
  for mode in 1 0; do
    d=$out/sq_$N; [ $mode = 0 ] && d=$out/sq_old_$N
    export SF_MARCH=$mode
    rocprofv3 --kernel-trace --stats --output-format csv -d $d/trace -- python3 tools/jacobi_sweep.py $N > $d.trace.log 2>&1 || exit 1
    rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_LDS SQ_INSTS_SMEM --output-format csv -d $d/insts -- python3 tools/jacobi_sweep.py $N > $d.insts.log 2>&1 || exit 1
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM --output-format csv -d $d/active -- python3 tools/jacobi_sweep.py $N > $d.active.log 2>&1 || exit 1
    rocprofv3 --pmc SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --output-format csv -d $d/wait -- python3 tools/jacobi_sweep.py $N > $d.wait.log 2>&1 || exit 1
  done
  unset SF_MARCH
done
unset SF_SWEEP_K SF_SWEEP_REPS
# one rank's share on this one GPU (loopback halo; see tools/rank_share.py): the 1024^3 strong-scaling grid and the
# weak-scaling grids of bench.py --weak
python3 tools/rank_share.py --ranks 1 --grid 1024 --steps 4 --warmup 1 >> $out/rank_share.jsonl 2>> $out/rank_share.err || exit 1
for R in 2 4 8; do python3 tools/rank_share.py --ranks $R --grid 1024 --steps 5 --warmup 2 >> $out/rank_share.jsonl 2>> $out/rank_share.err || exit 1; done
for R in 1 2 4 8; do python3 tools/rank_share.py --ranks $R --steps 10 >> $out/rank_share.jsonl 2>> $out/rank_share.err || exit 1; done
python3 tools/rank_share.py --ranks 1 --grid 512 --dtype f64 --iters 40 --steps 4 --warmup 1 >> $out/rank_share.jsonl 2>> $out/rank_share.err || exit 1
python3 tools/rank_share.py --ranks 8 --grid 512 --dtype f64 --iters 40 --steps 5 --warmup 2 >> $out/rank_share.jsonl 2>> $out/rank_share.err || exit 1
bash tools/cfg_bench.sh > $out/configs.txt 2>&1
echo "profiles $tag collected"
