#!/bin/bash
# tools/collect_profiles.sh <round-tag> — run ON THE GPU BOX (via gpurun). Produces under gpurun_out/profiles_<tag>/:
#   bench.json                      the default bench.py line
#   bench_stats/                    rocprofv3 --kernel-trace --stats of the same command (fewer steps)
#   pmc_<N>/{fetch,write,tcc,req}   separate --pmc passes for the lin_solve sweep at N = 256 and 512
#   rank_share.jsonl                per-rank share timing of the weak-scaling grids (loopback halo)
# Copy the summaries into profiles/ afterwards with tools/summarize_profiles.py.
tag=${1:-r01}
out=gpurun_out/profiles_$tag
rm -rf $out
mkdir -p $out
export TMPDIR=/tmp
python3 bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/bench_stats -- python3 bench.py --steps 5 --warmup 1 --no-cpu-baseline > $out/bench_stats.log 2>&1 || exit 1
for N in 256 512; do
  export SF_SWEEP_K=20 SF_SWEEP_REPS=1
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/pmc_$N/trace -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.trace.log 2>&1 || exit 1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_$N/fetch -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.fetch.log 2>&1 || exit 1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_$N/write -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.write.log 2>&1 || exit 1
  rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_$N/tcc -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.tcc.log 2>&1 || exit 1
  rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum --output-format csv -d $out/pmc_$N/req -- python3 tools/jacobi_sweep.py $N > $out/pmc_$N.req.log 2>&1 || exit 1
done
# one rank's share of the 2/4/8-rank weak-scaling grids on this one GPU (loopback halo; see tools/rank_share.py)
unset SF_SWEEP_K SF_SWEEP_REPS
for R in 1 2 4 8; do python3 tools/rank_share.py --ranks $R >> $out/rank_share.jsonl 2>> $out/rank_share.err || exit 1; done
echo "profiles $tag collected"
