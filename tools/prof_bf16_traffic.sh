# Round 5: HBM-side traffic of the configs[4] bf16 kernels (separate --pmc passes: FETCH_SIZE and WRITE_SIZE do not fit one pass)
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_stall; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  n=c3_bf16_$c
  timeout 240 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/$n -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --iters 1 > $O/$n.log 2>&1
  f=$(find $O/$n -name "*counter_collection.csv" | head -1); [ -n "$f" ] && python3 $R/tools/pmc_summary.py $f > $O/$n.csv
  rm -rf $O/$n
done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_bf16_kernel_stats_final.csv
rm -rf $O/prof
