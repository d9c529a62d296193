# round 6, first GPU call: full -m gpu suite, the grid-barrier probe, kernel trace of a small-batch sampler, bench line
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6a; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q -s -k "bf16_mode_at_the_judged" > $O/bf16_judged.log 2>&1; echo "rc=$?" >> $O/bf16_judged.log
timeout 120 $R/tools/native/grid_barrier_probe > $O/grid_barrier_probe.txt 2>&1; echo "rc=$?" >> $O/grid_barrier_probe.txt
timeout 1200 python3 -m pytest $R/tests -m gpu -x -q > $O/pytest_gpu.log 2>&1; echo "rc=$?" >> $O/pytest_gpu.log
MCG_SMALL_SHAPES=4,27,8,27 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_small -- python3 $R/tools/bench_small.py > $O/prof_small.log 2>&1
f=$(find $O/prof_small -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/small_kernel_stats.csv
MCG_SMALL_SHAPES=4,27,8,27 timeout 120 python3 $R/tools/bench_small.py > $O/bench_small.txt 2>&1
timeout 900 python3 $R/bench.py > $O/bench.json 2> $O/bench.err; echo "rc=$?" >> $O/bench.err
tail -5 $O/bf16_judged.log; cat $O/grid_barrier_probe.txt; tail -3 $O/pytest_gpu.log; cat $O/bench_small.txt
