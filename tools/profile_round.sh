set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $R/gpurun_out/bench_c2.json 2> $R/gpurun_out/bench_c2.err
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-x6-probe > $R/gpurun_out/prof_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_stats_x6 -- python3 $R/bench.py --dtype f32x6 --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/prof_stats_x6.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/gpurun_out/prof_fetch -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $R/gpurun_out/prof_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/gpurun_out/prof_write -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $R/gpurun_out/prof_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $R/gpurun_out/prof_mfma -- python3 $R/tools/bench_kernels.py --shape c2 --iters 2 > $R/gpurun_out/prof_mfma.log 2>&1
python3 $R/bench.py --n-samples 256 --variance 12 --no-cpu-baseline > $R/gpurun_out/bench_c3.json 2>/dev/null
find $R/gpurun_out -name "*.csv" | head -30
