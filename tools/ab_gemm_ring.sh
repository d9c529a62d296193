R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_gemm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 -m pytest $R/tests/test_hip_parity.py -q -x -k "gathers_partial or bf16_mode_vs or config5" 2>&1 | tail -2
for i in 1 2; do python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 2>/dev/null; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profr -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 > $O/profr.log 2>&1
f=$(find $O/profr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/final_c3_bf16_kernel_stats.csv && grep "gemm\|edge" $f | cut -c1-140
rm -rf $O/profr
