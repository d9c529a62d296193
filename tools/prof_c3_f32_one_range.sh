# fp32 kernels at the configs[2] shape WITHOUT overlap (one molecule range): per-kernel times of the node GEMMs at 6 895 rows
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_c3f32; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rg in 1 3; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof$rg -- python3 $R/tools/bench_kernels.py --shape c3 --dtype f32 --ranges $rg > $O/prof$rg.log 2>&1
f=$(find $O/prof$rg -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_f32_ranges${rg}_kernel_stats.csv
rm -rf $O/prof$rg
tail -1 $O/prof$rg.log; head -8 $O/c3_f32_ranges${rg}_kernel_stats.csv | cut -c1-120
done
