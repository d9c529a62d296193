R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r5_gemm; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rg in 1 2 3; do echo -n "ranges=$rg  "; python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges $rg 2>/dev/null; done
for mols in 96 128 160 192; do for lds in 1 2; do echo -n "mols=$mols lds=$lds  "; python3 $R/tools/bench_kernels.py --mols $mols --atoms 27 --dtype bf16 --bf16-lds $lds 2>/dev/null; done; done
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/profr -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --bf16-lds 2 > $O/profr.log 2>&1
f=$(find $O/profr -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && grep gemm $f | cut -c1-140
rm -rf $O/profr
