set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6g; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 900 python3 -m pytest $R/tests -m gpu -x -q -k "gathers_partial or bf16_mode_vs_emulation or randomized_batch or config5 or model_options" > $O/pytest_fused.log 2>&1; echo "rc=$?" >> $O/pytest_fused.log
tail -15 $O/pytest_fused.log
for f in 1 2; do for r in 1 2; do
  timeout 200 python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges $r --node-fused $f 2>&1 | grep -v amdgpu.ids >> $O/ab_fused.txt
done; done
for f in 1 2; do for mols in 64 96 128; do
  timeout 200 python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --ranges 1 --node-fused $f 2>&1 | grep -v amdgpu.ids >> $O/ab_fused.txt
done; done
cat $O/ab_fused.txt
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges 1 --node-fused 2 > $O/prof.log 2>&1
f=$(find $O/prof -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $O/c3_bf16_fused_kernel_stats.csv && python3 $R/tools/stats_csv.py $O/c3_bf16_fused_kernel_stats.csv 8
