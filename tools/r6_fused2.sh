R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6h; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for rep in 1 2; do for mols in 4 8 16 32 48; do for f in 1 2; do
  echo -n "fused=$f " >> $O/ab_fused_small.txt
  timeout 200 python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --ranges 1 --node-fused $f 2>&1 | grep -v amdgpu.ids >> $O/ab_fused_small.txt
done; done; done
cat $O/ab_fused_small.txt
