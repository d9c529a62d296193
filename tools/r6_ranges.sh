cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r6j; mkdir -p $O
for rep in 1 2; do for mols in 48 64 80 96 112 128 256 320; do for rg in 1 2 3; do
  if [ $rg = 3 ] && [ $mols -lt 256 ]; then continue; fi
  echo -n "mols=$mols x27 bf16 ranges=$rg  " >> $O/ranges.txt; python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --mt 4 --ranges $rg 2>&1 | grep -v amdgpu.ids >> $O/ranges.txt
done; done; done
awk '{print $1,$2,$3,$4,$NF}' $O/ranges.txt
