cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for rep in 1 2; do for rg in 1 2 3; do echo -n "c3 bf16 ranges=$rg  "; python3 $R/tools/bench_kernels.py --shape c3 --dtype bf16 --ranges $rg; done; done
for mols in 96 128 160 192; do for rg in 1 2; do echo -n "mols=$mols x27 bf16 ranges=$rg  "; python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --mt 4 --ranges $rg; done; done
