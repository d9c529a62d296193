# Round 5: molecule ranges of the bf16 mode beyond the judged batch size (27-atom molecules; ms per denoiser call)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mols in 288 384 512 768; do for rg in 1 2 3 4; do echo -n "mols=$mols x27 bf16 ranges=$rg  "; python3 $R/tools/bench_kernels.py --mols $mols --dtype bf16 --mt 4 --ranges $rg; done; done
