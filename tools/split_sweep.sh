#!/bin/bash
# On the GPU box: one denoiser call (phi) at N molecules of 27 atoms for several range splits:  tools/split_sweep.sh "64 96" "1 2 3"
R=$GRAFT_REPO_ROOT
for mols in $1; do
for sp in $2; do
  echo -n "[mols=$mols n_ranges=$sp] "; python3 $R/tools/bench_kernels.py --mols $mols --iters 10 --ranges $sp | sed 's/dtype=f32 shape=c2 mt=1//'
done
done
