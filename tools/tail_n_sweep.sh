#!/bin/bash
# On the GPU box: edge launch / denoiser call at N molecules of 27 atoms for several numbers of four-tile workgroups (MCG_TAIL)
R=$GRAFT_REPO_ROOT
mols=$1; shift
for t in "$@"; do
  echo -n "[mols=$mols MCG_TAIL=$t] "; MCG_TAIL=$t python3 $R/tools/bench_kernels.py --mols $mols --iters 20 | sed 's/dtype=f32 shape=c2 mt=1//'
done
