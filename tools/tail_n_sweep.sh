#!/bin/bash
# On the GPU box: edge launch / denoiser call at N molecules of 27 atoms for several numbers of four-tile workgroups
#   tools/tail_n_sweep.sh <mols> <four_tile_units> ...
R=$GRAFT_REPO_ROOT
mols=$1; shift
for t in "$@"; do
  echo -n "[mols=$mols four_tile_units=$t] "; python3 $R/tools/bench_kernels.py --mols $mols --iters 20 --four-tile-units $t | sed 's/dtype=f32 shape=c2 mt=1//'
done
