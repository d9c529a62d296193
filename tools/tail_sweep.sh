#!/bin/bash
# On the GPU box: edge-kernel time per launch over batch sizes, four-tile units only / default split / quarter-tile units only.
R=$GRAFT_REPO_ROOT
for mols in "$@"; do
  for t in all 0 -1; do
    echo -n "mols=$mols four_tile_units=$t  "
    python3 $R/tools/bench_kernels.py --mols $mols --iters 10 --four-tile-units $t | sed 's/dtype=f32 shape=c2 mt=1//'
  done
done
