#!/bin/bash
# On the GPU box: edge-kernel time per launch over batch sizes, four-tile body only / default split / quarter-tile body only.
R=$GRAFT_REPO_ROOT
for mols in "$@"; do
  for t in 0 default -1; do
    if [ $t = default ]; then unset MCG_TAIL; else export MCG_TAIL=$t; fi
    echo -n "mols=$mols MCG_TAIL=$t  "
    MCG_NS_MAX_TILES=${NS:-512} python3 $R/tools/bench_kernels.py --mols $mols --iters 10 | sed 's/dtype=f32 shape=c2 mt=1//'
  done
done
