#!/bin/bash
# On the GPU box: tools/variant_sweep.sh "<tags>" <mols...>  - edge time of measurement builds (tools/build_variants.sh)
R=$GRAFT_REPO_ROOT
tags=$1; shift
for mols in "$@"; do
  for tag in $tags; do
    if [ $tag = base ]; then unset MCG_LIB_PATH; else export MCG_LIB_PATH=$R/tools/native/variants/libmlconfgen_hip_$tag.so; fi
    echo -n "mols=$mols $tag  "
    MCG_NS_MAX_TILES=0 python3 $R/tools/bench_kernels.py --mols $mols --iters 5 | sed 's/dtype=f32 shape=c2 mt=1//'
  done
done
