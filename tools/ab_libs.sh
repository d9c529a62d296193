#!/bin/bash
# On the GPU box: tools/ab_libs.sh "<tags>" <bench_kernels args...> - the same micro-benchmark on several builds of the library
R=$GRAFT_REPO_ROOT
tags=$1; shift
for rep in 1 2; do
for tag in $tags; do
  if [ $tag = base ]; then unset MCG_LIB_PATH; else export MCG_LIB_PATH=$R/tools/native/variants/libmlconfgen_hip_$tag.so; fi
  echo -n "$tag  "
  python3 $R/tools/bench_kernels.py "$@" | sed 's/dtype=f32 //'
done
done
